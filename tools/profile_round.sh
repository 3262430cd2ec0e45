set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r01b
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01b/stats -o bench -- python3 bench.py > gpurun_out/r01b/bench_profiled.json 2> gpurun_out/r01b/bench_profiled.err
timeout -k 10 300 python3 bench.py > gpurun_out/r01b/bench.json 2> gpurun_out/r01b/bench.err
timeout -k 10 300 python3 bench.py --mode strict --steps 2 --spp 16 --cpu-seconds 0 > gpurun_out/r01b/bench_strict.json 2> gpurun_out/r01b/bench_strict.err
bash tools/run_pmc.sh gpurun_out/r01b/pmc > gpurun_out/r01b/pmc_summary.txt 2>&1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/r01b/fetch --output-format csv -- python3 bench.py --steps 1 --warmup 0 --spp 4 --cpu-seconds 0 > gpurun_out/r01b/fetch.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/r01b/write --output-format csv -- python3 bench.py --steps 1 --warmup 0 --spp 4 --cpu-seconds 0 > gpurun_out/r01b/write.log 2>&1
python3 tools/pmc_summary.py gpurun_out/r01b/fetch k_trace > gpurun_out/r01b/fetch_write.txt
python3 tools/pmc_summary.py gpurun_out/r01b/write k_trace >> gpurun_out/r01b/fetch_write.txt
cat gpurun_out/r01b/fetch_write.txt
tail -c 1500 gpurun_out/r01b/bench.json
