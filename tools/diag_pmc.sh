set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/diag1; mkdir -p $out
common="--workload cfg3 --steps 1 --warmup 0 --spp 4 --cpu-seconds 0 --no-extra --mode fast"
timeout -s KILL 240 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum -d $out/a --output-format csv -- python3 bench.py $common > $out/a.log 2>$out/a.err
python3 tools/pmc_summary.py $out k_trace > $out/summary.txt
cat $out/summary.txt | head -40
