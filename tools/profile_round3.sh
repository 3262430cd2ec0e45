#!/bin/bash
# Profiling evidence on one MI355X (run through gpurun): bench lines, kernel
# stats, and the counter passes bench.py's roofline constants come from.
#   RT_GIT_HEAD=<commit> bash tools/profile_round3.sh [cfg3|cfg4] [modes]
# Every GPU step has its own time limit; the script stops at the first failure.
set -e
wl=${1:-cfg3}
modes=${2:-"fast strict"}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RT_PROFILE_ROUND=r03
out=gpurun_out/r03_$wl
mkdir -p $out
spp=4; [ "$wl" = cfg4 ] && spp=2
common="--workload $wl --steps 1 --warmup 0 --spp $spp --cpu-seconds 0 --no-extra"
for mode in $modes; do
  timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE GRBM_COUNT \
      -d $out/sq_$mode --output-format csv -- python3 bench.py $common --mode $mode > $out/sq_$mode.log 2>$out/sq_$mode.err
  python3 tools/pmc_to_profiles.py valu $out/sq_$mode $out/sq_$mode.log $wl $mode > $out/valu_$mode.txt
  timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d $out/fetch_$mode --output-format csv -- python3 bench.py $common --mode $mode > $out/fetch_$mode.log 2>$out/fetch_$mode.err
  timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE -d $out/write_$mode --output-format csv -- python3 bench.py $common --mode $mode > $out/write_$mode.log 2>$out/write_$mode.err
  python3 tools/pmc_to_profiles.py traffic $out/fetch_$mode $out/write_$mode $out/fetch_$mode.log $wl $mode > $out/traffic_$mode.txt
  timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM_WR TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum \
      -d $out/mem1_$mode --output-format csv -- python3 bench.py $common --mode $mode > $out/mem1_$mode.log 2>&1
  timeout -s KILL 240 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
      -d $out/mem2_$mode --output-format csv -- python3 bench.py $common --mode $mode > $out/mem2_$mode.log 2>&1
done
python3 tools/pmc_summary.py $out k_ > $out/pmc_all_kernels.txt
cp profiles/r03_valu_per_ray.json profiles/r03_traffic.json $out/
if [ "$wl" = cfg3 ]; then
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 bench.py --cpu-seconds 0 --no-extra > $out/bench_profiled.json 2> $out/bench_profiled.err
  timeout -k 10 400 python3 bench.py > $out/bench.json 2> $out/bench.err
fi
tail -c 1500 $out/bench.json 2>/dev/null || true
