#!/bin/bash
# Fabric-traffic counters of the traversal kernel on the 10M-triangle
# configuration (run on the GPU box): one rocprofv3 run per counter group.
set -e
out=${1:-gpurun_out/pmc_cfg4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
pass() {
  n=$1; shift
  timeout -s KILL 400 rocprofv3 --pmc "$@" -d "$out/$n" --output-format csv -- \
    python3 bench.py --workload cfg4 --steps 1 --warmup 0 --spp 1 --cpu-seconds 0 > "$out/$n.log" 2>&1
  python3 tools/pmc_summary.py "$out/$n" k_trace
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
tail -c 600 "$out/tcc.log"
