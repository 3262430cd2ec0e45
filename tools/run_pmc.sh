#!/bin/bash
# Counter passes over one short frame of the benchmark (run on the GPU box):
#   bash tools/run_pmc.sh OUTDIR [extra bench.py args]
# One rocprofv3 run per pass; the profiled program comes right after `--`.
set -e
out=${1:-gpurun_out/pmc}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
pass() {
  n=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" -d "$out/p$n" --output-format csv -- \
    python3 bench.py --steps 1 --warmup 0 --spp 1 --cpu-seconds 0 $EXTRA > "$out/p$n.log" 2>&1
}
EXTRA="$*"
pass 1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE GRBM_COUNT
pass 2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM_WR TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum
pass 3 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
python3 tools/pmc_summary.py "$out" k_trace
