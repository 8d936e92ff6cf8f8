#!/bin/bash
# PMC passes for the headline bench (separate passes; --pmc never combined with --stats / sys traces).
# usage (on the GPU box, via gpurun): bash tools/pmc.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-pmc}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -o $name -- python $R/bench.py --steps 3 --warmup 1 --preroll-ms 60 --no-cpu-baseline > $OUT/$name.log 2>&1
}
run p1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD
run p2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run p3 FETCH_SIZE GRBM_GUI_ACTIVE
run p4 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
ls $OUT
