#!/bin/bash
# SQ counter passes for the step kernel.  bash tools/run_sq.sh <tag>
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_I8" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/set$i -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/set$i.json 2> $OUT/set$i.err || echo "set $i failed"
done
echo done
