#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ablate
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for D in 0 1 2 3 4 7; do
  FCM_DBG=$D timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $OUT/d$D -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/d$D.json 2> $OUT/d$D.err
done
echo done
