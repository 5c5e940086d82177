#!/bin/bash
# Collects HBM traffic counters for the step kernel (separate --pmc passes, as
# MI355X_MICROARCH.md prescribes) plus the calibration kernel in the same access
# pattern.  Run on the GPU box from the repo root: bash tools/run_pmc.sh <tag>
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/bench_$C -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_$C.json 2> $OUT/bench_$C.err || echo "bench $C failed"
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/calib_$C -- $ROOT/tools/pmc_calib 4096 4096 > $OUT/calib_$C.log 2>&1 || echo "calib $C failed"
done
find $OUT -name "*.csv" | head -20
