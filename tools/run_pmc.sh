#!/bin/bash
# HBM traffic counters of the step kernel for every single-GPU share of the BASELINE configs (separate --pmc passes, as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass) plus the calibration kernel in the same
# access pattern.  GPU box, from the repo root: bash tools/run_pmc.sh <tag> [configs...]; then tools/pmc_summary.py.
set -u
TAG=${1:-r02}; shift
CONFIGS=${@:-1 2 3 4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
[ -x $ROOT/tools/pmc_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $ROOT/tools/pmc_calib $ROOT/tools/pmc_calib.hip
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/calib_$C -- $ROOT/tools/pmc_calib 4096 4096 > $OUT/calib_$C.log 2>&1 || echo "calib $C failed"
  for K in $CONFIGS; do
    timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/c${K}_$C -- python3 $ROOT/bench.py --config $K --steps 2 --warmup 1 --no-cpu-baseline > $OUT/c${K}_$C.json 2> $OUT/c${K}_$C.err || echo "config $K $C failed"
  done
  # the default move mix (clique moves, one-wave kernel) on the headline config
  timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/cd2_$C -- python3 $ROOT/bench.py --config 2 --moves default --steps 2 --warmup 1 --no-cpu-baseline > $OUT/cd2_$C.json 2> $OUT/cd2_$C.err || echo "default mix $C failed"
done
echo done
