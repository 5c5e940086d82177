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
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $ROOT/tools/pmc_calib $ROOT/tools/pmc_calib.hip
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/calib_$C -- $ROOT/tools/pmc_calib 4096 4096 > $OUT/calib_$C.log 2>&1 || echo "calib $C failed"
  # the same for rows longer than a cache line, read as scattered lines (n = 4000: 512-B rows, 40 positions; n = 30000: 3840-B rows, 40 and 3 positions)
  for PAT in "512 40" "3840 40" "3840 3"; do
    set -- $PAT
    timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/calib$1x$2_$C -- $ROOT/tools/pmc_calib 4096 2048 $1 $2 > $OUT/calib$1x$2_$C.log 2>&1 || echo "calib $1 $2 $C failed"
  done
  for K in $CONFIGS; do
    timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/c${K}_$C -- python3 $ROOT/bench.py --config $K --steps 2 --warmup 1 --no-cpu-baseline --proposals 16384 > $OUT/c${K}_$C.json 2> $OUT/c${K}_$C.err || echo "config $K $C failed"
  done
  # the default move mix (clique moves) on the headline config, at its stated 4096 chains (one-wave kernel) and at 1024 (cooperative kernel, W = 4)
  timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/cd2_$C -- python3 $ROOT/bench.py --config 2 --moves default --steps 2 --warmup 1 --no-cpu-baseline > $OUT/cd2_$C.json 2> $OUT/cd2_$C.err || echo "default mix $C failed"
  timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/ce2_$C -- python3 $ROOT/bench.py --config 2 --moves default --chains 1024 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/ce2_$C.json 2> $OUT/ce2_$C.err || echo "default mix 1024 chains $C failed"
done
echo done
