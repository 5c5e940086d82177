#!/bin/bash
# GPU box, from the repo root: bash tools/collect_profiles.sh <tag>
# Produces under gpurun_out/prof_<tag>/ everything profiles/ keeps for one kernel build:
# the full bench line, the rocprofv3 --kernel-trace --stats summary of the same command, the
# FETCH_SIZE / WRITE_SIZE passes (+ calibration), the SQ counter passes, the default-mix bench
# line and the per-config lines.  Copy what is to be judged into profiles/ afterwards.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
[ -x $ROOT/tools/pmc_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $ROOT/tools/pmc_calib $ROOT/tools/pmc_calib.hip
cd $ROOT
echo "== bench (full)"; timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
echo "== bench default mix"; timeout -k 10 300 python bench.py --no-cpu-baseline --moves default --proposals 128 --steps 5 > $OUT/bench_default_mix.json 2> $OUT/bench_default_mix.err || exit 1
cd /tmp && export TMPDIR=/tmp
echo "== rocprofv3 stats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/stats_bench.json 2> $OUT/stats.err || exit 1
cd $ROOT
echo "== pmc"; bash tools/run_pmc.sh $TAG > $OUT/pmc.log 2>&1 && python tools/pmc_summary.py gpurun_out/pmc_$TAG $OUT/pmc_summary.json > /dev/null || exit 1
echo "== sq"; bash tools/run_sq.sh $TAG > $OUT/sq.log 2>&1 && python tools/sq_summary.py gpurun_out/sq_$TAG $OUT/sq_counters.json "rocprofv3 --pmc SQ counter passes of python3 bench.py --steps 2 --warmup 1 (kernel $TAG); per proposal (4096 chains x 1024 proposals per launch); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles" > /dev/null || exit 1
echo "== configs"; timeout -k 10 500 python tools/run_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err || exit 1
echo "== done"; ls $OUT
