#!/bin/bash
# GPU box, after profiles/pmc_summary.json of the current build is in place: the bench lines again (they replay the traffic
# of that summary), into gpurun_out/prof_<tag>/.  usage: bash tools/refresh_bench_lines.sh <tag>
set -u
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
for K in 1 3 4; do timeout -k 10 300 python bench.py --no-cpu-baseline --config $K --steps 8 > $OUT/bench_config$K.json 2> $OUT/bench_config$K.err || exit 1; done
timeout -k 10 300 python bench.py --no-cpu-baseline --chains 256 --steps 8 > $OUT/bench_config2_256chains.json 2> $OUT/b256.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --moves default --steps 8 > $OUT/bench_default_mix.json 2> $OUT/bdef.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --moves default --chains 1024 --steps 8 > $OUT/bench_default_mix_1024chains.json 2> $OUT/bdef1024.err || exit 1
for CH in 1024 4096; do timeout -k 10 300 python bench.py --no-cpu-baseline --config 4 --chains $CH --steps 8 > $OUT/bench_config4_${CH}chains.json 2> $OUT/bench_config4_$CH.err || exit 1; done
FCM_SPARSE=0 timeout -k 10 300 python bench.py --no-cpu-baseline --config 4 --steps 8 > $OUT/bench_config4_rowbitmaps.json 2> $OUT/bench_config4_rb.err || exit 1
echo done
