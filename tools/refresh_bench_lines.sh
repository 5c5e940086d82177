set -u
OUT=gpurun_out/prof_r02b; mkdir -p $OUT
export MW_STAMP_DIR=$GRAFT_REPO_ROOT/tools/_stamp/stampb
( bash tools/mw_stamps.sh 2 4096; bash tools/mw_stamps.sh 2 256; bash tools/mw_stamps.sh 3 1024; bash tools/mw_stamps.sh 4 256 ) > $OUT/mw_stamps.txt 2>&1
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
for K in 1 3 4; do timeout -k 10 300 python bench.py --no-cpu-baseline --config $K --steps 8 > $OUT/bench_config$K.json 2> $OUT/bench_config$K.err || exit 1; done
timeout -k 10 300 python bench.py --no-cpu-baseline --chains 256 --steps 8 > $OUT/bench_config2_256chains.json 2> $OUT/b256.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --config 3 --chains 256 --steps 8 > $OUT/bench_config3_256chains.json 2> $OUT/b3256.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --moves default --steps 8 > $OUT/bench_default_mix.json 2> $OUT/bdef.err || exit 1
echo done
