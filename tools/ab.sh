#!/bin/bash
# GPU box: parity tests, then the bench points used for A/B comparisons (config 2 simple, default mix, 256 chains, configs 3 and 4).
set -o pipefail
mkdir -p gpurun_out
if [ "${SKIP_TESTS:-0}" != "1" ]; then timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1; tail -3 gpurun_out/t.log; fi
show() { python -c "
import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[1],'%.4g'%d['value'],'%.3f ms'%d['kernel_ms_per_launch'],'frac %.3f'%d['roofline']['frac'], d['rare_paths_per_1e6'])" $1; }
timeout -k 10 200 python bench.py --no-cpu-baseline --steps ${STEPS:-4} > gpurun_out/b_simple.json 2>gpurun_out/b_simple.err && show gpurun_out/b_simple.json || exit 1
[ "${QUICK:-0}" = "1" ] && exit 0
timeout -k 10 200 python bench.py --no-cpu-baseline --moves default --steps 4 > gpurun_out/b_default.json 2>gpurun_out/b_default.err && show gpurun_out/b_default.json || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --chains 256 --steps 4 > gpurun_out/b_256.json 2>gpurun_out/b_256.err && show gpurun_out/b_256.json || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --config 3 --steps 4 > gpurun_out/b_c3.json 2>gpurun_out/b_c3.err && show gpurun_out/b_c3.json || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --config 4 --steps 4 > gpurun_out/b_c4.json 2>gpurun_out/b_c4.err && show gpurun_out/b_c4.json || exit 1
