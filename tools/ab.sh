#!/bin/bash
# GPU box: parity tests, then the three bench points used for A/B comparisons (config 3 simple, default mix, 256 chains).
set -o pipefail
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1; tail -3 gpurun_out/t.log
show() { python -c "
import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[1],'%.4g'%d['value'],'%.3f ms'%d['kernel_ms_per_launch'],'frac %.3f'%d['roofline']['frac'])" $1; }
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/b_simple.json 2>gpurun_out/b_simple.err && show gpurun_out/b_simple.json || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --moves default --proposals 128 --steps 5 > gpurun_out/b_default.json 2>gpurun_out/b_default.err && show gpurun_out/b_default.json || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --chains 256 --steps 5 > gpurun_out/b_256.json 2>gpurun_out/b_256.err && show gpurun_out/b_256.json || exit 1
