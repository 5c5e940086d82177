#!/bin/bash
# GPU box: quick check of the multi-wave kernel: its parity tests, then bench points (config 2 at 4096 and 256 chains, configs 3 and 4)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multi_wave or bench_kernel" > gpurun_out/t.log 2>&1; tail -3 gpurun_out/t.log
show() { python -c "
import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[1],'%.4g'%d['value'],'%.3f ms'%d['kernel_ms_per_launch'],'frac %.3f'%d['roofline']['frac'], 'W', d['roofline']['waves_per_chain'], 'acc %.2f'%d['accept_ratio'], {k:int(v) for k,v in d['rare_paths_per_1e6'].items()})" $1; }
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/b_simple.json 2>gpurun_out/b_simple.err && show gpurun_out/b_simple.json || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --chains 256 --steps 3 --warmup 1 > gpurun_out/b_256.json 2>gpurun_out/b_256.err && show gpurun_out/b_256.json || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --config 3 --steps 3 --warmup 1 > gpurun_out/b_c3.json 2>gpurun_out/b_c3.err && show gpurun_out/b_c3.json || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --config 4 --steps 3 --warmup 1 > gpurun_out/b_c4.json 2>gpurun_out/b_c4.err && show gpurun_out/b_c4.json || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --config 1 --steps 3 --warmup 1 > gpurun_out/b_c1.json 2>gpurun_out/b_c1.err && show gpurun_out/b_c1.json || exit 1
