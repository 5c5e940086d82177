#!/bin/bash
# GPU box: one short bench line per share that DESIGN 4.3's table quotes (no CPU baseline).  usage: bash tools/quick_configs.sh [tag]
T=${1:-qc}; mkdir -p gpurun_out
run() { name=$1; shift; r=$(timeout -k 10 240 python bench.py --no-cpu-baseline --steps 4 --warmup 1 "$@" 2>gpurun_out/${T}_$name.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('%.4g %.1f ms W=%s' % (d['value'], d['kernel_ms_per_launch'], d['config'].get('waves_per_chain')))"); echo "$T $name: $r" | tee -a gpurun_out/${T}.txt; }
run c2 --config 2
run c2_256 --config 2 --chains 256
run c1 --config 1
run c3 --config 3
run c4 --config 4
run c4_1024 --config 4 --chains 1024
run c4_4096 --config 4 --chains 4096
run c2_def --config 2 --moves default
run c2_def2048 --config 2 --moves default --chains 2048
run c2_def1024 --config 2 --moves default --chains 1024
run c1_def --config 1 --moves default
