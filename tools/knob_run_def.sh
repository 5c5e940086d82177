#!/bin/bash
# GPU box: default-mix bench (configs[2], 4096 chains) for the product library and knob builds.  usage: bash tools/knob_run_def.sh base name1 ...
mkdir -p gpurun_out
for v in "$@"; do
  L=""; [ "$v" != base ] && L=$PWD/tools/_stamp/knob_$v/libfcm.so
  r=$(FCM_LIB_PATH=$L timeout -k 10 200 python bench.py --no-cpu-baseline --moves default --steps 6 --warmup 1 2>gpurun_out/knobd_$v.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('%.4g %.1f ms' % (d['value'], d['kernel_ms_per_launch']))")
  echo "knob-def $v: $r" | tee -a gpurun_out/knobs_def.txt
done
