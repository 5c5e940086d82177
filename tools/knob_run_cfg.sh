#!/bin/bash
# GPU box: one bench share for the product library and knob builds.  usage: bash tools/knob_run_cfg.sh "<bench args>" base name1 ...
ARGS=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  L=""; [ "$v" != base ] && L=$PWD/tools/_stamp/knob_$v/libfcm.so
  r=$(FCM_LIB_PATH=$L timeout -k 10 200 python bench.py --no-cpu-baseline --steps 6 --warmup 1 $ARGS 2>gpurun_out/knobc_$v.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('%.4g %.1f ms' % (d['value'], d['kernel_ms_per_launch']))")
  echo "knob [$ARGS] $v: $r" | tee -a gpurun_out/knobs_cfg.txt
done
