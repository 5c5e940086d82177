#!/bin/bash
# GPU box: headline bench with the product library and with the ablation builds (tools/abl_build.sh).  usage: bash tools/ablate.sh base 1 3 7 ...
mkdir -p gpurun_out
for v in "$@"; do
  L=""; [ "$v" != base ] && L=tools/_stamp/abl$v/libfcm.so
  P=""; [ "$v" != base ] && P=1
  r=$(FCM_LIB_PATH=$L FCM_BENCH_PROBE=$P timeout -k 10 200 python bench.py --no-cpu-baseline --steps 4 --warmup 1 2>gpurun_out/abl_$v.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('%.4g %.1f ms accept %.3f empty %.3f' % (d['value'], d['kernel_ms_per_launch'], d['accept_ratio'], d['empty_fraction']))")
  echo "abl $v: $r" | tee -a gpurun_out/abl.txt
done
