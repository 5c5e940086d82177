#!/bin/bash
# GPU box: the few-chain bench points for the product library and for variant libraries (tools/variant_lib.sh), side by side.
# usage: bash tools/ab_variants.sh <variant names...>      ("base" = the product library)
mkdir -p gpurun_out
for v in "$@"; do
  L=""; [ "$v" != base ] && L=tools/_stamp/$v/libfcm.so
  P=""; [ "$v" != base ] && P=1   # (the product library runs verified; only the variant builds run as probes)
  line="$v:"
  for spec in "--config 1" "--config 4" "--chains 256" "--config 3" "--config 3 --chains 256" ""; do
    r=$(FCM_LIB_PATH=$L FCM_BENCH_PROBE=$P timeout -k 10 200 python bench.py --no-cpu-baseline --steps 6 $spec 2>/dev/null | python -c "import json,sys; print('%.4g' % json.loads(sys.stdin.read().strip().split(chr(10))[-1])['value'])")
    line="$line  [${spec:-headline}] $r"
  done
  echo "$line" | tee -a gpurun_out/ab_variants.txt
done
