#!/bin/bash
# GPU box: throughput against waves per chain.  usage: mw_sweep.sh <config> <chains> [W list]
CFG=${1:-2}; CH=${2:-256}; shift; shift
mkdir -p gpurun_out
for W in ${@:-1 2 4 8 16}; do
  FCM_MW=$W timeout -k 10 200 python bench.py --no-cpu-baseline --config $CFG --chains $CH --steps 2 --warmup 1 --proposals ${PROPS:-8192} > gpurun_out/sw_$W.json 2> gpurun_out/sw_$W.err || { echo "W=$W failed"; tail -3 gpurun_out/sw_$W.err; continue; }
  python -c "
import json,sys;d=json.load(open(sys.argv[1]));print('config',sys.argv[2],'chains',sys.argv[3],'W',d['roofline']['waves_per_chain'],'%.4g prop/s'%d['value'],'%.3f ms'%d['kernel_ms_per_launch'], {k:int(v) for k,v in d['rare_paths_per_1e6'].items()})" gpurun_out/sw_$W.json $CFG $CH
done
