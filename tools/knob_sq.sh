#!/bin/bash
# GPU box: vector / scalar instructions per proposal of the headline kernel for knob builds (tools/knob_build.sh).  usage: bash tools/knob_sq.sh base name1 ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/knobsq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf $OUT/$v
  if [ "$v" = base ]; then unset FCM_LIB_PATH; else export FCM_LIB_PATH=$ROOT/tools/_stamp/knob_$v/libfcm.so; fi
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/$v -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --proposals 4096 > $OUT/$v.json 2> $OUT/$v.err || echo "$v failed"
done
python3 - "$@" <<PY
import csv, glob, json, collections, sys
for V in sys.argv[1:]:
    f = glob.glob("$OUT/%s/*/*_counter_collection.csv" % V)
    if not f: print("knobsq", V, "no counters"); continue
    b = json.load(open("$OUT/%s.json" % V))
    nprop = b["config"]["chains_per_gpu"] * b["config"]["proposals_per_step"]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "fcm_step_" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    per = {k.replace("SQ_", ""): round(sum(v) / len(v) / nprop, 1) for k, v in acc.items()}
    print("knobsq", V, json.dumps(per))
PY
