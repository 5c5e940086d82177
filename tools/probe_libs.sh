#!/bin/bash
# GPU box: SQ counters of the step kernel for several library builds.  usage: probe_libs.sh "<lib names under tools/_stamp, or base>" [bench args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
LIBS=$1; shift
OUT=$ROOT/gpurun_out/probe
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for V in $LIBS; do
  [ $V = base ] && unset FCM_LIB_PATH || export FCM_LIB_PATH=$ROOT/tools/_stamp/$V/libfcm.so
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/$V -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --proposals 4096 "$@" > $OUT/$V.json 2> $OUT/$V.err || echo "$V failed"
done
python3 - <<PY
import csv, glob, json, collections
for V in "$LIBS".split():
    f = glob.glob("$OUT/%s/*/*_counter_collection.csv" % V)
    if not f: continue
    b = json.load(open("$OUT/%s.json" % V))
    nprop = b["config"]["chains_per_gpu"] * b["config"]["proposals_per_step"]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "fcm_step_" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(V, {k.replace("SQ_", ""): round(sum(v) / len(v) / nprop, 1) for k, v in sorted(acc.items())}, "%.4g prop/s" % b["value"])
PY
