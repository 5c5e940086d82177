#!/bin/bash
# GPU box: instruction-cache counters of the step kernel.  usage: icache.sh <bench.py args>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/icache
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/set1 -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --proposals 4096 "$@" > $OUT/set1.json 2> $OUT/set1.err || echo failed
python3 - <<PY
import csv, glob, json, collections
b = json.load(open("$OUT/set1.json"))
nprop = b["config"]["chains_per_gpu"] * b["config"]["proposals_per_step"]
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/set1/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "fcm_step_" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: round(sum(v) / len(v) / nprop, 2) for k, v in sorted(acc.items())}, "%.4g prop/s" % b["value"])
PY
