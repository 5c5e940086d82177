#!/bin/bash
# GPU box: the two SQ counter passes that matter (instruction mix, waits) for the bench kernel, per proposal.
# usage: sq_quick.sh [bench.py args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sqq
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/set$i -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --proposals 4096 "$@" > $OUT/set$i.json 2> $OUT/set$i.err || echo "set $i failed"
done
python3 - <<PY
import csv, glob, json, collections
per = {}
for sub in sorted(glob.glob("$OUT/set*/")):
    f = glob.glob(sub + "*/*_counter_collection.csv")
    if not f: continue
    b = json.load(open(sub.rstrip("/") + ".json"))
    nprop = b["config"]["chains_per_gpu"] * b["config"]["proposals_per_step"]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "fcm_step_" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items(): per[k] = round(sum(v) / len(v) / nprop, 1)
print(json.dumps(per))
PY
