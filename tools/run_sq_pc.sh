#!/bin/bash
# instruction-cache and issue counters for the step kernel, one-wave vs two-wave (flips only), GPU box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sq_pc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pc in 0 1; do
  i=0
  for SET in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
    i=$((i+1))
    FCM_PC=$pc timeout -k 10 120 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pc${pc}_set$i -- python3 $ROOT/tools/pc_check.py bench ${W0:-1} ${W1:-0} > $OUT/pc${pc}_set$i.log 2>&1 || echo "failed pc$pc set $i"
  done
done
python3 - <<PY
import csv, glob, collections
for pc in (0, 1):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/pc%d_set*/*/*_counter_collection.csv" % pc):
        for r in csv.DictReader(open(f)):
            if "fcm_step" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    n = 4096 * 1024
    print("pc=%d" % pc, {k: round(sum(v) / len(v) / n, 2) for k, v in sorted(acc.items())})
PY
