#!/bin/bash
# SQ instruction counters of the two-wave kernel for a diagnostic build of the library: bash tools/run_sq_alt.sh <lib.so> <w0> <w1>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sq_alt
rm -rf $OUT; mkdir -p $OUT
cp $1 $ROOT/flag_complex_mcmc_amd/libfcm.so
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  FCM_PC=1 timeout -k 10 120 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/set$i -- python3 $ROOT/tools/pc_proto.py bench $2 $3 > $OUT/set$i.log 2>&1 || echo "failed set $i"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/set*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "fcm_step" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
n = 4096 * 1024
print("$1 $2 $3", {k: round(sum(v) / len(v) / n, 1) for k, v in sorted(acc.items())})
PY
