#!/bin/bash
# GPU box: SQ instruction counters per proposal of the headline kernel for the product library and the ablation builds.
# usage: bash tools/sq_abl.sh base 1 3 63 ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sqabl
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf $OUT/$v
  if [ "$v" = base ]; then unset FCM_LIB_PATH FCM_BENCH_PROBE; else export FCM_LIB_PATH=$ROOT/tools/_stamp/abl$v/libfcm.so FCM_BENCH_PROBE=1; fi
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $OUT/$v -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --proposals 4096 > $OUT/$v.json 2> $OUT/$v.err || echo "$v failed"
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/${v}b -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --proposals 4096 > $OUT/${v}b.json 2> $OUT/${v}b.err || echo "$v b failed"
done
python3 - "$@" <<PY
import csv, glob, json, collections, sys
for V in sys.argv[1:]:
    per = {}
    for suf in ("", "b"):
        f = glob.glob("$OUT/%s%s/*/*_counter_collection.csv" % (V, suf))
        if not f: continue
        b = json.load(open("$OUT/%s%s.json" % (V, suf)))
        nprop = b["config"]["chains_per_gpu"] * b["config"]["proposals_per_step"]
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if "fcm_step_" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items(): per[k.replace("SQ_", "")] = round(sum(v) / len(v) / nprop, 1)
        per["prop/s" + suf] = "%.4g" % b["value"]
    print("abl", V, json.dumps(per))
PY
