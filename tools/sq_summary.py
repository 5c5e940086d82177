#!/usr/bin/env python3
"""Summarise a tools/run_sq.sh output directory: counters of the step kernel per proposal.
Usage: sq_summary.py gpurun_out/sq_<tag> [out.json] [note]"""
import csv, glob, json, sys, collections
d = sys.argv[1]
per = {}
for sub in sorted(glob.glob(d + "/set*/")):
    f = glob.glob(sub + "*/*_counter_collection.csv")
    if not f:
        continue
    bench = json.load(open(sub.rstrip("/") + ".json"))
    nprop = bench["config"]["chains_per_gpu"] * bench["config"]["proposals_per_step"]
    acc = collections.defaultdict(list)
    byid = collections.defaultdict(dict)
    for r in csv.DictReader(open(f[0])):
        if "fcm_step_" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        # a launch shows one row per dimension instance; rocprofv3 csv has one row per (dispatch, counter)
        per[k] = sum(v) / len(v) / nprop
out = {"note": sys.argv[3] if len(sys.argv) > 3 else "", "per_proposal": per}
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
