#!/usr/bin/env python3
"""Summarise a tools/run_pmc.sh output directory: per-launch FETCH_SIZE / WRITE_SIZE
of the step kernel, corrected with the calibration kernel of the same access
pattern (MI355X_MICROARCH.md "HBM": FETCH_SIZE counts 128-B requests at 64 B on
gfx950).  Usage: pmc_summary.py gpurun_out/pmc_<tag> [out.json]"""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]


def counters(sub, kernel_sub, counter):
    f = glob.glob("%s/%s/*/*_counter_collection.csv" % (d, sub))[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if kernel_sub in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return vals


calib_read_bytes = 4096 * 4096 * 128            # tools/pmc_calib.hip: waves * R * 128 B
calib_write_stores = 4096 * (4096 // 16)         # one dword store per 16 rows in write mode
cf = counters("calib_FETCH_SIZE", "calib_kernel", "FETCH_SIZE")
cw = counters("calib_WRITE_SIZE", "calib_kernel", "WRITE_SIZE")
fetch_corr = calib_read_bytes / (sum(cf) / len(cf) * 1024.0)
write_bytes_per_store = (sum(cw[3:]) / len(cw[3:]) * 1024.0) / calib_write_stores
sf = counters("bench_FETCH_SIZE", "fcm_step_", "FETCH_SIZE")
sw = counters("bench_WRITE_SIZE", "fcm_step_", "WRITE_SIZE")
bench = json.load(open("%s/bench_FETCH_SIZE.json" % d))
out = {
    "n_chains": bench["config"]["chains_per_gpu"], "proposals": bench["config"]["proposals_per_step"],
    "launches_seen": len(sf),
    "FETCH_SIZE_KB_per_launch": sum(sf) / len(sf), "WRITE_SIZE_KB_per_launch": sum(sw) / len(sw),
    "fetch_correction": fetch_corr,
    "calib_write_bytes_per_dword_store": write_bytes_per_store,
    "fetch_bytes_per_launch": sum(sf) / len(sf) * 1024.0 * fetch_corr,
    "write_bytes_per_launch": sum(sw) / len(sw) * 1024.0,
    "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
}
out["hbm_bytes_per_launch"] = out["fetch_bytes_per_launch"] + out["write_bytes_per_launch"]
out["traffic_over_algorithmic"] = out["hbm_bytes_per_launch"] / out["algorithmic_bytes_per_launch"]
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
