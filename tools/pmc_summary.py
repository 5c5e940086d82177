#!/usr/bin/env python3
"""Summarise a tools/run_pmc.sh output directory: per-launch FETCH_SIZE / WRITE_SIZE of the step kernel for every
config that was run, corrected with the calibration kernel of the same access pattern (MI355X_MICROARCH.md "HBM":
FETCH_SIZE counts 128-B requests at 64 B on gfx950; other widths must be calibrated).
Usage: pmc_summary.py gpurun_out/pmc_<tag> <tag> [out.json]   -> a list, one record per config (bench.py reads it)"""
import csv, glob, json, os, sys

d, tag = sys.argv[1], sys.argv[2]


def counters(sub, kernel_sub, counter):
    fs = glob.glob("%s/%s/*/*_counter_collection.csv" % (d, sub))
    if not fs:
        return []
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(fs[0])) if kernel_sub in r["Kernel_Name"] and r["Counter_Name"] == counter]


calib_read_bytes = 4096 * 4096 * 128            # tools/pmc_calib.hip: waves * R * 128 B
calib_write_stores = 4096 * (4096 // 16)         # one dword store per 16 rows in write mode
cf = counters("calib_FETCH_SIZE", "calib_kernel", "FETCH_SIZE")
cw = counters("calib_WRITE_SIZE", "calib_kernel", "WRITE_SIZE")
fetch_corr = calib_read_bytes / (sum(cf) / len(cf) * 1024.0)
write_bytes_per_store = (sum(cw[3:]) / len(cw[3:]) * 1024.0) / calib_write_stores


def long_row_corr(pat):
    """FETCH_SIZE correction for rows longer than a line read as scattered lines (pmc_calib <waves> <R> <row_bytes> <positions>), or None."""
    c = counters("calib%s_FETCH_SIZE" % pat, "calib_kernel", "FETCH_SIZE")
    try:
        known = json.loads(open("%s/calib%s_FETCH_SIZE.log" % (d, pat)).read().strip().split("\n")[-1])["read_bytes_per_launch"]
    except Exception:
        return None
    return known / (sum(c) / len(c) * 1024.0) if c else None


corr_long = {"512x40": long_row_corr("512x40"), "3840x40": long_row_corr("3840x40"), "3840x3": long_row_corr("3840x3")}
out = []
for key in ("1", "2", "3", "4", "d2", "e2"):       # d2 / e2: the default move mix on config 2 with 4096 / 1024 chains
    k, moves = int(key[-1]), ("default" if key[0] in "de" else "simple")
    sf, sw = counters("c%s_FETCH_SIZE" % key, "fcm_step_", "FETCH_SIZE"), counters("c%s_WRITE_SIZE" % key, "fcm_step_", "WRITE_SIZE")
    if not sf or not sw or not os.path.exists("%s/c%s_FETCH_SIZE.json" % (d, key)):
        continue
    bench = json.load(open("%s/c%s_FETCH_SIZE.json" % (d, key)))
    # the first launches are warm-up; all launches run the same number of proposals
    rec = {"tag": tag, "config": k, "moves": moves, "n_chains": bench["config"]["chains_per_gpu"], "proposals": bench["config"]["proposals_per_step"],
           "kernel": bench["roofline"]["kernel"], "waves_per_chain": bench["roofline"]["waves_per_chain"], "lib_sha16": bench.get("lib_sha16"), "sparse_state": bool(bench["roofline"].get("sparse_state", False)), "n": bench["config"]["n"],
           "launches_seen": len(sf), "FETCH_SIZE_KB_per_launch": sum(sf) / len(sf), "WRITE_SIZE_KB_per_launch": sum(sw) / len(sw),
           "fetch_correction": None, "fetch_correction_pattern": None, "fetch_corrections_measured": dict(corr_long, **{"128": fetch_corr}),
           "calib_write_bytes_per_dword_store": write_bytes_per_store,
           "fetch_bytes_per_launch": None, "write_bytes_per_launch": sum(sw) / len(sw) * 1024.0,
           "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
           "algorithmic_model": bench["roofline"]["algorithmic_model"],
           "survey_bytes_per_launch": bench["roofline"]["survey_bytes_per_launch"],
           "kernel_ms_per_launch_under_profiler": bench["kernel_ms_per_launch"], "proposals_per_s_under_profiler": bench["value"]}
    # the calibration of the access pattern the config's builds have: whole 128-B rows up to 1024 vertices; scattered lines of
    # 512-B rows (n = 4000, ~40 positions per row) or of 3840-B rows (n = 30000, ~3 positions per row) beyond
    n = bench["config"]["n"]
    pat = "128" if n <= 1024 else ("512x40" if n <= 8192 else "3840x3")
    corr = fetch_corr if pat == "128" else (corr_long.get(pat) or fetch_corr)
    rec["fetch_correction"], rec["fetch_correction_pattern"] = corr, pat if (pat == "128" or corr_long.get(pat)) else "128 (no long-row calibration in this run)"
    rec["fetch_bytes_per_launch"] = sum(sf) / len(sf) * 1024.0 * corr
    rec["hbm_bytes_per_launch"] = rec["fetch_bytes_per_launch"] + rec["write_bytes_per_launch"]
    rec["traffic_over_algorithmic"] = rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"]
    rec["counter_GBps"] = rec["hbm_bytes_per_launch"] / (rec["kernel_ms_per_launch_under_profiler"] * 1e-3) / 1e9
    out.append(rec)
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
