#!/usr/bin/env python3
"""fcm_count_kernel (flagser_count) on the BASELINE graphs: simplices per second, the bytes it has to read and the time
the whole call takes (host bitmap -> device -> counts).  GPU box."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flag_complex_mcmc_amd as fcm
from bench import build_workload
out = []
for cfg in (2, 3, 4):
    n, e = build_workload(fcm, cfg, 1000, 0.10, 0)
    g = fcm.Graph.from_edges(n, e)
    g.flagser_count()   # warm-up (module load, first allocation)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        fc = g.flagser_count()
    dt = (time.perf_counter() - t0) / reps
    W = 8 * ((n + 63) // 64)
    # one wave per directed edge u->v: rows u and v (intersection), then one dword per lane of every row of out(u) & out(v)
    simplices = sum(fc[2:])
    out.append({"config": cfg, "n": n, "m": int(len(e)), "flag_count": fc, "call_ms": dt * 1e3, "simplices_dim2plus": simplices,
                "simplices_per_s": simplices / dt, "row_bytes": W, "min_bytes_rows_u_v": 2 * W * len(e), "GBps_rows_u_v": 2 * W * len(e) / dt / 1e9})
print(json.dumps(out))
