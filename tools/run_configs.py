#!/usr/bin/env python3
"""Runs the BASELINE.json configs that fit one GPU (with the stated per-GPU chain counts) and
prints one JSON line per config: setup time, proposals/s, invariants.  GPU box only."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import flag_complex_mcmc_amd as fcm
from bench import algorithmic_bytes


def strip(v):
    v = list(v)
    while v and v[-1] == 0:
        v.pop()
    return v


def run(name, n, edges, chains, proposals, launches, dim_cap=0):
    t0 = time.perf_counter()
    g = fcm.Graph.from_edges(n, edges)
    fc = g.flagser_count()
    b, ncl = fcm.Bounds.calculate(g, fc, fcm.Bounds.target(fc, 0.01), return_ncliques=True)
    s = fcm.MCMCSampler(g, b, n_chains=chains, seed=0, dim_cap=dim_cap)
    t_setup = time.perf_counter() - t0
    s.step(min(proposals, 64))  # warm-up
    st0 = s.stats()
    t1 = time.perf_counter()
    ms = []
    for _ in range(launches):
        s.step(proposals, sync=False)
    s.sync()
    dt = time.perf_counter() - t1
    st1 = s.stats()
    d = {k: int((st1[k].astype(np.int64) - st0[k].astype(np.int64)).sum()) for k in ("sampled", "accepted", "n_empty", "n_flip", "n_dmove", "sum_k", "n_cperm", "n_cswap", "n_changes")}
    ok = True
    for c in (0, chains - 1):
        full = s.graph(c).flagser_count()
        got = s.flag_count(c)            # never shrinks in length: compare zero-padded, on the tracked dimensions
        nc = s.ncounts
        pad = lambda v: (list(v) + [0] * nc)[:nc]
        ok &= (full == strip(got)) if s.info["lossless"] else (pad(full) == pad(got))
        if s.info["lossless"]:   # (the full-length bounds are not meaningful against a truncated count vector)
            ok &= b.check(s.flag_count(c))
    ab = algorithmic_bytes(d, n)
    print(json.dumps({"config": name, "n": n, "m": int(len(edges)), "chains": chains, "proposals_per_launch": proposals,
                      "launches": launches, "setup_s": round(t_setup, 2), "flag_count": fc, "ncliques_len": len(ncl),
                      "ncounts": s.ncounts, "lossless": s.info["lossless"], "k_max": s.info["k_max"], "k_mean": round(s.info["k_mean"], 2),
                      "proposals_per_s": d["sampled"] / dt, "accept_ratio": d["accepted"] / d["sampled"],
                      "empty_fraction": d["n_empty"] / d["sampled"], "algorithmic_GBps": ab / dt / 1e9,
                      "recount_matches": bool(ok), "GB_state": round(s.info["bytes_per_chain"] * chains / 1e9, 2)}), flush=True)


which = sys.argv[1:] or ["2", "3", "4", "5"]
if "2" in which:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import load_flag_fixture
    n, e = load_flag_fixture("bug_calc_relax_de.flag")
    run("configs[1]: C. elegans stand-in (bug_calc_relax_de.flag), 1024 chains, lossless (cap 5 would truncate: omega=8)", n, e, 1024, 4096, 4)
    run("configs[1] truncated: same graph, dim cap 5 as BASELINE.json states", n, e, 1024, 4096, 4, dim_cap=5)
if "3" in which:
    run("configs[2]: ER n=1000 p=0.10, 4096 chains", 1000, fcm.graphs.random_with_p(1000, 0.10, 0), 4096, 1024, 4)
if "4" in which:
    run("configs[3]: ER n=4000 p=0.05, 1024 chains per GPU (8192 over 8)", 4000, fcm.graphs.random_with_p(4000, 0.05, 0), 1024, 1024, 4)
if "5" in which:
    run("configs[4]: n=30000, 1M directed edge draws, 256 chains per GPU (2048 over 8)", 30000, fcm.graphs.random_edge_draws(30000, 1000000, 0), 256, 4096, 4)
