#!/usr/bin/env python3
"""Soak for the multi-wave kernel: the same chains (same seed) run with W = 1 (one-wave kernel) and with every other W
must end in the same state -- counts, sampled/accepted and the other chain-level counters, double-slot lists and the
whole bitmap of a few chains -- after long runs on the BASELINE graphs.  usage: mw_soak.py [config ...]   (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import flag_complex_mcmc_amd as fcm
from bench import build_workload

PLAN = {1: (256, 1 << 16), 2: (256, 1 << 16), 3: (128, 1 << 15), 4: (64, 1 << 16)}   # config: (chains, proposals per chain)
KEYS = ("sampled", "accepted", "n_empty", "n_flip", "n_dmove", "sum_k", "count_len", "status")
DIAG = ("n_wide", "n_big")   # diagnostics: printed, not compared (the kernels tally them at different points)


def run(cfg, W, g, chains, props):
    os.environ["FCM_MW"] = str(W)
    s = fcm.initialize_new_sampler(g, n_chains=chains, seed=1234)
    assert s.info["waves_per_chain"] == W, (s.info["waves_per_chain"], W)
    for chunk in (1, 63, 1000, props - 1064):
        s.step(chunk)
    st = s.stats()
    return (s.flag_counts(), {k: st[k].copy() for k in KEYS + DIAG}, [s.edgebits(c) for c in (0, chains // 2, chains - 1)],
            [s.double_slots(c) for c in (0, chains - 1)], int(st["n_redo"].sum()), int(st["n_recheck"].sum()), int(st["n_held"].sum()))


def soak(cfg, chains, props, say=print):
    n, e = build_workload(fcm, cfg, 1000, 0.10, 0)
    g = fcm.Graph.from_edges(n, e)
    t0 = time.time()
    old = os.environ.get("FCM_MW")
    try:
        ref = run(cfg, 1, g, chains, props)
        assert (ref[1]["status"] == 0).all()
        for W in (2, 4, 8, 16):
            got = run(cfg, W, g, chains, props)
            assert (got[0] == ref[0]).all(), "counts differ (config %d, W %d)" % (cfg, W)
            for k in KEYS:
                assert (got[1][k] == ref[1][k]).all(), "%s differs (config %d, W %d)" % (k, cfg, W)
            for a, b in zip(got[2], ref[2]):
                assert np.array_equal(np.asarray(a), np.asarray(b)), "bitmap differs (config %d, W %d)" % (cfg, W)
            for a, b in zip(got[3], ref[3]):
                assert np.array_equal(np.asarray(a), np.asarray(b)), "slot list differs (config %d, W %d)" % (cfg, W)
            say("config %d W %2d: %d chains x %d proposals identical to W = 1 (re-run %d, re-checked %d, held %d; wide %d/%d, big %d/%d)" % (cfg, W, chains, props, got[4], got[5], got[6], got[1]["n_wide"].sum(), ref[1]["n_wide"].sum(), got[1]["n_big"].sum(), ref[1]["n_big"].sum()))
    finally:
        if old is None:
            os.environ.pop("FCM_MW", None)
        else:
            os.environ["FCM_MW"] = old
    say("config %d done in %.1f s" % (cfg, time.time() - t0))


if __name__ == "__main__":
    for cfg in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]:
        soak(cfg, *PLAN[cfg], say=lambda m: print(m, flush=True))
