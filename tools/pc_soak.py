#!/usr/bin/env python3
"""Soak: the two-wave kernel against the one-wave kernel, same seeds, many proposals, several graphs
(every chain's counts, statistics, edges and slot list must agree exactly).  GPU box only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import flag_complex_mcmc_amd as fcm
from flag_complex_mcmc_amd import graphs
bad = 0
for (n, pr, gseed, chains, launches, props, relax) in [(1000, 0.10, 0, 4096, 8, 1024, 0.01), (1000, 0.10, 0, 256, 8, 2048, 0.01), (600, 0.12, 3, 128, 4, 2048, 0.01),
                                                         (1000, 0.13, 5, 64, 4, 1024, 0.02), (300, 0.20, 6, 64, 4, 2048, 0.02), (1000, 0.10, 1, 64, 2, 4096, 0.0005),
                                                         (100, 0.25, 9, 64, 4, 4096, 0.05), (4000, 0.05, 0, 64, 2, 2048, 0.01), (2000, 0.06, 2, 64, 2, 2048, 0.01)]:
    e = graphs.random_with_p(n, pr, seed=gseed)
    g = fcm.Graph.from_edges(n, e)
    fc = g.flagser_count()
    b = fcm.Bounds.calculate(g, fc, fcm.Bounds.target(fc, relax))
    os.environ["FCM_PC"] = "2"
    s2 = fcm.MCMCSampler(g, b, n_chains=chains, seed=77)
    os.environ["FCM_PC"] = "0"
    s1 = fcm.MCMCSampler(g, b, n_chains=chains, seed=77)
    if not (s2.info["two_wave"] == 1 and s1.info["two_wave"] == 0):
        print("n=%d: two-wave kernel not applicable (%d count entries)" % (n, s2.ncounts)); continue
    t0 = time.time()
    ok = True
    for L in range(launches):
        s1.step(props); s2.step(props)
        st1, st2 = s1.stats(), s2.stats()
        for k in ("sampled", "accepted", "n_empty", "n_flip", "n_dmove", "sum_k", "status"):
            if not (st1[k] == st2[k]).all():
                ok = False; print("n=%d launch %d: %s differs on chains %s" % (n, L, k, np.nonzero(st1[k] != st2[k])[0][:8]))
        for c in range(chains):
            if s1.flag_count(c) != s2.flag_count(c):
                ok = False; print("n=%d launch %d: counts differ on chain %d" % (n, L, c)); break
        if not ok: break
    for c in range(0, chains, max(1, chains // 8)):
        if not ((s1.edges(c) == s2.edges(c)).all() and (s1.double_slots(c) == s2.double_slots(c)).all()):
            ok = False; print("n=%d: final graph differs on chain %d" % (n, c))
    st = s2.stats()
    print("n=%d p=%.2f relax %.4f: %d chains x %d proposals, accept %.4f, dmoves %d, k_mean %.1f k_max %d: %s (%.1f s)" % (
        n, pr, relax, chains, launches * props, st["accepted"].sum() / st["sampled"].sum(), st["n_dmove"].sum(), s2.info["k_mean"], s2.info["k_max"],
        "IDENTICAL" if ok else "MISMATCH", time.time() - t0))
    bad += 0 if ok else 1
print("SOAK", "OK" if bad == 0 else "FAILED")
