#!/usr/bin/env python3
"""Check: the cooperative clique-move kernel (fcm_step_cq) with W waves per chain against the one-wave kernel (FCM_CQ=0),
launch sizes 1, 7, 100.  usage: cq_vs_onewave.py [W ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import flag_complex_mcmc_amd as fcm
from flag_complex_mcmc_amd import graphs

ok = True
for (n, pr, gseed, rel) in ((40, 0.35, 1, 0.2), (300, 0.12, 3, 0.02), (1000, 0.10, 0, 0.01)):
    e = graphs.random_with_p(n, pr, seed=gseed)
    g = fcm.Graph.from_edges(n, e)
    fc = g.flagser_count()
    if len(fc) > 8:
        continue
    b = fcm.Bounds.calculate(g, fc, fcm.Bounds.target(fc, rel))
    for weights in ((0.1, 0.1, 0.6, 0.2), (0, 0, 1, 0), (0, 0, 0, 1)):
        os.environ["FCM_CQ"] = "0"
        o = fcm.MCMCSampler(g, b, n_chains=6, seed=4, move_weights=weights)
        for nstep in (1, 7, 100):
            o.step(nstep)
        os.environ["FCM_CQ"] = "1"
        for W in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
            os.environ["FCM_CQW"] = str(W)
            a = fcm.MCMCSampler(g, b, n_chains=6, seed=4, move_weights=weights)
            assert a.info["waves_per_chain"] == W, a.info
            for nstep in (1, 7, 100):
                a.step(nstep)
            sa, so = a.stats(), o.stats()
            same = (a.flag_counts() == o.flag_counts()).all() and all((sa[k] == so[k]).all() for k in ("accepted", "n_changes", "sum_k", "status", "count_len", "n_cperm", "n_cswap", "n_flip", "n_dmove"))
            same = same and all((a.edges(c) == o.edges(c)).all() and (a.double_slots(c) == o.double_slots(c)).all() for c in range(6))
            print("n=%d weights=%s W=%d: %s" % (n, weights, W, "identical" if same else "DIFFERENT"), flush=True)
            ok = ok and same
sys.exit(0 if ok else 1)
