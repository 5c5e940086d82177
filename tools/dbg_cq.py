#!/usr/bin/env python3
"""Debug aid: the clique-move kernel (fcm_step_cq) against the one-wave kernel (FCM_CQ=0), proposal by proposal."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import flag_complex_mcmc_amd as fcm
from flag_complex_mcmc_amd import graphs

n, pr, gseed, seed = 40, 0.35, 1, 4
weights = tuple(float(x) for x in (sys.argv[1:5] or (0, 0, 1, 0)))
e = graphs.random_with_p(n, pr, seed=gseed)
g = fcm.Graph.from_edges(n, e)
fc = g.flagser_count()
b = fcm.Bounds.calculate(g, fc, fcm.Bounds.target(fc, 0.2))
print("counts", fc)
os.environ["FCM_CQ"] = "1"
a = fcm.MCMCSampler(g, b, n_chains=3, seed=seed, move_weights=weights)
os.environ["FCM_CQ"] = "0"
o = fcm.MCMCSampler(g, b, n_chains=3, seed=seed, move_weights=weights)
for step in range(300):
    a.step(1); o.step(1)
    sa, so = a.stats(), o.stats()
    ca, co = a.flag_counts(), o.flag_counts()
    for c in range(3):
        if (ca[c] != co[c]).any() or any(sa[k][c] != so[k][c] for k in ("accepted", "n_changes", "sum_k", "status")):
            print("step", step, "chain", c)
            print(" cq :", ca[c].tolist(), {k: int(sa[k][c]) for k in sa})
            print(" old:", co[c].tolist(), {k: int(so[k][c]) for k in so})
            ea, eo = {tuple(x) for x in a.edges(c).tolist()}, {tuple(x) for x in o.edges(c).tolist()}
            print(" edges only cq:", sorted(ea - eo), "only old:", sorted(eo - ea))
            sys.exit(1)
print("identical for 300 proposals")
