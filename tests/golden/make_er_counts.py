#!/usr/bin/env python3
"""Regenerates tests/golden/er_counts.json: simplex counts of the seeded ER
graphs the BASELINE configs name, computed by the CPU oracle.  (The reference
publishes no such numbers and cannot be run here)"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_ffi as o  # noqa: E402
from flag_complex_mcmc_amd import graphs  # noqa: E402

out = {}
for n, p, seed in [(1000, 0.10, 0)]:
    e = graphs.random_with_p(n, p, seed)
    g = o.Graph.from_edges(n, e)
    und = g.undirected_edges()
    out["n%d_p%.2f_seed%d" % (n, p, seed)] = {
        "n": n, "m": int(len(e)), "undirected": int(len(und)), "reciprocal": int(len(e) - len(und)),
        "flag_count": g.flagser_count()}
    print(out)
json.dump(out, open(os.path.join(HERE, "er_counts.json"), "w"), indent=1, sort_keys=True)
