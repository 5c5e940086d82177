#!/usr/bin/env python3
"""Driver for the two-wave (producer/consumer) step kernel: parity against the oracle on small
graphs (two-wave kernel forced: FCM_PC=2) and timing on config 3 for a given move mix (FCM_PC from
the environment: 0 = one-wave kernel, unset = the library's choice).  GPU box only.
pc_check.py parity|bench  w0 w1"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import flag_complex_mcmc_amd as fcm
from flag_complex_mcmc_amd import graphs
mode = sys.argv[1]
w = [float(sys.argv[2]), float(sys.argv[3]), 0.0, 0.0]
if mode == "parity":
    os.environ["FCM_PC"] = "2"
    from oracle import oracle_ffi as oracle
    from helpers import setup_pair, compare_chain
    bad = 0
    for (n, pr, seed, steps) in [(300, 0.15, 1, [1, 63, 64, 65, 700]), (40, 0.5, 2, [5, 100, 1000]), (20, 0.3, 3, [3, 200, 2000]),
                                 (12, 0.3, 5, [1, 2, 3, 64, 1000]), (150, 0.15, 2, [100, 900]), (1000, 0.02, 4, [300])]:
        e = graphs.random_with_p(n, pr, seed=seed)
        gg, go, b_g, b_o = setup_pair(fcm, oracle, n, e, 0.05)
        s = fcm.MCMCSampler(gg, b_g, n_chains=6, seed=11, move_weights=w)
        tw = [oracle.Chain(go, b_o, weights=w, seed=11, chain_id=c) for c in range(6)]
        for ns in steps:
            s.step(ns)
            for c in range(6):
                tw[c].step(ns)
                try:
                    compare_chain(s, c, tw[c], ctx=(n, c, ns))
                except AssertionError as ex:
                    bad += 1
                    print("MISMATCH", ex.args[0] if ex.args else ex)
                    break
        st = s.stats()
        print("graph n=%d p=%.2f: status %s flips %d dmoves %d empties %d accepted %d" % (n, pr, sorted(set(int(x) for x in st["status"])),
              st["n_flip"].sum(), st["n_dmove"].sum(), st["n_empty"].sum(), st["accepted"].sum()))
    print("PARITY", "OK" if bad == 0 else "FAILED (%d)" % bad)
else:
    n, chains, props = 1000, int(os.environ.get("PC_CHAINS", "4096")), 1024
    e = graphs.random_with_p(n, 0.10, 0)
    g = fcm.Graph.from_edges(n, e)
    fc = g.flagser_count(0)
    b = fcm.Bounds.calculate(g, fc, fcm.Bounds.target(fc, 0.01), 0)
    s = fcm.MCMCSampler(g, b, n_chains=chains, seed=0, move_weights=w)
    s.step(64)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); s.step(props); ts.append(time.perf_counter() - t0)
    st = s.stats()
    print("FCM_PC=%s weights %s chains %d: %.3f ms per launch, %.4g proposals/s, status %s, accepted %.4f" % (
        os.environ.get("FCM_PC", "0"), w, chains, 1e3 * min(ts), chains * props / min(ts), sorted(set(int(x) for x in st["status"])),
        st["accepted"].sum() / st["sampled"].sum()))
