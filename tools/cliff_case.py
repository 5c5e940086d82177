#!/usr/bin/env python3
"""GPU box: the dense fuzz case the round-3 medium campaign hit its time limit on (seed 52, case 10, as it was drawn then:
n = 167, p = 0.298 -- the generator's cap on p came afterwards), with the GPU leg and the oracle leg timed SEPARATELY,
once.  Prints proposals/s of the step kernel the library selects for it (more than 8 count entries:
the generic one-wave kernel, run-time depth), which evaluator the proposals went through (n_wide, n_big) and the oracle's
time for the same chain.  usage: python tools/cliff_case.py [seed] [case]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def drawn_case(seed0, i):
    """tests/test_fuzz_parity.py::_case in its MEDIUM form, WITHOUT the cap on p."""
    rng = np.random.default_rng([seed0, i])
    n = int(rng.integers(100, 700))
    p = float(np.sqrt(rng.uniform(10.0, 60.0) / n) / 2.0)
    mix = int(rng.integers(0, 5))
    weights = [(0.5, 0.5, 0.0, 0.0), (1.0, 0.0, 0.0, 0.0), (0.0, 1.0, 0.0, 0.0), (0.1, 0.1, 0.6, 0.2), (0.25, 0.25, 0.25, 0.25)][mix]
    W = [1, 2, 4, 8, 16][int(rng.integers(0, 5))]
    steps = [int(x) for x in rng.integers(1, 400, size=int(rng.integers(2, 5)))] + [int(rng.integers(400, 4000))]
    c = dict(n=n, p=p, gseed=int(rng.integers(0, 1 << 30)), weights=weights, W=W, steps=steps, chains=int(rng.integers(1, 5)),
             seed=int(rng.integers(0, 1 << 30)), relaxation=float(rng.choice([0.01, 0.05, 0.3])), first=int(rng.integers(0, 1000)))
    c["cq"], c["cqw"], c["sparse"] = int(rng.integers(0, 4) > 0), [1, 2, 4, 8][int(rng.integers(0, 4))], int(rng.integers(0, 2))
    return c


def main():
    seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 52
    i = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    c = drawn_case(seed0, i)
    os.environ["FCM_MW"], os.environ["FCM_CQ"], os.environ["FCM_CQW"], os.environ["FCM_SPARSE"] = str(c["W"]), str(c["cq"]), str(c["cqw"]), str(c["sparse"])
    import flag_complex_mcmc_amd as fcm
    import oracle_ffi as oracle   # checker only
    from helpers import setup_pair, compare_chain
    e = fcm.graphs.random_with_p(c["n"], c["p"], seed=c["gseed"])
    print("case", {k: c[k] for k in ("n", "p", "weights", "W", "steps", "chains", "relaxation", "cq", "cqw")}, "edges", len(e), flush=True)
    t0 = time.perf_counter()
    gg, go, b_g, b_o = setup_pair(fcm, oracle, c["n"], e, c["relaxation"])
    t_setup = time.perf_counter() - t0
    fc = go.flagser_count()
    t0 = time.perf_counter()
    s = fcm.MCMCSampler(gg, b_g, n_chains=c["chains"], seed=c["seed"], move_weights=c["weights"], first_chain_id=c["first"])
    t_create = time.perf_counter() - t0
    print("flag_count", fc, "| count entries tracked", s.ncounts, "| info", {k: s.info[k] for k in ("k_max", "k_mean", "waves_per_chain", "cooperative_clique_kernel", "n_undirected", "n_double")}, flush=True)
    nprop = sum(c["steps"])
    # ---- GPU leg, alone
    t0 = time.perf_counter()
    for nstep in c["steps"]:
        s.step(nstep)
    st = s.stats()
    t_gpu = time.perf_counter() - t0
    assert (st["status"] == 0).all()
    print("GPU leg: %d proposals x %d chain(s) in %.3f s = %.3g proposals/s per chain (%.1f us per proposal); n_wide %s n_big %s n_cperm %s n_cswap %s n_changes %s n_pairs %s accepted %s"
          % (nprop, c["chains"], t_gpu, nprop / t_gpu, 1e6 * t_gpu / nprop, st["n_wide"].tolist(), st["n_big"].tolist(), st["n_cperm"].tolist(), st["n_cswap"].tolist(),
             st["n_changes"].tolist(), st["n_pairs"].tolist(), st["accepted"].tolist()), flush=True)
    # ---- oracle leg, alone (chain 0 only), then the comparison
    tw = oracle.Chain(go, b_o, weights=c["weights"], seed=c["seed"], chain_id=c["first"])
    t0 = time.perf_counter()
    tw.step(nprop)
    t_or = time.perf_counter() - t0
    print("oracle leg (one chain): %.3f s = %.3g proposals/s (%.1f us per proposal); setup of both sides %.2f s, sampler create %.2f s" % (t_or, nprop / t_or, 1e6 * t_or / nprop, t_setup, t_create), flush=True)
    compare_chain(s, 0, tw, ctx="cliff case")
    print("parity: chain 0 == oracle twin")


if __name__ == "__main__":
    main()
