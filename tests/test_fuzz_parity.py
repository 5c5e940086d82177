"""Randomized differential test: random small graphs, move mixes, bounds relaxations, chain counts, launch sizes and
waves per chain against oracle twins (tolerance 0: counts, counters, edges, slot lists after every launch).  A dozen
cases by default; FCM_FUZZ_CASES=<n> and FCM_FUZZ_SEED=<s> run a campaign (the round-2 campaign: 400 cases, seed 2; round 3:
both clique-move kernels with W = 1, 2, 4, 8 and the sparse per-chain state are drawn too)."""
import os
import numpy as np
import pytest

from test_gpu_parity import _run_parity

pytestmark = pytest.mark.gpu

NCASES = int(os.environ.get("FCM_FUZZ_CASES", "12"))
SEED0 = int(os.environ.get("FCM_FUZZ_SEED", "1"))


MEDIUM = os.environ.get("FCM_FUZZ_MEDIUM", "0") == "1"   # campaign option: graphs of 100..700 vertices, local sets of 10..60 vertices


def _case(i):
    rng = np.random.default_rng([SEED0, i])
    if MEDIUM:
        n = int(rng.integers(100, 700))
        p = float(np.sqrt(rng.uniform(10.0, 60.0) / n) / 2.0)     # |N(a) cap N(b)| about n (2p)^2.  Not capped: at n ~ 150, p ~ 0.3 pr(G) has
        # cliques of a dozen vertices and a sixth of the pairs is reciprocal -- the generic deep kernels (more than 8 count entries) and the
        # oracle's recounts then take a minute or two per case (seed 52, case 10: GPU leg 42 s since the deep kernels count in 64 bits, 143 s
        # before; oracle leg 48 s: tools/cliff_case.py), so the medium campaign runs under an explicit long per-test timeout
    else:
        n = int(rng.integers(6, 70))
        p = float(rng.uniform(0.08, 0.5)) if n < 30 else float(rng.uniform(0.05, 0.25))
    mix = int(rng.integers(0, 5))
    weights = [(0.5, 0.5, 0.0, 0.0), (1.0, 0.0, 0.0, 0.0), (0.0, 1.0, 0.0, 0.0), (0.1, 0.1, 0.6, 0.2), (0.25, 0.25, 0.25, 0.25)][mix]
    W = [1, 2, 4, 8, 16][int(rng.integers(0, 5))]
    steps = [int(x) for x in rng.integers(1, 400, size=int(rng.integers(2, 5)))] + [int(rng.integers(400, 4000 if MEDIUM else 1500))]
    c = dict(n=n, p=p, gseed=int(rng.integers(0, 1 << 30)), weights=weights, W=W, steps=steps, chains=int(rng.integers(1, 5)),
             seed=int(rng.integers(0, 1 << 30)), relaxation=float(rng.choice([0.01, 0.05, 0.3])), first=int(rng.integers(0, 1000)))
    # round 3: which clique-move kernel (cooperative with W waves per chain, or the one-wave one), and the sparse per-chain
    # state where the graph allows it (simple moves, local sets of at most 11 vertices)
    c["cq"], c["cqw"], c["sparse"] = int(rng.integers(0, 4) > 0), [1, 2, 4, 8][int(rng.integers(0, 4))], int(rng.integers(0, 2))
    return c


@pytest.mark.timeout(900 if MEDIUM else 120)
@pytest.mark.parametrize("i", range(NCASES))
def test_random_case_against_oracle_twins(fcm, oracle, monkeypatch, i):
    from flag_complex_mcmc_amd import graphs
    c = _case(i)
    e = graphs.random_with_p(c["n"], c["p"], seed=c["gseed"])
    if len(e) < 2:
        pytest.skip("empty graph")
    monkeypatch.setenv("FCM_MW", str(c["W"]))
    monkeypatch.setenv("FCM_CQ", str(c["cq"]))
    monkeypatch.setenv("FCM_CQW", str(c["cqw"]))
    monkeypatch.setenv("FCM_SPARSE", str(c["sparse"]))
    bounds = None
    rng = np.random.default_rng([SEED0, i, 7])
    if rng.random() < 0.35:
        # hand-made bounds around the initial counts: tight (most proposals rejected), and now and then not even containing
        # the initial state (the chain then accepts nothing but empty transitions until a move brings it inside -- it cannot:
        # every non-empty proposal is checked against the bounds -- src/lib.rs:185-191)
        fc = oracle.Graph.from_edges(c["n"], e).flagser_count()
        width = [int(rng.integers(0, 1 + max(1, v // int(rng.choice([4, 20, 200]))))) for v in fc]
        lo = [max(0, v - w) for v, w in zip(fc, width)]
        hi = [v + w for v, w in zip(fc, width)]
        lo[0], hi[0], lo[1], hi[1] = fc[0], fc[0], fc[1], fc[1]
        if rng.random() < 0.2 and len(fc) > 2:
            lo[2] = fc[2] + 1; hi[2] = fc[2] + 1 + width[2]
        bounds = (lo, hi)
    try:
        s, tw = _run_parity(fcm, oracle, c["n"], e, n_chains=c["chains"], steps=c["steps"], seed=c["seed"], weights=c["weights"],
                            relaxation=c["relaxation"], first_chain_id=c["first"], bounds=bounds)
    except ValueError as ex:
        if "would panic in the reference" in str(ex):   # (the oracle's Bounds::calculate: the reference panics on this input)
            pytest.skip(str(ex))
        raise
    except fcm.FcmError as ex:
        # inputs on which the reference itself panics (Bounds::calculate on some tiny graphs) or that this build refuses loudly
        if ex.code in (4, 6):
            pytest.skip("refused: %s" % ex)
        raise
    assert (s.stats()["status"] == 0).all(), c
