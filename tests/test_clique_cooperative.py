"""The cooperative clique-move kernel (fcm_step_cq.hpp; reference clique_permute / clique_swap, src/lib.rs:214-290):
every W against oracle twins, tolerance 0.  W waves of a chain's workgroup share the changed pairs of a move, each pair
counted on the pre-move bitmap with the earlier pairs' changes patched into its local masks; wave 0 proposes, decides and
commits.  The library picks W from the chain count (8 for the small batches of these tests) and leaves batches of more
than 2048 chains to the one-wave kernel; FCM_CQ / FCM_CQW force the one or the other here, so that both kernels and every
W stay in front of the oracle."""
import numpy as np
import pytest

from helpers import compare_chain, setup_pair

pytestmark = pytest.mark.gpu

MODES = [("0", "1"), ("1", "1"), ("1", "2"), ("1", "4"), ("1", "8")]   # (FCM_CQ, FCM_CQW)


def _twins(fcm, oracle, n, e, weights, n_chains, steps, seed, relaxation, expect_w):
    gg, go, b_g, b_o = setup_pair(fcm, oracle, n, e, relaxation=relaxation)
    s = fcm.MCMCSampler(gg, b_g, n_chains=n_chains, seed=seed, move_weights=weights)
    if s.ncounts <= 8:
        assert s.info["waves_per_chain"] == expect_w, s.info
    tw = [oracle.Chain(go, b_o, weights=weights, seed=seed, chain_id=c) for c in range(n_chains)]
    for nstep in steps:
        s.step(nstep)
        for c in range(n_chains):
            tw[c].step(nstep)
            compare_chain(s, c, tw[c], ctx=(n, weights, c, nstep))
    assert (s.stats()["status"] == 0).all()
    return s


@pytest.mark.parametrize("cq,w", MODES)
def test_every_W_against_oracle_twins_on_dense_small_graphs(fcm, oracle, monkeypatch, cq, w):
    monkeypatch.setenv("FCM_CQ", cq)
    monkeypatch.setenv("FCM_CQW", w)
    want = int(w) if cq == "1" else 1
    changes = rejected = 0
    for n, pr, gseed in ((40, 0.35, 1), (60, 0.2, 5)):
        e = fcm.graphs.random_with_p(n, pr, seed=gseed)
        for weights in ((0.0, 0.0, 1.0, 0.0), (0.0, 0.0, 0.0, 1.0), (0.1, 0.1, 0.6, 0.2)):
            s = _twins(fcm, oracle, n, e, weights, 2, [1, 7, 120], 4, 0.2, want)
            st = s.stats()
            changes += int(st["n_changes"].sum())
            rejected += int((st["sampled"] - st["accepted"]).sum())
    assert changes > 2000 and rejected > 20      # (rejected moves: nothing was written, nothing to put back)


@pytest.mark.parametrize("cq,w", [("1", "2"), ("1", "8")])
def test_every_W_on_the_bench_graph_default_mix(fcm, oracle, monkeypatch, cq, w):
    """configs[2]'s graph, the reference's default move mix: two chains x 600 proposals (about 460 clique moves of ~12
    changed edges each)."""
    monkeypatch.setenv("FCM_CQ", cq)
    monkeypatch.setenv("FCM_CQW", w)
    n = 1000
    e = fcm.graphs.random_with_p(n, 0.10, 0)
    s = _twins(fcm, oracle, n, e, (0.1, 0.1, 0.6, 0.2), 2, [600], 0, 0.01, int(w))
    assert (s.stats()["n_cperm"] > 250).all() and (s.stats()["n_cswap"] > 60).all()


@pytest.mark.parametrize("w", ["1", "4"])
def test_graphs_of_more_than_1024_vertices(fcm, oracle, monkeypatch, w):
    """Rows longer than a cache line: no vertex -> d-index table in LDS (it covers 1024 vertices), the patches find their
    positions by comparing; the simple moves' exact run takes the looped build."""
    monkeypatch.setenv("FCM_CQ", "1")
    monkeypatch.setenv("FCM_CQW", w)
    n = 1100
    e = fcm.graphs.random_with_p(n, 0.08, seed=2)
    s = _twins(fcm, oracle, n, e, (0.1, 0.1, 0.6, 0.2), 2, [1, 64, 400], 11, 0.05, int(w))
    assert s.info["row_words"] > 16 and (s.stats()["n_cswap"] > 30).all()


def _book_graph(t, p_page, seed):
    """Two hubs joined to t page vertices (local set of t + 2 vertices for the hub pair), pages sparsely joined."""
    rng = np.random.default_rng(seed)
    n = t + 2
    e = [(0, 1)]
    for v in range(2, n):
        e.append((0, v) if rng.random() < 0.5 else (v, 0))
        e.append((1, v) if rng.random() < 0.5 else (v, 1))
    for a in range(2, n):
        for b in range(a + 1, n):
            if rng.random() < p_page:
                e.append((a, b) if rng.random() < 0.5 else (b, a))
    return n, np.array(e, np.uint32)


@pytest.mark.parametrize("t,p_page,w", [(100, 0.03, "4"), (300, 0.008, "2"), (300, 0.008, "8")])
def test_pairs_left_to_the_wide_evaluators(fcm, oracle, monkeypatch, t, p_page, w):
    """Cliques through the hub pair: its local set has t + 2 vertices (wide evaluator; beyond 256 the workspace one), so the
    pair is deferred to wave 0 with the earlier pairs' changes set into its masks."""
    monkeypatch.setenv("FCM_CQ", "1")
    monkeypatch.setenv("FCM_CQW", w)
    n, e = _book_graph(t, p_page, 7)
    s = _twins(fcm, oracle, n, e, (0.0, 0.0, 0.7, 0.3), 2, [1, 64, 300], 3, 0.3, int(w))
    assert s.stats()["n_wide"].sum() > 0


def test_library_picks_the_kernel_by_chain_count(fcm, monkeypatch):
    monkeypatch.delenv("FCM_CQ", raising=False)
    monkeypatch.delenv("FCM_CQW", raising=False)
    n = 200
    e = fcm.graphs.random_with_p(n, 0.12, seed=1)
    g = fcm.Graph.from_edges(n, e)
    for chains, want in ((64, 8), (512, 8), (513, 4), (1024, 4), (2048, 2), (2049, 1)):
        s = fcm.initialize_new_sampler(g, n_chains=chains, seed=0, simple=False)
        assert s.info["waves_per_chain"] == want, (chains, s.info["waves_per_chain"])
        s.step(50)
        assert (s.stats()["status"] == 0).all()
        c = chains - 1
        assert s.graph(c).flagser_count() == s.flag_count(c)[: len(s.graph(c).flagser_count())]


def _sampler(fcm, g, weights, n_chains, seed, relaxation=0.01):
    fc = g.flagser_count()
    b = fcm.Bounds.calculate(g, fc, fcm.Bounds.target(fc, relaxation))
    return fcm.MCMCSampler(g, b, n_chains=n_chains, seed=seed, move_weights=weights)


def test_pair_and_shared_row_counters_agree_between_the_two_kernels(fcm, monkeypatch):
    """FCM_STAT_PAIRS / FCM_STAT_SHARED_ROWS (what bench.py's traffic model of a clique move is made of): a changed pair has
    one or two changed directions; a permutation of a clique of c vertices with m changed pairs shares (m - 1) c rows;
    both clique-move kernels count the same."""
    n = 120
    e = fcm.graphs.random_with_p(n, 0.2, seed=3)
    g = fcm.Graph.from_edges(n, e)
    got = []
    for cq, w in (("0", "1"), ("1", "4")):
        monkeypatch.setenv("FCM_CQ", cq)
        monkeypatch.setenv("FCM_CQW", w)
        s = _sampler(fcm, g, (0.0, 0.0, 0.7, 0.3), 3, 2)
        s.step(800)
        st = s.stats()
        assert (st["status"] == 0).all()
        assert (st["n_pairs"] <= st["n_changes"]).all() and (st["n_changes"] <= 2 * st["n_pairs"]).all() and (st["n_pairs"] > 500).all()
        assert (st["n_shared_rows"] > 0).all() and (st["n_shared_rows"] < 8 * st["n_pairs"]).all()   # (cliques of at most 8 vertices here)
        got.append((st["n_pairs"].tolist(), st["n_shared_rows"].tolist(), st["n_changes"].tolist()))
    assert got[0] == got[1]
