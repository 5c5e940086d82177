"""The sparse per-chain state: two bits per adjacent pair of pr(G) instead of row bitmaps (the reference's edgebits layout,
src/io.rs:152-159, as the device state; SURVEY.md 7 "memory footprint", F8).  Chosen by the library for graphs like
BASELINE configs[4] (n = 30000, k ~ 0.1: 250 KB per chain instead of 115 MB); FCM_SPARSE forces it here.  Everything a
caller can see must be identical to the row-bitmap layout and to the oracle: trajectories, counters, edges, edgebits,
state files, the State API."""
import numpy as np
import pytest

from helpers import compare_chain, setup_pair

pytestmark = pytest.mark.gpu


def _sparse_graph(fcm, n, p, seed, recip=40):
    """ER digraph with a few reciprocal pairs put in by hand (at these densities there are next to none by chance)."""
    e = fcm.graphs.random_with_p(n, p, seed=seed)
    rng = np.random.default_rng(seed)
    extra = e[rng.choice(len(e), size=recip, replace=False)][:, ::-1]
    return np.unique(np.concatenate([e, extra]), axis=0).astype(np.uint32)


@pytest.mark.parametrize("W", ["2", "8", "16"])
def test_sparse_state_against_oracle_twins(fcm, oracle, monkeypatch, W):
    monkeypatch.setenv("FCM_SPARSE", "1")
    monkeypatch.setenv("FCM_MW", W)
    n = 1500
    e = _sparse_graph(fcm, n, 0.008, 3)
    rejected = nonempty_k = 0
    for weights, rel in (((0.5, 0.5, 0.0, 0.0), 0.0), ((1.0, 0.0, 0.0, 0.0), 0.002), ((0.0, 1.0, 0.0, 0.0), 0.0)):
        gg, go, b_g, b_o = setup_pair(fcm, oracle, n, e, relaxation=rel)
        s = fcm.MCMCSampler(gg, b_g, n_chains=3, seed=7, move_weights=weights)
        assert s.info["sparse_state"] == 1 and s.info["waves_per_chain"] == int(W) and 1 <= s.info["k_max"] <= 9, s.info
        tw = [oracle.Chain(go, b_o, weights=weights, seed=7, chain_id=c) for c in range(3)]
        for nstep in (1, 2, 61, 3000):
            s.step(nstep)
            for c in range(3):
                tw[c].step(nstep)
                compare_chain(s, c, tw[c], ctx=("sparse", weights, c, nstep))
        st = s.stats()
        assert (st["status"] == 0).all()
        rejected += int((st["sampled"] - st["accepted"]).sum())
        nonempty_k += int(st["sum_k"].sum())
    assert rejected > 0 and nonempty_k > 100      # local sets beyond the pair itself were built, proposals were refused


def test_sparse_and_row_bitmap_layouts_agree_and_readers_work(fcm, oracle, monkeypatch, tmp_path):
    n = 3000
    e = _sparse_graph(fcm, n, 0.004, 5, recip=80)
    g = fcm.Graph.from_edges(n, e)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("FCM_SPARSE", mode)
        s = fcm.initialize_new_sampler(g, n_chains=24, seed=2, target_relaxation=0.0)
        assert s.info["sparse_state"] == int(mode)
        for nstep in (1, 63, 5000):
            s.step(nstep)
        st = s.stats()
        assert (st["status"] == 0).all()
        out[mode] = (s.flag_counts(), {k: st[k].copy() for k in ("sampled", "accepted", "n_empty", "n_flip", "n_dmove", "sum_k", "count_len")},
                     [s.edges(c) for c in (0, 11, 23)], [s.edgebits(c) for c in (0, 23)], [s.double_slots(c) for c in (0, 23)], s)
    a, b = out["0"], out["1"]
    assert (a[0] == b[0]).all() and all((a[1][k] == b[1][k]).all() for k in a[1])
    for x, y in zip(a[2] + a[3] + a[4], b[2] + b[3] + b[4]):
        assert np.array_equal(x, y)
    sp = b[5]
    assert sp.info["bytes_per_chain"] * 20 < a[5].info["bytes_per_chain"]
    for c in (0, 23):
        assert sp.graph(c).flagser_count() == sp.flag_count(c)[: len(sp.graph(c).flagser_count())]
    # state files: written from the sparse record, read back into it, and the run goes on as if never stopped
    path = str(tmp_path / "sparse.state")
    sp.save_state(path, 9)
    monkeypatch.setenv("FCM_SPARSE", "1")
    k, resumed = fcm.MCMCSampler.load_state(path)
    assert k == 9 and resumed.info["sparse_state"] == 1
    resumed.step(700); sp.step(700); a[5].step(700)
    assert (resumed.flag_counts() == sp.flag_counts()).all() and (sp.flag_counts() == a[5].flag_counts()).all()
    assert np.array_equal(resumed.edgebits(5), a[5].edgebits(5))
    # the State API on the sparse record
    so = oracle.State(oracle.Graph.from_edges(n, sp.edges(3)))
    sg = sp.state(3)
    rng = np.random.default_rng(1)
    for it in range(40):
        t = fcm.Transition.single_edge_flip(sg, int(rng.integers(0, 2 ** 63)) * 2)
        if not t.change_edges:
            continue
        got = sg.apply_transition(t)
        assert got == so.apply_transition(t.change_edges), it
        if it % 3 == 0:
            sg.revert_transition(t, got)
            so.revert_transition(t.change_edges, got)
        assert sg.flag_count == so.flag_count and (sp.edges(3) == so.graph_edges()).all()


def test_library_selects_the_layout(fcm, monkeypatch):
    monkeypatch.delenv("FCM_SPARSE", raising=False)
    g = fcm.Graph.from_edges(3000, _sparse_graph(fcm, 3000, 0.004, 5))
    assert fcm.initialize_new_sampler(g, n_chains=2).info["sparse_state"] == 1                 # long rows, tiny local sets
    assert fcm.initialize_new_sampler(g, n_chains=2, simple=False).info["sparse_state"] == 0   # clique moves: row bitmaps
    dense = fcm.Graph.from_edges(1500, fcm.graphs.random_with_p(1500, 0.05, seed=1))
    assert fcm.initialize_new_sampler(dense, n_chains=2).info["sparse_state"] == 0             # local sets of dozens of vertices
    small = fcm.Graph.from_edges(500, fcm.graphs.random_with_p(500, 0.01, seed=1))
    assert fcm.initialize_new_sampler(small, n_chains=2).info["sparse_state"] == 0             # rows of one cache line
    monkeypatch.setenv("FCM_MW", "1")
    assert fcm.initialize_new_sampler(g, n_chains=2).info["sparse_state"] == 0                 # no multi-wave kernel, no sparse state
