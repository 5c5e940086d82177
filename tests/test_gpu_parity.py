"""GPU: the HIP path (through the C ABI) against the oracle, bit for bit.

Tolerance is 0 everywhere: the accept rule is integer-only, so a
GPU chain and its oracle twin on the same Philox stream must agree on
sampled/accepted, every count, every edge and the reciprocal-pair slot list.
"""
import numpy as np
import pytest

from helpers import compare_chain, known_answers, load_flag_fixture, setup_pair

pytestmark = pytest.mark.gpu


def _case_graph(name, rec):
    if name.endswith(".flag"):
        return load_flag_fixture(name)
    return rec["n"], np.array(rec["edges"], np.uint32)


def test_extension_is_loaded_and_sees_the_gpu(fcm):
    assert fcm.device_count() >= 1
    import ctypes
    assert ctypes.CDLL(fcm.LIB_PATH).fcm_version


# ---------------------------------------------------------------- counter ----
def test_flagser_count_known_answers(fcm):
    for name, rec in known_answers().items():
        n, e = _case_graph(name, rec)
        g = fcm.Graph.from_edges(n, e)
        assert g.flagser_count() == rec["flag_count"], name
        # the legacy symbol the reference binds (src/flagser.rs:7-21)
        assert fcm.count_unweighted(n, e) == rec["flag_count"], name


def test_flagser_count_edge_cases(fcm):
    assert fcm.Graph.new_disconnected(0).flagser_count() == []
    assert fcm.Graph.new_disconnected(5).flagser_count() == [5]
    assert fcm.count_unweighted(2, [(0, 1), (0, 1), (1, 1)]) == [2, 1]   # duplicates and loops ignored
    assert fcm.count_unweighted(2, [(0, 1), (1, 0)]) == [2, 2]           # reciprocal pair counts twice
    with pytest.raises(fcm.FcmError):
        fcm.count_unweighted(2, [(0, 2)])


def test_flagser_count_vs_oracle_random(fcm, oracle):
    from flag_complex_mcmc_amd import graphs
    for n, p, seed in [(50, 0.3, 0), (130, 0.15, 1), (300, 0.1, 2), (64, 0.5, 3), (1000, 0.02, 4)]:
        e = graphs.random_with_p(n, p, seed)
        assert fcm.Graph.from_edges(n, e).flagser_count() == oracle.Graph.from_edges(n, e).flagser_count(), (n, p)


def test_flagser_count_high_dimension(fcm, oracle):
    from flag_complex_mcmc_amd import graphs
    # a transitive tournament on 14 vertices: one 13-simplex, C(14,d+1) d-simplices
    e = graphs.simplex(13)
    import math
    want = [math.comb(14, d + 1) for d in range(14)]
    assert fcm.Graph.from_edges(14, e).flagser_count() == want
    assert oracle.Graph.from_edges(14, e).flagser_count() == want
    # 17 vertices would need dimension 16 > 15: refused, not truncated
    with pytest.raises(fcm.FcmError) as ei:
        fcm.Graph.from_edges(17, graphs.simplex(16)).flagser_count()
    assert ei.value.code == 4


def test_count_config3_known_answer(fcm):
    """ER n=1000 p=0.10 seed 0 (BASELINE config 3): committed golden counts,
    produced by the oracle (golden/make_er_counts.py)."""
    import json, os
    from flag_complex_mcmc_amd import graphs
    from helpers import GOLDEN
    rec = json.load(open(os.path.join(GOLDEN, "er_counts.json")))["n1000_p0.10_seed0"]
    e = graphs.random_with_p(1000, 0.10, 0)
    assert len(e) == rec["m"]
    g = fcm.Graph.from_edges(1000, e)
    assert g.flagser_count() == rec["flag_count"]


# ----------------------------------------------------------------- bounds ----
def test_bounds_calculate_vs_oracle(fcm, oracle):
    from flag_complex_mcmc_amd import graphs
    cases = [load_flag_fixture(f) for f in ("counterexample_any_order.flag", "bug_calc_relax_de.flag")]
    cases += [(60, graphs.random_with_p(60, 0.3, 3)), (200, graphs.random_with_p(200, 0.1, 5))]
    for n, e in cases:
        go = oracle.Graph.from_edges(n, e)
        st = oracle.State(go)
        fc = st.flag_count
        bo, ncl_o = oracle.bounds_calculate(st, oracle.target_bounds(fc, 0.01))
        gg = fcm.Graph.from_edges(n, e)
        bg, ncl_g = fcm.Bounds.calculate(gg, fc, fcm.Bounds.target(fc, 0.01), return_ncliques=True)
        assert (bg.flag_count_min, bg.flag_count_max) == bo.lists()
        assert ncl_g == ncl_o


def test_bounds_calculate_panics_like_the_reference(fcm):
    # reciprocal pair but no 2-simplex: flag_count_max[2] is out of range (src/lib.rs:151)
    g = fcm.Graph.from_edges(3, [(0, 1), (1, 0), (1, 2)])
    fc = g.flagser_count()
    assert fc == [3, 3]
    with pytest.raises(fcm.FcmError) as ei:
        fcm.Bounds.calculate(g, fc, fcm.Bounds.target(fc, 0.01))
    assert ei.value.code == 6


# ------------------------------------------------------------ trajectories ----
def _run_parity(fcm, oracle, n, e, n_chains, steps, seed, weights=(0.5, 0.5, 0.0, 0.0), relaxation=0.01,
                first_chain_id=0, check_chains=None, bounds=None):
    if bounds is not None:   # hand-made bounds: Bounds::calculate is not involved (it panics on some tiny graphs)
        gg, go = fcm.Graph.from_edges(n, e), oracle.Graph.from_edges(n, e)
        b_g = fcm.Bounds(*bounds)
        b_o = oracle.Bounds.from_lists(*bounds)
    else:
        gg, go, b_g, b_o = setup_pair(fcm, oracle, n, e, relaxation)
    s = fcm.MCMCSampler(gg, b_g, n_chains=n_chains, seed=seed, move_weights=weights, first_chain_id=first_chain_id)
    check_chains = list(range(n_chains)) if check_chains is None else check_chains
    twins = {c: oracle.Chain(go, b_o, weights=weights, seed=seed, chain_id=first_chain_id + c) for c in check_chains}
    for nstep in steps:
        s.step(nstep)
        for c, tw in twins.items():
            tw.step(nstep)
            compare_chain(s, c, tw, ctx=(n, c, nstep))
    return s, twins


def test_trajectory_fixtures(fcm, oracle):
    for f in ("counterexample_any_order.flag", "counterexample_seo_greedy_5_start.flag", "bug_calc_relax_de.flag"):
        n, e = load_flag_fixture(f)
        s, tw = _run_parity(fcm, oracle, n, e, n_chains=3, steps=[1, 63, 64, 65, 500], seed=5)
        st = s.stats()
        assert (st["sampled"] == 693).all()
        # no reciprocal pairs: every double-move proposal is empty and accepted (src/lib.rs:324,185-187)
        assert (st["n_dmove"] == 0).all() and (st["n_empty"] > 250).all()


def test_trajectory_er_with_reciprocal_pairs(fcm, oracle):
    from flag_complex_mcmc_amd import graphs
    e = graphs.random_with_p(150, 0.15, seed=2)
    s, tw = _run_parity(fcm, oracle, 150, e, n_chains=4, steps=[100, 900, 1000], seed=9, first_chain_id=17)
    st = s.stats()
    assert (st["n_dmove"] > 0).all() and (st["n_flip"] > 0).all()
    # different chains took different paths
    assert len({tuple(s.flag_count(c)) for c in range(4)}) > 1


def test_trajectory_rejections_happen_and_match(fcm, oracle):
    """Tight bounds force rejections, exercising the drop path (src/lib.rs:189-191)."""
    from flag_complex_mcmc_amd import graphs
    e = graphs.random_with_p(120, 0.2, seed=4)
    go = oracle.Graph.from_edges(120, e)
    fc = go.flagser_count()
    mn = list(fc); mx = list(fc)
    for d in range(2, len(fc)):
        mn[d] = fc[d] - 3
        mx[d] = fc[d] + 3
    s, tw = _run_parity(fcm, oracle, 120, e, n_chains=2, steps=[300, 700], seed=1, bounds=(mn, mx))
    st = s.stats()
    assert (st["accepted"] < st["sampled"]).all()
    assert (st["accepted"] > st["n_empty"]).all()


def test_trajectory_only_flips_and_only_double_moves(fcm, oracle):
    from flag_complex_mcmc_amd import graphs
    e = graphs.random_with_p(100, 0.2, seed=6)
    _run_parity(fcm, oracle, 100, e, n_chains=2, steps=[400], seed=3, weights=(1.0, 0.0, 0.0, 0.0))
    _run_parity(fcm, oracle, 100, e, n_chains=2, steps=[400], seed=3, weights=(0.0, 1.0, 0.0, 0.0))
    _run_parity(fcm, oracle, 100, e, n_chains=2, steps=[400], seed=3, weights=(0.2, 0.8, 0.0, 0.0))


def test_new_top_dimension_and_count_len(fcm, oracle):
    """ex02-like growth: a flip can create a simplex of a dimension not present
    before; flag_count grows and never shrinks (src/lib.rs:72-74)."""
    # 0->1,0->2,1->2 plus 3 joined so that flipping creates / destroys 3-simplices
    e = np.array([(0, 1), (0, 2), (1, 2), (0, 3), (1, 3), (3, 2)], np.uint32)
    go = oracle.Graph.from_edges(4, e)
    fc = go.flagser_count()
    # generous hand-made bounds so every move is accepted
    mn = [4, 6, 0, 0]
    mx = [4, 6, 100, 100, 100]
    _run_parity(fcm, oracle, 4, e, n_chains=8, steps=[1, 1, 1, 5, 50], seed=2, weights=(1.0, 0.0, 0.0, 0.0), bounds=(mn, mx))


def test_chain_identity_is_independent_of_batching(fcm, oracle):
    """Chain c's stream is (seed, first_chain_id + c): splitting chains over
    handles (= over GPUs) must not change any trajectory."""
    from flag_complex_mcmc_amd import graphs
    e = graphs.random_with_p(100, 0.15, seed=8)
    gg, go, b_g, b_o = setup_pair(fcm, oracle, 100, e)
    whole = fcm.MCMCSampler(gg, b_g, n_chains=6, seed=4)
    lo = fcm.MCMCSampler(gg, b_g, n_chains=3, seed=4, first_chain_id=0)
    hi = fcm.MCMCSampler(gg, b_g, n_chains=3, seed=4, first_chain_id=3)
    for s in (whole, lo, hi):
        s.step(777)
    w = whole.flag_counts()
    assert (w[:3] == lo.flag_counts()).all() and (w[3:] == hi.flag_counts()).all()
    for c in range(3):
        assert (whole.edges(c) == lo.edges(c)).all() and (whole.edges(3 + c) == hi.edges(c)).all()
    # one launch of 777 == 777 launches of 1 (state is carried in HBM between launches)
    one = fcm.MCMCSampler(gg, b_g, n_chains=2, seed=4)
    for _ in range(130):
        one.step(1)
    ref = fcm.MCMCSampler(gg, b_g, n_chains=2, seed=4)
    ref.step(130)
    assert (one.flag_counts() == ref.flag_counts()).all() and (one.edges(1) == ref.edges(1)).all()


def test_next_uses_sample_distance(fcm, oracle):
    n, e = load_flag_fixture("counterexample_any_order.flag")
    gg, go, b_g, b_o = setup_pair(fcm, oracle, n, e)
    s = fcm.MCMCSampler(gg, b_g, n_chains=2, seed=0)
    assert s.sample_distance == oracle.default_sample_distance(18)          # src/bin/sample.rs:102
    s.next()
    assert (s.sampled == s.sample_distance).all()
    tw = oracle.Chain(go, b_o, seed=0, chain_id=1, sample_distance=s.sample_distance)
    tw.next()
    compare_chain(s, 1, tw)
    assert abs(s.acceptance_ratio()[1] - tw.acceptance_ratio()) == 0.0


def test_initialize_new_sampler_end_to_end(fcm, oracle, golden_dir):
    import os
    path = os.path.join(golden_dir, "bug_calc_relax_de.flag")
    s = fcm.initialize_new_sampler(path, target_relaxation=0.01, seed=3, n_chains=2, simple=True, sample_distance=200)
    go = oracle.Graph.read_flag_file(path)
    st = oracle.State(go)
    bo, _ = oracle.bounds_calculate(st, oracle.target_bounds(st.flag_count, 0.01))
    assert (s.bounds.flag_count_min, s.bounds.flag_count_max) == bo.lists()
    s.next()
    tw = oracle.Chain(go, bo, seed=3, chain_id=0, sample_distance=200)
    tw.next()
    compare_chain(s, 0, tw)


def test_unsupported_inputs_fail_loudly(fcm):
    from flag_complex_mcmc_amd import graphs
    g = fcm.Graph.from_edges(4, graphs.simplex(3))
    b = fcm.Bounds([4, 6, 0, 0], [4, 6, 9, 9])
    with pytest.raises(fcm.FcmError):
        fcm.MCMCSampler(g, b, move_weights=(0, 0, 0, 0))
    # more reachable dimensions than the 16 tracked count entries
    big = fcm.Graph.from_edges(20, graphs.simplex(19))
    with pytest.raises(fcm.FcmError) as ei:
        fcm.MCMCSampler(big, fcm.Bounds([20], [20]))
    assert ei.value.code == 4
    # a common neighbourhood beyond 1022 vertices, with either move mix
    t = 1100
    book = [(0, 1)] + [(0, w) for w in range(2, t)] + [(1, w) for w in range(2, t)]
    with pytest.raises(fcm.FcmError) as ei:
        fcm.MCMCSampler(fcm.Graph.from_edges(t, book), fcm.Bounds([t], [t]))
    assert ei.value.code == 4
    with pytest.raises(fcm.FcmError) as ei:
        fcm.MCMCSampler(fcm.Graph.from_edges(t, book), fcm.Bounds([t], [t]), move_weights=fcm.MOVE_DISTRIBUTION)
    assert ei.value.code == 4


def _book_graph(t, p_page, seed):
    """Edge {0,1} whose common neighbourhood is every other vertex ("pages"),
    random orientations, sparse random edges among the pages: local sets of
    t vertices with low dimension -- exercises the multi-word mask paths."""
    rng = np.random.default_rng(seed)
    e = [(0, 1)]
    for w in range(2, t):
        for a in (0, 1):
            r = rng.random()
            if r < 0.45:
                e.append((a, w))
            elif r < 0.9:
                e.append((w, a))
            else:
                e += [(a, w), (w, a)]
    for i in range(2, t):
        for j in range(2, t):
            if i != j and rng.random() < p_page:
                e.append((i, j))
    return np.array(e, np.uint32)


@pytest.mark.parametrize("t,p_page", [(60, 0.04), (63, 0.04), (64, 0.04), (65, 0.04), (66, 0.04), (100, 0.03), (129, 0.02), (200, 0.012)])
def test_wide_neighbourhoods(fcm, oracle, t, p_page):
    """|N(a) cap N(b)| + 2 = t around and above 64.  t > 64 needs two or four mask
    words per local vertex (wide path); t <= 64 fits one word, but the per-class
    copies of the many multi-class vertices of this graph do not fit in 64 nodes,
    which exercises the fast path's fallback to the wide one."""
    e = _book_graph(t, p_page, seed=t)
    go = oracle.Graph.from_edges(t, e)
    assert fcm.Graph.from_edges(t, e).flagser_count() == go.flagser_count()
    s, tw = _run_parity(fcm, oracle, t, e, n_chains=2, steps=[64, 2000], seed=t, relaxation=0.3)
    assert s.info["k_max"] == t - 2
    st = s.stats()
    assert (st["n_flip"] > 0).all() and (st["n_dmove"] > 0).all()


@pytest.mark.parametrize("t,p_page,mw", [(300, 0.008, "1"), (600, 0.004, "1"), (600, 0.004, "8"), (1000, 0.0015, "2")])
def test_local_sets_beyond_256_vertices(fcm, oracle, monkeypatch, t, p_page, mw):
    """the reference recounts whatever neighbourhood a pair has (src/lib.rs:62-71).  The pair {0,1} of
    this graph has every other vertex as a common neighbour: local sets of 300, 600 and 1000 vertices, which take the
    evaluator with its masks in the chain's HBM workspace (fcm_xwide.hpp), in the one-wave kernel and under the token of
    the multi-wave kernel.  Oracle twins, tolerance 0; the hub pair is proposed a few times per chain."""
    monkeypatch.setenv("FCM_MW", mw)
    e = _book_graph(t, p_page, seed=t)
    go = oracle.Graph.from_edges(t, e)
    s, tw = _run_parity(fcm, oracle, t, e, n_chains=3, steps=[64, 3000], seed=t + 1, relaxation=0.3)
    assert s.info["k_max"] == t - 2 and s.info["waves_per_chain"] == int(mw)
    st = s.stats()
    assert (st["status"] == 0).all() and st["n_wide"].sum() > 0, "the hub pair was never evaluated"


@pytest.mark.parametrize("t,p_page", [(300, 0.008), (700, 0.003)])
def test_clique_moves_with_local_sets_beyond_256_vertices(fcm, oracle, monkeypatch, t, p_page):
    """The same hub pair under the reference's default move mix: {0, 1, page} are maximal cliques, so clique moves change
    the pair {0, 1} itself, whose local set of t vertices takes the HBM-workspace evaluator inside a clique move
    (fcm_xwide.hpp: xw_edge).  Oracle twins, tolerance 0.  The one-wave kernel's variants 6_2 / 14_2 (FCM_CQ=0; the
    cooperative kernel's deferred pairs: tests/test_clique_cooperative.py)."""
    monkeypatch.setenv("FCM_CQ", "0")
    e = _book_graph(t, p_page, seed=t)
    s, tw = _run_parity(fcm, oracle, t, e, n_chains=3, steps=[64, 1500], seed=t + 2, weights=(0.0, 0.0, 0.75, 0.25), relaxation=0.3)   # clique moves only: every wide evaluation is a clique move's
    assert s.info["k_max"] == t - 2 and s.info["waves_per_chain"] == 1
    st = s.stats()
    assert (st["status"] == 0).all() and st["n_cperm"].sum() + st["n_cswap"].sum() > 0
    assert st["n_wide"].sum() > 0, "no clique move changed the hub pair"


@pytest.mark.parametrize("t", [202, 258, 602, 1026])
def test_count_wide_common_out_neighbourhood(fcm, oracle, t):
    # 0->1, 0->w, 1->w for t-2 pages: out(0) & out(1) has t-2 vertices (beyond 256: the counter's second pass)
    e = [(0, 1)] + [(0, w) for w in range(2, t)] + [(1, w) for w in range(2, t)]
    rng = np.random.default_rng(0)
    e += [(int(i), int(j)) for i in range(2, t) for j in range(2, t) if i != j and rng.random() < 4.0 / t]
    e = np.array(e, np.uint32)
    assert fcm.Graph.from_edges(t, e).flagser_count() == oracle.Graph.from_edges(t, e).flagser_count()
    # one more page than the counter takes: refused, not truncated
    if t == 1026:
        e2 = np.concatenate([e, np.array([(0, t), (1, t)], np.uint32)])
        with pytest.raises(fcm.FcmError) as ei:
            fcm.Graph.from_edges(t + 1, e2).flagser_count()
        assert ei.value.code == 4


# -------------------------------------------------- edgebits (src/io.rs) -----
def test_count_second_pass_refuses_dimensions_it_does_not_track(fcm, oracle):
    """A directed 17-clique (a simplex of dimension 16) whose first edge has 300 common out-neighbours: only the
    counter's second pass (fcm_count_xw_kernel, 257..1024 common out-neighbours) walks it, and it must refuse like the
    first pass does -- count vectors hold dimensions 0..15 -- instead of returning truncated counts.  One vertex fewer
    (dimension 15) is counted, and equals the oracle."""
    def graph(t):
        w = list(range(2, 302))
        e = [(0, 1)] + [(0, x) for x in w] + [(1, x) for x in w] + [(w[i], w[j]) for i in range(t) for j in range(i + 1, t)]
        return 302, np.array(e, np.uint32)
    n, e = graph(14)
    fc = fcm.Graph.from_edges(n, e).flagser_count()
    assert len(fc) == 16 and fc[15] == 1 and fc == oracle.Graph.from_edges(n, e).flagser_count()
    n, e = graph(15)
    with pytest.raises(fcm.FcmError) as ei:
        fcm.Graph.from_edges(n, e).flagser_count()
    assert ei.value.code == fcm._ffi.ERR_UNSUPPORTED and "dimension" in str(ei.value)


def test_edgebits_layout(fcm, tmp_path):
    from flag_complex_mcmc_amd import graphs
    e = graphs.random_with_p(40, 0.3, seed=1)
    g = fcm.Graph.from_edges(40, e)
    fc = g.flagser_count()
    b = fcm.Bounds.calculate(g, fc, fcm.Bounds.target(fc, 0.05))
    s = fcm.MCMCSampler(g, b, n_chains=2, seed=1)
    s.step(300)
    for c in range(2):
        cur = s.graph(c)
        # slot list per src/io.rs:152-159: every edge and its reverse, sorted by (max,min,a<b), dedup
        slots = sorted({(int(a), int(b)) for a, b in e} | {(int(b), int(a)) for a, b in e},
                       key=lambda ab: (max(ab), min(ab), ab[0] < ab[1]))
        bits = [1 if cur.has_edge(a, bb) else 0 for a, bb in slots]
        want = np.packbits(np.array(bits, np.uint8), bitorder="little")
        assert (s.edgebits(c) == want).all()
    out = fcm.BitOutput(g, str(tmp_path / "lab-001"))
    out.save(s, 0)
    out.save(s, 1)
    out.flush()
    data = (tmp_path / "lab-001" / "0.edgebits").read_bytes()
    assert len(data) == 2 * len(s.edgebits(0)) and (tmp_path / "lab-001" / "graph.flag").exists()


# ----------------------------------------- full-size, size-independent checks -
def test_full_size_config3_invariants(fcm):
    """BASELINE config 3 shape (ER n=1000 p=0.10), many chains, long enough that
    the oracle is not run: check properties that hold at any size.
      * incrementally maintained counts == from-scratch recount of the final graph
      * count[0], count[1] and pr(G) unchanged (reference README.md:3)
      * final counts within the bounds
      * sampled == requested; accepted <= sampled; empty+flip+dmove == sampled."""
    from flag_complex_mcmc_amd import graphs
    n = 1000
    e = graphs.random_with_p(n, 0.10, 0)
    s = fcm.initialize_new_sampler(fcm.Graph.from_edges(n, e), n_chains=256, seed=0)
    s.step(2000)
    st = s.stats()
    assert (st["sampled"] == 2000).all() and (st["accepted"] <= st["sampled"]).all()
    assert (st["n_empty"] + st["n_flip"] + st["n_dmove"] == 2000).all()
    counts = s.flag_counts()
    und0 = fcm.Graph.from_edges(n, e).undirected_edges()
    for c in (0, 100, 255):
        g = s.graph(c)
        fc = s.flag_count(c)
        while fc[-1] == 0:   # flag_count never shrinks in length (src/lib.rs:72-74)
            fc.pop()
        assert g.flagser_count() == fc
        assert (g.undirected_edges() == und0).all() and g.nedges() == len(e)
        assert s.bounds.check(s.flag_count(c))
    assert (counts[:, 0] == n).all() and (counts[:, 1] == len(e)).all()
    assert len({tuple(r) for r in counts.tolist()}) > 200   # chains diverged


def test_config4_scale_parity_with_oracle(fcm, oracle):
    """BASELINE config 4 shape (ER n=4000 p=0.05, rows of 512 B): two chains
    against their oracle twins, and the initial count against the oracle."""
    from flag_complex_mcmc_amd import graphs
    n = 4000
    e = graphs.random_with_p(n, 0.05, 0)
    gg, go, b_g, b_o = setup_pair(fcm, oracle, n, e)
    assert gg.flagser_count() == go.flagser_count()
    s = fcm.MCMCSampler(gg, b_g, n_chains=2, seed=3, first_chain_id=5)
    tw = [oracle.Chain(go, b_o, seed=3, chain_id=5 + c) for c in range(2)]
    for nstep in (70, 330):
        s.step(nstep)
        for c in range(2):
            tw[c].step(nstep)
            compare_chain(s, c, tw[c], ctx=("config4", c, nstep))
    assert s.info["row_words"] == 64 and s.info["k_max"] > 62   # wide-path edges exist in this graph


def test_config5_scale_invariants(fcm):
    """BASELINE config 5 shape (n=30000, 1M directed edge draws; rows of 3840 B,
    115 MB of bitmap per chain): size-independent checks, 4 chains."""
    from flag_complex_mcmc_amd import graphs
    n = 30000
    e = graphs.random_edge_draws(n, 1000000, 0)
    g = fcm.Graph.from_edges(n, e)
    s = fcm.initialize_new_sampler(g, n_chains=4, seed=1)
    s.step(3000)
    st = s.stats()
    assert (st["sampled"] == 3000).all() and (st["n_empty"] + st["n_flip"] + st["n_dmove"] == 3000).all()
    assert (st["n_dmove"] > 0).all()
    und0 = g.undirected_edges()
    for c in (0, 3):
        cur = s.graph(c)
        fc = s.flag_count(c)
        while fc[-1] == 0:
            fc.pop()
        assert cur.flagser_count() == fc
        assert cur.nedges() == len(e) and (cur.undirected_edges() == und0).all()
        assert s.bounds.check(s.flag_count(c))
    assert not (s.edges(0) == s.edges(3)).all()


def test_truncated_mode_tracks_only_the_capped_dimensions(fcm, oracle):
    """dim_cap (BASELINE configs name caps 5/6/7; the reference has none):
    dimensions above the cap are not tracked and not bounds-checked; the tracked
    ones still equal a full recount as long as both runs accept the same moves."""
    n, e = load_flag_fixture("bug_calc_relax_de.flag")
    g = fcm.Graph.from_edges(n, e)
    full = fcm.initialize_new_sampler(g, n_chains=2, seed=2)
    cap = fcm.initialize_new_sampler(g, n_chains=2, seed=2, dim_cap=4)
    assert full.info["lossless"] == 1 and full.ncounts == 8
    assert cap.info["lossless"] == 0 and cap.ncounts == 5
    cap.step(400)
    for c in range(2):
        recount = cap.graph(c).flagser_count()
        assert recount[:5] == cap.flag_count(c)[:5]


# ------------------------------------------- checkpoint / resume (src/io.rs:51-62)
def test_save_state_load_state_resumes_exactly(fcm, oracle, tmp_path):
    from flag_complex_mcmc_amd import graphs
    e = graphs.random_with_p(150, 0.15, seed=2)
    gg, go, b_g, b_o = setup_pair(fcm, oracle, 150, e)
    a = fcm.MCMCSampler(gg, b_g, n_chains=5, seed=7, first_chain_id=11, sample_distance=250)
    a.next()
    path = tmp_path / "sampler-lab-007.state"
    a.save_state(str(path), sample_number=42)
    assert path.exists() and not (tmp_path / "sampler-lab-007.state.tmp").exists()   # tmp-then-rename (io.rs:52-54)
    n, b = fcm.MCMCSampler.load_state(str(path))
    assert n == 42 and b.n_chains == 5 and b.sample_distance == 250
    assert (b.bounds.flag_count_min, b.bounds.flag_count_max) == (b_g.flag_count_min, b_g.flag_count_max)
    assert (a.flag_counts() == b.flag_counts()).all()
    a.next()
    b.next()
    straight = fcm.MCMCSampler(gg, b_g, n_chains=5, seed=7, first_chain_id=11, sample_distance=250)
    straight.step(500)
    for s in (a, b):
        assert (s.flag_counts() == straight.flag_counts()).all()
        for k, v in straight.stats().items():
            if k not in ("n_redo", "n_recheck", "n_held"):   # timing diagnostics of the multi-wave kernel (proposals re-run under the token, records re-checked or waited for), not chain state
                assert (s.stats()[k] == v).all(), k
        for c in range(5):
            assert (s.edges(c) == straight.edges(c)).all()
            assert (s.double_slots(c) == straight.double_slots(c)).all()
    tw = oracle.Chain(go, b_o, seed=7, chain_id=13)
    tw.step(500)
    compare_chain(b, 2, tw)


def test_load_state_rejects_garbage(fcm, tmp_path):
    p = tmp_path / "x.state"
    p.write_bytes(b"not a state file at all" * 10)
    with pytest.raises(fcm.FcmError) as ei:
        fcm.MCMCSampler.load_state(str(p))
    assert ei.value.code == 5
    with pytest.raises(fcm.FcmError):
        fcm.MCMCSampler.load_state(str(tmp_path / "missing.state"))


# ------------------------------------------- the `sample` CLI (src/bin/sample.rs)
def test_sample_cli_matches_library_and_resumes(fcm, golden_dir, tmp_path):
    import os, subprocess
    exe = os.path.join(os.path.dirname(fcm.LIB_PATH), "sample")
    assert os.path.exists(exe), "build it with make -C flag_complex_mcmc_amd/csrc"
    flag = os.path.join(golden_dir, "bug_calc_relax_de.flag")
    sdir, stdir = tmp_path / "samples", tmp_path / "state"
    base = [exe, "-i", flag, "-l", "lab", "-s", "4", "--simple", "--chains", "3", "--sample-distance", "150",
            "--samples-store-dir", str(sdir), "--state-store-dir", str(stdir), "--state-save-interval", "2"]
    r = subprocess.run(base + ["-n", "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "The sampling distance was set to 150." in r.stdout and r.stdout.count("flag count: [") == 3
    state = stdir / "sampler-lab-004.state"
    assert state.exists()
    # the same three samples through the library
    s = fcm.initialize_new_sampler(flag, target_relaxation=0.01, seed=4, n_chains=3, sample_distance=150)
    recs = {c: b"" for c in range(3)}
    for _ in range(3):
        s.next()
        for c in range(3):
            recs[c] += s.edgebits(c).tobytes()
    for c in range(3):
        d = sdir / "lab-004" / ("chain%05d" % c)
        assert (d / "graph.flag").exists()
        assert (d / "0.edgebits").read_bytes() == recs[c]
    assert "flag count: %s" % s.flag_count(0) in r.stdout
    # resume (-c) for two more samples == five samples straight
    r2 = subprocess.run([exe, "-c", str(state), "-l", "lab", "-s", "4", "-n", "2", "--samples-store-dir", str(tmp_path / "s2"),
                         "--state-store-dir", str(stdir)], capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0, r2.stdout + r2.stderr
    s.next(); s.next()
    assert "flag count: %s" % s.flag_count(0) in r2.stdout
    n, resumed = fcm.MCMCSampler.load_state(str(state))
    assert n == 5 and (resumed.flag_counts() == s.flag_counts()).all()
    # without --simple the CLI runs the reference's default mix with the clique moves (sample.rs:17,101)
    r3 = subprocess.run([exe, "-i", flag, "-l", "dflt", "-s", "1", "-n", "2", "--chains", "2", "--sample-distance", "120",
                         "--samples-store-dir", str(tmp_path / "s3"), "--state-store-dir", str(stdir)],
                        capture_output=True, text=True, timeout=300)
    assert r3.returncode == 0, r3.stdout + r3.stderr
    d = fcm.initialize_new_sampler(flag, seed=1, n_chains=2, sample_distance=120, simple=False)
    d.next(); d.next()
    assert (d.stats()["n_cperm"] > 0).all()
    assert "flag count: %s" % d.flag_count(0) in r3.stdout
    # ... and its state file resumes with the slot index rebuilt
    n2, res2 = fcm.MCMCSampler.load_state(str(stdir / "sampler-dflt-001.state"))
    res2.next(); d.next()
    assert n2 == 2 and (res2.flag_counts() == d.flag_counts()).all() and (res2.edges(1) == d.edges(1)).all()


# ----------------------------------- clique moves (src/lib.rs:214-290)
CLIQUE_WEIGHTS = [(0.0, 0.0, 1.0, 0.0), (0.0, 0.0, 0.0, 1.0), (0.1, 0.1, 0.6, 0.2)]


@pytest.mark.parametrize("weights", CLIQUE_WEIGHTS)
def test_clique_moves_trajectory_parity(fcm, oracle, weights):
    from flag_complex_mcmc_amd import graphs
    # dense small graph: big cliques, many reciprocal pairs, overlapping cliques for clique_swap
    e = graphs.random_with_p(40, 0.35, seed=1)
    s, tw = _run_parity(fcm, oracle, 40, e, n_chains=3, steps=[1, 1, 1, 61, 200], seed=4, weights=weights, relaxation=0.2)
    st = s.stats()
    if weights[2] > 0:
        assert (st["n_cperm"] > 0).all()
    if weights[3] > 0:
        assert (st["n_cswap"] > 0).all()
    assert (st["n_changes"] > 0).all() and (st["accepted"] < st["sampled"]).any()
    # sparser, larger: cliques of order 3-5
    e = graphs.random_with_p(150, 0.12, seed=2)
    _run_parity(fcm, oracle, 150, e, n_chains=2, steps=[300], seed=5, weights=weights)


def test_clique_moves_on_fixture_and_invariants(fcm, oracle):
    n, e = load_flag_fixture("bug_calc_relax_de.flag")
    s, tw = _run_parity(fcm, oracle, n, e, n_chains=2, steps=[400], seed=6, weights=(0.1, 0.1, 0.6, 0.2))
    und0 = fcm.Graph.from_edges(n, e).undirected_edges()
    for c in range(2):
        g = s.graph(c)
        fc = s.flag_count(c)
        while fc[-1] == 0:
            fc.pop()
        assert g.flagser_count() == fc and (g.undirected_edges() == und0).all() and g.nedges() == len(e)


def test_default_mix_config_scale_oracle_twins(fcm, oracle):
    """the reference's default move mix on the bench graph itself (ER n = 1000, p = 0.10, seed 0: 77 % clique
    moves of about a dozen directed changes each), two chains x 500 proposals against oracle twins, tolerance 0."""
    from flag_complex_mcmc_amd import graphs
    n = 1000
    e = graphs.random_with_p(n, 0.10, 0)
    s, tw = _run_parity(fcm, oracle, n, e, n_chains=2, steps=[1, 63, 436], seed=12, weights=(0.1, 0.1, 0.6, 0.2), relaxation=0.01)
    st = s.stats()
    assert (st["status"] == 0).all() and (st["n_cperm"] > 200).all() and (st["n_cswap"] > 50).all() and (st["n_changes"] > 3000).all()


def test_default_mix_config3_invariants(fcm):
    """The reference's default move mix [0.1,0.1,0.6,0.2] (src/bin/sample.rs:17) on
    the config-3 graph, 64 chains: size-independent checks."""
    from flag_complex_mcmc_amd import graphs
    n = 1000
    e = graphs.random_with_p(n, 0.10, 0)
    s = fcm.initialize_new_sampler(fcm.Graph.from_edges(n, e), n_chains=64, seed=0, simple=False)
    s.step(300)
    st = s.stats()
    assert (st["sampled"] == 300).all()
    assert (st["n_empty"] + st["n_flip"] + st["n_dmove"] + st["n_cperm"] + st["n_cswap"] == 300).all()
    assert (st["n_cperm"] > 100).all() and (st["n_cswap"] > 20).all()
    und0 = fcm.Graph.from_edges(n, e).undirected_edges()
    for c in (0, 63):
        g = s.graph(c)
        fc = s.flag_count(c)
        while fc[-1] == 0:
            fc.pop()
        assert g.flagser_count() == fc
        assert (g.undirected_edges() == und0).all() and g.nedges() == len(e)
        assert s.bounds.check(s.flag_count(c))


def _special_graphs():
    """Small graphs that stress structural corner cases of the moves."""
    rng = np.random.default_rng(11)
    out = {}
    # triangle-free: maximal cliques are edges (order 2) and isolated vertices (order 1)
    e = [(i, i + 1) for i in range(0, 14)] + [(3, 2), (7, 6), (14, 0), (0, 14)]
    out["cycle_with_doubles_and_isolated"] = (18, np.array(e, np.uint32))
    # two disjoint 5-cliques (clique_swap with an empty intersection) plus a bridge
    e = []
    for base in (0, 5):
        for i in range(5):
            for j in range(i):
                r = rng.random()
                e += [(base + i, base + j)] if r < 0.45 else ([(base + j, base + i)] if r < 0.9 else [(base + i, base + j), (base + j, base + i)])
    e += [(4, 5)]
    out["two_disjoint_5_cliques"] = (10, np.array(e, np.uint32))
    # one 6-clique made of reciprocal pairs only (every permutation is the identity move)
    e = [(i, j) for i in range(6) for j in range(6) if i != j]
    out["all_reciprocal_6_clique"] = (6, np.array(e, np.uint32))
    # overlapping cliques sharing a 3-clique core (clique_swap with a large intersection)
    e = []
    core = [0, 1, 2]
    groups = [core + [3, 4], core + [5, 6], core + [7, 8]]
    seen = set()
    for g in groups:
        for i in g:
            for j in g:
                if i < j and (i, j) not in seen:
                    seen.add((i, j))
                    r = rng.random()
                    e += [(i, j)] if r < 0.4 else ([(j, i)] if r < 0.8 else [(i, j), (j, i)])
    out["three_5_cliques_sharing_a_triangle"] = (9, np.array(e, np.uint32))
    return out


@pytest.mark.parametrize("name", sorted(_special_graphs()))
@pytest.mark.parametrize("weights", [(0.25, 0.25, 0.25, 0.25), (0.0, 0.0, 0.5, 0.5)])
def test_special_graphs_all_moves_parity(fcm, oracle, name, weights):
    n, e = _special_graphs()[name]
    go = oracle.Graph.from_edges(n, e)
    fc = go.flagser_count()
    # hand-made bounds: generous on every dimension the undirected cliques allow, so moves are mostly accepted
    # but some are rejected (tight on dimension 2)
    ncl = oracle.State(go).clique_counts()
    width = max(len(ncl), len(fc)) + 1
    mn = [fc[0], fc[1]] + [0] * (width - 2)
    mx = [fc[0], fc[1]] + [10 ** 6] * (width - 2)
    if len(fc) > 2:
        mn[2], mx[2] = max(0, fc[2] - 6), fc[2] + 6
    s, tw = _run_parity(fcm, oracle, n, e, n_chains=3, steps=[1, 2, 61, 64, 130], seed=8, weights=weights, bounds=(mn, mx))
    st = s.stats()
    assert (st["sampled"] == 258).all()
    assert (st["n_empty"] + st["n_flip"] + st["n_dmove"] + st["n_cperm"] + st["n_cswap"] == 258).all()


def test_single_chain_and_odd_launch_sizes(fcm, oracle):
    from flag_complex_mcmc_amd import graphs
    e = graphs.random_with_p(90, 0.2, seed=12)
    # n_chains = 1; launches of 0, 1, 63, 64, 65, 127 proposals; default mix
    _run_parity(fcm, oracle, 90, e, n_chains=1, steps=[0, 1, 63, 64, 65, 127, 0, 3], seed=21, weights=(0.1, 0.1, 0.6, 0.2), relaxation=0.05)


# ---- the multi-wave kernel (W waves per chain, in-order commit): simple moves --------------------------
def test_multi_wave_kernel_is_selected_and_switchable(fcm, monkeypatch):
    from flag_complex_mcmc_amd import graphs
    e = graphs.random_with_p(500, 0.12, seed=3)
    g = fcm.Graph.from_edges(500, e)
    fc = g.flagser_count()
    b = fcm.Bounds.calculate(g, fc, fcm.Bounds.target(fc, 0.05))
    assert fcm.MCMCSampler(g, b, n_chains=2, seed=1).info["waves_per_chain"] == 8         # few chains: many waves each
    eb = graphs.random_with_p(1500, 0.02, seed=3)                                          # rows longer than a cache line: 16 up to 512 chains
    gb = fcm.Graph.from_edges(1500, eb)
    fcb = gb.flagser_count()
    bb = fcm.Bounds.calculate(gb, fcb, fcm.Bounds.target(fcb, 0.05))
    assert fcm.MCMCSampler(gb, bb, n_chains=2, seed=1).info["waves_per_chain"] == (16 if len(fcb) - 2 >= 2 else 1)
    assert fcm.MCMCSampler(gb, bb, n_chains=257, seed=1).info["waves_per_chain"] == (8 if len(fcb) - 2 >= 2 else 1)   # (a sparse graph: few lines per build)
    assert fcm.MCMCSampler(g, b, n_chains=2048, seed=1).info["waves_per_chain"] == 4
    assert fcm.MCMCSampler(g, b, n_chains=1024, seed=1).info["waves_per_chain"] == 8
    assert fcm.MCMCSampler(g, b, n_chains=3000, seed=1).info["waves_per_chain"] == 2      # 3000 x 4 would not all be resident
    assert fcm.MCMCSampler(g, b, n_chains=4096, seed=1).info["waves_per_chain"] == 2      # chains x W = the chip's 8192 wave slots
    assert fcm.MCMCSampler(g, b, n_chains=2, seed=1, move_weights=fcm.MOVE_DISTRIBUTION).info["waves_per_chain"] == 8     # clique moves: the cooperative kernel, W by chain count
    assert fcm.MCMCSampler(g, b, n_chains=4096, seed=1, move_weights=fcm.MOVE_DISTRIBUTION).info["waves_per_chain"] == 1  # ... and the one-wave kernel beyond 2048 chains
    monkeypatch.setenv("FCM_MW", "1")
    assert fcm.MCMCSampler(g, b, n_chains=2, seed=1).info["waves_per_chain"] == 1
    monkeypatch.setenv("FCM_MW", "4")
    assert fcm.MCMCSampler(g, b, n_chains=2, seed=1).info["waves_per_chain"] == 4


@pytest.mark.parametrize("W", [2, 4, 8, 16])
def test_multi_wave_conflict_paths_trajectory_parity(fcm, oracle, monkeypatch, W):
    """Tiny graphs: consecutive proposals hit the same pair, the same slot of the reciprocal list or each other's local
    sets all the time, so a good share of the proposals is found in conflict with a commit that was in flight and is
    run again under the token (n_redo), for every W."""
    from flag_complex_mcmc_amd import graphs
    monkeypatch.setenv("FCM_MW", str(W))
    redo = recheck = held = 0
    for n, pr, gseed in ((12, 0.3, 5), (20, 0.3, 3), (9, 0.45, 7)):
        e = graphs.random_with_p(n, pr, seed=gseed)
        for w in ((0.5, 0.5, 0.0, 0.0), (0.0, 1.0, 0.0, 0.0), (1.0, 0.0, 0.0, 0.0)):
            s, tw = _run_parity(fcm, oracle, n, e, n_chains=4, steps=[1, 2, 3, 61, 64, 700], seed=21, weights=w, relaxation=0.3)
            assert s.info["waves_per_chain"] == W
            assert (s.stats()["status"] == 0).all()
            redo += int(s.stats()["n_redo"].sum())
            recheck += int(s.stats()["n_recheck"].sum())
            held += int(s.stats()["n_held"].sum())
    assert redo > 100
    if W >= 4:   # the staged-record protocol: both of its rare paths ran (and the trajectories above are the oracle's)
        assert recheck > 0 and held > 0, (recheck, held)


@pytest.mark.parametrize("cfg,chains,props", [(2, 96, 1 << 15), (3, 48, 1 << 14)])
def test_multi_wave_soak_every_W_ends_where_one_wave_ends(fcm, cfg, chains, props):
    """tools/mw_soak.py: BASELINE graphs (n = 1000: exact lookup maps; n = 4000: supersets), W = 2, 4, 8, 16 against the
    one-wave kernel -- counts, counters, slot lists and whole bitmaps of three chains after tens of thousands of proposals
    per chain, with thousands of records re-checked and staged conflicts waited for on the way."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("mw_soak", os.path.join(os.path.dirname(__file__), "..", "tools", "mw_soak.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.soak(cfg, chains, props, say=lambda m: None)


@pytest.mark.parametrize("W", [2, 16])
def test_multi_wave_equals_one_wave_at_scale(fcm, monkeypatch, W):
    """Config-3 graph (local sets of 65 vertices included: the wide path under the token): the multi-wave kernel must
    leave the chains the one-wave kernel leaves."""
    from flag_complex_mcmc_amd import graphs
    n = 1000
    e = graphs.random_with_p(n, 0.10, seed=0)
    g = fcm.Graph.from_edges(n, e)
    fc = g.flagser_count()
    b = fcm.Bounds.calculate(g, fc, fcm.Bounds.target(fc, 0.01))
    monkeypatch.setenv("FCM_MW", str(W))
    s2 = fcm.MCMCSampler(g, b, n_chains=96, seed=3)
    monkeypatch.setenv("FCM_MW", "1")
    s1 = fcm.MCMCSampler(g, b, n_chains=96, seed=3)
    assert s2.info["waves_per_chain"] == W and s1.info["waves_per_chain"] == 1
    for nstep in (1, 63, 1000, 3000):
        s1.step(nstep)
        s2.step(nstep)
        st1, st2 = s1.stats(), s2.stats()
        for k in ("sampled", "accepted", "n_empty", "n_flip", "n_dmove", "sum_k", "status", "count_len"):
            assert (st1[k] == st2[k]).all(), (nstep, k)
        for c in range(96):
            assert s1.flag_count(c) == s2.flag_count(c), (nstep, c)
    for c in (0, 17, 95):
        assert (s1.edges(c) == s2.edges(c)).all()
        assert (s1.double_slots(c) == s2.double_slots(c)).all()


def test_multi_wave_long_rows_oracle_twins(fcm, oracle, monkeypatch):
    """Rows longer than a cache line (n = 1500: the n-tagged kernels), a small-k and a larger-k graph, W = 8."""
    from flag_complex_mcmc_amd import graphs
    monkeypatch.setenv("FCM_MW", "8")
    for n, pr, gseed in ((1500, 0.02, 1), (1100, 0.08, 2)):
        e = graphs.random_with_p(n, pr, seed=gseed)
        s, tw = _run_parity(fcm, oracle, n, e, n_chains=3, steps=[1, 64, 2000], seed=11, relaxation=0.05)
        assert s.info["waves_per_chain"] == 8 and s.info["row_words"] > 16
        assert (s.stats()["status"] == 0).all()


# ---- the kernel bench.py times, in front of the oracle -----------------------
def _pick_chains(stats, want):
    """Chain indices whose run hit the rare paths: wide evaluations first, then REDOs, then big local sets."""
    order = np.lexsort((-stats["n_big"].astype(np.int64), -stats["n_redo"].astype(np.int64), -stats["n_wide"].astype(np.int64)))
    return [int(c) for c in order[:want]]


def test_bench_kernel_config3_oracle_twins(fcm, oracle):
    """The config-3 graph itself (ER n=1000 p=0.10 seed 0), the kernel variant bench.py times: oracle twins, tolerance 0,
    on chains chosen so that the rare paths run: local sets beyond 48 vertices (second build trip), beyond 64 (wide
    evaluator) and proposals that had to be re-run on the exact state (REDO).  The GPU picks the chains (128 chains x 3000
    proposals, counters n_big / n_wide / n_redo); the oracle then checks those chains edge for edge."""
    from flag_complex_mcmc_amd import graphs
    n, nprop = 1000, 3000
    e = graphs.random_with_p(n, 0.10, 0)
    gg, go, b_g, b_o = setup_pair(fcm, oracle, n, e)
    s = fcm.MCMCSampler(gg, b_g, n_chains=128, seed=0)
    assert s.info["waves_per_chain"] >= 2, "the headline workload must run on the multi-wave kernel"
    s.step(nprop)
    st = s.stats()
    assert (st["status"] == 0).all() and (st["sampled"] == nprop).all()
    assert st["n_big"].sum() > 0 and st["n_wide"].sum() > 0 and st["n_redo"].sum() > 0, \
        (int(st["n_big"].sum()), int(st["n_wide"].sum()), int(st["n_redo"].sum()))
    chosen = _pick_chains(st, 4)
    assert sum(int(st["n_wide"][c]) for c in chosen) > 0 and sum(int(st["n_redo"][c]) for c in chosen) > 0
    assert sum(int(st["n_big"][c]) for c in chosen) > 0
    for c in chosen:
        tw = oracle.Chain(go, b_o, seed=0, chain_id=c)
        tw.step(nprop)
        compare_chain(s, c, tw, ctx=("config3", c))
    # ... and the same chains again in a fresh sampler, launch sizes that are not multiples of anything
    s2 = fcm.MCMCSampler(gg, b_g, n_chains=2, seed=0, first_chain_id=chosen[0])
    tw = oracle.Chain(go, b_o, seed=0, chain_id=chosen[0])
    for nstep in (1, 31, 33, 1000, 1935):
        s2.step(nstep)
        tw.step(nstep)
        compare_chain(s2, 0, tw, ctx=("config3 split", nstep))
    assert s2.flag_count(0) == s.flag_count(chosen[0])


def test_bench_kernel_natural_selection_oracle_twins(fcm, oracle):
    """A graph on which the library itself selects the multi-wave kernel (n=500 p=0.12: mean neighbourhood about 25)."""
    from flag_complex_mcmc_amd import graphs
    n = 500
    e = graphs.random_with_p(n, 0.12, seed=3)
    s, tw = _run_parity(fcm, oracle, n, e, n_chains=4, steps=[3000], seed=2, relaxation=0.05, first_chain_id=40)
    assert s.info["waves_per_chain"] >= 2 and 12 <= s.info["k_mean"] <= 48
    assert (s.stats()["status"] == 0).all()


# ---- 32-bit local counts -------------------------------------------------------
@pytest.mark.parametrize("mw", ["1", "8"])
def test_local_count_guard_refuses_instead_of_wrapping(fcm, monkeypatch, mw):
    """Kernels that track 6 levels check the lanes' counts before the 32-bit wave sum, deeper ones bound every walk (arcs x
    max children^(t-2)); past 2^31 - 1 they raise a status bit.  No graph that small enough to count here gets there, so the test
    lowers the limit through the library's test hook and checks that the run then fails loudly at the next read-out,
    on both step kernels; with the real limit the same run is clean."""
    n, e = load_flag_fixture("bug_calc_relax_de.flag")     # 8 count entries: the 6-level kernels
    g = fcm.Graph.from_edges(n, e)
    monkeypatch.setenv("FCM_MW", mw)
    ok = fcm.initialize_new_sampler(g, n_chains=4, seed=1)
    assert ok.ncounts == 8
    ok.step(2000)
    assert (ok.stats()["status"] == 0).all()
    monkeypatch.setenv("FCM_TEST_GUARD_LIMIT", "40")
    bad = fcm.initialize_new_sampler(g, n_chains=4, seed=1)
    bad.step(2000)
    with pytest.raises(fcm.FcmError) as ei:
        bad.stats()
    assert ei.value.code == 8 and "0x100" in str(ei.value)


def test_deep_kernels_count_on_the_wide_path_when_the_bound_is_passed(fcm, oracle, monkeypatch):
    """Kernels that track more than 6 levels bound every walk's counts by arcs x (max children)^(t-2) -- generous: dense
    little graphs pass 2^31 by the bound long before their counts do (found by the randomized campaign,
    tests/test_fuzz_parity.py).  Such a proposal is then counted by the wide evaluator, whose counts are 64-bit, instead of
    being refused.  Here the limit is lowered through the test hook so that every proposal takes that way: oracle twins."""
    from flag_complex_mcmc_amd import graphs
    n = 22
    e = graphs.random_with_p(n, 0.6, seed=2)
    assert len(fcm.Graph.from_edges(n, e).flagser_count()) >= 9          # more than 6 tracked levels: the generic kernels
    monkeypatch.setenv("FCM_MW", "1")
    monkeypatch.setenv("FCM_TEST_GUARD_LIMIT", "40")
    s, tw = _run_parity(fcm, oracle, n, e, n_chains=2, steps=[1, 64, 800], seed=9, relaxation=0.3)
    st = s.stats()
    assert (st["status"] == 0).all() and st["n_wide"].sum() > 100
    # the same inside clique moves: a changed pair's direction that passes the bound is counted on the wide path
    s, tw = _run_parity(fcm, oracle, n, e, n_chains=2, steps=[1, 64, 100], seed=10, weights=(0.0, 0.0, 0.75, 0.25), relaxation=0.3)
    st = s.stats()
    assert (st["status"] == 0).all() and st["n_wide"].sum() > 100


def test_count_kernel_counts_past_2_to_31_per_edge(fcm):
    """Complete 7-partite graph, parts of 7, every pair reciprocal, on 49 vertices: count[d] = C(7,d+1) 7^(d+1) (d+1)!
    -- 4.2e9 6-simplices in all and far more than 2^31 / 64 per lane of the counting wave: the per-lane accumulators
    are 64 bits wide."""
    import math
    parts, size = 7, 7
    n = parts * size
    e = np.array([(a, b) for a in range(n) for b in range(n) if a // size != b // size], np.uint32)
    want = [math.comb(parts, d + 1) * size ** (d + 1) * math.factorial(d + 1) for d in range(parts)]
    assert want[6] > 2 ** 31
    assert fcm.Graph.from_edges(n, e).flagser_count() == want


def test_commit_guard_refuses_out_of_range_indices(fcm, monkeypatch):
    """The commit's flat stores (two bitmap words, one slot of the reciprocal list) take their indices out of the
    proposal's record; the kernels hold them against the chain's sizes before storing and raise status 0x200 instead
    (VERDICT r3 item 1).  The test hook FCM_TEST_COMMIT_LIMIT lowers the limit the word indices are held
    against to one word, so that every commit beyond word 0 is "out of range": nothing is stored, the run fails loudly
    with FCM_ERR_INTERNAL at the next read-out.  Multi-wave kernel (W = 2, 8), its sparse-state variant, and the simple
    moves of the cooperative clique kernel."""
    from flag_complex_mcmc_amd import graphs
    n = 300
    e = graphs.random_with_p(n, 0.12, seed=8)
    g = fcm.Graph.from_edges(n, e)
    for mw, weights in (("2", fcm.MOVE_DISTRIBUTION_SIMPLE), ("8", fcm.MOVE_DISTRIBUTION_SIMPLE), ("", fcm.MOVE_DISTRIBUTION)):
        if mw:
            monkeypatch.setenv("FCM_MW", mw)
        else:
            monkeypatch.delenv("FCM_MW", raising=False)
            monkeypatch.setenv("FCM_CQ", "1")
        monkeypatch.delenv("FCM_TEST_COMMIT_LIMIT", raising=False)
        fc = g.flagser_count()
        b = fcm.Bounds.calculate(g, fc, fcm.Bounds.target(fc, 0.01))
        ok = fcm.MCMCSampler(g, b, n_chains=3, seed=1, move_weights=weights)
        ok.step(400)
        assert (ok.stats()["status"] == 0).all() and (ok.stats()["accepted"] > 0).all()
        monkeypatch.setenv("FCM_TEST_COMMIT_LIMIT", "1")
        bad = fcm.MCMCSampler(g, b, n_chains=3, seed=1, move_weights=weights)
        before = bad.edges(0).copy()
        bad.step(400)
        with pytest.raises(fcm.FcmError) as ei:
            bad.stats()
        assert ei.value.code == 8 and "0x200" in str(ei.value), str(ei.value)
        monkeypatch.delenv("FCM_TEST_COMMIT_LIMIT", raising=False)
    # sparse state (two bits per adjacent pair): the same guard on the record's words
    monkeypatch.setenv("FCM_MW", "8")
    monkeypatch.setenv("FCM_SPARSE", "1")
    monkeypatch.delenv("FCM_CQ", raising=False)
    n2 = 1200
    e2 = graphs.random_edge_draws(n2, 9000, 3)
    g2 = fcm.Graph.from_edges(n2, e2)
    fc2 = g2.flagser_count()
    b2 = fcm.Bounds.calculate(g2, fc2, fcm.Bounds.target(fc2, 0.01))
    ok = fcm.MCMCSampler(g2, b2, n_chains=2, seed=1)
    if ok.info["sparse_state"]:
        ok.step(500)
        assert (ok.stats()["status"] == 0).all()
        monkeypatch.setenv("FCM_TEST_COMMIT_LIMIT", "1")
        bad = fcm.MCMCSampler(g2, b2, n_chains=2, seed=1)
        bad.step(500)
        with pytest.raises(fcm.FcmError) as ei:
            bad.stats()
        assert ei.value.code == 8 and "0x200" in str(ei.value)
