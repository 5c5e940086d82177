"""CPU: pin the oracle (oracle/fcm_oracle.c) against everything the reference
holds for this path, and against independent known answers.

What the reference pins: only `test_intersect` (src/util.rs:107-156).  The rest
of the oracle is "parity unpinned" against the reference program (it cannot be
built here); those parts are checked against the definition
through tests/golden/known_answers.json (made by golden/make_known_answers.py,
which shares no code with the oracle) and against the worked numbers in
SURVEY.md App. E.
"""
import itertools

import numpy as np
import pytest

from helpers import known_answers, load_flag_fixture


# --- the reference's own golden vectors: src/util.rs:107-156 ----------------
INTERSECT_CASES = [
    ([1, 2, 5, 8, 9], [0, 2, 2, 2, 3, 4, 5, 9, 9, 9], [2, 5, 9]),
    ([0, 2, 2, 2, 3, 4, 5, 9, 9, 9], [1, 2, 5, 8, 9], [2, 5, 9]),
    ([], [0, 2, 2, 2, 3, 4, 5, 9, 9, 9], []),
    ([2], [3], []),
    ([2], [0, 1, 2, 3, 4, 5], [2]),
    ([0], [0, 1, 2, 3, 4, 5], [0]),
    ([0], [0, 1, 2, 3, 4, 5], [0]),
    ([5], [0, 1, 2, 3, 4, 5], [5]),
]


@pytest.mark.parametrize("a,b,want", INTERSECT_CASES)
def test_intersect_sorted_reference_vectors(oracle, a, b, want):
    assert oracle.intersect_sorted(a, b) == want


def test_intersect_sorted_duplicates_on_both_sides(oracle):
    # advances both sides on equality: duplicates pair up (src/util.rs:13-17)
    assert oracle.intersect_sorted([2, 2, 2, 7], [2, 2, 9]) == [2, 2]


# --- Philox4x32-10 known answers (Random123 kat_vectors) ---------------------
PHILOX_KAT = [
    ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
    ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
    ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0],
     [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
]


@pytest.mark.parametrize("ctr,key,want", PHILOX_KAT)
def test_philox_known_answers(oracle, ctr, key, want):
    assert oracle.philox(ctr, key) == want


# --- simplex counter against the independent known answers -------------------
def _graph_for(oracle, name, rec):
    if name.endswith(".flag"):
        n, e = load_flag_fixture(name)
    else:
        n, e = rec["n"], np.array(rec["edges"], np.uint32)
    return oracle.Graph.from_edges(n, e), n, e


def test_counts_match_known_answers(oracle):
    for name, rec in known_answers().items():
        g, n, e = _graph_for(oracle, name, rec)
        assert g.flagser_count() == rec["flag_count"], name
        assert g.nnodes() == rec["n"] and len(g.edges()) == rec["m"], name
        # legacy entry point shape (src/flagser.rs:13-21)
        assert oracle.flagser_count_unweighted(n, e) == rec["flag_count"], name


def test_flag_reader_matches_fixture_text(oracle, golden_dir):
    import os
    for f in os.listdir(golden_dir):
        if f.endswith(".flag"):
            n, e = load_flag_fixture(f)
            g = oracle.Graph.read_flag_file(os.path.join(golden_dir, f))
            assert g.nnodes() == n
            assert sorted(map(tuple, g.edges().tolist())) == sorted(set(map(tuple, e.tolist())))


def _brute(n, edges):
    es = {(int(a), int(b)) for a, b in edges}
    counts = [n]
    for d in range(1, n):
        c = sum(1 for t in itertools.permutations(range(n), d + 1)
                if all((t[i], t[j]) in es for i in range(d + 1) for j in range(i + 1, d + 1)))
        if not c:
            break
        counts.append(c)
    return counts


def test_counts_match_bruteforce_on_random_small_graphs(oracle):
    rng = np.random.default_rng(7)
    for trial in range(12):
        n = int(rng.integers(2, 8))
        p = float(rng.choice([0.3, 0.6, 0.9]))
        adj = rng.random((n, n)) < p
        np.fill_diagonal(adj, False)
        e = np.argwhere(adj).astype(np.uint32)
        assert oracle.Graph.from_edges(n, e).flagser_count() == _brute(n, e), (trial, n, p)


# --- structural facts the reference's fixtures imply -------------------------
def test_reference_fixture_pair_properties(oracle):
    """counterexample_seo_greedy_5_{start,bad}.flag were written by
    seo_search_counterexample.rs:100-101 after a greedy run that only accepts
    flips with post[2] > pre[2] (:66-68) and stops early only when the flag
    count equals the undirected clique count (:92-96).  Hence: same pr(G), no
    reciprocal pairs (gen_seo_er), count[2] did not decrease, and `bad` did not
    reach the clique counts."""
    ka = known_answers()
    ns, es = load_flag_fixture("counterexample_seo_greedy_5_start.flag")
    nb, eb = load_flag_fixture("counterexample_seo_greedy_5_bad.flag")
    gs, gb = oracle.Graph.from_edges(ns, es), oracle.Graph.from_edges(nb, eb)
    assert (gs.undirected_edges() == gb.undirected_edges()).all()
    assert len(gs.undirected_edges()) == len(gs.edges())  # SEO
    cs, cb = gs.flagser_count(), gb.flagser_count()
    assert cb[2] >= cs[2] and len(cb) >= len(cs)
    cl = ka["counterexample_seo_greedy_5_bad.flag"]["undirected_cliques"]
    assert cb != cl
    assert all(x <= y for x, y in zip(cb, cl))  # an SEO graph never exceeds the clique counts (src/lib.rs:135-137)


# --- util.rs arithmetic, SURVEY.md App. E -------------------------------------
OEIS_A058298 = [2, 3, 6, 8, 12, 24, 30, 40, 60, 120, 144, 180, 240, 360, 720,
                840, 1008, 1260, 1680, 2520, 5040, 5760, 6720, 8064, 10080,
                13440, 20160, 40320, 45360, 51840, 60480, 72576, 90720,
                120960, 181440, 362880, 403200, 453600, 518400, 604800,
                725760, 907200, 1209600, 1814400, 3628800, 3991680, 4435200, 4989600,
                5702400, 6652800, 7983360, 9979200, 13305600, 19958400, 39916800,
                43545600, 47900160, 53222400, 59875200, 68428800,
                79833600, 95800320, 119750400, 159667200]  # the integer sequence, first 64 terms (src/util.rs:98-105)


def test_oeis_table(oracle):
    assert [oracle.lib().fo_oeis_a058298(i) for i in range(64)] == OEIS_A058298


def test_factorial_is_off_by_one_like_the_reference(oracle):
    L = oracle.lib()
    assert [L.fo_factorial(x) for x in range(7)] == [1, 1, 1, 2, 6, 24, 120]       # (x-1)!, src/util.rs:65-71
    assert [L.fo_binomial(4, k) for k in range(1, 5)] == [3, 6, 3, 1]


def test_all_le_zero_padding(oracle):
    assert oracle.all_le([1, 2], [1, 2, 0])          # src/util.rs:53-63
    assert oracle.all_le([1, 2], [1, 3])
    assert not oracle.all_le([1, 2, 1], [1, 2])      # longer left side compares against the pad
    assert oracle.all_le([], [])


def test_relax_vector_worked_example(oracle):
    # counts of the n=1000 scratch graph
    sc = [1000, 100151, 1005709, 1013125, 102824, 1084]
    a = np.array(sc, np.uint64)
    out = np.zeros(len(sc), np.uint64)
    import ctypes as C
    rc = oracle.lib().fo_calc_relax_de(a.ctypes.data_as(oracle.u64p), len(sc), out.ctypes.data_as(oracle.u64p))
    assert rc == 0
    assert [int(x) for x in out[2:]] == [2, 6, 24, 120]
    relax = [int(out[d]) * oracle.lib().fo_binomial(len(sc) - 2, d - 1) for d in range(2, len(sc))]
    assert relax == [6, 36, 72, 120]


def test_bounds_seo_shortcut_on_fixtures(oracle):
    ka = known_answers()
    for f in ("counterexample_any_order.flag", "bug_calc_relax_de.flag"):
        n, e = load_flag_fixture(f)
        g = oracle.Graph.from_edges(n, e)
        st = oracle.State(g)
        tb = oracle.target_bounds(st.flag_count, 0.01)
        b, ncl = oracle.bounds_calculate(st, tb)
        mn, mx = b.lists()
        assert ncl == ka[f]["undirected_cliques"]
        assert mx == ncl and mn == tb.lists()[0]      # src/lib.rs:135-137
        assert oracle.bounds_check(b, st.flag_count)


def test_bounds_general_branch(oracle):
    from flag_complex_mcmc_amd import graphs
    e = graphs.random_with_p(60, 0.3, seed=3)
    g = oracle.Graph.from_edges(60, e)
    st = oracle.State(g)
    fc = st.flag_count
    tb = oracle.target_bounds(fc, 0.01)
    b, ncl = oracle.bounds_calculate(st, tb)
    mn, mx = b.lists()
    tmn, tmx = tb.lists()
    assert len(mx) == len(fc) + 1 and mx[-1] == 10 and mx[2] == 2 ** 64 - 1       # src/lib.rs:151-152
    assert mn[:2] == fc[:2] and mx[:2] == fc[:2]
    for d in range(3, len(fc)):
        assert mx[d] >= tmx[d] and mn[d] <= tmn[d]


# --- apply / revert (src/lib.rs:61-95) -----------------------------------------
def test_apply_revert_ex04_to_ex05(oracle):
    """ex04 -> ex05 is the single flip 3->1 => 1->3 with zero net change
    (example_flag_generator.py:57-65)."""
    ka = known_answers()
    g = oracle.Graph.from_edges(4, np.array(ka["ex04"]["edges"], np.uint32))
    st = oracle.State(g)
    t = [((3, 1), False), ((1, 3), True)]
    pre, post = st.apply_transition(t)
    assert st.flag_count == ka["ex05"]["flag_count"]
    assert sorted(map(tuple, st.graph_edges().tolist())) == sorted(map(tuple, ka["ex05"]["edges"]))
    st.revert_transition(t, (pre, post))
    assert st.flag_count == ka["ex04"]["flag_count"]
    assert sorted(map(tuple, st.graph_edges().tolist())) == sorted(map(tuple, ka["ex04"]["edges"]))


def test_incremental_count_equals_full_recount(oracle):
    from flag_complex_mcmc_amd import graphs
    e = graphs.random_with_p(120, 0.12, seed=1)
    g = oracle.Graph.from_edges(120, e)
    st = oracle.State(g)
    tb = oracle.target_bounds(st.flag_count, 0.05)
    b, _ = oracle.bounds_calculate(st, tb)
    ch = oracle.Chain(g, b, seed=11, chain_id=3)
    und0 = ch.state.undirected_edges()
    for _ in range(5):
        ch.step(400)
        assert ch.state.graph().flagser_count() == [c for c in ch.state.flag_count]
        assert oracle.bounds_check(b, ch.state.flag_count)
    s = ch.stats()
    assert s["sampled"] == 2000 and s["n_empty"] + s["n_flip"] + s["n_dmove"] == 2000
    assert s["n_flip"] > 0 and s["n_dmove"] > 0
    # pr(G) and the edge count never change (reference README.md:3)
    assert (ch.state.graph().undirected_edges() == und0).all()
    assert len(ch.state.graph_edges()) == len(e)


def test_empty_transitions_are_accepted(oracle):
    # no reciprocal pairs => double_edge_move is always empty (src/lib.rs:324) and counts as accepted (:185-187)
    n, e = load_flag_fixture("counterexample_any_order.flag")
    g = oracle.Graph.from_edges(n, e)
    st = oracle.State(g)
    b, _ = oracle.bounds_calculate(st, oracle.target_bounds(st.flag_count, 0.01))
    ch = oracle.Chain(g, b, weights=(0.0, 1.0, 0.0, 0.0), seed=0)
    ch.step(50)
    s = ch.stats()
    assert s["sampled"] == 50 and s["accepted"] == 50 and s["n_empty"] == 50


def test_default_sample_distance(oracle):
    import math
    for m in (18, 1961, 100151):
        assert oracle.default_sample_distance(m) == math.ceil(2.0 * m * math.log2(m))   # src/bin/sample.rs:102
