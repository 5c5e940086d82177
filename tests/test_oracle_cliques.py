"""CPU: the oracle's maximal-clique enumeration and clique moves
(reference src/lib.rs:41-49, :214-290)."""
import itertools

import numpy as np
import pytest

from helpers import known_answers, load_flag_fixture


def _maximal_cliques_bruteforce(n, und):
    out = []
    for r in range(1, n + 1):
        for comb in itertools.combinations(range(n), r):
            if all(und[a][b] for a, b in itertools.combinations(comb, 2)):
                if not any(all(und[v][c] for c in comb) for v in range(n) if v not in comb):
                    out.append(comb)
    return out


def test_maximal_cliques_match_bruteforce(oracle):
    rng = np.random.default_rng(5)
    for trial in range(8):
        n = int(rng.integers(3, 11))
        adj = rng.random((n, n)) < rng.choice([0.2, 0.5, 0.8])
        np.fill_diagonal(adj, False)
        e = np.argwhere(adj).astype(np.uint32)
        und = adj | adj.T
        want = _maximal_cliques_bruteforce(n, und)
        st = oracle.State(oracle.Graph.from_edges(n, e))
        counts = st.clique_counts()
        got = []
        for o in range(1, len(counts) + 1):
            cl = st.cliques_of_order(o)
            assert len(cl) == counts[o - 1]
            rows = [tuple(int(x) for x in r) for r in cl]
            assert rows == sorted(rows) and all(list(r) == sorted(r) for r in rows)   # canonical order
            got += rows
        assert sorted(got) == sorted(want), (trial, n)
        assert counts[-1] > 0


def test_maximal_cliques_on_fixtures(oracle):
    ka = known_answers()
    for f in ("counterexample_any_order.flag", "bug_calc_relax_de.flag"):
        n, e = load_flag_fixture(f)
        st = oracle.State(oracle.Graph.from_edges(n, e))
        counts = st.clique_counts()
        # the largest maximal clique is the largest clique: same length as the undirected clique counts
        assert len(counts) == len(ka[f]["undirected_cliques"])
        # every maximal clique of the top order is a clique of that order, and all of those are maximal
        assert counts[-1] == ka[f]["undirected_cliques"][-1]
        g = oracle.Graph.from_edges(n, e)
        for o in (len(counts), 3):
            for cl in st.cliques_of_order(o)[:50]:
                assert all(g.has_edge(int(a), int(b)) or g.has_edge(int(b), int(a)) for a, b in itertools.combinations(cl, 2))


def _reciprocal(g):
    e = {tuple(x) for x in g.edges().tolist()}
    return sorted((a, b) for a, b in e if a > b and (b, a) in e)


@pytest.mark.parametrize("weights", [(0.0, 0.0, 1.0, 0.0), (0.0, 0.0, 0.0, 1.0), (0.1, 0.1, 0.6, 0.2)])
def test_clique_moves_keep_the_invariants(oracle, weights):
    """After any number of clique moves: pr(G), the edge count and the number of
    reciprocal pairs are unchanged (reference README.md:3), the incrementally
    maintained count equals a full recount, the slot list names exactly the
    reciprocal pairs, and the state is inside the bounds."""
    from flag_complex_mcmc_amd import graphs
    e = graphs.random_with_p(70, 0.25, seed=3)
    g = oracle.Graph.from_edges(70, e)
    st0 = oracle.State(g)
    b, _ = oracle.bounds_calculate(st0, oracle.target_bounds(st0.flag_count, 0.05))
    ch = oracle.Chain(g, b, weights=weights, seed=5, chain_id=2)
    und0 = ch.state.undirected_edges()
    pairs0 = {tuple(x) for x in und0.tolist()}
    for _ in range(4):
        ch.step(150)
        cur = ch.state.graph()
        assert cur.flagser_count() == [c for c in ch.state.flag_count if True][: len(cur.flagser_count())]
        assert (cur.undirected_edges() == und0).all() and len(cur.edges()) == len(e)
        rec = _reciprocal(cur)
        ue = [tuple(x) for x in und0.tolist()]
        assert sorted(ue[i] for i in ch.dbl()) == rec
        assert oracle.bounds_check(b, ch.state.flag_count)
    s = ch.stats()
    assert s["sampled"] == 600
    assert s["n_empty"] + s["n_flip"] + s["n_dmove"] + s["n_cperm"] + s["n_cswap"] == 600
    if weights[2] > 0:
        assert s["n_cperm"] > 0
    if weights[3] > 0:
        assert s["n_cswap"] > 0
    assert s["n_changes"] > 0 and s["accepted"] > s["n_empty"]


def test_clique_permute_relabels_the_orientation_pattern(oracle):
    """clique_permute on a graph that IS one clique (ex03-like tournament): the
    move relabels the orientation pattern by a permutation (src/lib.rs:222-230),
    so the sorted out-degree sequence -- and for a transitive tournament the
    whole count vector -- is invariant."""
    from flag_complex_mcmc_amd import graphs
    e = graphs.simplex(5)           # transitive tournament on 6 vertices: one maximal clique
    g = oracle.Graph.from_edges(6, e)
    st0 = oracle.State(g)
    assert st0.clique_counts() == [0, 0, 0, 0, 0, 1]
    fc = st0.flag_count
    b = oracle.Bounds.from_lists([0] * len(fc), [10 ** 9] * len(fc))
    ch = oracle.Chain(g, b, weights=(0, 0, 1, 0), seed=1)
    seen = set()
    for _ in range(30):
        ch.step(1)
        cur = ch.state.graph()
        assert ch.state.flag_count == fc                       # a relabelled transitive tournament
        outdeg = sorted(sum(1 for a, _ in cur.edges().tolist() if a == v) for v in range(6))
        assert outdeg == [0, 1, 2, 3, 4, 5]
        seen.add(tuple(map(tuple, cur.edges().tolist())))
    assert len(seen) > 10                                       # it really moves
    assert ch.stats()["accepted"] == 30
