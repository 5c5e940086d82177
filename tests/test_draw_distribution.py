"""CPU: the draw spec samples what the reference samples.

The reference draws a uniform directed edge (`sample_edge`), a uniform
reciprocal pair (`sample_double_edge`), a single edge by rejection and a fair
coin (src/lib.rs:292-325); clique moves draw an order by count^0.2, a uniform
clique and a uniform permutation (:214-218).  Bit-level parity with its
Xoshiro stream is impossible, so this checks the *distribution*
of the first proposal over many independent Philox streams from one fixed
state, with chi-square tests at a fixed seed (deterministic).  The GPU path
inherits the result through its bit-exact parity with the oracle.
"""
import collections

import numpy as np
from scipy import stats

N_STREAMS = 30000


def _graph():
    # 9 vertices: a mix of single edges and reciprocal pairs, several cliques
    single = [(0, 1), (2, 1), (2, 0), (3, 2), (1, 3), (4, 3), (4, 2), (5, 4), (3, 5), (6, 5), (4, 6), (7, 6), (5, 7), (8, 7), (6, 8), (0, 8)]
    double = [(0, 3), (2, 5), (4, 7)]
    e = single + double + [(b, a) for a, b in double]
    return 9, np.array(e, np.uint32), single, double


def _first_moves(oracle, weights):
    n, e, single, double = _graph()
    g = oracle.Graph.from_edges(n, e)
    st = oracle.State(g)
    fc = st.flag_count
    wide = oracle.Bounds.from_lists([0] * 8, [10 ** 9] * 8)     # accept everything
    e0 = {tuple(x) for x in e.tolist()}
    out = collections.Counter()
    for c in range(N_STREAMS):
        ch = oracle.Chain(g, wide, weights=weights, seed=12345, chain_id=c)
        ch.step(1)
        e1 = {tuple(x) for x in ch.state.graph_edges().tolist()}
        out[(tuple(sorted(e0 - e1)), tuple(sorted(e1 - e0)))] += 1
    return out, single, double


def test_single_edge_flip_is_uniform_over_directed_edges(oracle):
    out, single, double = _first_moves(oracle, (1.0, 0.0, 0.0, 0.0))
    m = len(single) + 2 * len(double)
    # empty transition iff the drawn directed edge belongs to a reciprocal pair (src/lib.rs:294-298)
    empty = out.pop(((), ()), 0)
    assert abs(empty / N_STREAMS - 2 * len(double) / m) < 0.01
    # every single edge is flipped equally often; nothing else ever happens
    assert set(out) == {(((a, b),), ((b, a),)) for a, b in single}
    chi2, pval = stats.chisquare(list(out.values()))
    assert pval > 1e-3, (chi2, pval)


def test_double_edge_move_is_uniform(oracle):
    out, single, double = _first_moves(oracle, (0.0, 1.0, 0.0, 0.0))
    assert ((), ()) not in out                      # a reciprocal pair and a single edge always exist here
    # outcome = (deleted direction of a reciprocal pair, single edge that got its reverse added):
    # uniform over 2 * D * (#single edges) combinations (src/lib.rs:306-322)
    want = set()
    for a, b in double:
        for x, y in single:
            want.add((((a, b),), ((y, x),)))
            want.add((((b, a),), ((y, x),)))
    assert set(out) == want
    chi2, pval = stats.chisquare(list(out.values()))
    assert pval > 1e-3, (chi2, pval)


def test_move_selection_follows_the_weights(oracle):
    n, e, single, double = _graph()
    g = oracle.Graph.from_edges(n, e)
    wide = oracle.Bounds.from_lists([0] * 8, [10 ** 9] * 8)
    kinds = collections.Counter()
    w = (0.1, 0.1, 0.6, 0.2)                        # src/bin/sample.rs:17
    ch = oracle.Chain(g, wide, weights=w, seed=7, chain_id=0)
    # count proposals by kind over one long chain: empties are attributed by elimination below
    ch.step(40000)
    s = ch.stats()
    assert s["sampled"] == 40000
    # flips are empty with probability 2D/m, double moves never here; clique moves can be empty (identity effect)
    m = len(single) + 2 * len(double)
    assert abs(s["n_dmove"] / 40000 - 0.1) < 0.006
    assert abs(s["n_flip"] / 40000 - 0.1 * (1 - 2 * len(double) / m)) < 0.02   # the pair count drifts a little as doubles move
    # clique moves whose permutation leaves the orientation pattern unchanged are empty transitions
    # (common for the 2- and 3-cliques of this graph), so only upper bounds are sharp
    assert 0.3 < s["n_cperm"] / 40000 <= 0.6 + 0.01 and 0.08 < s["n_cswap"] / 40000 <= 0.2 + 0.01
    assert s["n_empty"] + s["n_flip"] + s["n_dmove"] + s["n_cperm"] + s["n_cswap"] == 40000


def test_clique_permute_draws_a_uniform_permutation(oracle):
    # one maximal clique only (a transitive tournament on 4 vertices): 4! relabellings, all equally likely
    from flag_complex_mcmc_amd import graphs
    e = graphs.simplex(3)
    g = oracle.Graph.from_edges(4, e)
    wide = oracle.Bounds.from_lists([0] * 6, [10 ** 9] * 6)
    out = collections.Counter()
    for c in range(24000):
        ch = oracle.Chain(g, wide, weights=(0, 0, 1, 0), seed=99, chain_id=c)
        ch.step(1)
        out[tuple(map(tuple, ch.state.graph_edges().tolist()))] += 1
    assert len(out) == 24                           # every relabelled tournament is reached
    chi2, pval = stats.chisquare(list(out.values()))
    assert pval > 1e-3, (chi2, pval)
