"""Shared helpers for the parity tests."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_flag_fixture(name):
    """(n, edges) of a committed .flag data fixture (tests/golden/*.flag)."""
    path = os.path.join(GOLDEN, name)
    with open(path) as f:
        lines = f.read().split("\n")
    n = len([t for t in lines[1].split(" ") if t])
    edges = []
    for ln in lines[3:]:
        t = [x for x in ln.split(" ") if x]
        if len(t) >= 2:
            edges.append((int(t[0]), int(t[1])))
    return n, np.array(edges, np.uint32).reshape(-1, 2)


def known_answers():
    with open(os.path.join(GOLDEN, "known_answers.json")) as f:
        return json.load(f)


def setup_pair(fcm, oracle, n, edges, relaxation=0.01, device=0):
    """Build the same sampler inputs on both sides; returns (g_gpu, g_cpu, bounds_gpu, bounds_cpu)."""
    gg = fcm.Graph.from_edges(n, edges)
    go = oracle.Graph.from_edges(n, edges)
    fc = go.flagser_count()
    st = oracle.State(go)
    tb_o = oracle.target_bounds(fc, relaxation)
    b_o, _ = oracle.bounds_calculate(st, tb_o)
    mn, mx = b_o.lists()
    b_g = fcm.Bounds(mn, mx)
    return gg, go, b_g, b_o


def compare_chain(fcm_sampler, chain_idx, ochain, ctx=""):
    """Exact comparison of one GPU chain with its oracle twin (tolerance 0)."""
    ost = ochain.stats()
    gst = fcm_sampler.stats()
    for k in ("sampled", "accepted", "n_empty", "n_flip", "n_dmove", "sum_k", "n_cperm", "n_cswap", "n_changes"):
        assert int(gst[k][chain_idx]) == ost[k], (ctx, k, int(gst[k][chain_idx]), ost[k])
    assert fcm_sampler.flag_count(chain_idx) == ochain.state.flag_count, ctx
    ge = fcm_sampler.edges(chain_idx)
    oe = ochain.state.graph_edges()
    assert ge.shape == oe.shape and (ge == oe).all(), ctx
    assert [int(x) for x in fcm_sampler.double_slots(chain_idx)] == ochain.dbl(), ctx
