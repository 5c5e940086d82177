"""Synthetic input graphs for tests and bench.py.

Recipes follow the reference's generators (flag_file_examples/data_importer.py
and example_flag_generator.py), restated with an explicit seeded generator
because the reference uses the unseeded global numpy RNG.
"""
import numpy as np


def random_with_p(n, p, seed=0):
    """`random_with_p` (data_importer.py:102-106): entry (i,j), i != j, present
    iff U(0,1) < p * n^2 / (n^2 - n).  Returns the (m,2) directed edge list."""
    rng = np.random.default_rng(seed)
    thr = p * (n ** 2) / (n ** 2 - n)
    edges = []
    for i in range(n):  # row at a time: O(n) memory
        row = rng.random(n) < thr
        row[i] = False
        j = np.nonzero(row)[0]
        if len(j):
            edges.append(np.stack([np.full(len(j), i, np.uint32), j.astype(np.uint32)], axis=1))
    return np.concatenate(edges) if edges else np.zeros((0, 2), np.uint32)


def random_edge_draws(n, ndraws, seed=0):
    """BASELINE config 5: `ndraws` uniform ordered pairs i != j, deduplicated."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, n, ndraws, dtype=np.int64)
    b = rng.integers(0, n - 1, ndraws, dtype=np.int64)
    b = b + (b >= a)
    key = np.unique(a * n + b)
    return np.stack([(key // n).astype(np.uint32), (key % n).astype(np.uint32)], axis=1)


def simplex(d):
    """data_importer.py:59-61: edges i->j for i > j on d+1 vertices."""
    return np.array([(i, j) for i in range(d + 1) for j in range(i)], np.uint32).reshape(-1, 2)


def clique(d):
    """data_importer.py:64-69: all ordered pairs on d+1 vertices."""
    return np.array([(i, j) for i in range(d + 1) for j in range(d + 1) if i != j], np.uint32).reshape(-1, 2)


def densifier(li, lj):
    """data_importer.py:120-126."""
    return np.array(list(zip(li, lj)), np.uint32).reshape(-1, 2)


def seoify(edges, seed=0):
    """example_flag_generator.py:16-25: drop one direction of every reciprocal pair."""
    rng = np.random.default_rng(seed)
    s = {(int(a), int(b)) for a, b in edges}
    out = set(s)
    for a, b in sorted(s):
        if a < b and (b, a) in s:
            out.discard((a, b) if rng.random() < 0.5 else (b, a))
    return np.array(sorted(out), np.uint32).reshape(-1, 2)
