#!/usr/bin/env python3
"""Regenerates tests/golden/known_answers.json.

The reference holds NO expected simplex counts for any graph:
these known answers come from the definition of the directed flag complex
(SURVEY.md App. A.2), computed here by two methods that share no code with
oracle/fcm_oracle.c or the HIP kernels:
  * `count_dfs`   set-based depth-first enumeration in pure Python;
  * `count_brute` enumeration of all ordered vertex tuples (graphs with n <= 8).
Inputs: the four .flag data fixtures the reference ships in flag_file_examples/
(copied as data into tests/golden/) and the small graphs its generator script
defines (example_flag_generator.py:42-73; "boese testcases" 3-cycle from
Testcases.pdf).  They are NOT outputs of the reference program.
"""
import itertools
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from helpers import load_flag_fixture  # noqa: E402
from flag_complex_mcmc_amd import graphs  # noqa: E402


def count_dfs(n, edges):
    out = [set() for _ in range(n)]
    for a, b in edges:
        if a != b:
            out[int(a)].add(int(b))
    counts = [n, sum(len(s) for s in out)]

    def rec(cand, depth):
        for v in cand:
            nc = cand & out[v]
            if depth >= len(counts):
                counts.append(0)
            counts[depth] += 1
            if nc:
                rec(nc, depth + 1)

    for v in range(n):
        for w in out[v]:
            nc = out[v] & out[w]
            if nc:
                rec(nc, 2)
    while counts and counts[-1] == 0:
        counts.pop()
    return counts


def count_brute(n, edges):
    es = {(int(a), int(b)) for a, b in edges if a != b}
    counts = [n]
    for d in range(1, n):
        c = 0
        for tup in itertools.permutations(range(n), d + 1):
            if all((tup[i], tup[j]) in es for i in range(d + 1) for j in range(i + 1, d + 1)):
                c += 1
        if c == 0:
            break
        counts.append(c)
    return counts


def undirected_cliques(n, edges):
    und = {(max(int(a), int(b)), min(int(a), int(b))) for a, b in edges if a != b}
    return count_dfs(n, sorted(und))


def join(e1, n1, e2):
    import numpy as np
    return np.concatenate([e1, e2 + n1])


def main():
    import numpy as np
    cases = {}
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".flag"):
            n, e = load_flag_fixture(f)
            cases[f] = (n, e)
    ex00 = graphs.simplex(3)
    ex04 = graphs.densifier([0, 0, 1, 3, 3], [1, 2, 2, 1, 2])
    ex05 = graphs.densifier([0, 0, 1, 1, 3], [1, 2, 2, 3, 2])
    small = {
        "ex00": (4, ex00),
        "ex01": (4, np.concatenate([ex00, [[0, 3]]]).astype(np.uint32)),
        "ex02": (4, np.concatenate([ex00, [[2, 3]]]).astype(np.uint32)),
        "ex03": (4, graphs.clique(3)),
        "ex04": (4, ex04),
        "ex05": (4, ex05),
        "ex06": (8, join(ex04, 4, ex05)),
        "ex07": (10, graphs.densifier([0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 3, 4, 5, 6, 7, 8, 9],
                                      [1, 2, 2, 3, 4, 5, 6, 7, 8, 9, 2, 2, 2, 2, 2, 2, 2])),
        "cycle3": (3, np.array([[0, 1], [1, 2], [2, 0]], np.uint32)),
    }
    cases.update(small)
    out = {}
    for name, (n, e) in cases.items():
        c = count_dfs(n, e)
        if n <= 8:
            assert count_brute(n, e) == c, name
        rec = {"n": n, "m": int(len({(int(a), int(b)) for a, b in e})), "flag_count": c,
               "undirected_cliques": undirected_cliques(n, e)}
        if name in small:
            rec["edges"] = [[int(a), int(b)] for a, b in e]
        out[name] = rec
    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for k, v in out.items():
        print(k, v["n"], v["m"], v["flag_count"], v["undirected_cliques"])


if __name__ == "__main__":
    main()
