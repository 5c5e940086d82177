"""ctypes binding of the CPU oracle (oracle/fcm_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfcm_oracle.so")
MAXDIM = 64

u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
i32p = C.POINTER(C.c_int)


def build(force=False):
    src = os.path.join(_HERE, "fcm_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    return _LIB_PATH


class Bounds(C.Structure):
    _fields_ = [("min", C.c_uint64 * MAXDIM), ("min_len", C.c_int),
                ("max", C.c_uint64 * MAXDIM), ("max_len", C.c_int)]

    @classmethod
    def from_lists(cls, mn, mx):
        b = cls()
        for i, v in enumerate(mn):
            b.min[i] = v
        for i, v in enumerate(mx):
            b.max[i] = v
        b.min_len, b.max_len = len(mn), len(mx)
        return b

    def lists(self):
        return [int(self.min[i]) for i in range(self.min_len)], [int(self.max[i]) for i in range(self.max_len)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    vp = C.c_void_p
    sig = {
        "fo_philox": (None, [u32p, u32p, u32p]),
        "fo_graph_new": (vp, [C.c_uint32]),
        "fo_graph_free": (None, [vp]),
        "fo_graph_clone": (vp, [vp]),
        "fo_graph_nnodes": (C.c_uint32, [vp]),
        "fo_graph_has_edge": (C.c_int, [vp, C.c_uint32, C.c_uint32]),
        "fo_graph_set_edge": (None, [vp, C.c_uint32, C.c_uint32, C.c_int]),
        "fo_graph_add_edge": (None, [vp, C.c_uint32, C.c_uint32]),
        "fo_graph_edges": (C.c_uint64, [vp, u32p, C.c_uint64]),
        "fo_graph_undirected_edges": (C.c_uint64, [vp, u32p, C.c_uint64]),
        "fo_graph_subgraph": (vp, [vp, u32p, C.c_uint32]),
        "fo_graph_flagser_count": (C.c_int, [vp, u64p]),
        "fo_flagser_count_unweighted": (C.c_void_p, [C.c_uint64, C.c_uint64, u32p, u64p]),
        "fo_intersect_sorted": (C.c_uint64, [u32p, C.c_uint64, u32p, C.c_uint64, u32p]),
        "fo_all_le": (C.c_int, [u64p, C.c_int, u64p, C.c_int, C.c_uint64]),
        "fo_factorial": (C.c_uint64, [C.c_uint64]),
        "fo_binomial": (C.c_uint64, [C.c_uint64, C.c_uint64]),
        "fo_oeis_a058298": (C.c_uint64, [C.c_int]),
        "fo_calc_relax_de": (C.c_int, [u64p, C.c_int, u64p]),
        "fo_state_new": (vp, [vp]),
        "fo_state_free": (None, [vp]),
        "fo_state_graph": (vp, [vp]),
        "fo_state_flag_count": (C.c_int, [vp, u64p]),
        "fo_state_n_uedges": (C.c_uint64, [vp]),
        "fo_state_uedges": (u32p, [vp]),
        "fo_state_edge_neighborhood": (C.c_int64, [vp, C.c_uint32, C.c_uint32, u32p, C.c_uint64]),
        "fo_state_edgeset_neighborhood": (C.c_int64, [vp, u32p, C.c_uint32, u32p, C.c_uint64]),
        "fo_state_clique_counts": (C.c_int, [vp, u64p]),
        "fo_state_cliques_of_order": (C.c_int64, [vp, C.c_int, u32p, C.c_uint64]),
        "fo_state_apply_flat": (C.c_int, [vp, C.c_uint32, u32p, i32p, u64p, i32p, u64p, i32p]),
        "fo_state_revert_flat": (C.c_int, [vp, C.c_uint32, u32p, i32p, u64p, C.c_int, u64p, C.c_int]),
        "fo_target_bounds": (None, [u64p, C.c_int, C.c_double, C.POINTER(Bounds)]),
        "fo_bounds_calculate": (C.c_int, [vp, C.POINTER(Bounds), C.POINTER(Bounds), u64p, i32p]),
        "fo_bounds_check": (C.c_int, [C.POINTER(Bounds), u64p, C.c_int]),
        "fo_default_sample_distance": (C.c_uint64, [C.c_uint64]),
        "fo_move_thresholds": (None, [C.POINTER(C.c_double), u64p]),
        "fo_chain_new": (vp, [vp, C.POINTER(Bounds), C.POINTER(C.c_double), C.c_uint64, C.c_uint64, C.c_uint32]),
        "fo_chain_free": (None, [vp]),
        "fo_chain_state": (vp, [vp]),
        "fo_chain_stats": (None, [vp, u64p]),
        "fo_chain_n_double": (C.c_uint64, [vp]),
        "fo_chain_dbl": (u64p, [vp]),
        "fo_chain_step": (C.c_int, [vp, C.c_uint64]),
        "fo_chain_next": (C.c_int, [vp]),
        "fo_chain_acceptance_ratio": (C.c_double, [vp]),
        "fo_read_flag_file": (vp, [C.c_char_p]),
        "fo_chains_step_mt": (C.c_double, [C.POINTER(vp), C.c_int, C.c_uint64, C.c_int, i32p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    L._libc_free = C.CDLL(None).free
    L._libc_free.argtypes = [C.c_void_p]
    _lib = L
    return L


def _u32(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a, a.ctypes.data_as(u32p)


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(u64p)


def philox(ctr, key):
    c, cp = _u32(ctr)
    k, kp = _u32(key)
    o = np.zeros(4, np.uint32)
    lib().fo_philox(cp, kp, o.ctypes.data_as(u32p))
    return [int(x) for x in o]


def intersect_sorted(a, b):
    a_, ap = _u32(a)
    b_, bp = _u32(b)
    out = np.zeros(max(1, min(len(a_), len(b_))), np.uint32)
    n = lib().fo_intersect_sorted(ap, len(a_), bp, len(b_), out.ctypes.data_as(u32p))
    return [int(x) for x in out[:n]]


def all_le(a, b, z=0):
    a_, ap = _u64(a)
    b_, bp = _u64(b)
    return bool(lib().fo_all_le(ap, len(a_), bp, len(b_), z))


class Graph:
    """Oracle Graph (flag_complex::Graph surface)."""

    def __init__(self, n=None, _h=None):
        self._h = _h if _h is not None else lib().fo_graph_new(n)
        if not self._h:
            raise MemoryError

    def __del__(self):
        if getattr(self, "_h", None):
            lib().fo_graph_free(self._h)
            self._h = None

    @classmethod
    def from_edges(cls, n, edges):
        g = cls(n)
        for a, b in np.asarray(edges, dtype=np.int64).reshape(-1, 2):
            g.add_edge(int(a), int(b))
        return g

    @classmethod
    def read_flag_file(cls, fname):
        h = lib().fo_read_flag_file(os.fsencode(fname))
        if not h:
            raise ValueError("could not read flag file %r" % (fname,))
        return cls(_h=h)

    def clone(self):
        return Graph(_h=lib().fo_graph_clone(self._h))

    def nnodes(self):
        return int(lib().fo_graph_nnodes(self._h))

    def has_edge(self, a, b):
        return bool(lib().fo_graph_has_edge(self._h, a, b))

    def set_edge(self, a, b, present):
        lib().fo_graph_set_edge(self._h, a, b, int(bool(present)))

    def add_edge(self, a, b):
        lib().fo_graph_add_edge(self._h, a, b)

    def edges(self):
        m = lib().fo_graph_edges(self._h, None, 0)
        out = np.zeros((max(m, 1), 2), np.uint32)
        lib().fo_graph_edges(self._h, out.ctypes.data_as(u32p), m)
        return out[:m]

    def undirected_edges(self):
        m = lib().fo_graph_undirected_edges(self._h, None, 0)
        out = np.zeros((max(m, 1), 2), np.uint32)
        lib().fo_graph_undirected_edges(self._h, out.ctypes.data_as(u32p), m)
        return out[:m]

    def subgraph(self, nodes):
        n_, npn = _u32(nodes)
        return Graph(_h=lib().fo_graph_subgraph(self._h, npn, len(n_)))

    def flagser_count(self):
        out = np.zeros(MAXDIM, np.uint64)
        ln = lib().fo_graph_flagser_count(self._h, out.ctypes.data_as(u64p))
        if ln < 0:
            raise MemoryError
        return [int(x) for x in out[:ln]]


def flagser_count_unweighted(nvertices, edges):
    """Legacy C entry point shape (src/flagser.rs:7-21)."""
    e, ep = _u32(np.asarray(edges, dtype=np.uint32).reshape(-1, 2))
    n = C.c_uint64(0)
    p = lib().fo_flagser_count_unweighted(nvertices, len(e), ep, C.byref(n))
    if not p:
        raise ValueError("flagser_count_unweighted failed")
    res = [int(x) for x in C.cast(p, u64p)[: n.value]]
    lib()._libc_free(p)
    return res


class State:
    def __init__(self, graph, _h=None, _owner=None):
        self._owner = _owner
        self._h = _h if _h is not None else lib().fo_state_new(graph._h)

    def __del__(self):
        if self._owner is None and getattr(self, "_h", None):
            lib().fo_state_free(self._h)
            self._h = None

    @property
    def flag_count(self):
        out = np.zeros(MAXDIM, np.uint64)
        ln = lib().fo_state_flag_count(self._h, out.ctypes.data_as(u64p))
        return [int(x) for x in out[:ln]]

    def graph_edges(self):
        g = Graph(_h=lib().fo_graph_clone(lib().fo_state_graph(self._h)))
        return g.edges()

    def graph(self):
        return Graph(_h=lib().fo_graph_clone(lib().fo_state_graph(self._h)))

    def undirected_edges(self):
        n = lib().fo_state_n_uedges(self._h)
        p = lib().fo_state_uedges(self._h)
        return np.ctypeslib.as_array(p, shape=(max(n, 1), 2))[:n].copy()

    def clique_counts(self):
        """cliques_by_order[i].len() for i = 0.. (src/lib.rs:42-49; printed at sample.rs:85)"""
        out = np.zeros(MAXDIM, np.uint64)
        n = lib().fo_state_clique_counts(self._h, out.ctypes.data_as(u64p))
        if n < 0:
            raise MemoryError
        return [int(x) for x in out[:n]]

    def cliques_of_order(self, order):
        cnt = lib().fo_state_cliques_of_order(self._h, order, None, 0)
        if cnt < 0:
            raise KeyError(order)
        out = np.zeros((max(cnt, 1), order), np.uint32)
        lib().fo_state_cliques_of_order(self._h, order, out.ctypes.data_as(u32p), cnt * order)
        return out[:cnt]

    def edge_neighborhood(self, a, b):
        out = np.zeros(self_cap(self), np.uint32)
        n = lib().fo_state_edge_neighborhood(self._h, a, b, out.ctypes.data_as(u32p), len(out))
        if n < 0:
            raise KeyError((a, b))
        return [int(x) for x in out[:n]]

    def edgeset_neighborhood(self, edges):
        e, ep = _u32(np.asarray(edges, dtype=np.uint32).reshape(-1, 2))
        out = np.zeros(self_cap(self) * max(1, len(e)) + 2 * len(e) + 2, np.uint32)
        n = lib().fo_state_edgeset_neighborhood(self._h, ep, len(e), out.ctypes.data_as(u32p), len(out))
        if n < 0:
            raise KeyError("edge not in neighbourhood table")
        return [int(x) for x in out[:n]]

    def apply_transition(self, change_edges):
        """change_edges: list of ((a,b), add).  Returns (pre, post)."""
        n = len(change_edges)
        e, ep = _u32(np.array([c[0] for c in change_edges], dtype=np.uint32).reshape(-1, 2)) if n else _u32(np.zeros((1, 2)))
        add = (C.c_int * max(n, 1))(*[int(c[1]) for c in change_edges])
        pre = np.zeros(MAXDIM, np.uint64)
        post = np.zeros(MAXDIM, np.uint64)
        pl, ql = C.c_int(0), C.c_int(0)
        rc = lib().fo_state_apply_flat(self._h, n, ep, add, pre.ctypes.data_as(u64p), C.byref(pl),
                                        post.ctypes.data_as(u64p), C.byref(ql))
        if rc:
            raise AssertionError("apply_transition failed rc=%d" % rc)
        return [int(x) for x in pre[: pl.value]], [int(x) for x in post[: ql.value]]

    def revert_transition(self, change_edges, counters):
        n = len(change_edges)
        e, ep = _u32(np.array([c[0] for c in change_edges], dtype=np.uint32).reshape(-1, 2)) if n else _u32(np.zeros((1, 2)))
        add = (C.c_int * max(n, 1))(*[int(c[1]) for c in change_edges])
        pre = np.zeros(MAXDIM, np.uint64)
        post = np.zeros(MAXDIM, np.uint64)
        pre[: len(counters[0])] = counters[0]
        post[: len(counters[1])] = counters[1]
        rc = lib().fo_state_revert_flat(self._h, n, ep, add, pre.ctypes.data_as(u64p), len(counters[0]),
                                         post.ctypes.data_as(u64p), len(counters[1]))
        if rc:
            raise AssertionError("revert_transition failed rc=%d" % rc)


def self_cap(state):
    return int(lib().fo_graph_nnodes(lib().fo_state_graph(state._h))) + 2


def target_bounds(flag_count, relaxation):
    fc, fp = _u64(flag_count)
    b = Bounds()
    lib().fo_target_bounds(fp, len(fc), relaxation, C.byref(b))
    return b


def bounds_calculate(state, target):
    out = Bounds()
    ncl = np.zeros(MAXDIM, np.uint64)
    nl = C.c_int(0)
    rc = lib().fo_bounds_calculate(state._h, C.byref(target), C.byref(out), ncl.ctypes.data_as(u64p), C.byref(nl))
    if rc:
        raise ValueError("Bounds::calculate would panic in the reference (rc=%d)" % rc)
    return out, [int(x) for x in ncl[: nl.value]]


def bounds_check(bounds, flag_count):
    fc, fp = _u64(flag_count)
    return bool(lib().fo_bounds_check(C.byref(bounds), fp, len(fc)))


def default_sample_distance(nedges):
    return int(lib().fo_default_sample_distance(nedges))


def move_thresholds(weights):
    w = (C.c_double * 4)(*weights)
    out = np.zeros(4, np.uint64)
    lib().fo_move_thresholds(w, out.ctypes.data_as(u64p))
    return [int(x) for x in out]


SIMPLE_WEIGHTS = (0.5, 0.5, 0.0, 0.0)  # src/bin/sample.rs:16
DEFAULT_WEIGHTS = (0.1, 0.1, 0.6, 0.2)  # src/bin/sample.rs:17


class Chain:
    """One MCMCSampler (src/lib.rs:163-198) on the oracle."""

    def __init__(self, graph, bounds, weights=SIMPLE_WEIGHTS, sample_distance=0, seed=0, chain_id=0):
        w = (C.c_double * 4)(*weights)
        self._h = lib().fo_chain_new(graph._h, C.byref(bounds), w, sample_distance, seed, chain_id)
        if not self._h:
            raise MemoryError
        self.state = State(None, _h=lib().fo_chain_state(self._h), _owner=self)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().fo_chain_free(self._h)
            self._h = None

    def step(self, n):
        rc = lib().fo_chain_step(self._h, n)
        if rc:
            raise AssertionError("oracle step failed rc=%d" % rc)

    def next(self):
        rc = lib().fo_chain_next(self._h)
        if rc:
            raise AssertionError("oracle next failed rc=%d" % rc)
        return self.state

    def stats(self):
        out = np.zeros(9, np.uint64)
        lib().fo_chain_stats(self._h, out.ctypes.data_as(u64p))
        return dict(zip(("sampled", "accepted", "n_empty", "n_flip", "n_dmove", "sum_k", "n_cperm", "n_cswap", "n_changes"),
                        (int(x) for x in out)))

    def dbl(self):
        n = lib().fo_chain_n_double(self._h)
        p = lib().fo_chain_dbl(self._h)
        return [int(p[i]) for i in range(n)]

    def acceptance_ratio(self):
        return float(lib().fo_chain_acceptance_ratio(self._h))


def chains_step_mt(chains, nprop, nthreads):
    arr = (C.c_void_p * len(chains))(*[c._h for c in chains])
    rc = C.c_int(0)
    secs = lib().fo_chains_step_mt(arr, len(chains), nprop, nthreads, C.byref(rc))
    if rc.value:
        raise AssertionError("oracle mt step failed rc=%d" % rc.value)
    return secs
