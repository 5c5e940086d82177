"""Host-side mirror of the reference's interface for the hot path.

Names, argument meaning and error behaviour follow the reference crate
`directed-scm` (reference src/lib.rs, src/io.rs, src/bin/sample.rs) so a user
of the reference finds the same objects here:

    Graph                 flag_complex::Graph surface
    Bounds                src/lib.rs:113-161
    MCMCSampler           src/lib.rs:163-198 -- here a *batch* of independent
                          chains on one GPU (one persistent workgroup each)
    initialize_new_sampler  src/bin/sample.rs:80-105
    read_flag_file / save_flag_file / BitOutput   src/io.rs:18-48,128-212

Everything computes through libfcm.so (HIP); nothing here re-implements it.
"""
import ctypes as C
import math
import os

import numpy as np

from . import _ffi
from ._ffi import CBounds, CSamplerConfig, CSamplerInfo, FcmError, check, lib, u32p, u64p, i32p

MOVE_DISTRIBUTION_SIMPLE = (0.5, 0.5, 0.0, 0.0)  # src/bin/sample.rs:16
MOVE_DISTRIBUTION = (0.1, 0.1, 0.6, 0.2)         # src/bin/sample.rs:17


def _u32(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a, a.ctypes.data_as(u32p)


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(u64p)


class Graph:
    """Directed graph without loops; adjacency as out-row bitmaps."""

    def __init__(self, _h):
        self._h = _h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib().fcm_graph_destroy(h)

    @classmethod
    def new_disconnected(cls, nnodes):
        h = C.c_void_p()
        check(lib().fcm_graph_new_disconnected(nnodes, C.byref(h)))
        return cls(h)

    @classmethod
    def from_edges(cls, nnodes, edges):
        e, ep = _u32(np.asarray(edges, dtype=np.uint32).reshape(-1, 2))
        h = C.c_void_p()
        check(lib().fcm_graph_from_edges(nnodes, len(e), ep, C.byref(h)))
        return cls(h)

    @classmethod
    def from_adjacency(cls, adj):
        adj = np.asarray(adj)
        return cls.from_edges(adj.shape[0], np.argwhere(adj != 0))

    def clone(self):
        h = C.c_void_p()
        check(lib().fcm_graph_clone(self._h, C.byref(h)))
        return Graph(h)

    def nnodes(self):
        return int(lib().fcm_graph_nnodes(self._h))

    def nedges(self):
        return int(lib().fcm_graph_nedges(self._h))

    def has_edge(self, a, b):
        return bool(lib().fcm_graph_has_edge(self._h, a, b))

    def set_edge(self, a, b, present):
        check(lib().fcm_graph_set_edge(self._h, a, b, int(bool(present))))

    def add_edge(self, a, b):
        check(lib().fcm_graph_add_edge(self._h, a, b))

    def remove_edge(self, a, b):
        check(lib().fcm_graph_remove_edge(self._h, a, b))

    def edges(self):
        m = C.c_uint64(0)
        check(lib().fcm_graph_edges(self._h, None, 0, C.byref(m)))
        out = np.zeros((max(m.value, 1), 2), np.uint32)
        check(lib().fcm_graph_edges(self._h, out.ctypes.data_as(u32p), m.value, C.byref(m)))
        return out[: m.value]

    def undirected_edges(self):
        m = C.c_uint64(0)
        check(lib().fcm_graph_undirected_edges(self._h, None, 0, C.byref(m)))
        out = np.zeros((max(m.value, 1), 2), np.uint32)
        check(lib().fcm_graph_undirected_edges(self._h, out.ctypes.data_as(u32p), m.value, C.byref(m)))
        return out[: m.value]

    def flagser_count(self, device=0):
        """Directed-flag-complex cell counts, on the GPU."""
        out = np.zeros(_ffi.MAX_COUNTS, np.uint64)
        ln = C.c_int(0)
        check(lib().fcm_graph_flagser_count(self._h, device, out.ctypes.data_as(u64p), len(out), C.byref(ln)))
        return [int(x) for x in out[: ln.value]]


def count_unweighted(nvertices, edges):
    """The reference's legacy wrapper (src/flagser.rs:13-21) over the C symbol
    `flagser_count_unweighted`, here exported by libfcm.so."""
    e, ep = _u32(np.asarray(edges, dtype=np.uint32).reshape(-1, 2))
    n = C.c_size_t(0)
    p = lib().flagser_count_unweighted(nvertices, len(e), ep, C.byref(n))
    if not p:
        raise FcmError(-1, lib().fcm_last_error().decode())
    res = [int(x) for x in C.cast(p, u64p)[: n.value]]
    lib()._free(p)
    return res


def read_flag_file(fname):
    h = C.c_void_p()
    check(lib().fcm_read_flag_file(os.fsencode(fname), C.byref(h)))
    return Graph(h)


def save_flag_file(fname, graph):
    check(lib().fcm_save_flag_file(os.fsencode(fname), graph._h))


class Bounds:
    """`Bounds { flag_count_min, flag_count_max }` (src/lib.rs:113-117)."""

    def __init__(self, flag_count_min, flag_count_max):
        self.flag_count_min = [int(x) for x in flag_count_min]
        self.flag_count_max = [int(x) for x in flag_count_max]

    def _c(self):
        b = CBounds()
        for i, v in enumerate(self.flag_count_min):
            b.flag_count_min[i] = v
        for i, v in enumerate(self.flag_count_max):
            b.flag_count_max[i] = v
        b.min_len, b.max_len = len(self.flag_count_min), len(self.flag_count_max)
        return b

    @classmethod
    def _from_c(cls, b):
        return cls([b.flag_count_min[i] for i in range(b.min_len)], [b.flag_count_max[i] for i in range(b.max_len)])

    @classmethod
    def target(cls, flag_count, target_relaxation):
        """src/bin/sample.rs:89-95"""
        fc, fp = _u64(flag_count)
        b = CBounds()
        check(lib().fcm_target_bounds(fp, len(fc), target_relaxation, C.byref(b)))
        return cls._from_c(b)

    @classmethod
    def calculate(cls, graph, flag_count, target_bounds, device=0, return_ncliques=False):
        """Bounds::calculate (src/lib.rs:119-156)."""
        fc, fp = _u64(flag_count)
        out = CBounds()
        t = target_bounds._c()
        ncl = np.zeros(_ffi.MAX_COUNTS + 1, np.uint64)
        nl = C.c_int(0)
        check(lib().fcm_bounds_calculate(graph._h, fp, len(fc), C.byref(t), device, C.byref(out),
                                         ncl.ctypes.data_as(u64p), C.byref(nl)))
        b = cls._from_c(out)
        return (b, [int(x) for x in ncl[: nl.value]]) if return_ncliques else b

    def check(self, flag_count):
        fc, fp = _u64(flag_count)
        c = self._c()
        return bool(lib().fcm_bounds_check(C.byref(c), fp, len(fc)))

    def __repr__(self):
        return "Bounds(min=%r, max=%r)" % (self.flag_count_min, self.flag_count_max)


def default_sample_distance(nedges):
    return int(lib().fcm_default_sample_distance(nedges))


class MCMCSampler:
    """A batch of independent chains of the reference's MCMCSampler on one GPU.

    Chain i draws from the Philox stream (seed, first_chain_id + i), so a chain's
    trajectory does not depend on how chains are split over handles or GPUs.
    """

    def __init__(self, graph, bounds, n_chains=1, seed=0, move_weights=MOVE_DISTRIBUTION_SIMPLE,
                 sample_distance=0, dim_cap=0, device=0, first_chain_id=0, _h=None):
        if _h is not None:
            self._adopt(_h, bounds)
            return
        cfg = CSamplerConfig()
        cfg.n_chains, cfg.first_chain_id, cfg.seed = n_chains, first_chain_id, seed
        for i in range(4):
            cfg.move_weights[i] = move_weights[i]
        cfg.sample_distance, cfg.dim_cap, cfg.device = sample_distance, dim_cap, device
        h = C.c_void_p()
        check(lib().fcm_sampler_create(graph._h, C.byref(bounds._c()), C.byref(cfg), C.byref(h)))
        self._adopt(h, bounds)

    def _adopt(self, h, bounds):
        self._h = h
        if bounds is None:
            cb = CBounds()
            check(lib().fcm_sampler_get_bounds(h, C.byref(cb)))
            bounds = Bounds._from_c(cb)
        self.bounds = bounds
        self.ncounts = int(lib().fcm_sampler_ncounts(h))
        self.sample_distance = int(lib().fcm_sampler_sample_distance(h))
        info = CSamplerInfo()
        check(lib().fcm_sampler_get_info(h, C.byref(info)))
        self.info = {f: getattr(info, f) for f, _ in CSamplerInfo._fields_}
        self.n_chains = int(self.info["n_chains"])

    # ---- the State API on every chain at once (fcm_sampler_apply_transitions / _revert_transitions / _single_edge_flips) ----
    def _flat_transitions(self, transitions):
        if len(transitions) != self.n_chains:
            raise ValueError("one transition per chain: %d given, %d chains" % (len(transitions), self.n_chains))
        m_cap = max(1, max(len(t.change_edges) for t in transitions))
        e = np.zeros((self.n_chains, m_cap, 2), np.uint32)
        a = np.zeros((self.n_chains, m_cap), np.int32)
        m = np.zeros(self.n_chains, np.uint32)
        for c, t in enumerate(transitions):
            m[c] = len(t.change_edges)
            for i, (ed, ad) in enumerate(t.change_edges):
                e[c, i] = ed
                a[c, i] = 1 if ad else 0
        return e, a, m, m_cap

    def apply_transitions(self, transitions):
        """State::apply_transition (src/lib.rs:61-79) on every chain in one call: transitions[c] for chain c.  Returns
        (counters, status): counters[c] = (pre, post) as the reference returns them, status[c] = 0 or the error code of a
        chain whose transition was refused (that chain is unchanged)."""
        e, a, m, m_cap = self._flat_transitions(transitions)
        pre, post = np.zeros((self.n_chains, _ffi.MAX_COUNTS), np.uint64), np.zeros((self.n_chains, _ffi.MAX_COUNTS), np.uint64)
        pl, ql, st = np.zeros(self.n_chains, np.int32), np.zeros(self.n_chains, np.int32), np.zeros(self.n_chains, np.int32)
        check(lib().fcm_sampler_apply_transitions(self._h, e.ctypes.data_as(u32p), a.ctypes.data_as(i32p), m.ctypes.data_as(u32p), m_cap,
                                                  pre.ctypes.data_as(u64p), pl.ctypes.data_as(i32p), post.ctypes.data_as(u64p), ql.ctypes.data_as(i32p),
                                                  st.ctypes.data_as(i32p)))
        return [([int(v) for v in pre[c, : pl[c]]], [int(v) for v in post[c, : ql[c]]]) for c in range(self.n_chains)], st

    def revert_transitions(self, transitions, counters):
        """State::revert_transition (src/lib.rs:81-95) on every chain in one call; counters[c] = the (pre, post) of chain c.
        A chain whose transition is empty (Transition([])) is left alone."""
        e, a, m, m_cap = self._flat_transitions(transitions)
        pre, post = np.zeros((self.n_chains, _ffi.MAX_COUNTS), np.uint64), np.zeros((self.n_chains, _ffi.MAX_COUNTS), np.uint64)
        pl, ql, st = np.zeros(self.n_chains, np.int32), np.zeros(self.n_chains, np.int32), np.zeros(self.n_chains, np.int32)
        for c, (p_, q_) in enumerate(counters):
            pre[c, : len(p_)] = p_
            post[c, : len(q_)] = q_
            pl[c], ql[c] = len(p_), len(q_)
        check(lib().fcm_sampler_revert_transitions(self._h, e.ctypes.data_as(u32p), a.ctypes.data_as(i32p), m.ctypes.data_as(u32p), m_cap,
                                                   pre.ctypes.data_as(u64p), pl.ctypes.data_as(i32p), post.ctypes.data_as(u64p), ql.ctypes.data_as(i32p),
                                                   st.ctypes.data_as(i32p)))
        return st

    def single_edge_flips(self, xs):
        """Transition::single_edge_flip (src/lib.rs:292-299) drawn on every chain's current graph: xs[c] = one uniform 64-bit
        integer for chain c (the draw is DESIGN.md 3)."""
        x = np.array([int(v) & (2 ** 64 - 1) for v in xs], np.uint64)
        if len(x) != self.n_chains:
            raise ValueError("one number per chain")
        e = np.zeros((self.n_chains, 2, 2), np.uint32)
        a = np.zeros((self.n_chains, 2), np.int32)
        n = np.zeros(self.n_chains, np.uint32)
        check(lib().fcm_sampler_single_edge_flips(self._h, x.ctypes.data_as(u64p), e.ctypes.data_as(u32p), a.ctypes.data_as(i32p), n.ctypes.data_as(u32p)))
        return [Transition([((int(e[c, i, 0]), int(e[c, i, 1])), bool(a[c, i])) for i in range(int(n[c]))]) for c in range(self.n_chains)]

    def save_state(self, fname, sample_number=0):
        """io::save_state (src/io.rs:51-56): written to <fname>.tmp, then renamed."""
        check(lib().fcm_sampler_save_state(self._h, os.fsencode(fname), sample_number))

    @classmethod
    def load_state(cls, fname, device=0):
        """io::load_state (src/io.rs:58-62) -> (sample_number, sampler)."""
        h = C.c_void_p()
        n = C.c_uint64(0)
        check(lib().fcm_sampler_load_state(os.fsencode(fname), device, C.byref(h), C.byref(n)))
        return int(n.value), cls(None, None, _h=h)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib().fcm_sampler_destroy(h)

    def set_stream(self, hip_stream):
        check(lib().fcm_sampler_set_stream(self._h, C.c_void_p(hip_stream)))

    def step(self, n_proposals, sync=True):
        check(lib().fcm_sampler_step(self._h, n_proposals))
        if sync:
            self.sync()

    def sync(self):
        check(lib().fcm_sampler_sync(self._h))

    def next(self):
        """MCMCSampler::next: sample_distance proposals on every chain."""
        check(lib().fcm_sampler_next(self._h))
        return self

    def last_step_ms(self):
        ms = C.c_float(0)
        check(lib().fcm_sampler_last_step_ms(self._h, C.byref(ms)))
        return ms.value

    def flag_counts(self, with_len=False):
        out = np.zeros((self.n_chains, self.ncounts), np.uint64)
        ln = np.zeros(self.n_chains, np.int32)
        check(lib().fcm_sampler_get_counts(self._h, out.ctypes.data_as(u64p), ln.ctypes.data_as(i32p)))
        return (out, ln) if with_len else out

    def flag_count(self, chain=0):
        """The chain's `state.flag_count` as the reference would print it."""
        out, ln = self.flag_counts(with_len=True)
        return [int(x) for x in out[chain, : ln[chain]]]

    def stats(self):
        out = np.zeros((self.n_chains, _ffi.NSTATS), np.uint64)
        check(lib().fcm_sampler_get_stats(self._h, out.ctypes.data_as(u64p)))
        # (every slot has one meaning on every sampler, include/fcm.h FCM_STAT_*: n_recheck / n_held stay 0 on a sampler
        # with clique moves, n_pairs / n_shared_rows on a simple-move one)
        return {name: out[:, i].copy() for i, name in enumerate(_ffi.STAT_NAMES)}

    @property
    def sampled(self):
        return self.stats()["sampled"]

    @property
    def accepted(self):
        return self.stats()["accepted"]

    def acceptance_ratio(self):
        st = self.stats()
        return st["accepted"].astype(np.float64) / st["sampled"].astype(np.float64)

    def edges(self, chain=0):
        m = C.c_uint64(0)
        check(lib().fcm_sampler_get_edges(self._h, chain, None, 0, C.byref(m)))
        out = np.zeros((max(m.value, 1), 2), np.uint32)
        check(lib().fcm_sampler_get_edges(self._h, chain, out.ctypes.data_as(u32p), m.value, C.byref(m)))
        return out[: m.value]

    def graph(self, chain=0):
        return Graph.from_edges(self.info["n"], self.edges(chain))

    def edgebits(self, chain=0):
        n = C.c_uint64(0)
        check(lib().fcm_sampler_get_edgebits(self._h, chain, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), np.uint8)
        check(lib().fcm_sampler_get_edgebits(self._h, chain, out.ctypes.data_as(C.POINTER(C.c_uint8)), n.value, C.byref(n)))
        return out[: n.value]

    def state(self, chain=0):
        """The chain's `State` (MCMCSampler::state, src/lib.rs:166)."""
        return State(self, chain)

    def double_slots(self, chain=0):
        n = C.c_uint64(0)
        check(lib().fcm_sampler_get_double_slots(self._h, chain, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), np.uint32)
        check(lib().fcm_sampler_get_double_slots(self._h, chain, out.ctypes.data_as(u32p), n.value, C.byref(n)))
        return out[: n.value]


class Transition:
    """`Transition { change_edges: Vec<([Node; 2], bool)> }` (src/lib.rs:200-204); `True` = add the edge."""

    def __init__(self, change_edges=()):
        self.change_edges = [((int(e[0]), int(e[1])), bool(add)) for e, add in change_edges]

    def _flat(self):
        e = np.array([c[0] for c in self.change_edges], np.uint32).reshape(-1, 2)
        a = np.array([1 if c[1] else 0 for c in self.change_edges], np.int32)
        return np.ascontiguousarray(e), np.ascontiguousarray(a)

    @classmethod
    def single_edge_flip(cls, state, x):
        """Transition::single_edge_flip (src/lib.rs:292-299) on `state`'s current graph.  The reference takes an rng;
        here the caller passes one uniform 64-bit integer `x` (the draw is DESIGN.md 3)."""
        e = np.zeros((2, 2), np.uint32)
        a = np.zeros(2, np.int32)
        n = C.c_uint32(0)
        check(lib().fcm_sampler_single_edge_flip(state._s._h, state._chain, int(x) & (2 ** 64 - 1), e.ctypes.data_as(u32p), a.ctypes.data_as(i32p), C.byref(n)))
        return cls([((int(e[i, 0]), int(e[i, 1])), bool(a[i])) for i in range(n.value)])

    def __repr__(self):
        return "Transition(%r)" % (self.change_edges,)


class State:
    """The reference's `State` (src/lib.rs:29-112) of one chain of a sampler: a view, the data stays on the GPU."""

    def __init__(self, sampler, chain):
        self._s, self._chain = sampler, int(chain)

    @property
    def flag_count(self):
        return self._s.flag_count(self._chain)

    @property
    def graph(self):
        return self._s.graph(self._chain)

    def edgeset_neighborhood(self, edges):
        """State::edgeset_neighborhood (src/lib.rs:99-111)."""
        e, ep = _u32(np.asarray(edges, np.uint32).reshape(-1, 2))
        k = C.c_uint64(0)
        check(lib().fcm_sampler_edgeset_neighborhood(self._s._h, ep, len(e), None, 0, C.byref(k)))
        out = np.zeros(max(k.value, 1), np.uint32)
        check(lib().fcm_sampler_edgeset_neighborhood(self._s._h, ep, len(e), out.ctypes.data_as(u32p), k.value, C.byref(k)))
        return [int(v) for v in out[: k.value]]

    def apply_transition(self, t):
        """State::apply_transition (src/lib.rs:61-79) -> (pre, post), the reference's two count vectors."""
        e, a = t._flat()
        pre, post = np.zeros(_ffi.MAX_COUNTS, np.uint64), np.zeros(_ffi.MAX_COUNTS, np.uint64)
        pl, ql = C.c_int32(0), C.c_int32(0)
        check(lib().fcm_sampler_apply_transition(self._s._h, self._chain, e.ctypes.data_as(u32p), a.ctypes.data_as(i32p), len(a),
                                                 pre.ctypes.data_as(u64p), C.byref(pl), post.ctypes.data_as(u64p), C.byref(ql)))
        return [int(v) for v in pre[: pl.value]], [int(v) for v in post[: ql.value]]

    def revert_transition(self, t, counters):
        """State::revert_transition (src/lib.rs:81-95); `counters` = the (pre, post) apply_transition returned."""
        e, a = t._flat()
        pre, pp = _u64(list(counters[0]) + [0])
        post, qp = _u64(list(counters[1]) + [0])
        check(lib().fcm_sampler_revert_transition(self._s._h, self._chain, e.ctypes.data_as(u32p), a.ctypes.data_as(i32p), len(a),
                                                  pp, len(counters[0]), qp, len(counters[1])))


class MultiDeviceSampler:
    """Chains sharded over several devices of ONE process: one `MCMCSampler` handle per entry of `devices`, each
    driven by its own host thread (libfcm handles are independent; the reference's multi-chain precedent runs its
    `State`s on OS threads, src/bin/all_cxs.rs:33-38).  Global chain c lives on shard c // ceil(C/G) and draws from
    stream (seed, c): every result equals the single-handle run's chain for chain, whatever the device list is (a
    device may be named twice).  The gather is host-side concatenation in global chain order."""

    def __init__(self, graph, bounds, n_chains, devices, seed=0, move_weights=MOVE_DISTRIBUTION_SIMPLE,
                 sample_distance=0, dim_cap=0):
        from .distributed import shard_range
        self.devices, self.ranges = [], []
        for r, dev in enumerate(devices):
            lo, hi = shard_range(n_chains, r, len(devices))
            if hi > lo:
                self.devices.append(dev)
                self.ranges.append((lo, hi))
        self.shards = self._parallel(lambda r: MCMCSampler(graph, bounds, n_chains=self.ranges[r][1] - self.ranges[r][0], seed=seed,
                                                           move_weights=move_weights, sample_distance=sample_distance, dim_cap=dim_cap,
                                                           device=self.devices[r], first_chain_id=self.ranges[r][0]))
        self.n_chains = n_chains
        self.bounds = bounds

    def _parallel(self, fn):
        """fn(r) for every shard, each on its own thread (ctypes releases the GIL inside libfcm)."""
        import threading
        n = len(self.ranges)
        out, err = [None] * n, [None] * n

        def run(r):
            try:
                out[r] = fn(r)
            except BaseException as e:  # noqa: BLE001 -- carried to the caller's thread
                err[r] = e
        th = [threading.Thread(target=run, args=(r,)) for r in range(1, n)]
        for t in th:
            t.start()
        run(0)
        for t in th:
            t.join()
        for e in err:
            if e is not None:
                raise e
        return out

    def step(self, n_proposals):
        self._parallel(lambda r: self.shards[r].step(n_proposals))

    def next(self):
        self._parallel(lambda r: self.shards[r].next())
        return self

    @staticmethod
    def shard_file(fname, r, n_shards):
        """<fname> for a single shard (the single-device layout), <fname>.shard<r> otherwise."""
        return fname if n_shards == 1 else "%s.shard%d" % (fname, r)

    def save_state(self, fname, sample_number=0):
        """One state file per shard, each carrying its place in the set (fcm_sampler_save_state_shard): all of them are
        written to <file>.new first and moved into place once every one is complete."""
        import time
        G = len(self.shards)
        set_id = ((sample_number << 32) ^ int(time.time()) ^ (os.getpid() << 20)) & 0xFFFFFFFFFFFFFFFF
        self._parallel(lambda r: check(lib().fcm_sampler_save_state_shard(
            self.shards[r]._h, os.fsencode(self.shard_file(fname, r, G) + ".new"), sample_number, r, G, self.n_chains, set_id)))
        for r in range(G):
            os.replace(self.shard_file(fname, r, G) + ".new", self.shard_file(fname, r, G))

    @staticmethod
    def state_file_info(fname):
        info = _ffi.CStateInfo()
        check(lib().fcm_state_file_info(os.fsencode(fname), C.byref(info)))
        return {f: int(getattr(info, f)) for f, _ in _ffi.CStateInfo._fields_}

    @classmethod
    def load_state(cls, fname, devices):
        """-> (sample_number, sampler).  The number of shards comes from the files: shard r goes to devices[r % len(devices)]
        (a state saved on 4 handles resumes on 2 devices or on 1).  Raises if a shard is missing, if the files are not of
        one save, or if their chain ranges do not tile 0 .. total-1."""
        if not devices:
            raise FcmError(_ffi.ERR_INVALID, "no devices")
        sharded = os.path.exists(fname + ".shard0")
        if sharded and os.path.exists(fname):   # both layouts on disk: the newer save
            one, many = cls.state_file_info(fname), cls.state_file_info(fname + ".shard0")
            first = many if many["sample_number"] >= one["sample_number"] else one
        else:
            first = cls.state_file_info(fname + ".shard0" if sharded else fname)
        G = first["shard_count"]
        at = first["first_chain_id"] if first["shard_index"] == 0 else 0   # (a single handle's file may start at any global chain id)
        for r in range(G):
            f = cls.shard_file(fname, r, G)
            if not os.path.exists(f):
                raise FcmError(_ffi.ERR_IO, "%s: shard %d of %d is missing" % (f, r, G))
            h = cls.state_file_info(f)
            if h["shard_index"] != r or any(h[k] != first[k] for k in ("shard_count", "total_chains", "set_id", "sample_number", "seed", "n")):
                raise FcmError(_ffi.ERR_IO, "%s: not a shard of the same save as %s" % (f, cls.shard_file(fname, 0, G)))
            if h["first_chain_id"] != at:
                raise FcmError(_ffi.ERR_IO, "%s: its chains start at %d, the shards before it end at %d" % (f, h["first_chain_id"], at))
            at += h["n_chains"]
        if at != first["total_chains"]:
            raise FcmError(_ffi.ERR_IO, "%s: the shards hold %d chains, the run had %d" % (fname, at, first["total_chains"]))
        self = cls.__new__(cls)
        self.devices = [devices[r % len(devices)] for r in range(G)]
        self.ranges = [None] * G
        self.shards = self._parallel(lambda r: MCMCSampler.load_state(cls.shard_file(fname, r, G), self.devices[r])[1])
        self.ranges, lo = [], 0
        for s in self.shards:
            self.ranges.append((lo, lo + s.n_chains))
            lo += s.n_chains
        self.n_chains = lo
        self.bounds = None
        return first["sample_number"], self

    def flag_counts(self):
        return np.concatenate(self._parallel(lambda r: self.shards[r].flag_counts()), axis=0)

    def stats(self):
        parts = self._parallel(lambda r: self.shards[r].stats())
        return {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}

    def locate(self, chain):
        for r, (lo, hi) in enumerate(self.ranges):
            if lo <= chain < hi:
                return r, chain - lo
        raise IndexError(chain)

    def flag_count(self, chain=0):
        r, c = self.locate(chain)
        return self.shards[r].flag_count(c)

    def edges(self, chain=0):
        r, c = self.locate(chain)
        return self.shards[r].edges(c)

    def edgebits(self, chain=0):
        r, c = self.locate(chain)
        return self.shards[r].edgebits(c)

    def double_slots(self, chain=0):
        r, c = self.locate(chain)
        return self.shards[r].double_slots(c)


def initialize_new_sampler(input, target_relaxation=0.01, seed=0, sample_distance=0, simple=True,
                           n_chains=1, device=0, dim_cap=0, first_chain_id=0):
    """initialize_new_sampler (src/bin/sample.rs:80-105) for a batch of chains."""
    g = input if isinstance(input, Graph) else read_flag_file(input)
    flag_count = g.flagser_count(device)                       # State::new, src/lib.rs:51
    target = Bounds.target(flag_count, target_relaxation)      # sample.rs:89-95
    bounds = Bounds.calculate(g, flag_count, target, device)   # sample.rs:96-100
    weights = MOVE_DISTRIBUTION_SIMPLE if simple else MOVE_DISTRIBUTION
    return MCMCSampler(g, bounds, n_chains=n_chains, seed=seed, move_weights=weights,
                       sample_distance=sample_distance, dim_cap=dim_cap, device=device,
                       first_chain_id=first_chain_id)


class BitOutput:
    """`io::BitOutput` (src/io.rs:128-212): <dir>/graph.flag plus N.edgebits
    files, each a run of fixed-size records (2 bits per adjacent pair)."""

    def __init__(self, graph, dir):
        os.makedirs(dir, exist_ok=True)
        save_flag_file(os.path.join(dir, "graph.flag"), graph)
        nslots = 2 * len(graph.undirected_edges())
        if nslots // 8 == 0:
            # the reference divides by zero here (src/io.rs:161)
            raise FcmError(_ffi.ERR_PANIC, "fewer than 8 edge slots: reference BitOutput::new divides by zero")
        self.chunk_size = max(2_000_000_000 // (nslots // 8), 1)
        self.index_in_file = 0
        self.index_in_dir = 0
        self.current_file = None
        self.dir = dir

    def save(self, sampler, chain=0):
        if self.index_in_file == 0:
            self.current_file = open(os.path.join(self.dir, "%d.edgebits" % self.index_in_dir), "wb")
        self.current_file.write(sampler.edgebits(chain).tobytes())
        self.index_in_file += 1
        if self.index_in_file == self.chunk_size:
            self.current_file.close()
            self.current_file = None
            self.index_in_file = 0
            self.index_in_dir += 1

    def flush(self):
        if self.current_file:
            self.current_file.flush()
