"""The State API through the C ABI (fcm_sampler_apply_transition / _revert_transition / _single_edge_flip /
_edgeset_neighborhood) against the oracle's restatement of State::apply_transition / revert_transition
(reference src/lib.rs:61-111, 292-299; callers: src/bin/seo_search_counterexample.rs:51-89,
seo_bt_flip_only_once.rs:65-69).  (pre, post) are compared as the reference returns them -- the counts of the
induced subgraph before and after, lengths included -- not only their difference."""
import numpy as np
import pytest

from helpers import setup_pair

pytestmark = pytest.mark.gpu


def _sampler(fcm, oracle, n, e, weights=(0.5, 0.5, 0.0, 0.0), n_chains=2):
    gg, go, bg, bo = setup_pair(fcm, oracle, n, e)
    s = fcm.MCMCSampler(gg, bg, n_chains=n_chains, seed=5, move_weights=weights)
    return s, oracle.State(go), go, bo


def _same(st_g, st_o, ctx=""):
    assert st_g.flag_count == st_o.flag_count, ctx
    ge, oe = st_g._s.edges(st_g._chain), st_o.graph_edges()
    assert ge.shape == oe.shape and (ge == oe).all(), ctx


def _reciprocal_ids(ue, edges):
    have = {(int(a), int(b)) for a, b in edges}
    return [i for i, (a, b) in enumerate(ue) if (int(a), int(b)) in have and (int(b), int(a)) in have]


def test_apply_revert_ex04_to_ex05(fcm, oracle):
    # ex04 -> ex05 is the flip of 3->1; both have counts [4, 5, 2]
    e = fcm.graphs.densifier([0, 0, 1, 3, 3], [1, 2, 2, 1, 2])
    s, so, _, _ = _sampler(fcm, oracle, 4, e)
    sg = s.state(1)
    t = fcm.Transition([((3, 1), False), ((1, 3), True)])
    assert sg.edgeset_neighborhood([(3, 1)]) == so.edgeset_neighborhood([(3, 1)])
    pre, post = sg.apply_transition(t)
    assert (pre, post) == so.apply_transition(t.change_edges)
    _same(sg, so)
    assert sg.flag_count == [4, 5, 2]
    assert sorted(map(tuple, s.edges(1).tolist())) == sorted(map(tuple, fcm.graphs.densifier([0, 0, 1, 1, 3], [1, 2, 2, 3, 2]).tolist()))
    assert s.flag_count(0) == [4, 5, 2] and (s.edges(0) == np.array(sorted(map(tuple, e.tolist())), np.uint32)).all()   # the other chain is untouched
    sg.revert_transition(t, (pre, post))
    so.revert_transition(t.change_edges, (pre, post))
    _same(sg, so)
    # the empty transition (src/lib.rs:298): empty neighbourhood, nothing changes
    pe, qe = sg.apply_transition(fcm.Transition([]))
    assert (pe, qe) == so.apply_transition([])
    _same(sg, so)


def test_random_transitions_against_the_oracle(fcm, oracle):
    n = 70
    e = fcm.graphs.random_with_p(n, 0.22, seed=11)
    s, so, go, _ = _sampler(fcm, oracle, n, e, weights=(0.1, 0.1, 0.6, 0.2))     # clique-move sampler: slot_of is kept too
    sg = s.state(0)
    ue = so.undirected_edges()
    rng = np.random.default_rng(3)
    history = []
    nontrivial = 0
    for it in range(200):
        cur = {(int(a), int(b)) for a, b in so.graph_edges()}
        single = [(int(a), int(b)) for a, b in ue if ((int(a), int(b)) in cur) != ((int(b), int(a)) in cur)]
        recip = [(int(a), int(b)) for a, b in ue if (int(a), int(b)) in cur and (int(b), int(a)) in cur]
        kind = it % 4
        ch = []
        if kind == 0:      # a flip, written as the reference's generators do
            a, b = single[rng.integers(len(single))]
            f, t_ = (a, b) if (a, b) in cur else (b, a)
            ch = [((f, t_), False), ((t_, f), True)]
        elif kind == 1:    # the shape of double_edge_move: one direction of a reciprocal pair goes, a single edge gets its reverse
            a, b = recip[rng.integers(len(recip))]
            c, d = single[rng.integers(len(single))]
            f, t_ = (c, d) if (c, d) in cur else (d, c)
            ch = [((t_, f), True), (((a, b) if rng.integers(2) else (b, a)), False)]
        elif kind == 2:    # several flips at once (a union neighbourhood, like a clique move's)
            for j in rng.choice(len(single), size=4, replace=False):
                a, b = single[j]
                f, t_ = (a, b) if (a, b) in cur else (b, a)
                ch += [((f, t_), False), ((t_, f), True)]
        else:              # undo something
            if history:
                tt, cnt = history.pop()
                sg.revert_transition(fcm.Transition(tt), cnt)
                so.revert_transition(tt, cnt)
                _same(sg, so, "revert %d" % it)
            continue
        assert sg.edgeset_neighborhood([c[0] for c in ch]) == so.edgeset_neighborhood([c[0] for c in ch])
        got = sg.apply_transition(fcm.Transition(ch))
        want = so.apply_transition(ch)
        assert got == want, (it, ch, got, want)
        nontrivial += got[0] != got[1]
        _same(sg, so, "apply %d" % it)
        history = [(ch, got)]     # only the latest transition can be reverted exactly (the reference keeps no stack either)
    assert nontrivial > 50
    # the sampler goes on from the state the API left: slot list = the reciprocal pairs, counts = a full recount
    assert sorted(int(x) for x in s.double_slots(0)) == _reciprocal_ids(ue, s.edges(0))
    s.step(400)
    assert (s.stats()["status"] == 0).all()
    for c in (0, 1):
        assert s.graph(c).flagser_count() == s.flag_count(c)[: len(s.graph(c).flagser_count())]
        assert sorted(int(x) for x in s.double_slots(c)) == _reciprocal_ids(ue, s.edges(c))


def test_greedy_search_loop_like_the_reference_tool(fcm, oracle):
    """The loop of src/bin/seo_search_counterexample.rs:51-89 on both sides: draw a flip, apply, keep it only if the
    number of 2-simplices grew, else revert."""
    n = 60
    e = fcm.graphs.seoify(fcm.graphs.random_with_p(n, 0.3, seed=2), seed=2)
    s, so, go, _ = _sampler(fcm, oracle, n, e, n_chains=1)
    sg = s.state(0)
    ue = so.undirected_edges()
    U = len(ue)
    rng = np.random.default_rng(9)
    kept = 0
    for it in range(300):
        x = int(rng.integers(0, 2 ** 63)) * 2 + int(rng.integers(0, 2))
        t = fcm.Transition.single_edge_flip(sg, x)
        # the same draw on the oracle's graph (no reciprocal pairs here, so D = 0)
        r = (x * U) >> 64
        big, small = int(ue[r][0]), int(ue[r][1])
        g_o = so.graph()
        f, t_ = (big, small) if g_o.has_edge(big, small) else (small, big)
        assert t.change_edges == [((f, t_), False), ((t_, f), True)], it
        pre, post = sg.apply_transition(t)
        assert (pre, post) == so.apply_transition(t.change_edges), it
        accept = not (len(post) < len(pre)) and not (len(post) > 2 and len(pre) > 2 and post[2] <= pre[2])
        if not accept:
            sg.revert_transition(t, (pre, post))
            so.revert_transition(t.change_edges, (pre, post))
        kept += accept
        assert sg.flag_count == so.flag_count, it
    assert 10 < kept < 290
    _same(sg, so)
    assert sg.flag_count[2] > go.flagser_count()[2]


def test_transitions_the_api_refuses_change_nothing(fcm, oracle):
    n = 40
    e = fcm.graphs.random_with_p(n, 0.25, seed=4)
    s, so, go, _ = _sampler(fcm, oracle, n, e)
    sg = s.state(0)
    before = (sg.flag_count, s.edges(0).tolist(), s.double_slots(0).tolist())
    und = {(int(a), int(b)) for a, b in so.undirected_edges()}
    cur = {(int(a), int(b)) for a, b in e}
    non_adjacent = next((a, b) for a in range(n) for b in range(a) if (a, b) not in und)
    with pytest.raises(fcm.FcmError) as ei:       # the reference panics on the HashMap index (src/lib.rs:104)
        sg.apply_transition(fcm.Transition([(non_adjacent, True)]))
    assert ei.value.code == fcm._ffi.ERR_PANIC
    a, b = next((a, b) for a, b in und if ((a, b) in cur) != ((b, a) in cur))
    f, t_ = (a, b) if (a, b) in cur else (b, a)
    with pytest.raises(fcm.FcmError) as ei:       # one more reciprocal pair
        sg.apply_transition(fcm.Transition([((t_, f), True)]))
    assert ei.value.code == fcm._ffi.ERR_UNSUPPORTED
    with pytest.raises(fcm.FcmError) as ei:       # the pair's only edge goes
        sg.apply_transition(fcm.Transition([((f, t_), False)]))
    assert ei.value.code == fcm._ffi.ERR_UNSUPPORTED
    with pytest.raises(fcm.FcmError):
        sg.apply_transition(fcm.Transition([((n, 0), True)]))
    with pytest.raises(fcm.FcmError) as ei:       # counts that are not there: the reference's assert!(*s >= *p)
        sg.revert_transition(fcm.Transition([((f, t_), False), ((t_, f), True)]), ([n, 10 ** 9], [n, 10 ** 9, 10 ** 9]))
    assert ei.value.code == fcm._ffi.ERR_PANIC
    assert before == (sg.flag_count, s.edges(0).tolist(), s.double_slots(0).tolist())


def test_greedy_search_on_256_chains_at_once(fcm, oracle):
    """the loop of src/bin/seo_search_counterexample.rs:51-89 -- draw a flip, apply it, keep it only if
    the number of 2-simplices grew, else revert -- on 256 chains of one handle in lockstep through the batched calls
    (fcm_sampler_single_edge_flips / _apply_transitions / _revert_transitions: one launch per call for all chains), against
    256 oracle States driven one by one.  Every (pre, post), every draw and every final state equal."""
    import time
    n, C, iters = 60, 256, 40
    e = fcm.graphs.seoify(fcm.graphs.random_with_p(n, 0.3, seed=2), seed=2)
    gg, go, bg, bo = setup_pair(fcm, oracle, n, e)
    s = fcm.MCMCSampler(gg, bg, n_chains=C, seed=5)
    sos = [oracle.State(go) for _ in range(C)]
    ue = sos[0].undirected_edges()
    U = len(ue)
    rng = np.random.default_rng(17)
    kept = np.zeros(C, np.int64)
    t_gpu = 0.0
    for it in range(iters):
        xs = [int(rng.integers(0, 2 ** 63)) * 2 + int(rng.integers(0, 2)) for _ in range(C)]
        t0 = time.perf_counter()
        ts = s.single_edge_flips(xs)
        counters, st = s.apply_transitions(ts)
        t_gpu += time.perf_counter() - t0
        assert (st == 0).all()
        back, back_counters = [], []
        for c in range(C):
            r = (xs[c] * U) >> 64                                         # the same draw on the oracle's graph (no reciprocal pairs: D = 0)
            big, small = int(ue[r][0]), int(ue[r][1])
            f, t_ = (big, small) if sos[c].graph().has_edge(big, small) else (small, big)
            assert ts[c].change_edges == [((f, t_), False), ((t_, f), True)], (it, c)
            pre, post = counters[c]
            assert (pre, post) == sos[c].apply_transition(ts[c].change_edges), (it, c)
            accept = not (len(post) < len(pre)) and not (len(post) > 2 and len(pre) > 2 and post[2] <= pre[2])
            kept[c] += accept
            if accept:
                back.append(fcm.Transition([])); back_counters.append(([], []))
            else:
                sos[c].revert_transition(ts[c].change_edges, (pre, post))
                back.append(ts[c]); back_counters.append((pre, post))
        t0 = time.perf_counter()
        st = s.revert_transitions(back, back_counters)
        t_gpu += time.perf_counter() - t0
        assert (st == 0).all()
    fcs = s.flag_counts(with_len=True)
    for c in range(C):
        assert [int(v) for v in fcs[0][c, : fcs[1][c]]] == sos[c].flag_count, c
    for c in (0, 1, 100, 255):
        assert (s.edges(c) == sos[c].graph_edges()).all()
    assert kept.min() >= 1 and kept.max() < iters and len(set(kept.tolist())) > 3        # the chains went different ways
    print("batched State API: %d chains x %d iterations, %.1f us per transition (draw + apply + revert calls)" % (C, iters, 1e6 * t_gpu / (C * iters)))
    # the sampler steps on from the states the search left
    s.step(300)
    assert (s.stats()["status"] == 0).all()
    for c in (3, 200):
        assert s.graph(c).flagser_count() == s.flag_count(c)[: len(s.graph(c).flagser_count())]


def test_batched_transitions_refuse_per_chain_and_fall_back(fcm, oracle):
    """Per-chain outcomes of one batched call: an empty transition, a flip, a transition on two pairs (taken by the one-chain
    path), one the API refuses (the pair's only edge goes) and one off pr(G) (the reference's HashMap panic) -- the refused
    chains are left as they were, the others are applied, and each chain's code says which."""
    n = 40
    e = fcm.graphs.random_with_p(n, 0.25, seed=4)
    gg, go, bg, bo = setup_pair(fcm, oracle, n, e)
    s = fcm.MCMCSampler(gg, bg, n_chains=5, seed=1)
    so = [oracle.State(go) for _ in range(5)]
    und = {(int(a), int(b)) for a, b in so[0].undirected_edges()}
    cur = {(int(a), int(b)) for a, b in e}
    singles = [(a, b) for a, b in sorted(und) if ((a, b) in cur) != ((b, a) in cur)]
    def flip(p):
        a, b = p
        f, t_ = (a, b) if (a, b) in cur else (b, a)
        return [((f, t_), False), ((t_, f), True)]
    non_adjacent = next((a, b) for a in range(n) for b in range(a) if (a, b) not in und)
    two_pairs = flip(singles[0]) + flip(singles[5])
    only_edge_goes = [flip(singles[1])[0]]
    ts = [fcm.Transition([]), fcm.Transition(flip(singles[2])), fcm.Transition(two_pairs), fcm.Transition(only_edge_goes), fcm.Transition([(non_adjacent, True)])]
    before = [(s.flag_count(c), s.edges(c).tolist()) for c in range(5)]
    counters, st = s.apply_transitions(ts)
    assert st.tolist() == [0, 0, 0, fcm._ffi.ERR_UNSUPPORTED, fcm._ffi.ERR_PANIC]
    for c in (0, 1, 2):
        assert counters[c] == so[c].apply_transition(ts[c].change_edges), c
        assert s.flag_count(c) == so[c].flag_count and (s.edges(c) == so[c].graph_edges()).all()
    for c in (3, 4):
        assert (s.flag_count(c), s.edges(c).tolist()) == before[c]
    st = s.revert_transitions([ts[0], ts[1], ts[2], fcm.Transition([]), fcm.Transition([])], [counters[0], counters[1], counters[2], ([], []), ([], [])])
    assert (st == 0).all()
    for c in range(5):
        assert (s.flag_count(c), s.edges(c).tolist()) == before[c], c


def test_apply_transition_on_a_dim_cap_sampler_stays_consistent_with_step(fcm, oracle):
    """A sampler in truncated mode (dim_cap) tracks fewer count entries than the complex has.  The State API still returns the
    reference's full-length (pre, post) -- compared with the oracle -- while the handle's own flag_count moves only in the
    tracked entries; the sampler then goes on from that state and its tracked counts still equal a full recount."""
    n = 60
    e = fcm.graphs.random_with_p(n, 0.3, seed=5)
    gg, go, bg, bo = setup_pair(fcm, oracle, n, e)
    full_len = len(go.flagser_count())
    cap = 3
    s = fcm.MCMCSampler(gg, bg, n_chains=2, seed=9, dim_cap=cap)
    assert s.info["lossless"] == 0 and s.ncounts == cap + 1 < full_len
    so = oracle.State(go)
    sg = s.state(1)
    ue = so.undirected_edges()
    rng = np.random.default_rng(8)
    deeper = 0
    for it in range(40):
        cur = {(int(a), int(b)) for a, b in so.graph_edges()}
        single = [(int(a), int(b)) for a, b in ue if ((int(a), int(b)) in cur) != ((int(b), int(a)) in cur)]
        a, b = single[rng.integers(len(single))]
        f, t_ = (a, b) if (a, b) in cur else (b, a)
        ch = [((f, t_), False), ((t_, f), True)]
        got = sg.apply_transition(fcm.Transition(ch))
        want = so.apply_transition(ch)
        assert got == want, (it, got, want)            # the full vectors, whatever the handle tracks
        deeper += max(len(got[0]), len(got[1])) > s.ncounts
        assert list(s.flag_count(1))[: s.ncounts] == list(so.flag_count)[: s.ncounts], it
        ge, oe = s.edges(1), so.graph_edges()
        assert ge.shape == oe.shape and (ge == oe).all()
    assert deeper > 0                                   # some neighbourhoods were deeper than the cap: the case the test is for
    s.step(300)
    assert (s.stats()["status"] == 0).all()
    for c in range(2):
        assert s.graph(c).flagser_count()[: s.ncounts] == list(s.flag_count(c))[: s.ncounts]
