"""Chains sharded over several handles in ONE process, one host thread per handle (north_star: "independent chains
shard trivially across the 8 GPUs"; reference precedent for threads: src/bin/all_cxs.rs:33-38).  On the 1-GPU box
the device list names GPU 0 twice: two handles, two threads, one GPU -- results must equal the single-handle run chain
for chain, which also exercises include/fcm.h's threading promise (distinct handles from distinct threads,
thread-local fcm_last_error, hipSetDevice per calling thread)."""
import os
import subprocess
import threading

import numpy as np
import pytest

from helpers import load_flag_fixture, setup_pair

pytestmark = pytest.mark.gpu


def test_two_handles_on_two_threads_equal_one_handle(fcm, oracle):
    n = 300
    e = fcm.graphs.random_with_p(n, 0.12, seed=8)
    gg, go, bg, bo = setup_pair(fcm, oracle, n, e)
    one = fcm.MCMCSampler(gg, bg, n_chains=7, seed=3)
    multi = fcm.MultiDeviceSampler(gg, bg, 7, devices=[0, 0, 0], seed=3)
    assert multi.ranges == [(0, 3), (3, 6), (6, 7)]
    for _ in range(3):
        one.step(700)
        multi.step(700)
        assert (multi.flag_counts() == one.flag_counts()).all()
    st1, stm = one.stats(), multi.stats()
    for k in ("sampled", "accepted", "n_empty", "n_flip", "n_dmove", "sum_k", "status"):
        assert (st1[k] == stm[k]).all(), k
    for c in range(7):
        assert (multi.edges(c) == one.edges(c)).all() and (multi.edgebits(c) == one.edgebits(c)).all()
        assert (multi.double_slots(c) == one.double_slots(c)).all()
    # ... and chain 4 against its oracle twin: stream (seed, global chain id) whatever the sharding
    tw = oracle.Chain(go, bo, seed=3, chain_id=4)
    tw.step(2100)
    assert multi.flag_count(4) == tw.state.flag_count and (multi.edges(4) == tw.state.graph_edges()).all()


def test_two_handles_default_move_mix(fcm, oracle):
    """The reference's default mix (clique moves: the cooperative kernel and its per-chain slot index) sharded over two
    handles on two threads: chain for chain the single-handle run."""
    n = 200
    e = fcm.graphs.random_with_p(n, 0.12, seed=4)
    gg, go, bg, bo = setup_pair(fcm, oracle, n, e)
    one = fcm.MCMCSampler(gg, bg, n_chains=5, seed=8, move_weights=fcm.MOVE_DISTRIBUTION)
    multi = fcm.MultiDeviceSampler(gg, bg, 5, devices=[0, 0], seed=8, move_weights=fcm.MOVE_DISTRIBUTION)
    for _ in range(2):
        one.step(300)
        multi.step(300)
    assert (multi.flag_counts() == one.flag_counts()).all() and (multi.stats()["n_cperm"] == one.stats()["n_cperm"]).all()
    for c in range(5):
        assert (multi.edges(c) == one.edges(c)).all() and (multi.double_slots(c) == one.double_slots(c)).all()
    tw = oracle.Chain(go, bo, weights=fcm.MOVE_DISTRIBUTION, seed=8, chain_id=3)
    tw.step(600)
    assert multi.flag_count(3) == tw.state.flag_count and (multi.edges(3) == tw.state.graph_edges()).all()


def test_errors_are_per_thread(fcm, oracle):
    """fcm_last_error is thread-local: a failing call on one thread does not disturb a handle stepping on another."""
    n = 200
    e = fcm.graphs.random_with_p(n, 0.1, seed=1)
    gg, go, bg, bo = setup_pair(fcm, oracle, n, e)
    s = fcm.MCMCSampler(gg, bg, n_chains=4, seed=9)
    ref = fcm.MCMCSampler(gg, bg, n_chains=4, seed=9)
    ref.step(3000)
    errs = []

    def bad():
        for _ in range(50):
            try:
                fcm.MCMCSampler(gg, bg, n_chains=0)
            except fcm.FcmError as ex:
                errs.append(str(ex))
    t = threading.Thread(target=bad)
    t.start()
    for _ in range(30):
        s.step(100)
    t.join()
    assert len(errs) == 50 and all("n_chains" in m for m in errs)
    assert (s.flag_counts() == ref.flag_counts()).all()


def test_sample_cli_devices_list_equals_one_device(fcm, golden_dir, tmp_path):
    exe = os.path.join(os.path.dirname(fcm.LIB_PATH), "sample")
    flag = os.path.join(golden_dir, "bug_calc_relax_de.flag")
    outs = {}
    for tag, devargs in (("one", ["--device", "0"]), ("two", ["--devices", "0,0"])):
        cmd = [exe, "-i", flag, "-l", "lab", "-s", "6", "--simple", "--chains", "5", "--sample-distance", "200", "-n", "3",
               "--samples-store-dir", str(tmp_path / tag / "samples"), "--state-store-dir", str(tmp_path / tag / "state"),
               "--state-save-interval", "2"] + devargs
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        outs[tag] = r.stdout
    assert "5 chains on 2 handles: device 0: 3 device 0: 2" in outs["two"]
    pick = lambda txt: [ln for ln in txt.split("\n") if ln.startswith("flag count:") or ln.startswith("[sample")]
    assert pick(outs["one"]) == pick(outs["two"]) and len(pick(outs["one"])) == 6
    for c in range(5):
        a = tmp_path / "one" / "samples" / "lab-006" / ("chain%05d" % c) / "0.edgebits"
        b = tmp_path / "two" / "samples" / "lab-006" / ("chain%05d" % c) / "0.edgebits"
        assert a.read_bytes() == b.read_bytes() and len(a.read_bytes()) > 0
    # the sharded run resumes from its per-shard state files and ends where five samples straight end
    st = tmp_path / "two" / "state" / "sampler-lab-006.state"
    assert (tmp_path / "two" / "state" / "sampler-lab-006.state.shard0").exists() and (tmp_path / "two" / "state" / "sampler-lab-006.state.shard1").exists()
    r2 = subprocess.run([exe, "-c", str(st), "-l", "lab", "-s", "6", "-n", "2", "--devices", "0,0", "--samples-store-dir", str(tmp_path / "two" / "s2"),
                         "--state-store-dir", str(tmp_path / "two" / "state")], capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0, r2.stdout + r2.stderr
    s = fcm.initialize_new_sampler(flag, target_relaxation=0.01, seed=6, n_chains=5, sample_distance=200)
    for _ in range(5):
        s.next()
    assert "flag count: %s" % s.flag_count(0) in r2.stdout
    n1, sh1 = fcm.MCMCSampler.load_state(str(st) + ".shard1")
    assert n1 == 5 and (sh1.flag_counts() == s.flag_counts()[3:]).all()
