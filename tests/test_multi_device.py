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


def test_resume_takes_the_shard_count_from_the_files(fcm, oracle, tmp_path):
    """a state saved on 2 handles resumes on 1 device and on 3 -- the number of shards, their order and
    the run's chain count come from the files (fcm_sampler_save_state_shard / fcm_state_file_info), not from the device
    list given at resume time -- and ends where the unsharded run ends.  A missing shard, shards of two different saves
    and shards that do not tile the run's chains are refused loudly."""
    n = 250
    e = fcm.graphs.random_with_p(n, 0.1, seed=5)
    gg, go, bg, bo = setup_pair(fcm, oracle, n, e)
    straight = fcm.MCMCSampler(gg, bg, n_chains=7, seed=4)
    straight.step(1500)
    two = fcm.MultiDeviceSampler(gg, bg, 7, devices=[0, 0], seed=4)
    two.step(600)
    f = str(tmp_path / "run.state")
    two.save_state(f, sample_number=3)
    assert os.path.exists(f + ".shard0") and os.path.exists(f + ".shard1") and not os.path.exists(f) and not os.path.exists(f + ".shard0.new")
    info = fcm.MultiDeviceSampler.state_file_info(f + ".shard1")
    assert (info["shard_index"], info["shard_count"], info["total_chains"], info["first_chain_id"], info["n_chains"], info["sample_number"]) == (1, 2, 7, 4, 3, 3)
    for devices in ([0], [0, 0, 0], [0, 0]):
        num, m = fcm.MultiDeviceSampler.load_state(f, devices)
        assert num == 3 and m.n_chains == 7 and m.ranges == [(0, 4), (4, 7)] and len(m.shards) == 2
        m.step(900)
        assert (m.flag_counts() == straight.flag_counts()).all()
        for c in (0, 3, 4, 6):
            assert (m.edges(c) == straight.edges(c)).all() and (m.double_slots(c) == straight.double_slots(c)).all()
    # a state saved on 4 handles and resumed with two devices: all four shards, two per device
    four = fcm.MultiDeviceSampler(gg, bg, 7, devices=[0, 0, 0, 0], seed=4)
    four.step(600)
    f4 = str(tmp_path / "run4.state")
    four.save_state(f4, sample_number=9)
    num, m = fcm.MultiDeviceSampler.load_state(f4, [0, 0])
    assert num == 9 and len(m.shards) == 4 and m.n_chains == 7
    m.step(900)
    assert (m.flag_counts() == straight.flag_counts()).all()
    # a missing shard
    os.rename(f4 + ".shard2", f4 + ".gone")
    with pytest.raises(fcm.FcmError) as ei:
        fcm.MultiDeviceSampler.load_state(f4, [0, 0])
    assert "shard 2 of 4 is missing" in str(ei.value)
    os.rename(f4 + ".gone", f4 + ".shard2")
    # shards of two different saves (same layout, another sample number)
    four.step(10)
    four.save_state(str(tmp_path / "later.state"), sample_number=10)
    os.replace(str(tmp_path / "later.state.shard3"), f4 + ".shard3")
    with pytest.raises(fcm.FcmError) as ei:
        fcm.MultiDeviceSampler.load_state(f4, [0])
    assert "not a shard of the same save" in str(ei.value)
    # a single-handle file resumes through the same door, on any device list
    straight2 = fcm.MCMCSampler(gg, bg, n_chains=7, seed=4)
    straight2.step(600)
    f1 = str(tmp_path / "one.state")
    straight2.save_state(f1, 2)
    num, m = fcm.MultiDeviceSampler.load_state(f1, [0, 0, 0])
    assert num == 2 and len(m.shards) == 1 and m.n_chains == 7
    m.step(900)
    assert (m.flag_counts() == straight.flag_counts()).all()


def test_sample_cli_resumes_two_shards_on_one_device(fcm, golden_dir, tmp_path):
    """The same through the `sample` binary: saved with --devices 0,0, continued with --device 0 and with --devices 0,0,0."""
    exe = os.path.join(os.path.dirname(fcm.LIB_PATH), "sample")
    flag = os.path.join(golden_dir, "bug_calc_relax_de.flag")
    base = [exe, "-l", "lab", "-s", "2", "--samples-store-dir", str(tmp_path / "samples"), "--state-store-dir", str(tmp_path / "state")]
    r = subprocess.run(base + ["-i", flag, "--simple", "--chains", "5", "--sample-distance", "150", "-n", "2", "--devices", "0,0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    st = str(tmp_path / "state" / "sampler-lab-002.state")
    s = fcm.initialize_new_sampler(flag, target_relaxation=0.01, seed=2, n_chains=5, sample_distance=150)
    for _ in range(4):
        s.next()
    import shutil
    for tag, devargs in (("one", ["--device", "0"]), ("three", ["--devices", "0,0,0"])):
        d = tmp_path / ("state_" + tag)
        shutil.copytree(tmp_path / "state", d)
        r2 = subprocess.run([exe, "-c", str(d / "sampler-lab-002.state"), "-l", "lab", "-s", "2", "-n", "2", "--samples-store-dir", str(tmp_path / ("s_" + tag)),
                             "--state-store-dir", str(d)] + devargs, capture_output=True, text=True, timeout=300)
        assert r2.returncode == 0, r2.stdout + r2.stderr
        assert "5 chains on 2 handles" in r2.stdout and "flag count: %s" % s.flag_count(0) in r2.stdout, r2.stdout
    assert os.path.exists(st + ".shard0")
