"""CPU, world_size 2, gloo: the N>1 path of bench.py -- chain sharding and the
report-time gather of per-chain count histograms."""
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_chains():
    from flag_complex_mcmc_amd.distributed import shard_range
    for total, world in [(8192, 8), (2048, 8), (10, 4), (3, 8), (4096, 1)]:
        got = []
        for r in range(world):
            lo, hi = shard_range(total, r, world)
            assert 0 <= lo <= hi <= total
            got += list(range(lo, hi))
        assert got == list(range(total))
    assert shard_range(8192, 3, 8) == (3072, 4096)


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, %r)
    from flag_complex_mcmc_amd.distributed import shard_range, gather_counts, count_histogram
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    total, nc, ns = %d, 6, 8
    lo, hi = shard_range(total, rank, world)
    # chain c holds counts c*10+d (with a value above 2^63 to check the u64 bit pattern survives)
    counts = np.array([[c * 10 + d for d in range(nc)] for c in range(lo, hi)], np.uint64).reshape(hi - lo, nc)
    counts[:, 2] += np.uint64(2**63)
    stats = np.array([[c * 100 + j for j in range(ns)] for c in range(lo, hi)], np.uint64).reshape(hi - lo, ns)
    ac, ast = gather_counts(counts, stats)
    assert ac.shape == (total, nc) and ast.shape == (total, ns), (ac.shape, ast.shape)
    want = np.array([[c * 10 + d for d in range(nc)] for c in range(total)], np.uint64)
    want[:, 2] += np.uint64(2**63)
    assert (ac == want).all()
    assert (ast[:, 0] == np.arange(total) * 100).all()
    vals, mult = count_histogram(ac, 3)
    assert len(vals) == total and mult.sum() == total
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(%r, "rank%%d.ok" %% rank), "w").write("ok")   # stdout of the two ranks may interleave
""")


def _run_gather(tmp_path, total, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, total, str(tmp_path)))
    import socket
    with socket.socket() as sk:          # a free port, so reruns never collide with a lingering socket
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert all((tmp_path / ("rank%d.ok" % k)).exists() for k in range(world)), r.stdout + r.stderr


def test_gather_counts_world2_gloo(tmp_path):
    _run_gather(tmp_path, 12, 2)


def test_gather_counts_unequal_shards_world4_gloo(tmp_path):
    """total = 10 over 4 ranks: shard_range gives blocks of 3, 3, 3, 1; 3 over 4: the last rank holds none."""
    _run_gather(tmp_path, 10, 4)
    sub = tmp_path / "b"
    sub.mkdir()
    _run_gather(sub, 3, 4)


def test_gather_is_identity_without_process_group():
    from flag_complex_mcmc_amd.distributed import gather_counts
    c = np.arange(12, dtype=np.uint64).reshape(3, 4)
    s = np.arange(24, dtype=np.uint64).reshape(3, 8)
    ac, ast = gather_counts(c, s)
    assert (ac == c).all() and (ast == s).all()
