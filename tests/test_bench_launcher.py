"""CPU: `python bench.py --gpus N` starts its own ranks.  With FCM_BENCH_LAUNCH_ONLY=1 every
rank reports its coordinates and leaves before anything touches a GPU, so this runs without one."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launches_its_ranks():
    env = dict(os.environ, FCM_BENCH_LAUNCH_ONLY="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    import re
    recs = [json.loads(m) for m in re.findall(r"\{[^{}]*\}", r.stdout)]   # (two ranks may share a line of the pipe)
    assert sorted((x["rank"], x["world"]) for x in recs) == [(0, 2), (1, 2)], r.stdout
    assert "starting 2 ranks" in r.stderr


def test_bench_under_a_launcher_does_not_relaunch():
    env = dict(os.environ, FCM_BENCH_LAUNCH_ONLY="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True,
                       timeout=120, env=env, cwd=ROOT)
    assert r.returncode == 0 and json.loads(r.stdout.strip())["world"] == 1 and "starting" not in r.stderr
    # a launcher that disagrees with --gpus is an error, not a silent relaunch
    env["WORLD_SIZE"] = "2"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                       timeout=120, env=env, cwd=ROOT)
    assert r.returncode != 0


import pytest


@pytest.mark.gpu
def test_two_rank_rehearsal_line_carries_the_cpu_baseline():
    """north_star asks for the CPU figure "in the same run" at every N.  Two ranks sharing GPU 0 over gloo
    (FCM_BENCH_REHEARSE=1; RCCL refuses two ranks on one device): rank 0's line holds `cpu_baseline` next to n_gpus = 2."""
    env = dict(os.environ, FCM_BENCH_REHEARSE="1", FCM_BENCH_CPU_SECONDS="1", FCM_BENCH_CPU_THREADS="4")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--chains", "256", "--proposals", "1024"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.split("\n") if ln.startswith("{") and '"metric"' in ln]
    assert len(line) == 1, r.stdout
    rec = json.loads(line[0])
    assert rec["n_gpus"] == 2 and rec["gathered_chains"] == 512 and rec["scaling"] == "weak"
    cb = rec["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "proposals/s"
    assert rec["roofline"]["kernel"] == "fcm_step_mw_kernel" and "probe" not in rec
