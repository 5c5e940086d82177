#!/usr/bin/env python3
"""bench.py -- headline benchmark: edge-flip proposals/sec on the BASELINE
configs[2] workload (Erdos-Renyi digraph n=1000, p=0.10, 4096 independent chains
per GPU, `--simple` move mix, target_relaxation 0.01).

  python bench.py --gpus N --steps K --warmup W            (any N: launches its own ranks)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (the driver's form)

A "step" is one launch of the persistent step kernel: every chain runs
--proposals iterations of the reference loop src/lib.rs:182-192.  Inputs are
resident in HBM before the timed region.  Prints ONE JSON line on rank 0.

--config 1..4 selects the other single-GPU shares of BASELINE.json's configs
(parity-test cases; bench lines for DESIGN.md's table, not the headline).
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "edge-flip proposals/sec/GPU (n=1k graph); bit-exact simplex counts"  # BASELINE.json
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

# BASELINE.json configs[i] -> (description, per-GPU chains, default proposals per chain per step)
CONFIGS = {
    1: ("configs[1]: C. elegans stand-in (tests/golden/bug_calc_relax_de.flag, n=279), 1024 chains", 1024, 65536),
    2: ("configs[2]: Erdos-Renyi digraph n=1000 p=0.10 seed 0, 4096 chains per GPU", 4096, 65536),
    3: ("configs[3]: Erdos-Renyi digraph n=4000 p=0.05 seed 0, 1024 chains per GPU (8192 over 8)", 1024, 16384),
    4: ("configs[4]: n=30000, 1M directed edge draws seed 0, 256 chains per GPU (2048 over 8)", 256, 32768),
}


def algorithmic_bytes(d, n, clique_pairs=None):
    """W = 8*ceil(n/64) bytes per row; non-empty flip reads
    (k+4) rows + 64 B of word updates, double-edge move (k1+k2+8) rows + 128 B,
    an empty proposal 16 B.  Clique moves, as the survey's formula stands: every changed directed edge one (k+2)-row
    evaluation.  With clique_pairs = (changed vertex pairs, shared rows) -- the counters of the clique-move kernels,
    FCM_STAT_PAIRS / FCM_STAT_SHARED_ROWS -- rows that are read once are charged once: a pair whose two directions both
    change is ONE local build (the mean k over pairs taken as the mean over changed directions: k belongs to the pair),
    and the rows of a permuted clique's own vertices, which every pair of that move has in its local set, count once per
    move."""
    W = 8 * ((n + 63) // 64)
    nchg = int(d.get("n_changes", 0))
    rows_simple = int(d["sum_k"]) + 4 * int(d["n_flip"]) + 8 * int(d["n_dmove"]) + 2 * nchg   # (sum_k includes k per changed direction)
    if clique_pairs is not None and nchg > 0:
        pairs, shared = clique_pairs
        # sum_k = (simple moves' k) + (clique moves' k per changed direction); the stats do not split it, the mean k does
        k_simple = float(d["sum_k"]) / max(1.0, float(int(d["n_flip"]) + 2 * int(d["n_dmove"]) + nchg))
        rows_clique_dir = (k_simple + 2.0) * nchg
        rows_clique_pair = (k_simple + 2.0) * pairs - shared
        rows_simple = rows_simple - rows_clique_dir + max(0.0, rows_clique_pair)
    return rows_simple * W + 64 * int(d["n_flip"]) + 128 * int(d["n_dmove"]) + 64 * nchg + 16 * int(d["n_empty"])


def needed_bytes(d, n, mean_k, clique_pairs=None):
    """Bytes the kernels actually have to move when rows are longer than one 128-B line (n > 1024): a local
    build reads, of each of its k+2 rows, only the lines that hold the k+2 bit positions it tests.
    Expected distinct lines per row for k+2 uniform positions among nl lines: nl * (1 - (1 - 1/nl)^(k+2)).
    Plus the static side: 16 B of table entry and 4k B of vertex list per evaluated edge.  Equals the SURVEY
    figure (whole rows) up to the static terms once k+2 is several times nl."""
    nl = max(1, (8 * ((n + 63) // 64) + 127) // 128)
    s = mean_k + 2.0
    lines = nl * (1.0 - (1.0 - 1.0 / nl) ** s)
    evals = int(d["n_flip"]) + 2 * int(d["n_dmove"]) + int(d.get("n_changes", 0))
    shared = 0.0
    if clique_pairs is not None and int(d.get("n_changes", 0)) > 0:   # a clique move: one build per changed PAIR, the clique's own rows once per move
        pairs, shared = clique_pairs
        evals = int(d["n_flip"]) + 2 * int(d["n_dmove"]) + pairs
    return evals * (s * lines * 128.0 + 16.0 + 4.0 * mean_k + 64.0) - shared * lines * 128.0 + 16.0 * int(d["n_empty"])


def sparse_bytes(d, mean_k):
    """Bytes a proposal has to move on the sparse per-chain state (two bits per adjacent pair): per evaluated
    pair its 16-B table entry, 4k B of vertex list, its (k+2)(k+1)/2 local pair entries of 8 B (none for k = 0) and as many
    dword gathers from the chain's record, plus the commit's read-modify-write; an empty proposal 16 B."""
    t = (mean_k + 2.0) * (mean_k + 1.0) / 2.0
    evals = int(d["n_flip"]) + 2 * int(d["n_dmove"])
    return evals * (16.0 + 4.0 * mean_k + (8.0 * t if mean_k > 0 else 0.0) + 4.0 * t + 8.0) + 16.0 * int(d["n_empty"])


def sparse_sector_bytes(d, mean_k):
    """The same accesses at the granularity the memory system moves them: 64-B sectors on the read side, 32-B on the write
    side (tools/pmc_calib.hip measures 32 B per scattered dword store).  A flip: its table entry, its vertex list, the
    record sector of its pair (+ the sectors of its local pair entries and of their record gathers when k > 0); a double-edge
    move: the slot word, the slot pair's and the candidate's table entries, and list + record sector for both pairs; an
    empty proposal its table entry; a non-empty one two 32-B writes.  What the FETCH_SIZE / WRITE_SIZE counters should see."""
    t = (mean_k + 2.0) * (mean_k + 1.0) / 2.0
    extra = (math.ceil(8.0 * t / 64.0) * 64.0 + 64.0 * min(t, 4.0)) * min(1.0, mean_k)   # local pair entries + their gathers, in the share of pairs with k > 0
    flips, dmoves = int(d["n_flip"]), int(d["n_dmove"])
    return flips * (3 * 64.0 + extra + 64.0) + dmoves * (7 * 64.0 + 2 * extra + 64.0) + 64.0 * int(d["n_empty"])


def cpu_baseline(n, edges, bounds_lists, seed, target_seconds=15.0):
    """Times the CPU oracle (reference-faithful port: neighbourhood lookup,
    induced-subgraph recount before/after, revert from saved vectors) on the
    host cores, one chain per thread, on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_ffi as oracle  # checker / baseline only, never the product path
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, int(os.environ.get("FCM_BENCH_CPU_THREADS", "16"))))  # a 1-GPU box's CPU share is 16
    go = oracle.Graph.from_edges(n, edges)
    bo = oracle.Bounds.from_lists(*bounds_lists)
    probe = oracle.Chain(go, bo, seed=seed, chain_id=0)
    t0 = time.perf_counter()
    probe.step(2000)
    per_prop = (time.perf_counter() - t0) / 2000
    nprop = max(1000, int(target_seconds / per_prop))
    log("cpu baseline: %.1f us/proposal single thread; %d threads x %d proposals" % (per_prop * 1e6, cores, nprop))
    chains = [oracle.Chain(go, bo, seed=seed, chain_id=c) for c in range(cores)]
    secs = oracle.chains_step_mt(chains, nprop, cores)
    total = nprop * cores
    return {"value": total / secs, "unit": "proposals/s", "cores": cores, "kind": "port",
            "sample": "%d chains x %d proposals of the same n=%d graph, seeds as GPU chains 0..%d, %.1f s"
                      % (cores, nprop, n, cores - 1, secs),
            "single_thread_proposals_per_s": 1.0 / per_prop}


def lib_sha16(path):
    """What ties a profile to the build it was taken on: the first 16 hex digits of the SHA-256 over the library's SOURCES
    (csrc/*.hpp, *.hip, *.cpp, the Makefile, include/fcm.h, in name order) -- the same after a rebuild, different after any
    kernel or host edit.  A library loaded from somewhere else (FCM_LIB_PATH: a diagnostic build) gets its file's own hash
    behind a "variant:" tag, which no committed profile carries."""
    import glob
    import hashlib
    h = hashlib.sha256()
    if os.environ.get("FCM_LIB_PATH"):
        with open(path, "rb") as f:
            for blk in iter(lambda: f.read(1 << 20), b""):
                h.update(blk)
        return "variant:" + h.hexdigest()[:16]
    src = os.path.join(ROOT, "flag_complex_mcmc_amd", "csrc")
    files = sorted(glob.glob(os.path.join(src, "*.hpp")) + glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.cpp"))
                   + [os.path.join(src, "Makefile"), os.path.join(ROOT, "include", "fcm.h")])
    for fn in files:
        h.update(os.path.basename(fn).encode() + b"\0")
        with open(fn, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def load_traffic(config, n_chains, proposals, moves, this_lib, sparse=None):
    """(HBM bytes per launch, source) from the committed rocprofv3 PMC summary of this same command
    (profiles/pmc_summary.json, written by tools/pmc_summary.py from separate --pmc FETCH_SIZE / WRITE_SIZE
    passes), or (None, why).  A replay of a profiled run, not a measurement of this one: the source says so.  The
    counters scale with the proposals per launch (every launch runs the same loop), so a record taken at another launch
    length is scaled; a record taken on ANOTHER BUILD of the library is not used at all."""
    path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        recs = json.load(open(path))
    except Exception:
        return None, None
    stale = None
    for rec in (recs if isinstance(recs, list) else [recs]):
        if rec.get("config", 2) == config and rec.get("n_chains") == n_chains and rec.get("moves", "simple") == moves:
            if sparse is not None and bool(rec.get("sparse_state", False)) != bool(sparse):
                continue            # (the record is of the other state layout: FCM_SPARSE)
            if rec.get("lib_sha16") != this_lib:   # (a later record of the same command may be of this build: keep looking)
                stale = stale or "STALE: profiled build %s != this build %s (profiles/pmc_summary.json, %s)" % (rec.get("lib_sha16"), this_lib, rec.get("tag", "?"))
                continue
            scale = float(proposals) / float(rec.get("proposals") or proposals)
            return rec.get("hbm_bytes_per_launch") * scale, ("fabric-side fetch + write-back (Infinity Cache + HBM) by FETCH_SIZE / WRITE_SIZE, replayed from profiles/pmc_summary.json (%s, build %s%s): rocprofv3 --pmc passes of this command, not this run"
                                                             % (rec.get("tag", "?"), this_lib, "" if scale == 1.0 else ", scaled x%g from %d proposals per launch" % (scale, rec.get("proposals"))))
    return None, stale


def log(msg):
    print("[bench %7.1fs] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


T_START = time.perf_counter()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks with torch.distributed.run as a CHILD
    process -- before this process has made any GPU call, and never an exec -- and leave with its code."""
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "4")
    log("no launcher in the environment: starting %d ranks: %s" % (args.gpus, " ".join(cmd[1:8]) + " ..."))
    return subprocess.call(cmd, env=env)


def build_workload(fcm, config, n_arg, p_arg, seed):
    if config == 1:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import load_flag_fixture
        n, edges = load_flag_fixture("bug_calc_relax_de.flag")
    elif config == 2:
        n = n_arg
        edges = fcm.graphs.random_with_p(n, p_arg, seed)
    elif config == 3:
        n = 4000
        edges = fcm.graphs.random_with_p(n, 0.05, seed)
    else:
        n = 30000
        edges = fcm.graphs.random_edge_draws(n, 1000000, seed)
    return n, edges


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=2, help="BASELINE.json configs[i]; 2 is the headline")
    ap.add_argument("--chains", type=int, default=0, help="chains per GPU (0 = the config's stated per-GPU count)")
    ap.add_argument("--proposals", type=int, default=0, help="proposals per chain per step (0 = the config's default; one launch)")
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--p", type=float, default=0.10)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--moves", choices=["simple", "default"], default="simple",
                    help="simple = [0.5,0.5,0,0] (the headline path); default = the reference's [0.1,0.1,0.6,0.2] with clique moves")
    ap.add_argument("--weights", default="", help="explicit move weights a,b,c,d (profiling aid: isolates one kind of move); overrides --moves")
    args = ap.parse_args()
    if args.chains <= 0:
        args.chains = CONFIGS[args.config][1]
    if args.proposals <= 0:
        args.proposals = CONFIGS[args.config][2] if (args.moves == "simple" and not args.weights) else 2048

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if os.environ.get("FCM_BENCH_LAUNCH_ONLY", "0") == "1":   # launcher test (tests/test_bench_launcher.py): no GPU needed
        sys.stdout.write(json.dumps({"launch_only": True, "rank": rank, "local_rank": local_rank, "world": world}) + "\n")
        sys.stdout.flush()
        return

    import numpy as np
    import torch
    import torch.distributed as dist
    import flag_complex_mcmc_amd as fcm
    from flag_complex_mcmc_amd import distributed as fdist

    if not torch.cuda.is_available() or fcm.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: libfcm has no CPU path")
    # Rehearsal on a 1-GPU box: FCM_BENCH_REHEARSE=1 lets all ranks share GPU 0 and talk over gloo
    # (RCCL refuses two ranks on one device).  The driver's multi-GPU run uses one GPU per rank and RCCL.
    rehearse = os.environ.get("FCM_BENCH_REHEARSE", "0") == "1"
    if rehearse:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d: LOCAL_RANK %d but only %d GPU(s) visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if rehearse else "nccl"
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    # ---- workload: synthetic digraph, same on every rank -----------------------
    n, edges = build_workload(fcm, args.config, args.n, args.p, args.seed)
    g = fcm.Graph.from_edges(n, edges)
    log("graph built: n=%d, %d edges" % (n, len(edges)))
    flag_count = g.flagser_count(local_rank)
    log("initial count %s" % flag_count)
    bounds = fcm.Bounds.calculate(g, flag_count, fcm.Bounds.target(flag_count, 0.01), device=local_rank)
    total_chains = args.chains * world                      # weak scaling: fixed chains per GPU
    lo, hi = fdist.shard_range(total_chains, rank, world)
    weights = fcm.MOVE_DISTRIBUTION_SIMPLE if args.moves == "simple" else fcm.MOVE_DISTRIBUTION
    if args.weights:
        weights = tuple(float(x) for x in args.weights.split(","))
        args.moves = "custom"
    s = fcm.MCMCSampler(g, bounds, n_chains=hi - lo, seed=args.seed, move_weights=weights,
                        device=local_rank, first_chain_id=lo)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    s.set_stream(stream.cuda_stream)   # kernels and the timing events share this one HIP stream
    log("sampler ready: %s" % s.info)

    def barrier():
        if world > 1:
            dist.barrier()

    def gather():
        """The path's one exchange: every rank's per-chain count vectors and counters to every rank."""
        st = s.stats()
        mat = np.stack([st[k] for k in fcm._ffi.STAT_NAMES], axis=1)
        return st, fdist.gather_counts(s.flag_counts(), mat, device=None if rehearse else dev)

    for _ in range(args.warmup):
        s.step(args.proposals, sync=False)
    torch.cuda.synchronize()
    if world > 1:
        gather()     # warm the collective too (communicator set-up is not part of a step)
    log("warmup done")
    st0 = s.stats()

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        s.step(args.proposals, sync=False)
    ev1.record()
    torch.cuda.synchronize()
    st1, (all_counts, all_stats) = gather()   # inside the timed region: the histogram gather is part of the job
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / max(1, args.steps)  # avg launch duration, HIP events on the launch stream
    log("timed region done: %.3f s, %.2f ms per launch" % (elapsed, kernel_ms))

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- parity gate on this rank's result (outside the timed region) ----------
    keys = ("sampled", "accepted", "n_empty", "n_flip", "n_dmove", "sum_k", "n_cperm", "n_cswap", "n_changes", "n_redo", "n_wide", "n_big", "n_recheck", "n_held", "n_pairs", "n_shared_rows")
    probe = bool(os.environ.get("FCM_BENCH_PROBE"))   # instruction-cost probe builds (wrong results by design): no parity gate, and the line says so
    d = {k: (st1[k].astype(np.int64) - st0[k].astype(np.int64)).sum() for k in keys}
    assert d["sampled"] == (hi - lo) * args.steps * args.proposals, "proposal count mismatch"
    assert probe or (st1["status"] == 0).all(), "device-side consistency check failed"

    def strip(v):  # flag_count never shrinks in length (src/lib.rs:72-74): ignore trailing zeros
        v = list(v)
        while v and v[-1] == 0:
            v.pop()
        return v

    for c in (() if probe else (0, (hi - lo) // 2, hi - lo - 1)):
        assert s.graph(c).flagser_count(local_rank) == strip(s.flag_count(c)), "incremental counts != full recount (chain %d)" % c
        assert bounds.check(s.flag_count(c)), "chain %d left the bounds" % c
    # the gathered matrix holds every rank's chains, in global chain order
    assert all_counts.shape[0] == total_chains and (all_stats[:, 0] == all_stats[0, 0]).all()
    assert (all_counts[lo:hi] == s.flag_counts()).all()

    if rank == 0:
        total_prop = total_chains * args.steps * args.proposals
        mean_k = float(d["sum_k"]) / max(1.0, float(d["n_flip"] + 2 * d["n_dmove"] + d["n_changes"]))
        survey_bytes = algorithmic_bytes(d, n) / args.steps             # per launch, this rank
        long_rows = n > 1024
        sparse = bool(s.info.get("sparse_state", 0))
        clique = weights[2] > 0 or weights[3] > 0
        # clique moves: rows that are read once are charged once (FCM_STAT_PAIRS, FCM_STAT_SHARED_ROWS)
        clique_bytes = algorithmic_bytes(d, n, (float(d["n_pairs"]), float(d["n_shared_rows"]))) / args.steps if clique else None
        cpairs = (float(d["n_pairs"]), float(d["n_shared_rows"])) if clique else None
        abytes = sparse_bytes(d, mean_k) / args.steps if sparse else (needed_bytes(d, n, mean_k, cpairs) / args.steps if long_rows else (clique_bytes if clique else survey_bytes))
        achieved = abytes / (kernel_ms * 1e-3) / 1e9
        this_lib = lib_sha16(fcm.LIB_PATH)
        traffic, traffic_source = load_traffic(args.config, args.chains, args.proposals, args.moves, this_lib, sparse)
        out = {
            "metric": METRIC, "value": total_prop / elapsed, "unit": "proposals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32/u64 bitset", "data": "synthetic",
            "config": {"workload": "BASELINE %s, %d chains per GPU, %d proposals per chain per step, moves %s, target_relaxation 0.01"
                                   % (CONFIGS[args.config][0], args.chains, args.proposals, list(weights)),
                       "baseline_config": args.config, "n": n, "chains_per_gpu": args.chains, "proposals_per_step": args.proposals,
                       "edges": int(len(edges)), "initial_flag_count": flag_count, "parallelism": "chains sharded, %d rank(s)" % world},
            "per_gpu_value": total_prop / elapsed / world,
            "lib_sha16": this_lib,
            "kernel_ms_per_launch": kernel_ms,
            "accept_ratio": float(d["accepted"]) / float(d["sampled"]),
            "empty_fraction": float(d["n_empty"]) / float(d["sampled"]),
            "mean_k": mean_k,
            "clique_move_fraction": float(d["n_cperm"] + d["n_cswap"]) / float(d["sampled"]),
            "changed_edges_per_clique_move": float(d["n_changes"]) / max(1.0, float(d["n_cperm"] + d["n_cswap"])),
            "rare_paths_per_1e6": {k: 1e6 * float(d[k]) / float(d["sampled"]) for k in (("n_redo", "n_wide", "n_big") + (() if clique else ("n_recheck", "n_held")))},
            **({"changed_pairs_per_clique_move": float(d["n_pairs"]) / max(1.0, float(d["n_cperm"] + d["n_cswap"]))} if clique else {}),
            "count_histogram_dim2_distinct": int(len(fdist.count_histogram(all_counts, 2)[0])),
            "parity": ("NOT CHECKED: FCM_BENCH_PROBE run (probe build, results wrong by design)" if probe
                       else "counts == full GPU recount on 3 chains; oracle parity in tests/ -m gpu"),
            **({"probe": True} if probe else {}),
            "gathered_chains": int(all_counts.shape[0]),
            "collective": {"backend": backend, "world_size_seen": dist.get_world_size() if world > 1 else 1,
                           "what": "all_gather of per-chain count vectors + counters, inside the timed region"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": abytes,
                         "algorithmic_model": ("sparse state: table entry + list + local pair entries + record gathers (DESIGN.md 2; the record lives in L2 / MALL)" if sparse
                                               else ("lines touched (rows longer than one 128-B line; DESIGN.md 4.3)" if long_rows
                                                     else ("SURVEY.md 8(d) whole rows, a clique move's rows charged once: one build per changed pair, the clique's own rows once per move (DESIGN.md 4.3)" if clique
                                                           else "SURVEY.md 8(d): whole rows"))),
                         "sparse_state": sparse,
                         "survey_bytes_per_launch": survey_bytes,
                         **({"sector_model_bytes_per_launch": sparse_sector_bytes(d, mean_k) / args.steps,
                             "traffic_over_sector_model": (traffic / (sparse_sector_bytes(d, mean_k) / args.steps)) if traffic else None} if sparse else {}),
                         "kernel": ("fcm_step_cq_kernel" if s.info["cooperative_clique_kernel"] else "fcm_step_kernel") if clique
                                   else ("fcm_step_mw_kernel" if s.info["waves_per_chain"] >= 2 else "fcm_step_kernel"),
                         "waves_per_chain": int(s.info["waves_per_chain"])},
        }
        # the reference-faithful CPU port beside it, in the same run, on rank 0's host cores, whatever the number of ranks
        # (outside the timed region; north_star: "next to the reference Rust CPU path timed on the box's host cores")
        if not args.no_cpu_baseline and args.config == 2 and args.moves == "simple":
            out["cpu_baseline"] = cpu_baseline(n, edges, (bounds.flag_count_min, bounds.flag_count_max), args.seed,
                                               target_seconds=float(os.environ.get("FCM_BENCH_CPU_SECONDS", "15")))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
