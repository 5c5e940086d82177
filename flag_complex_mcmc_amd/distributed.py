"""Multi-GPU plumbing: chains are independent, so they shard over ranks with no
data-path collective.  The only exchange is the gather of the
per-chain simplex-count histograms at report time, one all_gather over
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests)."""
import numpy as np


def shard_range(total_chains, rank, world_size):
    """Contiguous block of global chain ids owned by `rank` (chain c -> rank
    c // ceil(C/G)); chain ids, and therefore trajectories, do not depend on
    the number of ranks."""
    per = (total_chains + world_size - 1) // world_size
    lo = min(rank * per, total_chains)
    hi = min(lo + per, total_chains)
    return lo, hi


def gather_counts(counts, stats, device=None):
    """All-gather per-chain flag_count vectors [c_local, nc] and counters
    [c_local, ns] (uint64) from every rank; returns numpy arrays ordered by
    global chain id.  Ranks may hold different numbers of chains (shard_range's
    blocks are unequal when the total does not divide, and the last ranks may hold
    none): every rank's block is padded to the largest before the one all_gather
    and cut back to its own length afterwards."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return np.asarray(counts), np.asarray(stats)
    world = dist.get_world_size()
    # uint64 -> int64 bit pattern: collectives do not care, and RCCL has no u64 all_gather dtype in torch
    local = np.concatenate([np.asarray(counts, np.uint64), np.asarray(stats, np.uint64)], axis=1).view(np.int64)
    dev = device if device is not None else "cpu"
    rows = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    all_rows = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_rows, rows)
    all_rows = [int(x) for x in all_rows.cpu().tolist()]
    most = max(all_rows)
    t = torch.zeros((most, local.shape[1]), dtype=torch.int64, device=dev)
    if local.shape[0]:
        t[: local.shape[0]] = torch.from_numpy(np.ascontiguousarray(local)).to(dev)
    out = torch.empty((world * most, local.shape[1]), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(out, t)
    arr = out.cpu().numpy().view(np.uint64)
    if any(r != most for r in all_rows):
        arr = np.concatenate([arr[r * most: r * most + all_rows[r]] for r in range(world)], axis=0)
    nc = np.asarray(counts).shape[1]
    return arr[:, :nc].copy(), arr[:, nc:].copy()


def count_histogram(all_counts, dim):
    """Histogram of the dimension-`dim` simplex count over all chains:
    (values, multiplicities)."""
    return np.unique(np.asarray(all_counts)[:, dim], return_counts=True)
