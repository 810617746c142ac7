"""Multi-GPU sharding of the hot path.

(gas, band) pairs of find_g_points are independent partition problems
(find_g_points.cpp:655, :1152), so the work is dealt round-robin to ranks, one process per
GPU; nothing is exchanged on the data path.  The only collective is ONE all-reduce at the end
that combines the ranks' scalars (max of elapsed time, sum of wavenumber passes, sum of the
final per-g-point errors = "the final cost"), RCCL over xGMI on the GPU box, gloo in the CPU
tests.
"""
import numpy as np


def shard_tasks(ntasks, rank, world_size):
    """Indices of the (gas, band) tasks owned by `rank`: round-robin, deterministic."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return list(range(rank, ntasks, world_size))


def task_table(gases, nband):
    """[(gas, band), ...] in the order the reference loops (gas outer, band inner)."""
    return [(g, b) for g in gases for b in range(nband)]


def reduce_scalars(elapsed_s, passes, cost, device=None, group=None):
    """One all-reduce of [elapsed, passes, cost] -> (max elapsed, total passes, total cost).

    MAX and SUM are folded into a single SUM all-reduce of a (world, 3) one-hot matrix so that
    exactly one collective is issued (latency-bound on xGMI: a few KB at most)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(elapsed_s), float(passes), float(cost)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    buf = torch.zeros((world, 3), dtype=torch.float64, device=device)
    buf[rank, 0], buf[rank, 1], buf[rank, 2] = float(elapsed_s), float(passes), float(cost)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    b = buf.cpu().numpy()
    # fixed rank order: every rank computes bit-identical totals
    return float(b[:, 0].max()), float(np.add.reduce(b[:, 1])), float(np.add.reduce(b[:, 2]))
