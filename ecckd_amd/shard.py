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


def deal_tasks(ntasks, rank, world_size):
    """Indices of the (gas, band) tasks owned by `rank` when the task table (gas outer, band inner, the order of the
    reference's loops) is cut into `world_size` CONTIGUOUS shares whose sizes differ by at most one: a process then touches
    as few gases as possible - it reads a gas's spectra only if it searches one of its bands - and the bands of a gas that
    it does own are searched side by side in one launch train.  An interval's error does not depend on what else is in
    its batch (ecckd_calc_error_multi), so every band ends at the same g points however the table is cut."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, extra = divmod(ntasks, world_size)
    begin = rank * base + min(rank, extra)
    return list(range(begin, begin + base + (1 if rank < extra else 0)))


def world(group=None):
    """(rank, world_size) of the torch.distributed job, (0, 1) when there is none."""
    try:
        import torch.distributed as dist
    except ImportError:
        return 0, 1
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def gather_to_root(obj, group=None, dst=0):
    """The ranks' picklable results on rank `dst`, in rank order (None elsewhere); [obj] without a job.  The only data
    the (gas, band) searches hand back: a few numbers per g point."""
    rank, ws = world(group)
    if ws == 1:
        return [obj]
    import torch.distributed as dist
    out = [None] * ws if rank == dst else None
    dist.gather_object(obj, out, dst=dst, group=group)
    return out


def task_table(gases, nband):
    """[(gas, band), ...] in the order the reference loops (gas outer, band inner)."""
    return [(g, b) for g in gases for b in range(nband)]


def reduce_scalars(elapsed_s, passes, cost, device=None, group=None, count=False):
    """One all-reduce of [elapsed, passes, cost] -> (max elapsed, total passes, total cost).

    MAX and SUM are folded into a single SUM all-reduce of a (world, 4) one-hot matrix so that
    exactly one collective is issued (latency-bound on xGMI: a few KB at most).  The fourth column
    carries a one per rank: with `count=True` the number of ranks the collective actually saw is
    returned as a fourth value (bench.py prints it as `ranks_seen`)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        out = (float(elapsed_s), float(passes), float(cost))
        return out + (1,) if count else out
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    buf = torch.zeros((world, 4), dtype=torch.float64, device=device)
    buf[rank, 0], buf[rank, 1], buf[rank, 2], buf[rank, 3] = float(elapsed_s), float(passes), float(cost), 1.0
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    b = buf.cpu().numpy()
    # fixed rank order: every rank computes bit-identical totals
    out = (float(b[:, 0].max()), float(np.add.reduce(b[:, 1])), float(np.add.reduce(b[:, 2])))
    return out + (int(round(float(np.add.reduce(b[:, 3])))),) if count else out


# ---- reorder_spectrum of ONE band on several GPUs: the key sweep split by wavenumber range (reorder row) ----------

def wavenumber_range(nwav, rank, world_size, align=256):
    """Contiguous share [begin, end) of the nwav wavenumbers owned by `rank`: shares of whole `align`-point tiles (the key
    kernel's block) whose sizes differ by at most one tile; the last rank takes the ragged end."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    ntile = (nwav + align - 1) // align
    base, extra = divmod(ntile, world_size)
    t0 = rank * base + min(rank, extra)
    t1 = t0 + base + (1 if rank < extra else 0)
    return min(t0 * align, nwav), min(t1 * align, nwav)


def reorder_single_band(key_of_range, nwav, sort_on_root, group=None, device=None):
    """reorder_spectrum.cpp:111-300 for ONE band (the fsck structure) on several processes: the sorting key of a wavenumber
    depends on that wavenumber's column alone (:162-228), so every rank sweeps its own range - `key_of_range(begin, end)` ->
    (key, column optical depth) of [begin, end), float64 tensors or arrays - the pieces are gathered on rank 0 in rank order
    (the exchange step of this stage: 16 B per wavenumber, once) and rank 0 sorts: `sort_on_root(key)` -> rank (the per-band
    stable sort, :262-300, is one sort for one band).  Returns (key, col_od, rank) on rank 0, (None, None, None) elsewhere.
    Without a process group: everything on this process."""
    import torch
    r, ws = world(group)
    b, e = wavenumber_range(nwav, r, ws)
    key, col = key_of_range(b, e)
    key = torch.as_tensor(key, dtype=torch.float64)
    col = torch.as_tensor(col, dtype=torch.float64)
    if ws > 1:
        import torch.distributed as dist
        nccl = dist.get_backend(group) == "nccl"
        mine = torch.stack([key, col]).contiguous()                      # (2, n_r)
        if not nccl:
            mine = mine.cpu()
        sizes = [wavenumber_range(nwav, q, ws)[1] - wavenumber_range(nwav, q, ws)[0] for q in range(ws)]
        pieces = [torch.empty((2, n), dtype=torch.float64, device=mine.device) for n in sizes] if r == 0 else None
        dist.gather(mine, pieces, dst=0, group=group) if len(set(sizes)) == 1 else _gather_ragged(mine, pieces, sizes, r, ws, group)
        if r != 0:
            return None, None, None
        whole = torch.cat(pieces, dim=1)
        key, col = whole[0], whole[1]
        if device is not None:
            key, col = key.to(device), col.to(device)
    return key, col, sort_on_root(key)


def _gather_ragged(mine, pieces, sizes, r, ws, group):
    """gather of pieces of different lengths: point-to-point to rank 0 in rank order"""
    import torch.distributed as dist
    if r == 0:
        pieces[0].copy_(mine)
        for q in range(1, ws):
            dist.recv(pieces[q], src=q, group=group)
    else:
        dist.send(mine, dst=0, group=group)


# ---- optimize_lut: training profiles sharded over the ranks (optimize_lut row) -------------------

_PER_COLUMN = ("pressure_hl", "temperature_hl", "vmr_fl", "flux_dn", "flux_up", "spectral_flux_dn_surf", "spectral_flux_up_toa",
               "surf_emissivity", "mu0", "temperature_fl", "relative_flux_dn", "relative_flux_up")


def column_range(ncol, rank, world_size):
    """Contiguous share [begin, end) of `ncol` training profiles owned by `rank` (sizes differ by at most one)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, extra = divmod(ncol, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_scene_columns(scene, rank, world_size):
    """The scene dict of api.Optimizer restricted to this rank's profiles.  Everything indexed by column is sliced;
    per-band / per-g-point quantities (effective albedo, boundary weights, tsi, gas_present) describe the whole
    training set and are shared.  Every rank must keep at least one profile."""
    ncol = np.asarray(scene["pressure_hl"]).shape[0]
    b, e = column_range(ncol, rank, world_size)
    if e <= b:
        raise ValueError(f"rank {rank} of {world_size} would get none of the {ncol} profiles")
    out = dict(scene)
    for k in _PER_COLUMN:
        if scene.get(k) is not None:
            out[k] = np.ascontiguousarray(np.asarray(scene[k])[b:e])
    return out


def make_allreduce_callback(group=None):
    """The ecckd_allreduce_fn for ecckd_opt_set_allreduce: ONE all-reduce (SUM) of the buffer [gradient, cost] per
    cost-function evaluation, over torch.distributed - backend "nccl" is RCCL over xGMI on the GPU box; with the
    "gloo" backend (CPU tests, or several ranks on one GPU) the buffer is staged through the host.
    Returns (ctypes callback, keep-alive); a Python exception inside the callback becomes a non-zero return code,
    which the library reports as PROCESSING_ERROR."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from . import _lib

    backend = dist.get_backend(group)

    class _DevView:                                            # a (count,) float64 view of a raw device pointer
        def __init__(self, ptr, count):
            self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}

    def reduce(ptr, count, on_device):
        if on_device:
            t = torch.as_tensor(_DevView(ptr, count), device="cuda")
            if backend == "gloo":
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            torch.cuda.synchronize()
        else:
            a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(count,))
            dist.all_reduce(torch.from_numpy(a), op=dist.ReduceOp.SUM, group=group)

    def cb(d_buf, count, stream, user):
        try:
            reduce(d_buf, count, on_device=not bool(user))
            return 0
        except Exception as exc:                                # never let an exception cross the C boundary
            import sys
            print(f"ecckd all-reduce callback failed: {exc!r}", file=sys.stderr)
            return 1

    fn = _lib.ALLREDUCE_FN(cb)
    return fn, (cb, reduce)
