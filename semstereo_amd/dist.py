"""Multi-GPU execution of the hot path: one process per GPU, pairs sharded over ranks.

The reference scales with nn.DataParallel (single process, batch scatter / output gather,
main_us3d.py:100, test_us3d.py:58).  Every op on the path is per-pair (BatchNorm uses running
statistics in eval), so the MI355X-native form is embarrassingly parallel: rank r owns a contiguous
block of the batch, runs the identical kernels, and no collective is needed INSIDE the forward.
RCCL (torch.distributed backend "nccl" on ROCm) is used only around it: a one-time weight
broadcast, and after the forward either an all_gather of the [b,H,W] disparities or an all_reduce
of a few scalars.  All helpers work with the gloo backend too (CPU tests, world_size 2).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, force=False):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns
    (rank, world_size, local_rank).  A single process (no env) needs no group; `force` (or SS_DIST_FORCE_INIT=1)
    initialises one anyway -- a world of ONE rank over RCCL runs the same communicator set-up, broadcast, all_gather
    and all_reduce code as N ranks do (tests/test_nccl_world1_gpu.py: the nccl path on a one-GPU box)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = force or os.environ.get("SS_DIST_FORCE_INIT", "0") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_bounds(n_items, rank, world):
    """Contiguous block [lo, hi) of `n_items` owned by `rank`: sizes differ by at most one, the
    first (n_items % world) ranks get the extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(tensors, rank, world):
    """Slice dim 0 of every tensor to this rank's block."""
    n = tensors[0].shape[0]
    lo, hi = shard_bounds(n, rank, world)
    return [t[lo:hi] for t in tensors]


def broadcast_module(module, src=0):
    """One-time weight broadcast so every rank holds rank `src`'s parameters and buffers."""
    if not (dist.is_available() and dist.is_initialized()):          # (a world of one rank still runs the collective)
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)


def gather_batch(local, n_items):
    """all_gather of per-rank blocks back into the full batch order.  Blocks may differ in length
    by one (shard_bounds), so they are padded to the largest block for the collective."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_bounds(n_items, r, world) for r in range(world)]
    biggest = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((biggest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0)


def reduce_metrics(n_pairs, abs_err_sum, n_pixels, seconds, device):
    """all_reduce of the benchmark scalars: SUM of pairs / |error| / pixels, MAX of elapsed time."""
    sums = torch.tensor([float(n_pairs), float(abs_err_sum), float(n_pixels)], dtype=torch.float64, device=device)
    tmax = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    return sums[0].item(), sums[1].item(), sums[2].item(), tmax[0].item()
