"""Instance sharding of a QP batch over ranks (one process per GPU) and the ONLY collectives the path
needs: a gather of solutions / statistics and a max-reduction of timings, both outside the solve.
OCP-QP instances are independent, so the data path itself has no collective.
Works with any torch.distributed backend: "nccl" (= RCCL over xGMI on MI355X) in production, "gloo"
in the CPU tests."""
import numpy as np


def shard_range(n_total: int, rank: int, world: int):
    """contiguous, balanced instance range [lo, hi) owned by `rank` (remainder to the first ranks)"""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_by_class(class_sizes, rank: int, world: int):
    """mixed-shape batches (C5): every shape class is split over all ranks; returns per-class ranges"""
    return [shard_range(n, rank, world) for n in class_sizes]


def gather_instances(local, n_total: int, dist=None, device=None):
    """all_gather of per-instance rows (uneven shards padded) -> array [n_total, ...] in instance order"""
    import torch
    local = np.ascontiguousarray(local)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    cap = max(hi - lo for lo, hi in sizes)
    pad = np.zeros((cap,) + local.shape[1:], dtype=local.dtype)
    pad[: local.shape[0]] = local
    t = torch.from_numpy(pad)
    if device is not None:
        t = t.to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return np.concatenate([o.cpu().numpy()[: hi - lo] for o, (lo, hi) in zip(outs, sizes)], axis=0)


def reduce_max(value: float, dist=None, device=None) -> float:
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
