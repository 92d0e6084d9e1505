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


def gather_solutions(gb, dist=None, rank: int = 0, world: int = 1, counts=None):
    """The path's only collective, through the library's own entry: ONE RCCL all-gather over xGMI of the
    full solution payload {u x sl su pi lam t} + (status, iter) of every instance + the solve time of every rank,
    from device buffers on the batch's stream -- no host bounce.  torch is plumbing here: it broadcasts the 128-byte
    RCCL unique id and owns the receive buffers.  `counts`: instances per rank when the shards are uneven (the library then
    moves exact-size point-to-point transfers in one group, ocp_qp_gpu_batch_gather_v); the gathered arrays are in rank
    order, rank r's instances at offset sum(counts[:r]).  Returns timing and a consistency check, or None without a GPU
    library entry (host simulation)."""
    import ctypes as C
    import time
    import torch
    L, h, B = gb._L, gb._h, gb.n_batch
    dev = torch.device("cuda", torch.cuda.current_device())
    uid = torch.zeros(128, dtype=torch.uint8, device=dev)
    if rank == 0:
        buf = (C.c_char * 128)()
        if L.ocp_qp_gpu_comm_unique_id(buf) != 0:
            return None
        uid.copy_(torch.frombuffer(bytearray(bytes(buf)), dtype=torch.uint8))
    if dist is not None and dist.is_initialized() and world > 1:
        dist.broadcast(uid, 0)
    idb = bytes(uid.cpu().numpy().tobytes())
    comm = L.ocp_qp_gpu_comm_create(idb, int(world), int(rank), -1)
    if not comm:
        return None
    comm = C.c_void_p(comm)
    Lout = L.ocp_qp_gpu_batch_bulk_len(h, 1)
    cnts = np.full(world, B, dtype=np.int32) if counts is None else np.ascontiguousarray(counts, dtype=np.int32)
    assert cnts.size == world and int(cnts[rank]) == B
    first = int(cnts[:rank].sum())
    total = int(cnts.sum())
    sol = torch.empty((total, Lout), dtype=torch.float64, device=dev)
    info = torch.empty((total, 2), dtype=torch.int32, device=dev)
    tm = torch.empty(world, dtype=torch.float64, device=dev)
    cptr = cnts.ctypes.data_as(C.c_void_p)
    torch.cuda.synchronize()
    best = None
    for _ in range(2):      # first call pays RCCL's lazy channel setup
        t0 = time.perf_counter()
        rc = L.ocp_qp_gpu_batch_gather_v(h, comm, -1, cptr, C.c_void_p(sol.data_ptr()), C.c_void_p(info.data_ptr()), C.c_void_p(tm.data_ptr()))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    # the same payload to rank 0 only (ncclSend / ncclRecv in one group): 1x per link instead of n_ranks x per rank
    root_ms, root_ok = None, None
    if hasattr(L, "ocp_qp_gpu_batch_gather_root") and rc == 0:
        sol_r = torch.zeros_like(sol) if rank == 0 else None
        info_r = torch.zeros_like(info) if rank == 0 else None
        tm_r = torch.zeros_like(tm) if rank == 0 else None
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rr = L.ocp_qp_gpu_batch_gather_v(h, comm, 0, cptr, ptr(sol_r), ptr(info_r), ptr(tm_r))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if rr == 0:
                root_ms = dt * 1e3 if root_ms is None else min(root_ms, dt * 1e3)
        if rank == 0 and root_ms is not None:
            root_ok = bool(torch.equal(sol_r, sol)) and bool(torch.equal(info_r, info))
    L.ocp_qp_gpu_comm_destroy(comm)
    if rc != 0:
        return None
    # consistency: this rank's slice of the gathered payload is what the getters return
    n0 = C.c_int(0)
    off = L.ocp_qp_gpu_batch_bulk_offset(h, 1, b"u", 0, C.byref(n0))
    own = sol[first:first + B, off:off + n0.value].cpu().numpy()
    ok = bool(np.array_equal(own, gb.get("u", 0))) and bool(np.array_equal(info[first:first + B, 0].cpu().numpy(), gb.info("status")))
    payload = total * (Lout * 8 + 8) + world * 8
    return {"ms": best * 1e3, "bytes_per_instance": Lout * 8 + 8, "payload_bytes_received_per_rank": payload,
            "GBps_received_per_rank": payload / best / 1e9, "ranks": world, "slice_matches_getters": ok,
            "iter_mean_all_ranks": float(info[:, 1].double().mean().item()),
            "nonzero_status_all_ranks": int((info[:, 0] != 0).sum().item()), "instances_per_rank": [int(v) for v in cnts], "solve_s_per_rank": [float(v) for v in tm.cpu().numpy()],
            "collective": ("ncclAllGather x3 (solutions f64, status/iter i32, time f64)" if len(set(int(v) for v in cnts)) == 1 else
                           "ncclSend / ncclRecv with exact counts in one group (uneven shards)") + " via ocp_qp_gpu_batch_gather_v",
            "gather_to_root_ms": root_ms, "gather_to_root_equals_all_gather": root_ok,
            "gather_to_root": "ncclSend / ncclRecv group to rank 0 via ocp_qp_gpu_batch_gather_root (1x payload per link)"}


def reduce_max(value: float, dist=None, device=None) -> float:
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
