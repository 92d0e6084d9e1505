"""N > 1: the guarded solutions gather, the self-launch under torch.distributed.run, the dry run, --config c5 (a part of bench.py)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from benchlib.configs import tol_setup
from benchlib.line import emit, git_head


GATHER_LIMIT_S = 240.0


def guarded_gather(fn, world, limit_s=None):
    """The solutions gather is the one step of an N > 1 run that has never met more than one device (RCCL has only
    ever run as one rank here).  It runs after everything the line needs has been measured; with more than one rank it runs on a
    helper thread under a time limit, so that a collective that never completes costs the gather record, not the run: returns
    (result or None, error string or None).  The caller emits its line and, if the error says the collective is still in flight, leaves
    the process with os._exit (the helper thread cannot be joined)."""
    if world <= 1:
        return fn(), None
    import threading
    import torch
    limit_s = float(os.environ.get("ACADOS_AMD_GATHER_LIMIT_S", GATHER_LIMIT_S)) if limit_s is None else limit_s
    box, dev_idx = {}, torch.cuda.current_device() if torch.cuda.is_available() else None

    def run():
        try:
            if dev_idx is not None:
                torch.cuda.set_device(dev_idx)      # the current device is per thread
            box["v"] = fn()
        except Exception as e:      # noqa: BLE001 -- reported on the line
            box["e"] = f"{type(e).__name__}: {e}"
    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(limit_s)
    if th.is_alive():
        return None, f"in flight after {limit_s:.0f} s"
    return box.get("v"), box.get("e")


def leave_after_stuck_gather(err):
    """a collective still in flight holds the stream and a thread: no destroy_process_group, no interpreter shutdown"""
    if err and err.startswith("in flight"):
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


def relaunch(n):
    """N ranks of this script on one node (the command line the driver uses for N > 1): rank 0's JSON line is the last line of
    stdout, the exit code is the launcher's"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py")] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def dry_run(args):
    """the launch / sharding path without a GPU: a gloo group of the ranks that were started"""
    import torch.distributed as dist
    from acados_amd.generators import C5_CLASSES
    from acados_amd.sharding import shard_range
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    if args.config == "c5":
        per_class = args.c5_total // len(C5_CLASSES)
        # every class (the nine shapes + the multi-phase one) is split the same way: one range per rank says it all; gather_counts is
        # what main_c5 hands to the library's exact-count gather (uneven shards: 58,254 = 6 x 7,282 + 2 x 7,281)
        mine = {"classes": len(C5_CLASSES) + 1, "per_class": list(shard_range(per_class, rank, max(world, 8))),
                "gather_counts": [hi - lo for lo, hi in (shard_range(per_class, r, max(world, 8)) for r in range(world))]}
    else:
        mine = {"instances": [rank * args.batch, (rank + 1) * args.batch]}
    mine.update(rank=rank, local_rank=int(os.environ.get("LOCAL_RANK", "0")), pid=os.getpid())
    ranks = [mine]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "config": args.config, "ranks": ranks,
                          "gather": {"ranks": world, "collective": "ocp_qp_gpu_batch_gather (RCCL) after the timed region"}},
                         separators=(",", ":")), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main_c5(args):
    """BASELINE configs[4]: the nine shape classes nx in {4,12,24} x N in {20,50,100} plus the multi-phase class, --c5-total
    instances split evenly over the classes and every class over the ranks (identical work per rank: ranks finish together);
    each rank solves its share of every class as one device batch, the classes concurrently (acados_amd/shape_classes.py).
    Timed region = `steps` solves of everything a rank holds, data resident; MAX over ranks; afterwards every class's
    solutions are gathered through the library's collective (ocp_qp_gpu_batch_gather, RCCL)."""
    import torch
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import (C5_CLASSES, fill_lqr_batch, fill_multiphase_batch, lqr_dims, multiphase_batch, multiphase_dims,
                                       random_lqr_batch)
    from acados_amd.shape_classes import ConcurrentClasses
    from acados_amd.sharding import gather_solutions, reduce_max, shard_range
    ranks_total = max(world, 8)          # weak scaling: a rank holds the share of the 8-GPU job whatever the number of ranks present
    per_class = args.c5_total // len(C5_CLASSES)      # split equally over the nine classes; the multi-phase class comes on top
    lo, hi = shard_range(per_class, rank, ranks_total)
    batches = []
    for ci, (nx, nu, N) in enumerate(C5_CLASSES):
        data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=hi - lo, seed=200 + ci, first=lo)
        gb = OcpQpGpuBatch(lqr_dims(N, nx, nu), hi - lo, device=local_rank)
        fill_lqr_batch(gb, data, N, xp=lambda a: torch.from_numpy(a).to(dev))
        batches.append((f"nx={nx} nu={nu} N={N}", gb))
    dm = multiphase_batch(N=50, batch=hi - lo, first=lo)
    gm = OcpQpGpuBatch(multiphase_dims(50), hi - lo, device=local_rank)
    fill_multiphase_batch(gm, dm)
    batches.append(("multi-phase nx=12->4 at k=25 nu=3 N=50", gm))
    for _, gb in batches:
        tol_setup(gb)
        gb.solve()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with ConcurrentClasses([gb for _, gb in batches]) as cc:
        for _ in range(max(args.warmup, 1)):
            cc.solve()
        barrier()
        t0 = time.perf_counter()
        bad = 0
        for _ in range(args.steps):
            bad += cc.solve()
        barrier()
        elapsed = reduce_max(time.perf_counter() - t0, dist, dev)
    count = sum(gb.n_batch for _, gb in batches)
    per = [{"class": c, "instances": gb.n_batch, "kernel": gb.kernel_name, "ms": gb.scalar("time_tot") * 1e3,
            "iters_mean": float(gb.info("iter").mean()), "failures": int((gb.info("status") != 0).sum()),
            "max_kkt_residual_independent": float(gb.res_compute().max())} for c, gb in batches]
    # shards of a class are uneven when per_class is not a multiple of the rank count (58,254 = 6 x 7,282 + 2 x 7,281): the
    # library's exact-count gather (ocp_qp_gpu_batch_gather_v) needs every rank's count
    counts = [hi_r - lo_r for lo_r, hi_r in (shard_range(per_class, r, ranks_total) for r in range(world))]
    tot = torch.tensor([count, bad], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tot)
    gathers, gerr = guarded_gather(lambda: [gather_solutions(gb, dist, rank, world, counts=counts) for _, gb in batches], world)
    if rank == 0:
        ok = [g for g in (gathers or []) if g]
        out = {"metric": "OCP-QP solves/sec, mixed shape classes (BASELINE configs[4])", "value": float(tot[0]) * args.steps / elapsed,
               "unit": "OCP-QP solves/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1),
               "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"ten shape classes (nx in {{4,12,24}} x N in {{20,50,100}} + multi-phase nx 12->4), "
                                      f"{args.c5_total} instances per 8 GPUs over the nine classes + the same share of the multi-phase class, {count} on this rank, "
                                      f"classes solved concurrently",
                          "global_batch": int(tot[0]), "parallelism": f"every class instance-sharded x{world}", "commit": git_head()},
               "failures": int(tot[1]), "per_class_rank0": per,
               "gather": {"ranks": world, "ms": sum(g["ms"] for g in ok), "ms_all_classes": sum(g["ms"] for g in ok), "classes_gathered": len(ok),
                          "slice_matches_getters": all(g["slice_matches_getters"] for g in ok) if ok else None,
                          "instances_per_rank": counts,
                          "collective": ok[0]["collective"] if ok else None}}
        if gerr:
            out["gather"]["error"] = gerr
        emit(out, args)
    leave_after_stuck_gather(gerr)
    if dist is not None:
        dist.destroy_process_group()
