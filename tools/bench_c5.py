"""C5: mixed shape classes nx in {4,12,24} x N in {20,50,100} (nu = ceil(nx/4)) plus the
multi-phase class, 524,288 instances in total, every class split over all ranks (one process per GPU).
    python tools/bench_c5.py [--total 524288]                      # 1 GPU: this rank holds everything it is given
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_c5.py
Each rank solves its share of every class as one device batch, the classes concurrently (acados_amd/shape_classes.py);
timed region = all solves of the rank, data resident
in HBM; MAX over ranks; one JSON line on rank 0.  No collective on the data path: an all_gather of statistics after
the timed region."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--total", type=int, default=524288, help="instances over all ranks and classes")
    ap.add_argument("--world-share", type=int, default=8, help="with fewer ranks than this, every rank still holds total/world_share")
    args = ap.parse_args()
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
    from acados_amd.generators import C5_CLASSES, fill_lqr_batch, lqr_dims, multiphase_qp, random_lqr_batch
    from acados_amd.sharding import reduce_max, shard_range

    # weak scaling: the per-rank share is that of the 8-GPU job whatever the number of ranks present
    ranks_total = max(world, args.world_share)
    per_class = args.total // (len(C5_CLASSES) + 1)
    batches, count = [], 0
    for ci, (nx, nu, N) in enumerate(C5_CLASSES):
        lo, hi = shard_range(per_class, rank, ranks_total)
        data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=hi - lo, seed=200 + ci, first=lo)
        gb = OcpQpGpuBatch(lqr_dims(N, nx, nu), hi - lo, device=local_rank)
        fill_lqr_batch(gb, data, N, xp=lambda a: torch.from_numpy(a).to(dev))
        batches.append(((nx, nu, N), gb)); count += hi - lo
    lo, hi = shard_range(min(per_class, 4096), rank, ranks_total)   # multi-phase class: built instance by instance
    if hi > lo:
        gb = OcpQpGpuBatch.from_qps([multiphase_qp(i, N=20) for i in range(lo, hi)], device=local_rank)
        batches.append((("12->4", 3, 20), gb)); count += hi - lo
    for _, gb in batches:
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            gb.opts_set(f, 1e-8)
        gb.solve()                                                   # warm-up
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    from acados_amd.shape_classes import ConcurrentClasses
    cc = ConcurrentClasses([gb for _, gb in batches])   # one host thread per class, the longest class on a high-priority stream
    cc.solve()                                          # warm-up of the concurrent path
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    bad = cc.solve()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = reduce_max(time.perf_counter() - t0, dist, dev)
    per = [{"class": list(c) if not isinstance(c[0], str) else c, "instances": gb.n_batch, "kernel": gb.kernel_name,
            "ms": gb.scalar("time_tot") * 1e3, "iters_mean": float(gb.info("iter").mean()), "failures": int((gb.info("status") != 0).sum())}
           for c, gb in batches]
    tot = torch.tensor([count, bad], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tot)
    if rank == 0:
        print(json.dumps({"metric": "OCP-QP solves/sec, mixed shape classes (C5)", "value": float(tot[0]) / elapsed, "unit": "OCP-QP solves/s",
                          "n_gpus": world, "instances": int(tot[0]), "failures": int(tot[1]), "seconds": elapsed, "scaling": "weak",
                          "per_class_rank0": per}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
