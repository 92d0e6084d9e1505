#!/usr/bin/env python3
"""bench.py -- OCP-QP solves/sec on MI355X (BASELINE.json metric), contract of the driver.

A "step" = one cold-start solve of the whole batch of synthetic OCP-QPs (every IPM iteration of
every instance), inputs already packed and resident in HBM when the timed region starts.
Workload at N=1 GPU: BASELINE.json configs[1] -- random LQR-like OCP-QP, N=50, nx=8, nu=3,
batch=65,536, u-box + x0 equality, tolerances 1e-8, iter_max 50 (SURVEY.md 8d "C2 input").
N>1: one process per GPU (torch.distributed.run), the batch is sharded by instance -- every rank
solves its own 65,536 instances (weak scaling), no data-path collective; an RCCL all_gather of the
first-stage controls + per-rank statistics runs AFTER the timed region (its time is reported).

Extra objects on the JSON line:
  roofline     dominant kernel (by accumulated HIP-event time inside the timed region, events on the
               stream the kernels are launched on): algorithmic bytes per launch / avg duration vs the
               8 TB/s HBM peak.  Algorithmic bytes per launch = active instances x 98,056 B
               (unique QP input 85,336 B + iterate/solution 12,720 B per solve).
  cpu_baseline the oracle (restated CPU port, NOT HPIPM: its sources are absent from the reference
               tree) on a bounded sample of the same workload on the host cores, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_IN, BYTES_OUT = 85336, 12720
HBM_PEAK_GBS = 8000.0                   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_bytes(N, nx, nu):
    """SURVEY.md 8d formula for the u-box + x0-equality LQR shape (nb_k = nu (k<N), nb_0 += nx)"""
    dbl_in = N * (nx * nx + nx * nu + nx) + (N + 1) * (nx * nx + nx) + N * (nu * nx + nu * nu + nu) \
        + 2 * (N * nu + nx)
    int_in = N * nu + nx            # idxb only (idxs_rev is empty when ns = 0): 158 ints, SURVEY 8d
    dbl_out = (N + 1) * nx + N * nu + N * nx + 4 * (N * nu + nx)
    return 8 * dbl_in + 4 * int_in, 8 * dbl_out


def cpu_baseline(data, N, sample, threads):
    from acados_amd.generators import lqr_instance_qp
    from oracle.oracle import OracleQp, default_opts, solve_batch
    qps = [OracleQp(lqr_instance_qp(data, i, N)) for i in range(sample)]
    opts = default_opts(tol_stat=1e-8, tol_eq=1e-8, tol_ineq=1e-8, tol_comp=1e-8, iter_max=50)
    best = 1e300
    for _ in range(3):      # min over repeats, as mass_spring_example.c:336-363 does
        t0 = time.perf_counter()
        st = solve_batch(qps, opts, nthreads=threads)
        best = min(best, time.perf_counter() - t0)
    assert np.all(st == 0)
    iters = float(np.mean([q.iter for q in qps]))
    cpu_baseline.solved = qps     # the same solutions double as the >= 1,024-instance parity sample
    return {"value": sample / best, "unit": "OCP-QP solves/s", "cores": threads, "kind": "port",
            "sample": f"{sample} instances of the same workload (seed 0, first instances), min of 3 repeats, "
                      f"OpenMP over instances as acados_solver.in.c:3232 does; restated CPU oracle, not HPIPM",
            "mean_iter": iters}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=65536, help="instances per GPU")
    ap.add_argument("--horizon", type=int, default=50)
    ap.add_argument("--nx", type=int, default=8)
    ap.add_argument("--nu", type=int, default=3)
    ap.add_argument("--cpu-sample", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--compact-min", type=int, default=None, help="override the library default of the compaction threshold")
    ap.add_argument("--check", type=int, default=8, help="instances per rank checked against the oracle (outside timing)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch

    N, nx, nu, B = args.horizon, args.nx, args.nu, args.batch
    # instance ids are global: rank r owns [r*B, (r+1)*B) of one counter-based stream
    data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=B, seed=0, first=rank * B)

    gb = OcpQpGpuBatch(lqr_dims(N, nx, nu), B, device=local_rank)

    def to_dev(a):
        t = torch.from_numpy(a).to(dev)
        torch.cuda.synchronize()
        return t

    t0 = time.perf_counter()
    fill_lqr_batch(gb, data, N, xp=to_dev)
    t_pack = time.perf_counter() - t0
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        gb.opts_set(f, 1e-8)
    gb.opts_set("iter_max", 50)
    gb.opts_set("warm_start", 0)
    if args.compact_min is not None:
        gb.opts_set("compact_min", args.compact_min)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        gb.solve()
    gb.opts_set("profile", 1)
    gb.scalar("prof_reset")
    barrier()
    t0 = time.perf_counter()
    bad = 0
    for _ in range(args.steps):
        bad += gb.solve()
    barrier()
    elapsed = time.perf_counter() - t0
    gb.opts_set("profile", 0)

    from acados_amd.sharding import gather_instances, reduce_max
    elapsed = reduce_max(elapsed, dist, dev)      # MAX over ranks

    iters = gb.info("iter")
    status = gb.info("status")
    res_max = max(float(gb.info(n).max()) for n in ("res_stat", "res_eq", "res_ineq", "res_comp"))

    # ---- gather of solutions + statistics over RCCL/xGMI, outside the timed region ----
    gather_ms = None
    if dist is not None:
        u0 = gb.get("u", 0)
        stats = np.array([[float(iters.mean()), float(iters.max()), float((status != 0).sum()), res_max]])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        u_all = gather_instances(u0, world * B, dist, dev)          # RCCL all_gather over xGMI
        s_all = gather_instances(stats, world, dist, dev)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - t0) * 1e3
        assert u_all.shape == (world * B, nu)
        mean_iter, max_iter = float(s_all[:, 0].mean()), int(s_all[:, 1].max())
        failures, res_max = int(s_all[:, 2].sum()), float(s_all[:, 3].max())
    else:
        mean_iter, max_iter, failures = float(iters.mean()), int(iters.max()), int((status != 0).sum())

    # ---- parity spot check against the oracle (checker only, outside timing) ----
    err = None
    if args.check > 0:
        from acados_amd.generators import lqr_instance_qp
        from oracle.oracle import OracleQp, default_opts
        xs = [gb.get("x", k) for k in range(N + 1)]
        us = [gb.get("u", k) for k in range(N)]
        err = 0.0
        for i in np.linspace(0, B - 1, args.check).astype(int):
            o = OracleQp(lqr_instance_qp(data, int(i), N))
            o.solve(default_opts(tol_stat=1e-8, tol_eq=1e-8, tol_ineq=1e-8, tol_comp=1e-8))
            for k in range(N + 1):
                r = o.get(k, "x")
                err = max(err, float(np.max(np.abs(xs[k][i] - r) / np.maximum(1.0, np.abs(r)))))
                if k < N:
                    r = o.get(k, "u")
                    err = max(err, float(np.max(np.abs(us[k][i] - r) / np.maximum(1.0, np.abs(r)))))

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (HIP events recorded on the launch stream) ----
    classes = ("back_fact", "fwd_aff", "back_rhs", "fwd_corr", "init", "finalize")
    prof = {c: (gb.scalar(f"prof_ms_{c}"), int(gb.scalar(f"prof_cnt_{c}"))) for c in classes}
    dom = max(prof, key=lambda c: prof[c][0])
    dom_ms, dom_cnt = prof[dom]
    b_in, b_out = algorithmic_bytes(N, nx, nu)
    # units one launch processes: launch j of the factor kernel sees the instances that have not converged before
    # iteration j (iter >= j), the other sweeps those with iter > j; only the launches of the full batch are timed
    # (the last survivors continue on a small wave-per-instance sub-batch)
    launches_per_solve = max(dom_cnt // max(args.steps, 1), 1)
    hist = np.bincount(iters, minlength=launches_per_solve + 1)
    still = B - np.cumsum(hist)                      # still[j] = instances with iter > j
    units = [(B if j == 0 else int(still[j - 1])) if dom == "back_fact" else int(still[j]) for j in range(launches_per_solve)]
    per_launch_bytes = float(np.mean(units)) * (b_in + b_out)
    avg_s = dom_ms * 1e-3 / max(dom_cnt, 1)
    achieved = per_launch_bytes / avg_s / 1e9
    solves_per_s = world * B * args.steps / elapsed
    # HBM traffic of the dominant kernel from the committed PMC passes (rocprofv3 cannot run inside
    # this process); corrected as MI355X_MICROARCH.md prescribes, see profiles/r01_v6_pmc_traffic.json
    kern_sym = {"back_fact": "kb_factor", "fwd_aff": "kb_forward", "back_rhs": "kb_backrhs", "fwd_corr": "kb_forward"}[dom] \
        if dom in ("back_fact", "fwd_aff", "back_rhs", "fwd_corr") else dom
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_v6_pmc_traffic.json")))
        want = {"back_fact": f"kb_factor<{nx}, {nu}, false>", "back_rhs": f"kb_backrhs<{nx}, {nu}, false>",
                "fwd_aff": f"kb_forward<{nx}, {nu}, false, false>", "fwd_corr": f"kb_forward<{nx}, {nu}, false, true>"}.get(dom)
        if want in pmc and B == 65536 and N == 50:
            traffic = pmc[want]["hbm_bytes_per_launch_avg"]
    except Exception:
        traffic = None
    out = {
        "metric": "OCP-QP solves/sec (batch) at N=50 nx=8 nu=3",
        "value": solves_per_s,
        "unit": "OCP-QP solves/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"random LQR OCP-QP (BASELINE configs[1]): N={N} nx={nx} nu={nu}, u-box + x0 equality, "
                               f"full-space Riccati IPM, cold start, tol 1e-8, iter_max 50",
                   "batch_per_gpu": B, "global_batch": world * B, "parallelism": f"instance-sharded x{world}",
                   "kernel": gb.kernel_name},
        "ipm": {"mean_iter": mean_iter, "max_iter": max_iter, "failures": failures, "max_kkt_residual": res_max,
                "max_rel_primal_err_vs_oracle": err, "oracle_checked_instances": args.check,
                "launches_per_step": int(gb.scalar("launches"))},
        "roofline": {"bound": "hbm", "kernel": f"{kern_sym} ({dom}) of {gb.kernel_name}",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic,
                     "bytes_per_launch": per_launch_bytes, "units_per_launch": units, "avg_launch_ms": avg_s * 1e3,
                     "launches_timed": dom_cnt,
                     "kernel_ms_share": {c: prof[c][0] for c in classes},
                     "whole_solve_GBps": solves_per_s / world * (b_in + b_out) / 1e9,
                     "whole_solve_frac": solves_per_s / world * (b_in + b_out) / 1e9 / HBM_PEAK_GBS},
        "pack_s": t_pack,
        "hbm_bytes_per_gpu": gb.bytes,
        "gather_ms": gather_ms,
    }
    if world == 1 and not args.no_cpu_baseline:
        threads = os.cpu_count() or 1
        out["cpu_baseline"] = cpu_baseline(data, N, min(args.cpu_sample, B), threads)
        if args.check > 0:
            # max relative primal error of the GPU solution against the oracle over the whole CPU sample
            err_s = 0.0
            for i, o in enumerate(cpu_baseline.solved):
                for k in range(N + 1):
                    r = o.get(k, "x")
                    err_s = max(err_s, float(np.max(np.abs(xs[k][i] - r) / np.maximum(1.0, np.abs(r)))))
                    if k < N:
                        r = o.get(k, "u")
                        err_s = max(err_s, float(np.max(np.abs(us[k][i] - r) / np.maximum(1.0, np.abs(r)))))
            out["ipm"]["max_rel_primal_err_vs_oracle"] = max(err, err_s)
            out["ipm"]["oracle_checked_instances"] = args.check + len(cpu_baseline.solved)
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
