#!/usr/bin/env python3
"""bench.py -- OCP-QP solves/sec on MI355X (BASELINE.json metric), contract of the driver.

A "step" = one cold-start solve of the whole batch of synthetic OCP-QPs (every IPM iteration of
every instance), inputs already packed and resident in HBM when the timed region starts.
Workload at N=1 GPU: BASELINE.json configs[1] (C2) -- random LQR-like OCP-QP, N=50, nx=8, nu=3,
batch=65,536, u-box + x0 equality, tolerances 1e-8, iter_max 50 (SURVEY.md 8d "C2 input").
N>1: one process per GPU (torch.distributed.run), the batch is sharded by instance -- every rank
solves its own 65,536 instances (weak scaling), no data-path collective; AFTER the timed region the
full solution payload {ux, pi, lam, t, status, iter, solve time} of every rank is gathered with ONE
RCCL all-gather over xGMI from device buffers through the library's own collective entry
(ocp_qp_gpu_batch_gather; its time is reported as gather_ms).

Extra objects on the JSON line:
  roofline     dominant kernel (by accumulated HIP-event time inside the timed region, events on the
               stream the kernels are launched on): algorithmic bytes per launch / avg duration vs the
               8 TB/s HBM peak.  Algorithmic bytes per launch = instances the launch still processes x
               98,056 B (unique QP input 85,336 B + iterate/solution 12,720 B per solve).
               `traffic` = HBM bytes per launch of the SAME kernel over the SAME set of launches from the
               rocprofv3 PMC passes of tools/profile_round.sh (FETCH_SIZE / WRITE_SIZE, separate passes,
               gfx950 correction of MI355X_MICROARCH.md), read from the newest profiles/*_pmc_traffic.json
               whose recorded commit is reported next to it; `full_launch` repeats both for the launches in
               which every instance is still iterating.
  configs      the other single-GPU configurations of BASELINE.json (C3, C4, the per-GPU share of C5),
               each with solves/s, iterations, failures, independently recomputed KKT residual, oracle
               error on a sample, dominant kernel and its roofline fraction (rank 0, N=1 only).
  cpu_baseline the oracle (restated CPU port, NOT HPIPM: its sources are absent from the reference
               tree) on a bounded sample of the same workload on the host cores, rank 0, N=1 only:
               one-thread rate, best thread count of a sweep, cores visible / allowed.
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# the parts (tests import them through this module: `from bench import oracle_error, compact_line ...`)
from benchlib.line import git_head, LINE_LIMIT, _r, _pick, DIST_KEYS, _dist, compact_line, emit  # noqa: F401
from benchlib.roofline import HBM_PEAK_GBS, HBM_COPY_GBS, SWEEPS, CLASSES, algorithmic_bytes_dims, kernel_symbol, sweep_roofline, pmc_traffic, config_traffic, mark_stale, mfma_util, MFMA_PROBES, MFMA_NOTE_C2, mfma_note_c3  # noqa: F401
from benchlib.cpu import cpu_caps, threads_allowed, cpu_baseline  # noqa: F401
from benchlib.parity import TIGHT, oracle_error  # noqa: F401
from benchlib.configs import tol_setup, run_config, polish_leg, other_configs, PCIE_PEAK_GBS, write_driver_qp, boundary_c3  # noqa: F401
from benchlib.multi import GATHER_LIMIT_S, guarded_gather, leave_after_stuck_gather, relaunch, dry_run, main_c5  # noqa: F401


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=65536, help="instances per GPU")
    ap.add_argument("--horizon", type=int, default=50)
    ap.add_argument("--nx", type=int, default=8)
    ap.add_argument("--nu", type=int, default=3)
    ap.add_argument("--cpu-sample", type=int, default=2048, help="unique instances built for the CPU baseline / parity sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip C3 / C4 / C5 (the headline line only)")
    ap.add_argument("--c4-batch", type=int, default=16384)
    ap.add_argument("--check-configs", type=int, default=1024,
                    help="instances per configuration checked against the oracle (OpenMP batch on the host, outside timing; SURVEY 8d asks >= 1,024)")
    ap.add_argument("--boundary-n", type=int, default=4096, help="capsules of the through-the-boundary leg (configs.boundary_C3)")
    ap.add_argument("--polish-legs", action="store_true",
                    help="also time the opt-in terminal polishing step on C2 / C4 (measured and dominated by a tighter tol_comp: profiles/r06_polish_sweep.txt)")
    ap.add_argument("--compact-min", type=int, default=None, help="override the library default of the compaction threshold")
    ap.add_argument("--check", type=int, default=8, help="instances per rank checked against the oracle (outside timing)")
    ap.add_argument("--config", choices=("c2", "c5"), default="c2",
                    help="c2: the headline workload (BASELINE configs[1]); c5: the mixed-shape-class batch of configs[4], 524,288 instances "
                         "split over the ranks present (a rank of a smaller job still holds the share of an 8-GPU job: weak scaling)")
    ap.add_argument("--c5-total", type=int, default=524288)
    ap.add_argument("--detail-file", default=None,
                    help="where the full result object goes (default gpurun_out/bench_detail.json); stdout's last line is the compact one")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch path only: every rank joins a gloo group, reports the instance ranges it would own, rank 0 prints one JSON "
                         "line; no GPU is touched (the CPU tier checks that --gpus N starts N ranks)")
    args = ap.parse_args()

    # `python bench.py --gpus N` started as ONE process: become N ranks (one process per GPU) under torch.distributed.run.
    # Started by the driver under torch.distributed.run (WORLD_SIZE set), the ranks are already there.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch(args.gpus))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus and int(os.environ.get("RANK", "0")) == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: the launcher's world size is what runs", file=sys.stderr)
    if args.dry_run:
        return dry_run(args)
    if args.config == "c5":
        return main_c5(args)

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
    from acados_amd.generators import fill_lqr_batch, lqr_dims, lqr_instance_qp, random_lqr_batch

    N, nx, nu, B = args.horizon, args.nx, args.nu, args.batch
    # instance ids are global: rank r owns [r*B, (r+1)*B) of one counter-based stream
    data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=B, seed=0, first=rank * B)
    dims = lqr_dims(N, nx, nu)
    gb = OcpQpGpuBatch(dims, B, device=local_rank)

    def to_dev(a):
        t = torch.from_numpy(a).to(dev)
        torch.cuda.synchronize()
        return t

    t0 = time.perf_counter()
    fill_lqr_batch(gb, data, N, xp=to_dev)
    t_pack = time.perf_counter() - t0
    tol_setup(gb)
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

    from acados_amd.sharding import gather_solutions, reduce_max
    elapsed = reduce_max(elapsed, dist, dev)      # MAX over ranks

    iters = gb.info("iter")
    status = gb.info("status")
    res_max = max(float(gb.info(n).max()) for n in ("res_stat", "res_eq", "res_ineq", "res_comp"))
    res_indep = float(gb.res_compute().max())

    if dist is not None:
        stats = torch.tensor([[float(iters.mean()), float(iters.max()), float((status != 0).sum()), res_max, res_indep]],
                             dtype=torch.float64, device=dev)
        s_all = [torch.empty_like(stats) for _ in range(world)]
        dist.all_gather(s_all, stats)
        s_all = torch.cat(s_all).cpu().numpy()
        mean_iter, max_iter = float(s_all[:, 0].mean()), int(s_all[:, 1].max())
        failures, res_max, res_indep = int(s_all[:, 2].sum()), float(s_all[:, 3].max()), float(s_all[:, 4].max())
    else:
        mean_iter, max_iter, failures = float(iters.mean()), int(iters.max()), int((status != 0).sum())

    # ---- parity spot check against the oracle (checker only, outside timing) ----
    err = None
    if args.check > 0:
        err = oracle_error(gb, lambda i: lqr_instance_qp(data, i, N), np.linspace(0, B - 1, args.check).astype(int), N, tight=False)["same_tol_max"]

    # ---- gather of the full solution payload + statistics over RCCL/xGMI, outside the timed region, LAST: everything the line
    # reports is in hand by then (guarded_gather) ----
    def the_gather():
        return gather_solutions(gb, dist, rank, world)       # device buffers, library collective (no host bounce)

    if rank != 0:
        _, gerr = guarded_gather(the_gather, world)
        leave_after_stuck_gather(gerr)
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (HIP events recorded on the launch stream) ----
    b_in, b_out = algorithmic_bytes_dims(dims)
    dom, prof, roof = sweep_roofline(gb, args.steps, b_in + b_out)
    solves_per_s = world * B * args.steps / elapsed
    tr = mark_stale(pmc_traffic(dom, nx, nu, B, N))
    roof["mfma"] = MFMA_NOTE_C2
    roof["traffic"] = tr["avg_main"] if tr else None
    roof["traffic_source"] = tr
    roof["traffic_note"] = ("HBM bytes per launch from the committed rocprofv3 PMC summary named in traffic_source (counter passes "
                            "cannot run inside this process); the time it is divided by is measured live in this run")
    roof["traffic_over_algorithmic"] = (tr["avg_main"] / roof["bytes_per_launch"]) if tr else None
    # the REAL rate of the dominant sweep (PMC bytes of the committed summary over the launch time measured here) against the copy
    # rate the part sustains: how much of the gap between `frac` and 1 is bytes the sweep also carries, how much is rate
    roof["traffic_GBps"] = (tr["avg_main"] / (roof["avg_launch_ms"] * 1e-3) / 1e9) if tr and roof.get("avg_launch_ms") else None
    roof["traffic_frac_of_sustained_copy"] = (roof["traffic_GBps"] / HBM_COPY_GBS) if roof["traffic_GBps"] else None
    full_units = [u for u in roof["units_per_launch"] if u == B]
    roof["full_launch"] = {"bytes": float(B * (b_in + b_out)), "traffic": tr["full"] if tr else None,
                           "traffic_over_algorithmic": (tr["full"] / (B * (b_in + b_out))) if tr else None,
                           "launches_per_solve_with_every_instance_active": len(full_units)}
    roof["whole_solve_GBps"] = solves_per_s / world * (b_in + b_out) / 1e9
    roof["whole_solve_frac"] = roof["whole_solve_GBps"] / HBM_PEAK_GBS
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
                   "kernel": gb.kernel_name, "commit": git_head()},
        "ipm": {"mean_iter": mean_iter, "max_iter": max_iter, "failures": failures, "max_kkt_residual": res_max,
                "max_kkt_residual_independent": res_indep,
                "max_rel_primal_err_vs_oracle": err, "oracle_checked_instances": args.check,
                "launches_per_step": int(gb.scalar("launches")),
                # the structural waste of one instance per lane: a wave streams its tiles until its LAST lane has converged.
                # iter_hist[j] = instances (this rank) that took j iterations; wave_max_iter_mean = mean over the 64-instance tiles
                # of the tile's largest count (what every sweep's traffic is proportional to) against mean_iter
                "iter_hist": np.bincount(iters).tolist(),
                "wave_max_iter_mean": float(iters[:(iters.size // 64) * 64].reshape(-1, 64).max(axis=1).mean()) if iters.size >= 64 else float(iters.max())},
        "roofline": roof,
        "pack_s": t_pack,
        "hbm_bytes_per_gpu": gb.bytes,
        "gather_ms": None,
        "gather": None,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(data, N, min(args.cpu_sample, B))
        if args.check > 0:
            # max relative primal error of the GPU solution against the oracle over the whole CPU sample
            xs = [gb.get("x", k) for k in range(N + 1)]
            us = [gb.get("u", k) for k in range(N)]
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
        cpu_baseline.solved = None
    if world == 1 and args.check_configs and not args.no_configs:
        idx = np.unique(np.linspace(0, B - 1, args.check_configs).astype(int))
        oe = oracle_error(gb, lambda i: lqr_instance_qp(data, i, N), idx, N)
        out["ipm"]["oracle_check"] = oe
        out["ipm"]["max_rel_primal_err_vs_oracle"] = max(out["ipm"]["max_rel_primal_err_vs_oracle"] or 0.0, oe["same_tol_max"])
        out["ipm"]["oracle_checked_instances"] = int(out["ipm"]["oracle_checked_instances"]) + int(idx.size)
    if world == 1 and not args.no_configs:
        out["configs"] = other_configs(gb, data, args)
        out["configs"]["boundary_C3"] = boundary_c3(args.boundary_n)
    gather, gerr = guarded_gather(the_gather, world)
    out["gather_ms"] = gather["ms"] if gather else None
    out["gather"] = gather if gather or not gerr else {"ranks": world, "error": gerr}
    emit(out, args)
    leave_after_stuck_gather(gerr)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
