#!/usr/bin/env python3
"""What the opt-in terminal polishing step (option "polish", gpu_batch.hip polish_pass) buys and costs, on the BASELINE batches C2
(65,536) and C4 (16,384): for every (iterations, polish_ratio, polish_min) variant the rate and the distance to THE solution (oracle at
complementarity 1e-12, solved once per configuration for a 1,024-instance sample) -- beside the plain exit and the tighter-tolerance
exits it competes with.  Run on the GPU box:  python tools/polish_sweep.py > gpurun_out/r06_polish_sweep.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import TIGHT, threads_allowed, tol_setup  # noqa: E402


def reference(qp_of, idx):
    from oracle.oracle import OracleQp, default_opts, solve_batch_handles
    qps = [OracleQp(qp_of(int(i))) for i in idx]
    st = solve_batch_handles([q.h.value for q in qps], default_opts(**TIGHT), nthreads=threads_allowed())
    assert (st == 0).all()
    return qps


def dist(gb, qps, idx, N):
    xs = [gb.get("x", k) for k in range(N + 1)]
    us = [gb.get("u", k) for k in range(N)]
    e = np.zeros(len(qps))
    for j, (i, o) in enumerate(zip(idx, qps)):
        for k in range(N + 1):
            r = o.get(k, "x")
            e[j] = max(e[j], float(np.max(np.abs(xs[k][i][:r.size] - r) / np.maximum(1.0, np.abs(r)))))
            if k < N:
                r = o.get(k, "u")
                e[j] = max(e[j], float(np.max(np.abs(us[k][i][:r.size] - r) / np.maximum(1.0, np.abs(r)))))
    return e


def run(gb, label, qps, idx, N, reps=3):
    gb.solve()
    t0 = time.perf_counter()
    for _ in range(reps):
        bad = gb.solve()
    dt = (time.perf_counter() - t0) / reps
    e = dist(gb, qps, idx, N)
    print(f"  {label:58s} {gb.n_batch / dt:10.4g} solves/s  {dt * 1e3:7.2f} ms  failures {bad}  kkt {gb.res_compute().max():.3e}  "
          f"polished {int(gb.scalar('polished')):6d} reverted {int(gb.scalar('polish_reverted')):4d}  dist median {np.median(e):.2e} "
          f"q99 {np.quantile(e, 0.99):.2e} max {e.max():.2e} above 1e-6: {(e > 1e-6).sum()}", flush=True)


def sweep(name, gb, qp_of, N, soft):
    idx = np.unique(np.linspace(0, gb.n_batch - 1, 1024).astype(int))
    qps = reference(qp_of, idx)
    print(f"{name}: {gb.kernel_name}, {gb.n_batch} instances, sample {idx.size}")
    tol_setup(gb)
    run(gb, "plain exit (1e-8 x 4)", qps, idx, N)
    for it in (1, 2):
        for ratio, vmin in ((1e-3, 0.0), (1e-4, 0.0), (1e-5, 0.0), (0.0, 1e-7), (0.0, 1e-8), (0.0, 0.0)):
            gb.opts_set("polish", it)
            gb.opts_set("polish_ratio", float(ratio))
            gb.opts_set("polish_min", float(vmin))
            run(gb, f"polish {it} iteration(s), ratio {ratio:g}, min {vmin:g}", qps, idx, N)
    gb.opts_set("polish", 0)
    for tc in (1e-9, 1e-10, 1e-11):
        if soft:
            gb.opts_set("tol_comp_soft_scale", tc / 1e-8)
        else:
            gb.opts_set("tol_comp", tc)
        run(gb, f"complementarity exit at {tc:g}", qps, idx, N)


def main():
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import (chain_soft_batch, chain_soft_dims, chain_soft_instance_qp, fill_chain_soft_batch, fill_lqr_batch,
                                       lqr_dims, lqr_instance_qp, random_lqr_batch)
    N, B = 50, 65536
    data = random_lqr_batch(N=N, batch=B, seed=0)
    gb = OcpQpGpuBatch(lqr_dims(N, 8, 3), B)
    fill_lqr_batch(gb, data, N)
    sweep("C2", gb, lambda i: lqr_instance_qp(data, i, N), N, False)
    del gb, data
    N4, B4 = 40, 16384
    d4 = chain_soft_batch(N=N4, batch=B4, seed=1)
    g4 = OcpQpGpuBatch(chain_soft_dims(N4), B4)
    fill_chain_soft_batch(g4, d4, N4)
    sweep("C4", g4, lambda i: chain_soft_instance_qp(d4, i, N4), N4, True)


if __name__ == "__main__":
    main()
