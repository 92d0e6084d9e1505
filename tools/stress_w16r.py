"""Repeatability stress of the LDS-DMA kernels: the C3 path, an nx=24 nu=6 class and the C4 class solved `reps` times each;
every repeat must give status 0, KKT residuals (independent kernel) within tolerance and BIT-IDENTICAL solutions and iteration
counts (the placement of an instance in the dense list of live instances varies from run to run, its arithmetic must not).
Usage: python tools/stress_w16r.py [reps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acados_amd import OcpQpGpuBatch
from acados_amd.generators import (chain_soft_batch, chain_soft_dims, fill_chain_soft_batch, fill_lqr_batch, lqr_dims,
                                   random_lqr_batch)

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def lqr(nx, nu, N, B, cond=0):
    d = random_lqr_batch(N=N, nx=nx, nu=nu, batch=B, seed=3)
    g = OcpQpGpuBatch(lqr_dims(N, nx, nu), B)
    fill_lqr_batch(g, d, N)
    if cond:
        g.opts_set("cond_N", cond)
    return g, N


def c4(B):
    d = chain_soft_batch(N=40, batch=B, seed=1)
    g = OcpQpGpuBatch(chain_soft_dims(40), B)
    fill_chain_soft_batch(g, d, 40)
    return g, 40


for tag, (g, N), tol in (("C3 65536", lqr(8, 3, 50, 65536, 10), 2e-8), ("nx=24 nu=6 N=50 x 16384", lqr(24, 6, 50, 16384), 1.002e-8),
                         ("C4 16384", c4(16384), 1.002e-8)):
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        g.opts_set(f, 1e-8)
    ref = None
    for r in range(reps):
        bad = g.solve()
        kkt = float(np.max(g.res_compute()))
        sol = (np.concatenate([g.get("x", k).ravel() for k in (0, N // 2, N)] + [g.get("u", k).ravel() for k in (0, N - 1)]), g.info("iter").copy())
        if ref is None:
            ref = sol
        same = np.array_equal(ref[0], sol[0]) and np.array_equal(ref[1], sol[1])
        print(f"{tag:26s} {g.condensed_kernel_name() or g.kernel_name:30s} rep {r}: failures {bad} kkt {kkt:.3e} identical {same}", flush=True)
        assert bad == 0 and kkt <= tol and same
print("stress passed")
