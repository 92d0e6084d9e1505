"""Full condensing on the device (cond_N = 1: one block, the reference's FULL_CONDENSING_* path) where the condensed stage
fits the condensing / IPM kernels (nx + N nu <= 64 columns): C2-shaped QPs (nx = 8, nu = 3, input box, x0 fixed) with
N = 6 ... 18, solve rate against the uncondensed run of the same batch, KKT residuals of the ORIGINAL QP at the expanded
point, largest difference of the two solutions; N = 19 (65 columns) shows the refusal.  python tools/full_condensing_rate.py [batch]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acados_amd import OcpQpGpuBatch
from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
for N in (6, 10, 14, 18, 19):
    data = random_lqr_batch(N=N, nx=8, nu=3, batch=B, seed=11)
    sol = {}
    for cn in (N, 1):
        g = OcpQpGpuBatch(lqr_dims(N, 8, 3), B)
        fill_lqr_batch(g, data, N)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"): g.opts_set(f, 1e-8)
        g.opts_set("cond_N", cn)
        g.solve()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); bad = g.solve(); ts.append(time.perf_counter() - t0)
        active = int(g.scalar("cond_N_active"))
        sol[cn] = np.concatenate([g.get("u", k) for k in range(N)] + [g.get("x", k) for k in range(N + 1)], axis=1)
        print(f"N {N:2d}  nx + N nu = {8 + 3 * N:2d} columns  requested N2 {cn:2d}  active N2 {active:2d}  kernel {(g.condensed_kernel_name() or g.kernel_name):28s}"
              f" {min(ts) * 1e3:7.2f} ms  {B / min(ts):9.0f} solves/s  iters {g.info('iter').mean():.2f}  failures {bad}"
              f"  KKT (original QP) {g.res_compute().max():.3e}  condense + expand {g.scalar('time_xcond') * 1e3:.2f} ms", flush=True)
        del g
    print(f"      max |full-space - condensed| over u, x: {np.max(np.abs(sol[N] - sol[1])):.2e}", flush=True)
