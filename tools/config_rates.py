"""Solve rates of the non-headline configurations (C3, C4, C5 classes) at moderate batch sizes.
Not bench lines: these shapes still run the rolled-loop / scratch kernels."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acados_amd import OcpQpGpuBatch
from acados_amd.generators import (chain_soft_qp, fill_lqr_batch, lqr_dims, random_lqr_batch)


def rate(gb, tag, n):
    gb.solve()
    t0 = time.perf_counter(); bad = gb.solve(); dt = time.perf_counter() - t0
    it = gb.info("iter")
    print(f"{tag:44s} kernel {gb.kernel_name:34s} batch {n:6d}  {dt*1e3:9.1f} ms/solve  {n/dt:11.0f} solves/s  iters {it.mean():.1f}/{it.max()}  failures {bad}")
    gb.scalar("prof_reset"); gb.opts_set("profile", 1); gb.solve(); gb.opts_set("profile", 0)
    cls = ("back_fact", "fwd_aff", "back_rhs", "fwd_corr")
    print("    avg ms per launch: " + "  ".join(f"{c} {gb.scalar('prof_ms_' + c) / max(gb.scalar('prof_cnt_' + c), 1):.3f}" for c in cls))


B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for (nx, nu, N) in ((4, 1, 50), (12, 3, 50), (24, 6, 50)):
    data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=B, seed=1)
    gb = OcpQpGpuBatch(lqr_dims(N, nx, nu), B)
    fill_lqr_batch(gb, data, N)
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"): gb.opts_set(f, 1e-8)
    rate(gb, f"C5 class nx={nx} nu={nu} N={N}", B)
data = random_lqr_batch(N=50, batch=B, seed=0)
gb = OcpQpGpuBatch(lqr_dims(50, 8, 3), B)
fill_lqr_batch(gb, data, 50)
for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"): gb.opts_set(f, 1e-8)
gb.opts_set("cond_N", 10)
rate(gb, "C3 = C2 data, partial condensing N2=10", B)
nb = min(B, 4096)
qps = [chain_soft_qp(i, N=40) for i in range(nb)]
gb = OcpQpGpuBatch.from_qps(qps)
rate(gb, "C4 chain nx=24 nu=3 ng=4 ns=8 N=40", nb)
