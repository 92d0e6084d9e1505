"""The cliff next to the box fast path: the C2 shape (N=50, nx=8, nu=3) with SOFT state bounds (one slack per state
row, stages 1..N) leaves the box kernels for the general wave-per-instance ones.  Rate of both at the same batch."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acados_amd import OcpQpGpuBatch
from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch

N, nx, nu = 50, 8, 3
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=B, seed=0)
for soft in (0, 1):
    d = lqr_dims(N, nx, nu)
    d.nbx[1:] = nx
    d.nb[:] = d.nbu + d.nbx
    if soft:
        d.ns[1:] = nx
    gb = OcpQpGpuBatch(d, B)
    if soft:
        for k in range(1, N + 1):
            nbu = int(d.nbu[k])
            gb.set_int("idxs_rev", k, np.concatenate([-np.ones(nbu, dtype=int), np.arange(nx)]))
    fill_lqr_batch(gb, data, N)
    for k in range(1, N + 1):
        gb.set("lbx", k, np.full((B, nx), -8.0)); gb.set("ubx", k, np.full((B, nx), 8.0))
        if soft:
            for f, v in (("Zl", 1e2), ("Zu", 1e2), ("zl", 1e1), ("zu", 1e1), ("lls", 0.0), ("lus", 0.0)):
                gb.set(f, k, np.full((B, nx), v))
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        gb.opts_set(f, 1e-8)
    bad = gb.solve()
    t0 = time.perf_counter(); bad = gb.solve(); dt = time.perf_counter() - t0
    it = gb.info("iter")
    gb.scalar("prof_reset"); gb.opts_set("profile", 1); gb.solve(); gb.opts_set("profile", 0)
    cls = ("back_fact", "fwd_aff", "back_rhs", "fwd_corr")
    print("    avg ms per launch: " + "  ".join(f"{c} {gb.scalar('prof_ms_' + c) / max(gb.scalar('prof_cnt_' + c), 1):.3f}" for c in cls))
    print(f"{'soft' if soft else 'hard'} state bounds: kernel {gb.kernel_name:40s} batch {B}  {dt*1e3:8.1f} ms  {B/dt:10.0f} solves/s  "
          f"iters {it.mean():.1f}/{it.max()}  failures {bad}", flush=True)
