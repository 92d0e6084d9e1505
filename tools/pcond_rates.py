"""Partial condensing beyond the box class: solve rate of the C4 class (soft state bounds + soft general rows) and
of the mass-spring unit-test QP (state bounds at every stage) for several N2, full-space run beside it."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acados_amd import OcpQpGpuBatch
from acados_amd.generators import chain_soft_batch, chain_soft_dims, fill_chain_soft_batch, mass_spring_qp

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096


def rate(gb, tag, cn):
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"): gb.opts_set(f, 1e-8)
    gb.opts_set("cond_N", cn)
    gb.solve()
    t0 = time.perf_counter(); bad = gb.solve(); dt = time.perf_counter() - t0
    it = gb.info("iter")
    print(f"{tag:34s} N2 {int(gb.scalar('cond_N_active')):3d}  {dt*1e3:8.1f} ms  {B/dt:10.0f} solves/s  iters {it.mean():.1f}/{it.max()}  "
          f"failures {bad}  condense+expand {gb.scalar('time_xcond')*1e3:.2f} ms", flush=True)


N = 40
data = chain_soft_batch(N=N, batch=B, seed=1)
for cn in (40, 20):
    gb = OcpQpGpuBatch(chain_soft_dims(N), B)
    fill_chain_soft_batch(gb, data, N)
    rate(gb, "C4 class nx=24 nu=3 ng=4 ns=8 N=40", cn)
    del gb
qp = mass_spring_qp(N=15)
for cn in (15, 5, 3):
    gb = OcpQpGpuBatch.from_qps([qp] * B)
    rate(gb, "mass-spring nx=8 nu=3 nb=11 N=15", cn)
    del gb
