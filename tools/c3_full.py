"""C3 at the BASELINE size: C2 data, batch 65,536, partial condensing N2=10; time split condense / solve / expand."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acados_amd import OcpQpGpuBatch
from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = 50
data = random_lqr_batch(N=N, batch=B, seed=0)
gb = OcpQpGpuBatch(lqr_dims(N, 8, 3), B)
fill_lqr_batch(gb, data, N)
for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"): gb.opts_set(f, 1e-8)
gb.opts_set("cond_N", 10)
gb.solve()
t0 = time.perf_counter(); bad = gb.solve(); dt = time.perf_counter() - t0
it = gb.info("iter")
print(f"C3 batch {B}: {dt*1e3:.1f} ms/solve  {B/dt:.0f} solves/s  iters {it.mean():.2f}/{it.max()}  failures {bad}  "
      f"time_tot {gb.scalar('time_tot')*1e3:.1f} ms of which condense+expand {gb.scalar('time_xcond')*1e3:.1f} ms")
for n in ("res_stat", "res_eq", "res_ineq", "res_comp"): print(n, gb.info(n).max())
