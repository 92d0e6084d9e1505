"""Instances still iterating after every IPM iteration of the C2 batch, and the time of every iteration."""
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
if len(sys.argv) > 2: gb.opts_set("tail_max", int(sys.argv[2]))
gb.solve()
ts = [gb.solve() * 0 + gb.scalar("time_tot") for _ in range(3)]
print("time_tot over 3 solves (ms):", ["%.2f" % (t * 1e3) for t in ts], "tail switches", gb.scalar("tail_switches"))
it = gb.info("iter")
h = np.bincount(it, minlength=it.max() + 1)
act = B - np.cumsum(h)
print("kernel", gb.kernel_name, "time_tot %.2f ms" % (gb.scalar("time_tot") * 1e3))
for k in range(len(h)):
    print(f"finished at iter {k:2d}: {h[k]:6d}   still active afterwards: {act[k]:6d} ({100.0 * act[k] / B:5.1f} %)")
