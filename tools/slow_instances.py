"""Which instances make the tail of a class, and what do their iterations look like?  nx=24 nu=6 N=100 (the longest C5
class): iteration histogram, and the per-iteration statistics (step lengths, sigma, mu, residual norms) of the slowest of
the instances the library keeps statistics for.   python tools/slow_instances.py [nx nu N batch]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acados_amd import OcpQpGpuBatch
from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch

nx, nu, N, B = (int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (24, 6, 100, 7281)))
data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=B, seed=200)
g = OcpQpGpuBatch(lqr_dims(N, nx, nu), B)
fill_lqr_batch(g, data, N)
for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
    g.opts_set(f, 1e-8)
g.solve()
it = g.info("iter")
h = np.bincount(it)
print("iteration histogram:", {int(i): int(c) for i, c in enumerate(h) if c})
print(f"mean {it.mean():.2f}, instances above mean + 5: {(it > it.mean() + 5).sum()}")
ns = 0
for i in range(256):
    try:
        g.stat(i); ns = i + 1
    except Exception:
        break
cand = np.argsort(-it[:ns])[:2]
np.set_printoptions(linewidth=200, precision=3, suppress=False)
for i in cand:
    st = g.stat(int(i))
    print(f"instance {i}: {it[i]} iterations; columns: alpha_aff(prim, dual) mu_aff sigma alpha(prim, dual) mu res_g res_b res_d res_m")
    for r in range(min(len(st), it[i] + 1)):
        print("  %2d  %.3f %.3f  %.2e %.2e  %.3f %.3f  %.2e  %.1e %.1e %.1e %.1e" % ((r,) + tuple(st[r, :11])))
