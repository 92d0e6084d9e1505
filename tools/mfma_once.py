"""One solve of the C4 batch and of the nx=24 nu=6 N=50 class (7,281 instances) for profiler passes (matrix-pipe counters of
kt_factor<24,3,4> / kt_factor<24,6>): python tools/mfma_once.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acados_amd import OcpQpGpuBatch
from acados_amd.generators import chain_soft_batch, chain_soft_dims, fill_chain_soft_batch, fill_lqr_batch, lqr_dims, random_lqr_batch

d = chain_soft_batch(N=40, batch=16384, seed=1)
g = OcpQpGpuBatch(chain_soft_dims(40), 16384)
fill_chain_soft_batch(g, d, 40)
for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"): g.opts_set(f, 1e-8)
print("C4 failures", g.solve(), g.kernel_name, int(g.scalar("w16_tiles")))
del g
d = random_lqr_batch(N=50, nx=24, nu=6, batch=7281, seed=200)
g = OcpQpGpuBatch(lqr_dims(50, 24, 6), 7281)
fill_lqr_batch(g, d, 50)
for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"): g.opts_set(f, 1e-8)
print("nx=24 failures", g.solve(), g.kernel_name, int(g.scalar("w16_tiles")))
