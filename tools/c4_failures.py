"""Development aid: which C4 instances (batch generator) do not reach status 0, and how close they get."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acados_amd import OcpQpGpuBatch
from acados_amd.generators import chain_soft_batch, chain_soft_dims, fill_chain_soft_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N = 40
data = chain_soft_batch(N=N, batch=B, seed=1)
gb = OcpQpGpuBatch(chain_soft_dims(N), B)
fill_chain_soft_batch(gb, data, N)
for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"): gb.opts_set(f, 1e-8)
gb.solve()
st, it = gb.info("status"), gb.info("iter")
print(gb.kernel_name, "failures", int((st != 0).sum()), "iter mean/max", it.mean(), it.max())
for i in np.nonzero(st != 0)[0]:
    print(i, "status", st[i], "iter", it[i], {f: float(gb.info(f)[i]) for f in ("res_stat", "res_eq", "res_ineq", "res_comp", "mu")})
print("iters histogram", np.bincount(it)[:51])
