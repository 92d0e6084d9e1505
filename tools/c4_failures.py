"""Development aid: which C4 instances do not reach status 0, and how close they get."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acados_amd import OcpQpGpuBatch
from acados_amd.generators import chain_soft_qp
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
gb = OcpQpGpuBatch.from_qps([chain_soft_qp(i, N=40) for i in range(B)])
gb.solve()
st, it = gb.info("status"), gb.info("iter")
print(gb.kernel_name, "failures", int((st != 0).sum()), "iter mean/max", it.mean(), it.max())
for i in np.nonzero(st != 0)[0]:
    print(i, "status", st[i], "iter", it[i], {f: float(gb.info(f)[i]) for f in ("res_stat", "res_eq", "res_ineq", "res_comp", "mu")})
print("iters histogram", np.bincount(it)[:51])
