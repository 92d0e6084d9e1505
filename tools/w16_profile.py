"""Per-sweep launch times of the 16-lanes-per-instance kernels on the C2 shape against batch size."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ACADOS_AMD_WPI"] = "1"
from acados_amd import OcpQpGpuBatch
from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch
N, nx, nu = 50, 8, 3
for B in (4, 64, 1024, 4096, 16384):
    data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=B, seed=0)
    gb = OcpQpGpuBatch(lqr_dims(N, nx, nu), B)
    fill_lqr_batch(gb, data, N)
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"): gb.opts_set(f, 1e-8)
    gb.solve()
    gb.scalar("prof_reset"); gb.opts_set("profile", 1); gb.solve(); gb.opts_set("profile", 0)
    cls = ("back_fact", "fwd_aff", "back_rhs", "fwd_corr")
    print(f"{gb.kernel_name} batch {B:6d}: " + "  ".join(f"{c} {gb.scalar('prof_ms_' + c) / max(gb.scalar('prof_cnt_' + c), 1) * 1e3:8.1f} us" for c in cls))
