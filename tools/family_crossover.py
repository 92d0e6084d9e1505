"""Solve time of the two kernel families (one instance per lane / one wave per instance) against batch size
on the C2 shape (N=50, nx=8, nu=3): where the dispatch rule should switch."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acados_amd import OcpQpGpuBatch
from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch

N, nx, nu = 50, 8, 3
for B in ([int(a) for a in sys.argv[1:]] or [1, 16, 64, 256, 1024, 4096, 16384]):
    data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=B, seed=0)
    row = []
    for fam in ("0", "1"):
        os.environ["ACADOS_AMD_WPI"] = fam
        gb = OcpQpGpuBatch(lqr_dims(N, nx, nu), B)
        fill_lqr_batch(gb, data, N)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"): gb.opts_set(f, 1e-8)
        gb.solve()
        t = min(gb.solve() * 0 + gb.scalar("time_tot") for _ in range(3))
        row.append((gb.kernel_name.split("(")[0].split("<")[0], t))
    print(f"batch {B:6d}: " + "   ".join(f"{n_} {t*1e3:8.3f} ms ({B/t:9.0f}/s)" for n_, t in row))
