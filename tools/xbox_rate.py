"""State-bounded variant of the C2 shape (box rows on all states at every stage, the mass-spring class): rate of the
two kernel families against batch size."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acados_amd import OcpQpGpuBatch
from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch
N, nx, nu = 50, 8, 3
BND = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
for B in (4096, 16384, 65536, 131072):
    data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=B, seed=0)
    row = []
    for fam in ("0", "1", None):
        if fam is None:
            os.environ.pop("ACADOS_AMD_WPI")
        else:
            os.environ["ACADOS_AMD_WPI"] = fam
        d = lqr_dims(N, nx, nu)
        d.nbx[:] = nx
        d.nb[:] = d.nbu + d.nbx
        gb = OcpQpGpuBatch(d, B)
        fill_lqr_batch(gb, data, N)
        for k in range(1, N + 1):
            gb.set("lbx", k, np.full((B, nx), -BND)); gb.set("ubx", k, np.full((B, nx), BND))
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"): gb.opts_set(f, 1e-8)
        bad = gb.solve()
        t = min(gb.solve() * 0 + gb.scalar("time_tot") for _ in range(2))
        row.append((gb.kernel_name.split("(")[0].split("<")[0] + ("/XBOX" if "XBOX=1" in gb.kernel_name else ""), t, bad))
    print(f"batch {B:6d}: " + "   ".join(f"{n_} {t*1e3:8.2f} ms ({B/t:9.0f}/s) fail {bad}" for n_, t, bad in row))
