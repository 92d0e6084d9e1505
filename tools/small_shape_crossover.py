"""nx = 4, nu = 1 (the smallest C5 shape): the sixteen-lanes kernels (ACADOS_AMD_WPI=1) against the pipelined
one-instance-per-lane kernels (ACADOS_AMD_WPI=0, ipm_kernels_box_small.hpp) over the batch size, with bounds on the
inputs only and with bounds on every state too.  The dispatch threshold of gpu_batch.hip comes from here."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acados_amd import OcpQpGpuBatch
from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch
nx, nu = 4, 1
for N in (20, 100):
    for xbox in (False, True):
        for B in (256, 1024, 2048, 4096, 7281, 16384, 65536):
            data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=B, seed=0)
            row = []
            for fam in ("1", "0"):
                os.environ["ACADOS_AMD_WPI"] = fam
                d = lqr_dims(N, nx, nu)
                if xbox:
                    d.nbx[1:] = nx
                    d.nb[:] = d.nbu + d.nbx
                gb = OcpQpGpuBatch(d, B)
                fill_lqr_batch(gb, data, N)
                if xbox:
                    for k in range(1, N + 1):
                        gb.set("lbx", k, np.full((B, nx), -4.0)); gb.set("ubx", k, np.full((B, nx), 4.0))
                for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"): gb.opts_set(f, 1e-8)
                bad = gb.solve()
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter(); gb.solve(); ts.append(time.perf_counter() - t0)
                row.append((gb.kernel_name.split("<")[0] + ("/XBOX" if "XBOX=1" in gb.kernel_name else ""), min(ts), bad, int(gb.info("iter").sum())))
                del gb
            print(f"N {N:3d} state bounds {int(xbox)} batch {B:6d}: " + "   ".join(f"{n_} {t*1e3:8.2f} ms ({B/t:9.0f}/s) fail {bad} iters {it}" for n_, t, bad, it in row), flush=True)
