"""C4 against the oracle at two tolerances: is the 4.6e-6 of the 1,024-instance sample at tol 1e-8 the size of the KKT ball of
this class (soft rows with Z = 1e2 leave the primal solution flat) or a defect of the kernels?  Both solvers are run at
1e-8 and at 1e-10 on the same 1,024 instances: if the difference shrinks with the tolerance, both converge to the same
point.  python tools/c4_tolerance_ball.py [instances]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acados_amd import OcpQpGpuBatch
from acados_amd.generators import chain_soft_batch, chain_soft_dims, chain_soft_instance_qp, fill_chain_soft_batch
from oracle.oracle import OracleQp, default_opts, solve_batch_handles

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = 40
data = chain_soft_batch(N=N, batch=B, seed=1)
for tol in (1e-8, 1e-10):
    g = OcpQpGpuBatch(chain_soft_dims(N), B)
    fill_chain_soft_batch(g, data, N)
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"): g.opts_set(f, tol)
    g.opts_set("iter_max", 100)
    bad = g.solve()
    xs = [g.get("x", k) for k in range(N + 1)]; us = [g.get("u", k) for k in range(N)]
    qps = [OracleQp(chain_soft_instance_qp(data, i, N)) for i in range(B)]
    solve_batch_handles([q.h.value for q in qps], default_opts(tol_stat=tol, tol_eq=tol, tol_ineq=tol, tol_comp=tol, iter_max=100),
                        nthreads=min(16, os.cpu_count() or 1))
    err = np.zeros(B)
    for i, o in enumerate(qps):
        for k in range(N + 1):
            r = o.get(k, "x"); err[i] = max(err[i], float(np.max(np.abs(xs[k][i][:r.size] - r) / np.maximum(1.0, np.abs(r)))))
            if k < N:
                r = o.get(k, "u"); err[i] = max(err[i], float(np.max(np.abs(us[k][i][:r.size] - r) / np.maximum(1.0, np.abs(r)))))
    it_o = np.array([o.iter for o in qps])
    print(f"tol {tol:.0e}: device failures {bad}  KKT (independent kernel) {g.res_compute().max():.3e}  iterations device {g.info('iter').mean():.2f} oracle {it_o.mean():.2f}"
          f"  max rel primal difference {err.max():.2e}  median {np.median(err):.2e}  instances above 1e-6: {int((err > 1e-6).sum())} of {B}", flush=True)
