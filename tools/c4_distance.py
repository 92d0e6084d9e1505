"""C4: how far from the SOLUTION does the device stop, and what does it cost to get closer?  The device is run at the
acados tolerances (1e-8 x 4) with tol_comp_soft_scale in {1, 1e-3 (default), 1e-4}; every result is compared with (a) the
oracle at the same effective tolerances and (b) the oracle at a tight tolerance (complementarity 1e-12; within ~1e-11 of
the exact solution: tests/dense_ref.py::solve_exact).  python tools/c4_distance.py [instances]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acados_amd import OcpQpGpuBatch
from acados_amd.generators import chain_soft_batch, chain_soft_dims, chain_soft_instance_qp, fill_chain_soft_batch
from oracle.oracle import OracleQp, default_opts, solve_batch_handles

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
BT = 16384
N = 40
data = chain_soft_batch(N=N, batch=BT, seed=1)
idx = np.unique(np.linspace(0, BT - 1, B).astype(int))
qps = [OracleQp(chain_soft_instance_qp(data, int(i), N)) for i in idx]
hs = [q.h.value for q in qps]
nth = min(16, os.cpu_count() or 1)


def osol():
    return np.stack([np.concatenate([o.get(k, "x") for k in range(N + 1)] + [o.get(k, "u") for k in range(N)]) for o in qps])


st = solve_batch_handles(hs, default_opts(tol_stat=1e-9, tol_eq=1e-11, tol_ineq=1e-11, tol_comp=1e-12, iter_max=100), nthreads=nth)
tight = osol()
print(f"tight reference: {int((st != 0).sum())} of {len(qps)} not converged, mean iterations {np.mean([q.iter for q in qps]):.2f}", flush=True)
g = OcpQpGpuBatch(chain_soft_dims(N), BT)
fill_chain_soft_batch(g, data, N)
for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
    g.opts_set(f, 1e-8)
for scale in (1.0, 1e-3, 1e-4):
    g.opts_set("tol_comp_soft_scale", scale)
    g.opts_set("iter_max", 50)
    g.solve()
    t0 = time.perf_counter(); bad = g.solve(); dt = time.perf_counter() - t0
    dev = np.concatenate([g.get("x", k)[idx] for k in range(N + 1)] + [g.get("u", k)[idx] for k in range(N)], axis=1)
    it = g.info("iter")
    solve_batch_handles(hs, default_opts(tol_stat=1e-8, tol_eq=1e-8, tol_ineq=1e-8, tol_comp=1e-8 * scale, iter_max=50), nthreads=nth)
    same = osol()
    rel = lambda a, r: np.max(np.abs(a - r) / np.maximum(1.0, np.abs(r)), axis=1)
    es, et = rel(dev, same), rel(dev, tight)
    print(f"scale {scale:.0e}: {BT / dt:.4e} solves/s ({dt * 1e3:.1f} ms), failures {bad}, iterations mean {it.mean():.2f} max {it.max()}, "
          f"KKT (independent kernel) {g.res_compute().max():.3e} | vs oracle at the same tolerance: max {es.max():.2e} median {np.median(es):.2e} "
          f"above 1e-6: {int((es > 1e-6).sum())} | distance to the solution: max {et.max():.2e} q99 {np.quantile(et, 0.99):.2e} "
          f"median {np.median(et):.2e} above 1e-6: {int((et > 1e-6).sum())} of {len(idx)}", flush=True)
