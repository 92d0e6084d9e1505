"""A/B of the step update of the sixteen-lanes families: a launch of its own (k_step_update, the default) against the pass at the end of
the corrector sweep (ACADOS_AMD_EXT_UPDATE=0), same batches, same data:   python tools/ext_update_ab.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acados_amd import OcpQpGpuBatch
from acados_amd.generators import (chain_soft_batch, chain_soft_dims, fill_chain_soft_batch, fill_lqr_batch, lqr_dims, random_lqr_batch)


def tols(g):
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        g.opts_set(f, 1e-8)


def ab(name, g):
    out = {}
    for mode in ("1", "0", "1", "0"):
        os.environ["ACADOS_AMD_EXT_UPDATE"] = mode
        g.solve()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); bad = g.solve(); ts.append(time.perf_counter() - t0)
        out.setdefault(mode, []).append(min(ts) * 1e3)
        it = g.info("iter").copy()
        out.setdefault("it" + mode, it)
        out.setdefault("x" + mode, g.get("x", 1).copy())
    same = bool(np.array_equal(out["it1"], out["it0"])) and float(np.max(np.abs(out["x1"] - out["x0"])))
    print(f"{name:44s} own launch {min(out['1']):8.2f} ms   pass in the sweep {min(out['0']):8.2f} ms   ratio {min(out['0']) / min(out['1']):.3f}   "
          f"kernel {g.condensed_kernel_name() or g.kernel_name}  failures {bad}  max |dx| between the two {same}", flush=True)


N = 50
d = random_lqr_batch(N=N, nx=8, nu=3, batch=65536, seed=0)
g = OcpQpGpuBatch(lqr_dims(N, 8, 3), 65536); fill_lqr_batch(g, d, N); tols(g); g.opts_set("cond_N", 10)
ab("C3 (C2 data, N2 = 10), 65,536", g); del g, d
d = chain_soft_batch(N=40, batch=16384, seed=1)
g = OcpQpGpuBatch(chain_soft_dims(40), 16384); fill_chain_soft_batch(g, d, 40); tols(g)
ab("C4 chain nx=24 nu=3 soft, 16,384", g); del g, d
for nx, nu, Nc in ((24, 6, 100), (24, 6, 20), (12, 3, 100), (12, 3, 20)):
    d = random_lqr_batch(N=Nc, nx=nx, nu=nu, batch=7281, seed=200)
    g = OcpQpGpuBatch(lqr_dims(Nc, nx, nu), 7281); fill_lqr_batch(g, d, Nc); tols(g)
    ab(f"C5 class nx={nx} nu={nu} N={Nc}, 7,281", g); del g, d
