"""Repeatability stress of the LDS-DMA kernels: the C3 path, an nx=24 nu=6 class and the C4 class solved `reps` times each;
every repeat must give status 0, KKT residuals (independent kernel) within tolerance and BIT-IDENTICAL solutions and iteration
counts (the placement of an instance in the dense list of live instances varies from run to run, its arithmetic must not).
Usage: python tools/stress_w16r.py [reps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acados_amd import OcpQpGpuBatch
from acados_amd.generators import (chain_soft_batch, chain_soft_dims, fill_chain_soft_batch, fill_lqr_batch, lqr_dims,
                                   random_lqr_batch)

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
# a development build of the library (make variant TAG=...): python tools/stress_w16r.py 3 libacados_amd_qp_<tag>.so
CLIB = None
if len(sys.argv) > 2:
    import ctypes
    from acados_amd import _lib
    CLIB = _lib.bind(ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab", sys.argv[2])))


def lqr(nx, nu, N, B, cond=0):
    d = random_lqr_batch(N=N, nx=nx, nu=nu, batch=B, seed=3)
    g = OcpQpGpuBatch(lqr_dims(N, nx, nu), B, _clib=CLIB)
    fill_lqr_batch(g, d, N)
    if cond:
        g.opts_set("cond_N", cond)
    return g, N


def c4(B):
    d = chain_soft_batch(N=40, batch=B, seed=1)
    g = OcpQpGpuBatch(chain_soft_dims(40), B, _clib=CLIB)
    fill_chain_soft_batch(g, d, 40)
    return g, 40


CASES = (("C3 65536", lambda: lqr(8, 3, 50, 65536, 10), 2e-8), ("nx=24 nu=6 N=50 x 16384", lambda: lqr(24, 6, 50, 16384), 1.002e-8),
         ("C4 16384", lambda: c4(16384), 1.002e-8))
for tag, mk, tol in (CASES if CLIB is None else CASES[:1]):
    g, N = mk()
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        g.opts_set(f, 1e-8)
    ref = None
    for r in range(reps):
        bad = g.solve()
        kkt = float(np.max(g.res_compute()))
        sol = (np.concatenate([g.get("x", k).ravel() for k in (0, N // 2, N)] + [g.get("u", k).ravel() for k in (0, N - 1)]), g.info("iter").copy())
        if ref is None:
            ref = sol
        same = np.array_equal(ref[0], sol[0]) and np.array_equal(ref[1], sol[1])
        print(f"{tag:26s} {g.condensed_kernel_name() or g.kernel_name:30s} rep {r}: failures {bad} kkt {kkt:.3e} identical {same}", flush=True)
        if CLIB is not None and tag.startswith("C3"):
            # development build: is the QP data in HBM still what was packed?  (stray stores show up here)
            d0 = random_lqr_batch(N=50, nx=8, nu=3, batch=65536, seed=3)
            dA = np.abs(g.get("A", 7).reshape(65536, 8, 8).transpose(0, 2, 1) - d0["A"]).max(axis=(1, 2))
            dQ = np.abs(g.get("Q", 31).reshape(65536, 8, 8) - d0["Q"]).max(axis=(1, 2))
            st = g.info("status")
            print(f"    instances with changed A[7] {int((dA > 0).sum())}, changed Q[31] {int((dQ > 0).sum())}, status != 0 {int((st != 0).sum())},"
                  f" first failing instances {np.nonzero(st)[0][:12].tolist()}", flush=True)
        if CLIB is None:
            assert bad == 0 and kkt <= tol and same
print("stress passed" if CLIB is None else "done (development build: nothing asserted)")
