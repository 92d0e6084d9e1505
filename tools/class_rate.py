"""Development aid: solve time and launch time per sweep of C5 classes (7,281 instances: the per-GPU share of one class)
on the product library and on development builds of it (make variant TAG=... DEFS=...); KX_CLASSES="nx,nu,N;..."
selects the classes, ACADOS_AMD_WPI / ACADOS_AMD_WPI_BATCH_MAX the kernel family:
  python tools/kx_prefetch_rate.py [libacados_amd_qp_<tag>.so ...]"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acados_amd import OcpQpGpuBatch, _lib
from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch

B = 7281
CLASSES = (tuple(tuple(int(v) for v in c.split(",")) for c in os.environ["KX_CLASSES"].split(";")) if os.environ.get("KX_CLASSES")
           else ((4, 1, 20), (4, 1, 100), (8, 3, 50), (12, 3, 100)))
datas = {c: random_lqr_batch(N=c[2], nx=c[0], nu=c[1], batch=B, seed=200) for c in CLASSES}
for name in [None] + sys.argv[1:]:
    clib = None if name is None else _lib.bind(ctypes.CDLL(os.path.join(ROOT, "tools", "ab", name)))
    for (nx, nu, N) in CLASSES:
        g = OcpQpGpuBatch(lqr_dims(N, nx, nu), B, _clib=clib)
        fill_lqr_batch(g, datas[(nx, nu, N)], N)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            g.opts_set(f, 1e-8)
        g.solve()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); bad = g.solve(); ts.append(time.perf_counter() - t0)
        it = g.info("iter")
        x = np.array(g.get_all("ux")) if hasattr(g, "get_all") else None
        g.scalar("prof_reset"); g.opts_set("profile", 1); g.solve(); g.opts_set("profile", 0)
        ms = {c: g.scalar("prof_ms_" + c) / max(g.scalar("prof_cnt_" + c), 1) for c in ("back_fact", "fwd_aff", "back_rhs", "fwd_corr")}
        kkt = float(np.max(g.res_compute()))
        print(f"{name or 'product library':30s} nx={nx:2d} nu={nu} N={N:3d} {g.kernel_name:20s} solve {min(ts) * 1e3:7.2f} ms  {B / min(ts):9.0f} /s  iters {it.sum()}"
              f"  failures {bad}  kkt {kkt:.2e}  launch us: fact {ms['back_fact'] * 1e3:.0f} faff {ms['fwd_aff'] * 1e3:.0f} rhs {ms['back_rhs'] * 1e3:.0f}"
              f" fcor {ms['fwd_corr'] * 1e3:.0f}", flush=True)
        del g
