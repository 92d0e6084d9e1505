"""Development aid: C3 (65,536 instances, partial condensing to N2 = 10) -- or, with `c2` as the first argument, C2 itself --
solve time on the product library and on development builds of it (make variant TAG=...):
  python tools/variant_rate.py [c2] [libacados_amd_qp_<tag>.so ...]"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acados_amd import OcpQpGpuBatch, _lib
from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch

N, B = 50, 65536
C2 = len(sys.argv) > 1 and sys.argv[1] == "c2"
data = random_lqr_batch(N=N, nx=8, nu=3, batch=B, seed=3)
for name in [None] + sys.argv[(2 if C2 else 1):]:
    clib = None if name is None else _lib.bind(ctypes.CDLL(os.path.join(ROOT, "tools", "ab", name)))
    g = OcpQpGpuBatch(lqr_dims(N, 8, 3), B, _clib=clib)
    fill_lqr_batch(g, data, N)
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        g.opts_set(f, 1e-8)
    if not C2:
        g.opts_set("cond_N", 10)
    g.opts_set("profile", 1)
    bad = g.solve()
    ts = []
    g.opts_set("profile", 0)
    for _ in range(5 if C2 else 3):
        t0 = time.perf_counter(); bad = g.solve(); ts.append(time.perf_counter() - t0)
    g.scalar("prof_reset"); g.opts_set("profile", 1); g.solve()
    kkt = float(np.max(g.res_compute()))
    ms = {c: g.scalar("prof_ms_" + c) / max(g.scalar("prof_cnt_" + c), 1) for c in ("back_fact", "fwd_aff", "back_rhs", "fwd_corr")}
    print(f"{name or 'product library':34s} {'C2' if C2 else 'C3'} solve {min(ts) * 1e3:7.2f} ms  failures {bad}  kkt {kkt:.3e}  kernel {g.kernel_name if C2 else g.condensed_kernel_name()}"
          f"  per launch: fact {ms['back_fact']:.3f} faff {ms['fwd_aff']:.3f} rhs {ms['back_rhs']:.3f} fcor {ms['fwd_corr']:.3f} ms"
          f" ({int(g.scalar('prof_cnt_fwd_corr'))} launches)  condense + expand {g.scalar('time_xcond') * 1e3:.2f} ms", flush=True)
