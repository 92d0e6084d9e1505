"""Factor sweep of the two-rows family on 4 x 4 MFMA tiles (kt_factor, ACADOS_AMD_W16T=1, the default) against the register-row
sweep (ky_factor, ACADOS_AMD_W16T=0), same box, same data: C3 (C2 data condensed to N2 = 10: the <8,15> shape at 65,536) and
C5 classes of the <24,6> shape at 7,281 instances.  Reports the whole solve, the launch time of every sweep, the iteration
sum and the largest difference between the two solutions.   python tools/w16t_ab.py [c3] [24,6,20] [24,6,100] ..."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acados_amd import OcpQpGpuBatch, _lib
from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch

# development builds of the library to run beside the product (make variant TAG=... DEFS=...): libacados_amd_qp_<tag>.so
extra = [a for a in sys.argv[1:] if a.endswith(".so")]
cases = [a for a in sys.argv[1:] if not a.endswith(".so")] or ["c3", "24,6,20", "24,6,100"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = {name: _lib.bind(ctypes.CDLL(os.path.join(ROOT, "tools", "ab", name))) for name in extra}
for case in cases:
    if case == "c3":
        nx, nu, N, B, cond = 8, 3, 50, 65536, 10
        data = random_lqr_batch(N=N, batch=B, seed=0)
    else:
        nx, nu, N = (int(v) for v in case.split(","))
        B, cond = 7281, 0
        data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=B, seed=200)
    ref = None
    for t, ln in [("1", None), ("0", None)] + [("1", n_) for n_ in extra] + [("1", None), ("0", None)] + [("1", n_) for n_ in extra]:
        os.environ["ACADOS_AMD_W16T"] = t
        g = OcpQpGpuBatch(lqr_dims(N, nx, nu), B, _clib=libs[ln] if ln else None)
        fill_lqr_batch(g, data, N)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            g.opts_set(f, 1e-8)
        if cond: g.opts_set("cond_N", cond)
        g.solve()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); bad = g.solve(); ts.append(time.perf_counter() - t0)
        it = g.info("iter")
        x = np.concatenate([g.get("x", k) for k in (1, N // 2, N)] + [g.get("u", k) for k in (0, N // 2)], axis=1)
        if ref is None: ref = x
        g.scalar("prof_reset"); g.opts_set("profile", 1); g.solve(); g.opts_set("profile", 0)
        ms = {c: g.scalar("prof_ms_" + c) / max(g.scalar("prof_cnt_" + c), 1) for c in ("back_fact", "fwd_aff", "back_rhs", "fwd_corr")}
        print(f"{case:10s} {ln or 'product':32s} W16T={t} ({'kt_factor, MFMA tiles' if t == '1' else 'ky_factor, DPP rows    '}) solve {min(ts)*1e3:7.2f} ms {B/min(ts):9.0f}/s  iters {int(it.sum())} max {int(it.max())} "
              f"failures {bad} KKT {g.res_compute().max():.2e}  launch us: fact {ms['back_fact']*1e3:.0f} faff {ms['fwd_aff']*1e3:.0f} rhs {ms['back_rhs']*1e3:.0f} fcor {ms['fwd_corr']*1e3:.0f}"
              f"  max |d| vs first {np.abs(x - ref).max():.1e}", flush=True)
        del g
