"""C4 (chain nx=24 nu=3, 4 soft state bounds + 4 soft general rows, N=40, 16,384 instances): factor sweep of the GEN
instantiation on 4 x 4 MFMA tiles (kt_factor<24,3,4>, ACADOS_AMD_W16T_GEN=1, the default) against register rows
(ky_factor<24,3,4>), same box, same data."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acados_amd import OcpQpGpuBatch, _lib
from acados_amd.generators import chain_soft_batch, chain_soft_dims, fill_chain_soft_batch

extra = [a for a in sys.argv[1:] if a.endswith(".so")]   # development builds of the library to run beside the product
args = [a for a in sys.argv[1:] if not a.endswith(".so")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = {n_: _lib.bind(ctypes.CDLL(os.path.join(ROOT, "tools", "ab", n_))) for n_ in extra}
N, B = 40, int(args[0]) if args else 16384
data = chain_soft_batch(N=N, batch=B, seed=1)
ref = None
for t, ln in [("1", None), ("0", None)] + [("1", n_) for n_ in extra] + [("1", None), ("0", None)] + [("1", n_) for n_ in extra]:
    os.environ["ACADOS_AMD_W16T_GEN"] = t
    g = OcpQpGpuBatch(chain_soft_dims(N), B, _clib=libs[ln] if ln else None)
    fill_chain_soft_batch(g, data, N)
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        g.opts_set(f, 1e-8)
    g.solve()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); bad = g.solve(); ts.append(time.perf_counter() - t0)
    it = g.info("iter")
    x = np.concatenate([g.get("x", k) for k in (1, N // 2, N)] + [g.get("u", k) for k in (0, N // 2)], axis=1)
    if ref is None: ref = x
    g.scalar("prof_reset"); g.opts_set("profile", 1); g.solve(); g.opts_set("profile", 0)
    ms = {c: g.scalar("prof_ms_" + c) / max(g.scalar("prof_cnt_" + c), 1) for c in ("back_fact", "fwd_aff", "back_rhs", "fwd_corr")}
    print(f"C4 {ln or 'product':30s} {g.kernel_name} tiles {int(g.scalar('w16_tiles'))}: solve {min(ts)*1e3:7.2f} ms {B/min(ts):9.0f}/s  iters {int(it.sum())} max {int(it.max())} failures {bad} "
          f"KKT {g.res_compute().max():.2e}  launch us: fact {ms['back_fact']*1e3:.0f} faff {ms['fwd_aff']*1e3:.0f} rhs {ms['back_rhs']*1e3:.0f} fcor {ms['fwd_corr']*1e3:.0f}  max |d| vs first {np.abs(x - ref).max():.1e}", flush=True)
    del g
