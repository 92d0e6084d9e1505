"""One C3 solve (C2 data, N2 = 10) per condensing kernel for profiler runs: python tools/c3_once.py [batch] [mfma: 1|0|both]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acados_amd import OcpQpGpuBatch
from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch

N, B = 50, int(sys.argv[1]) if len(sys.argv) > 1 else 65536
which = sys.argv[2] if len(sys.argv) > 2 else "both"
data = random_lqr_batch(N=N, batch=B, seed=0)
for mf in (("1", "0") if which == "both" else (which,)):
    os.environ["ACADOS_AMD_PCOND_MFMA"] = mf
    gb = OcpQpGpuBatch(lqr_dims(N, 8, 3), B)
    fill_lqr_batch(gb, data, N)
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        gb.opts_set(f, 1e-8)
    gb.opts_set("cond_N", 10)
    bad = gb.solve()
    print(f"pcond kernel {int(gb.scalar('pcond_kernel'))}: failures {bad}, condense+expand {gb.scalar('time_xcond')*1e3:.2f} ms")
    del gb
