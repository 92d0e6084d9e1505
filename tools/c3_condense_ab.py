"""C3 (C2 data, partial condensing to N2 = 10, 65,536 instances): the condensing kernel on the FP64 matrix pipe (km_pcond,
v_mfma_f64_4x4x4_4b_f64) against the same contraction on register rows with DPP broadcasts (kz_pcond); same box, same data.
Reports condense + expand time (HIP events inside the library), whole solve, and the largest difference of the solutions."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acados_amd import OcpQpGpuBatch
from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch

N, B = 50, int(sys.argv[1]) if len(sys.argv) > 1 else 65536
data = random_lqr_batch(N=N, batch=B, seed=0)
ref = None
for mf in ("1", "0", "1", "0"):
    os.environ["ACADOS_AMD_PCOND_MFMA"] = mf
    gb = OcpQpGpuBatch(lqr_dims(N, 8, 3), B)
    fill_lqr_batch(gb, data, N)
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        gb.opts_set(f, 1e-8)
    gb.opts_set("cond_N", 10)
    gb.solve()
    ts, tx = [], []
    for _ in range(4):
        t0 = time.perf_counter(); bad = gb.solve(); ts.append(time.perf_counter() - t0); tx.append(gb.scalar("time_xcond"))
    x = np.concatenate([gb.get("x", k) for k in (1, N // 2, N)], axis=1)
    if ref is None: ref = x
    print(f"pcond kernel {int(gb.scalar('pcond_kernel'))} ({'km_pcond, MFMA 4x4x4' if mf == '1' else 'kz_pcond, DPP rows'}): solve {min(ts)*1e3:7.2f} ms "
          f"{B/min(ts):10.0f} solves/s  condense+expand {min(tx)*1e3:6.2f} ms  failures {bad}  KKT {gb.res_compute().max():.2e}  max |dx| vs first {np.abs(x-ref).max():.1e}", flush=True)
    del gb
