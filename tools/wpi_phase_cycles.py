"""Development aid: cycles per phase of the wave-per-instance factor kernel (instance 0), from the
`make timing` build of the library.  Usage: python tools/wpi_phase_cycles.py [nx nu N batch]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (HIP runtime first)
from acados_amd import OcpQpGpuBatch, _lib
from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch

c4 = len(sys.argv) > 1 and sys.argv[1] == "c4"      # the C4 class (general rows + slacks) instead of a box-constrained LQR shape
nx, nu, N, B = (24, 3, 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1536) if c4 else \
    (int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (24, 6, 50, 1280)))
L = _lib.bind(ctypes.CDLL(os.path.join(ROOT, "tools", "ab", "libacados_amd_qp_timing.so")))
L.gqp_wpi_cycles_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
if c4:
    from acados_amd.generators import chain_soft_batch, chain_soft_dims, fill_chain_soft_batch
    data = chain_soft_batch(N=N, batch=B, seed=1)
    gb = OcpQpGpuBatch(chain_soft_dims(N), B, _clib=L)
    fill_chain_soft_batch(gb, data, N)
else:
    data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=B, seed=1)
    gb = OcpQpGpuBatch(lqr_dims(N, nx, nu), B, _clib=L)
    fill_lqr_batch(gb, data, N)
for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
    gb.opts_set(f, 1e-8)
gb.solve()
buf = np.zeros(16, dtype=np.uint64)
L.gqp_wpi_cycles_read(buf.ctypes.data, 1)
gb.solve()
L.gqp_wpi_cycles_read(buf.ctypes.data, 1)
it = int(gb.info("iter")[0]) + 1
# GQP_TICK slots of kw_factor / kw_factor_m, in the order the phases run inside a stage
order = [0, 6, 8, 9, 10, 11, 5, 1, 2, 12, 13, 14, 3, 4]
names = {0: "loads -> LDS (+ barrier)", 6: "touch next stage (prefetch)", 5: "vector part (rb, BA pi, Hv, rows, slacks)", 1: "W = [B A]' Lx+",
         2: "H tiles + w0 + M += W W' (+ G' Gamma G)", 3: "m + Cholesky (+ rhs) [rest]", 4: "factor -> LDS -> HBM",
         8: " vector: rb, BA pi, Hv (mfma kernel only)", 9: " vector: box row", 10: " vector: general row, slack rows + barrier",
         11: " vector: slack sums + barrier", 12: " chol: publish panel + barrier", 13: " chol: 4x4 block, own row + barrier", 14: " chol: trailing MFMA"}
tot = buf.sum()
print(f"kernel {gb.kernel_name}  batch {B}  instance 0: {it} factor sweeps, {N + 1} stages each")
for q in order:
    c = buf[q]
    print(f"  {names[q]:44s} {int(c) / it / (N + 1):10.0f} cycles/stage  {100.0 * int(c) / int(tot):5.1f} %")
print(f"  total                  {int(tot) / it / (N + 1):10.0f} cycles/stage (clock64 ticks)")
