"""Development aid: cycles per phase of the two-rows-per-lane factor kernel (ky_factor, instance 0), from the `make timing`
build of the library.  Usage: python tools/w16r_phase_cycles.py [nx nu N batch]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (HIP runtime first)
from acados_amd import OcpQpGpuBatch, _lib
from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch

c4 = len(sys.argv) > 1 and sys.argv[1] == "c4"     # the C4 class (general rows + slacks): python tools/w16r_phase_cycles.py c4 [batch]
L = _lib.bind(ctypes.CDLL(os.path.join(ROOT, "tools", "ab", os.environ.get("GQP_TIMING_LIB", "libacados_amd_qp_timing.so"))))
L.gqp_wpi_cycles_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
if c4:
    from acados_amd.generators import chain_soft_batch, chain_soft_dims, fill_chain_soft_batch
    N, B = 40, int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    data = chain_soft_batch(N=N, batch=B, seed=1)
    gb = OcpQpGpuBatch(chain_soft_dims(N), B, _clib=L)
    fill_chain_soft_batch(gb, data, N)
else:
    nx, nu, N, B = (int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (24, 6, 50, 4096)))
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
names = ["loads -> LDS staging, rows of H", "rb += [B A] v, H v", "W = [B A]' Lx+ (rolled), [B A]' pi+", "box rows, w0, m", "M += W W'",
         "Cholesky (+ rhs)", "factor -> HBM, x-block -> LDS"]
extra = {7: "  (rows: values to the lanes, general-row sums)", 8: "  (rows: the row functions)", 9: "  (rows: results back, rank-one terms)",
         10: "  (after W: stationarity residual, stores)", 11: "  (x-block transposed through LDS, w0)", 12: "  (descriptor of the stage after the next)",
         13: "  (DMA of the next [B A]': drain + issue)", 14: "  (prefetch of the next stage's vectors: issue)"}
tot = buf[:15].sum()
print(f"kernel {gb.kernel_name}  batch {B}  instance 0: {it} factor sweeps, {N + 1} stages each")
for q, nm in enumerate(names):
    print(f"  {nm:44s} {int(buf[q]) / it / (N + 1):10.0f} cycles/stage  {100.0 * int(buf[q]) / int(tot):5.1f} %")
for q, nm in extra.items():
    if buf[q]:
        print(f"  {nm:44s} {int(buf[q]) / it / (N + 1):10.0f} cycles/stage  {100.0 * int(buf[q]) / int(tot):5.1f} %")
print(f"  total                  {int(tot) / it / (N + 1):10.0f} cycles/stage (clock64 ticks)")
