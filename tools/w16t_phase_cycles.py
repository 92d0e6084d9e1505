"""Development aid: cycles per phase of the MFMA-tile factor sweep of the two-rows family (kt_factor, instance 0), from the
`make timing` build of the library.  Usage: python tools/w16t_phase_cycles.py [nx nu N batch]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (HIP runtime first)
from acados_amd import OcpQpGpuBatch, _lib
from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch

L = _lib.bind(ctypes.CDLL(os.path.join(ROOT, "tools", "ab", os.environ.get("GQP_TIMING_LIB", "libacados_amd_qp_timing.so"))))
L.gqp_wpi_cycles_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
if len(sys.argv) > 1 and sys.argv[1] == "c4":     # the C4 class (general rows + slacks): python tools/w16t_phase_cycles.py c4 [batch]
    from acados_amd.generators import chain_soft_batch, chain_soft_dims, fill_chain_soft_batch
    N, B = 40, int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    data = chain_soft_batch(N=N, batch=B, seed=1)
    gb = OcpQpGpuBatch(chain_soft_dims(N), B, _clib=L)
    fill_chain_soft_batch(gb, data, N)
else:
    nx, nu, N, B = (int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (24, 6, 20, 7281)))
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
names = ["wait for the DMA'd blocks / prefetched vectors (GEN: + the loads of this stage's vectors)", "exchange 1 (v, pi+ by x; b - x+ by y; GEN: + the inequality rows)", "tiles of H -> MT, H v (+ DMA of the next H)",
         "tiles of [B A], [B A] v", "[B A]' pi+", "DMA of the next [B A]' / prefetch of the next vectors (early variant)", "rb out, per-variable work (compact), late prefetch",
         "exchange 3 (m, diagonal terms by y)", "W' and M += W W' (MFMA)", "DMA of the next [B A]' (late variant), fixed-variable masking", "blocked Cholesky (MFMA + 4 x 4 diagonal blocks)",
         "factor -> HBM", "natural tiles of the state block (MFMA transposes)"]
tot = buf[:13].sum()
print(f"kernel {gb.kernel_name} (w16_tiles {int(gb.scalar('w16_tiles'))})  batch {B}  instance 0: {it} factor sweeps, {N + 1} stages each")
for q, nm in enumerate(names):
    print(f"  {nm:75s} {int(buf[q]) / it / (N + 1):8.0f} cycles/stage  {100.0 * int(buf[q]) / max(int(tot), 1):5.1f} %")
print(f"  total {int(tot) / it / (N + 1):10.0f} cycles/stage (clock64 ticks)")
