"""C2 (65,536 instances): solve time against the two hand-over rules of run_ipm -- `compact_min` (same-family compaction
of the survivors once at most half of a level is still iterating; off by default) and `tail_max` (switch of the last
survivors to the sixteen-lanes kernels) -- with the iteration histogram that explains it.
    python tools/compact_sweep.py [compact_min:tail_max ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acados_amd import OcpQpGpuBatch
from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch

N, B = 50, 65536
data = random_lqr_batch(N=N, batch=B, seed=0)
gb = OcpQpGpuBatch(lqr_dims(N, 8, 3), B)
fill_lqr_batch(gb, data, N)
for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
    gb.opts_set(f, 1e-8)
gb.solve()
it = gb.info("iter")
h = np.bincount(it)
alive = B - np.cumsum(h)
print("iteration histogram:", {int(i): int(c) for i, c in enumerate(h) if c})
print("still iterating after k iterations:", {int(i): int(a) for i, a in enumerate(alive) if 0 < a < B})
w = it.reshape(-1, 64).max(axis=1)
print(f"mean iterations {it.mean():.2f}; mean over waves of the wave maximum {w.mean():.2f} (what a one-instance-per-lane launch streams)")
ref = None
cases = [a for a in sys.argv[1:]] or ["0:12288", "0:20480", "16384:12288", "32768:12288", "65536:12288", "65536:8192", "65536:4096", "65536:20480"]
for c in cases:
    cm, tm = (int(x) for x in c.split(":"))
    gb.opts_set("compact_min", cm if cm > 0 else 1 << 30)
    gb.opts_set("tail_max", tm)
    gb.solve()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); bad = gb.solve(); ts.append(time.perf_counter() - t0)
    x = gb.get("x", N // 2).copy()
    if ref is None: ref = x
    print(f"compact_min {cm:6d} tail_max {tm:6d}: {min(ts)*1e3:7.2f} ms  {B/min(ts):10.0f} solves/s  compactions {int(gb.scalar('compactions'))} "
          f"tail switches {int(gb.scalar('tail_switches'))} failures {bad}  max |dx| vs first case {np.abs(x - ref).max():.1e}", flush=True)
