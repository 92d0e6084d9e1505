"""C2 (65,536 instances): solve time against the hand-over threshold of the tail switch (`tail_max`)."""
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
# (tail_max, tail_div): hand over when at most tail_max instances AND at most 1 / tail_div of the level remain
combos = [(0, 4), (8192, 4), (12288, 4), (16384, 4), (16384, 3), (20480, 3), (20480, 2), (24576, 2), (32768, 2)]
if len(sys.argv) > 1:
    combos = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]]
for tm, td in combos:
    gb.opts_set("tail_max", tm)
    gb.opts_set("tail_div", td)
    gb.solve()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); bad = gb.solve(); ts.append(time.perf_counter() - t0)
    print(f"tail_max {tm:6d} tail_div {td}: {min(ts)*1e3:7.2f} ms  {B/min(ts):10.0f} solves/s  tail switches {int(gb.scalar('tail_switches'))}  failures {bad}", flush=True)
