"""The per-GPU share of C5 (nine classes x 7,281 instances) solved concurrently from nine host threads, as bench.py does
it: makespan against the stream priorities of the classes (option `stream_priority`) and the order in which the threads
start.  python tools/c5_concurrent.py"""
import os, sys, time
import numpy as np
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acados_amd import OcpQpGpuBatch
from acados_amd.generators import C5_CLASSES, fill_lqr_batch, lqr_dims, random_lqr_batch

per_class = (524288 // 8) // len(C5_CLASSES)
batches = []
for ci, (nx, nu, N) in enumerate(C5_CLASSES):
    d = random_lqr_batch(N=N, nx=nx, nu=nu, batch=per_class, seed=200 + ci)
    g = OcpQpGpuBatch(lqr_dims(N, nx, nu), per_class)
    fill_lqr_batch(g, d, N)
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"): g.opts_set(f, 1e-8)
    g.solve()
    batches.append(((nx, nu, N), g))
seq = []
for c, g in batches:
    t0 = time.perf_counter(); g.solve(); seq.append(time.perf_counter() - t0)
print("one after the other: %.1f ms  " % (sum(seq) * 1e3) + "  ".join(f"{c}:{t*1e3:.1f}" for (c, _), t in zip(batches, seq)), flush=True)
cost = {c: t for (c, _), t in zip(batches, seq)}
schemes = {"no priorities": lambda c: 0,
           "longest class high, rest normal": lambda c: -1 if c == (24, 6, 100) else 0,
           "longest class high, other nx=24 normal, rest low": lambda c: -1 if c == (24, 6, 100) else (0 if c[0] == 24 else 1),
           "longest class high, rest low": lambda c: -1 if c == (24, 6, 100) else 1,
           "the two longest high, rest normal": lambda c: -1 if c in ((24, 6, 100), (24, 6, 50)) else 0,
           "by cost: >100 ms high, >20 ms normal, rest low": lambda c: -1 if cost[c] > 0.1 else (0 if cost[c] > 0.02 else 1)}
with ThreadPoolExecutor(max_workers=len(batches)) as pool:
    for name, pr in schemes.items():
        for c, g in batches: g.opts_set("stream_priority", pr(c))
        for order in ("longest first",):
            bs = batches if order == "as listed" else sorted(batches, key=lambda b: -cost[b[0]])
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); bad = sum(pool.map(lambda b: b[1].solve(), bs)); ts.append(time.perf_counter() - t0)
            print(f"{name:34s} threads {order:14s}: {min(ts)*1e3:7.1f} ms  {per_class*len(batches)/min(ts):9.0f} solves/s  failures {bad}", flush=True)
