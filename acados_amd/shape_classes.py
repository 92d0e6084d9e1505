"""Mixed-shape batches (BASELINE configs[4], "C5"): one device batch per shape class, the classes solved concurrently.

acados batches QPs of ONE structure (acados_solver.in.c:3222-3243 loops over capsules of one generated solver); a fleet
with several OCP structures runs several of those loops.  Here every structure is one `OcpQpGpuBatch` with its own HIP
stream, and the classes overlap on the chip: one host thread per class drives its IPM loop (the solve call releases the
GIL).  The sweeps of the larger blocks fill a SIMD's register file, so two classes never share a SIMD -- what overlaps is
one class's ramp-down (the iterations its last few instances still need) with another class's bulk.  That only pays if
the class that takes longest is never the one that waits: it gets a high-priority stream, the others fill the gaps
(measured on the per-GPU share of C5, nine classes x 7,281 instances: 303 ms without priorities, 278 ms with the
longest class on a high-priority stream, 369 ms one class after the other; tools/c5_concurrent.py)."""
from concurrent.futures import ThreadPoolExecutor
import numpy as np


def estimated_work(batch):
    """relative cost of one solve of a device batch: instances x sum over the stages of (nu + nx)^3"""
    d = batch.dims
    n = np.asarray(d.nx, dtype=np.float64) + np.asarray(d.nu, dtype=np.float64)
    return float(batch.n_batch) * float(np.sum(n ** 3))


def prioritise(batches):
    """the class with the most work on a high-priority stream, every other class on a normal one (call once, batches idle)"""
    if len(batches) < 2:
        return
    work = [estimated_work(b) for b in batches]
    longest = int(np.argmax(work))
    for i, b in enumerate(batches):
        b.opts_set("stream_priority", -1 if i == longest else 0)


class ConcurrentClasses:
    """solve() of several device batches at once; returns the number of instances that did not converge"""

    def __init__(self, batches):
        self.batches = list(batches)
        prioritise(self.batches)
        # threads start in the order of decreasing work
        self._order = sorted(range(len(self.batches)), key=lambda i: -estimated_work(self.batches[i]))
        self._pool = ThreadPoolExecutor(max_workers=max(1, len(self.batches)))

    def solve(self):
        return int(sum(self._pool.map(lambda i: self.batches[i].solve(), self._order)))

    def close(self):
        self._pool.shutdown()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
