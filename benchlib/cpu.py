"""cpu_baseline: the restated CPU oracle (checker; kind "port", NOT HPIPM) timed on the host cores the process may use (a part of bench.py)."""
import os
import time

import numpy as np


def cpu_caps():
    caps = {"logical": os.cpu_count() or 1, "affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None}
    try:
        import psutil
        caps["physical"] = psutil.cpu_count(logical=False)
    except Exception:
        caps["physical"] = None
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        caps["cgroup_quota_cpus"] = None if q[0] == "max" else float(q[0]) / float(q[1])
    except Exception:
        caps["cgroup_quota_cpus"] = None
    return caps


def threads_allowed():
    caps = cpu_caps()
    allowed = caps["affinity"] or caps["logical"]
    if caps["cgroup_quota_cpus"]:
        allowed = max(1, min(allowed, int(round(caps["cgroup_quota_cpus"]))))
    return allowed


def cpu_baseline(data, N, unique, budget_s=25.0):
    """the oracle on the host cores: `unique` instances of the same workload built once, cloned so that every thread of
    every probe has >= 64 independent solves; thread counts swept in powers of two up to the cores the process may use"""
    from acados_amd.generators import lqr_instance_qp
    from oracle.oracle import OracleQp, clone_handle, default_opts, free_handle, solve_batch_handles
    t_begin = time.perf_counter()
    qps = [OracleQp(lqr_instance_qp(data, i, N)) for i in range(unique)]
    opts = default_opts(tol_stat=1e-8, tol_eq=1e-8, tol_ineq=1e-8, tol_comp=1e-8, iter_max=50)
    caps = cpu_caps()
    allowed = caps["affinity"] or caps["logical"]
    if caps["cgroup_quota_cpus"]:
        allowed = max(1, min(allowed, int(round(caps["cgroup_quota_cpus"]))))
    handles = [q.h.value for q in qps]
    clones = []

    def pool(n):
        while len(handles) + len(clones) < n:
            clones.append(clone_handle(qps[len(clones) % unique].h))
        return (handles + [c.value for c in clones])[:n]

    # one thread: 512 solves (~0.5 s)
    def run(threads, n, reps):
        hs = pool(n)
        best = 1e300
        for _ in range(reps):
            t0 = time.perf_counter()
            st = solve_batch_handles(hs, opts, nthreads=threads)
            best = min(best, time.perf_counter() - t0)
        assert np.all(st == 0)
        return n / best

    one = run(1, min(512, max(unique, 64)), 2)
    sweep = {1: one}
    th = 2
    while th <= allowed and time.perf_counter() - t_begin < budget_s:
        n = max(64 * th, 1024)                       # >= 64 QPs per thread
        sweep[th] = run(th, n, 2)
        th *= 2
    if allowed not in sweep and time.perf_counter() - t_begin < budget_s:
        sweep[allowed] = run(allowed, max(64 * allowed, 1024), 2)
    best_t = max(sweep, key=lambda k: sweep[k])
    # every unique instance solved at least once (the probes above may have touched only the first ones): one pass over
    # the whole sample with the best thread count -- its solutions are the parity sample
    t0 = time.perf_counter()
    st = solve_batch_handles(handles, opts, nthreads=best_t)
    full_pass = unique / (time.perf_counter() - t0)
    assert np.all(st == 0)
    iters = float(np.mean([q.iter for q in qps]))
    for c in clones:
        free_handle(c)
    cpu_baseline.solved = qps     # the same solutions double as the parity sample
    return {"value": sweep[best_t], "unit": "OCP-QP solves/s", "cores": best_t, "kind": "port", "unique": unique,
            "kind_note": "port = this repository's restated CPU oracle (plain C, scalar loops, no BLASFEO micro-kernels); HPIPM + BLASFEO sources "
                         "are absent from the reference tree, so the reference itself cannot be timed here.  Expect HPIPM on the same cores to be "
                         "several times faster than this port (its dpotrf / dsyrk / dtrmm run on AVX-512 panel-major kernels): the GPU / CPU "
                         "ratio of this line would shrink by that factor and says nothing about kernel quality -- the roofline fraction does",
            "sample": f"{unique} instances of the same workload (seed 0, first instances) built once and cloned to >= 64 "
                      f"independent solves per thread, min of 2 repeats per thread count, OpenMP over instances as "
                      f"acados_solver.in.c:3232 does; restated CPU oracle, not HPIPM",
            "one_thread": one, "full_sample_pass": full_pass, "thread_sweep": {str(k): v for k, v in sorted(sweep.items())},
            "host": caps, "threads_allowed": allowed, "mean_iter": iters, "seconds": time.perf_counter() - t_begin}
