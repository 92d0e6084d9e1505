"""Whole solve in one launch (kx_solve) against the launch-per-sweep loop on small batches: time per solve, launches.
    gpurun -- python tools/single_launch_latency.py"""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from acados_amd import OcpQpGpuBatch
from acados_amd.generators import mass_spring_qp, lqr_instance_qp, random_lqr_batch
data = random_lqr_batch(N=50, nx=8, nu=3, batch=256, seed=1)
cases = [("mass-spring N=20 x1", [mass_spring_qp(N=20)]), ("C2 shape x1", [lqr_instance_qp(data, 0, 50)]),
         ("C2 shape x16", [lqr_instance_qp(data, i, 50) for i in range(16)]), ("C2 shape x64", [lqr_instance_qp(data, i, 50) for i in range(64)]),
         ("C2 shape x256", [lqr_instance_qp(data, i, 50) for i in range(256)])]
for name, qps in cases:
    for smax in (0, 256):
        b = OcpQpGpuBatch.from_qps(qps)
        b.opts_set("solve_max", smax)
        b.solve()
        t0 = time.perf_counter()
        for _ in range(20): b.solve()
        dt = (time.perf_counter() - t0) / 20 * 1e3
        print("%-22s solve_max %3d: %.3f ms per solve  kernel %s single_launch %d launches %d iter max %d" % (name, smax, dt, b.kernel_name, b.scalar("single_launch_solves"), b.scalar("launches"), int(b.info("iter").max())))
