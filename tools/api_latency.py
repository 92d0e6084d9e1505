"""Latency of the acados-shaped entry points (ocp_qp_solve, ocp_qp_solve_batch) on the GPU: pack + solve + unpack."""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acados_amd import AcadosOcpQpOptions, AcadosOcpQpSolver, AcadosOcpQpBatchSolver
from acados_amd.generators import mass_spring_qp, random_lqr_batch, lqr_instance_qp
qp = mass_spring_qp(N=20)
s = AcadosOcpQpSolver(qp, AcadosOcpQpOptions())
s.solve()
t0 = time.perf_counter()
for _ in range(10): s.solve()
print('single QP (C1 mass-spring N=20) ocp_qp_solve: %.2f ms per call, iter %d, time_tot %.3f ms, solver_call %.3f ms' % ((time.perf_counter()-t0)/10*1e3, s.get_stats('iter'), s.get_stats('time_tot')*1e3, s.get_stats('time_qp_solver_call')*1e3))
d = random_lqr_batch(N=50, batch=1024, seed=0)
qps = [lqr_instance_qp(d, i, 50) for i in range(1024)]
bs = AcadosOcpQpBatchSolver(qps, AcadosOcpQpOptions())
bs.solve()
t0 = time.perf_counter()
for _ in range(5): bs.solve()
print('ocp_qp_solve_batch n=1024 (C2 shape): %.2f ms per call' % ((time.perf_counter()-t0)/5*1e3), 'status ok', int((bs.status==0).sum()))
