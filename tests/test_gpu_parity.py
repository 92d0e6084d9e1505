"""GPU tier (-m gpu): parity tests proper.  Everything goes through the C-ABI of the hipcc-built
library on a real MI355X; the oracle (CPU) is the checker only.  Tolerances are written here:
  * vs oracle: 1e-8 relative on x,u,sl,su,pi,lam (north_star target: <= 1e-6 relative primal)
  * vs the reference's golden vectors: atol 1e-5 on lam, pi (test_ocpqp_solver.py:43)
  * KKT residuals <= 1e-8 (test_qpsolvers.cpp:83-86, 240-251)
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN_PAIRS, INPUT_ONLY, bulk_chunk_case, compare_condensed_with_oracle, compare_with_oracle, limit_cycle_case, load_qp, load_sol
from oracle.oracle import OracleQp, default_opts

# tolerance of an INDEPENDENTLY recomputed residual for a solve at tol 1e-8: the IPM judges complementarity by
# |lam t - tau| with the barrier floor tau = 1e-3 tol_comp, the residual kernel reports lam t itself
KKT_TOL = 1e-8 * (1.0 + 1e-3) + 1e-13

pytestmark = pytest.mark.gpu

ALL_QPS = [p for p, _ in GOLDEN_PAIRS] + INPUT_ONLY


@pytest.mark.parametrize("qp_file", ALL_QPS)
def test_batch_abi_matches_oracle_gpu(gpu_lib, qp_file):
    from acados_amd import OcpQpGpuBatch
    qp = load_qp(qp_file)
    o = OracleQp(qp)
    assert o.solve(default_opts(iter_max=100, tol_stat=1e-8)) == 0
    b = OcpQpGpuBatch.from_qps([qp] * 65)          # ragged: one wave + one lane
    b.opts_set("tol_stat", 1e-8)
    b.opts_set("iter_max", 100)
    assert b.solve() == 0
    assert np.all(np.abs(b.info("iter") - o.iter) <= 1)
    for inst in (0, 63, 64):
        compare_with_oracle(lambda k, f: b.get(f, k)[inst], o, qp, 1e-8, fields=("x", "u", "sl", "su", "pi", "lam"))
    assert max(b.info(n).max() for n in ("res_stat", "res_eq", "res_ineq", "res_comp")) <= 1e-8
    # all instances identical input -> bitwise identical output across lanes
    for k in range(qp.N + 1):
        x = b.get("x", k)
        assert np.all(x == x[0])


@pytest.mark.parametrize("qp_file,sol_file", GOLDEN_PAIRS)
def test_acados_api_golden_gpu(gpu_lib, qp_file, sol_file):
    """the reference's own QP test (test_ocpqp_solver.py:15-53) against this backend"""
    from acados_amd import AcadosOcpQpOptions, AcadosOcpQpSolver
    qp, sol = load_qp(qp_file), load_sol(sol_file)
    opts = AcadosOcpQpOptions()
    opts.iter_max = 500
    for name in ("PARTIAL_CONDENSING_GPU_IPM", "PARTIAL_CONDENSING_HPIPM"):
        opts.qp_solver = name
        solver = AcadosOcpQpSolver(qp, opts=opts)
        assert solver.solve() == 0, "QP solver returned non-zero status"
        for stage in range(qp.N + 1):
            ref = sol.get(f"lam_{stage}", np.zeros(0))
            if ref.size:
                assert np.allclose(solver.get(stage, "lam"), ref, atol=1e-5), f"lam mismatch at stage {stage}"
            if stage < qp.N:
                assert np.allclose(solver.get(stage, "pi"), sol[f"pi_{stage}"], atol=1e-5), f"pi mismatch at stage {stage}"


@pytest.mark.parametrize("N", [15, 20])
def test_mass_spring_gpu(gpu_lib, N):
    """C1 / the reference's unit-test QP: status 0 and max KKT residual <= 1e-8"""
    from acados_amd import AcadosOcpQpBatchSolver, AcadosOcpQpOptions
    from acados_amd.generators import mass_spring_qp
    qp = mass_spring_qp(N=N)
    o = OracleQp(qp)
    assert o.solve(default_opts(tol_stat=1e-8)) == 0
    opts = AcadosOcpQpOptions()
    opts.tol_stat = opts.tol_eq = opts.tol_ineq = opts.tol_comp = 1e-8
    for cond_N in (N, 5, 3):    # N2 in {N, 5, 3} as in test_qpsolvers.cpp:117-268
        opts.cond_N = cond_N
        s = AcadosOcpQpBatchSolver([qp, qp], opts)
        assert s.solve() == 0
        # condensed runs follow another iterate path to the same solution: multipliers of nearly active rows
        # (lam ~ 1e-6) then agree to a few 1e-8 only
        compare_with_oracle(lambda k, f: s.get_batch(1, k, f, unique_duals=False), o, qp, 1e-8 if cond_N == N else 2e-7)
        it0 = s.get_iter(0)
        assert s.solve() == 0 and s.get_iter(0) == it0   # cold start every time (mass_spring_example.c:352)


def test_lqr_sample_vs_oracle_gpu(gpu_lib):
    """C2 at full horizon (N=50, nx=8, nu=3): 1024-instance batch, 32 instances checked
    against the oracle; <= 1e-6 relative primal error is the north_star bar, we hold 1e-8"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, lqr_instance_qp, random_lqr_batch
    N, B = 50, 1024
    data = random_lqr_batch(N=N, batch=B, seed=0)
    gb = OcpQpGpuBatch(lqr_dims(N, 8, 3), B)
    fill_lqr_batch(gb, data, N)
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        gb.opts_set(f, 1e-8)
    assert gb.solve() == 0
    x = [gb.get("x", k) for k in range(N + 1)]
    u = [gb.get("u", k) for k in range(N)]
    pi = [gb.get("pi", k) for k in range(N)]
    worst = 0.0
    for i in range(0, B, 32):
        o = OracleQp(lqr_instance_qp(data, i, N))
        assert o.solve(default_opts(tol_stat=1e-8)) == 0
        for k in range(N + 1):
            rx = o.get(k, "x")
            worst = max(worst, np.max(np.abs(x[k][i] - rx) / np.maximum(1.0, np.abs(rx))))
            if k < N:
                ru, rp = o.get(k, "u"), o.get(k, "pi")
                worst = max(worst, np.max(np.abs(u[k][i] - ru) / np.maximum(1.0, np.abs(ru))))
                worst = max(worst, np.max(np.abs(pi[k][i] - rp) / np.maximum(1.0, np.abs(rp))))
        assert abs(gb.info("iter")[i] - o.iter) <= 1
    assert worst <= 1e-8, worst


def test_full_size_properties_gpu(gpu_lib):
    """BASELINE size (batch 65,536, N=50): size-independent properties instead of the oracle:
    status 0 everywhere, the four KKT residual norms <= tol for every instance, dynamics
    round trip x+ = A x + B u + b, input bounds respected, x_0 equals the given initial state,
    and a solution computed in a 65,536 batch equals the same instance solved in a 64 batch."""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch
    N, B = 50, 65536
    data = random_lqr_batch(N=N, batch=B, seed=0)
    gb = OcpQpGpuBatch(lqr_dims(N, 8, 3), B)
    fill_lqr_batch(gb, data, N)
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        gb.opts_set(f, 1e-8)
    assert gb.solve() == 0
    assert np.all(gb.info("status") == 0)
    for n in ("res_stat", "res_eq", "res_ineq", "res_comp"):
        assert gb.info(n).max() <= 1e-8
    # ... and the same statement from the kernel that shares nothing with the solver: ocp_qp_res_compute + _nrm_inf on the
    # (data, solution) in HBM, stationarity and complementarity included (mirror of test_qpsolvers.cpp:240-251)
    assert gb.res_compute().max() <= KKT_TOL
    it = gb.info("iter")
    assert it.min() >= 1 and it.max() <= 50
    x0 = gb.get("x", 0)
    assert np.array_equal(x0, data["x0"])
    xk = x0
    for k in range(N):
        uk = gb.get("u", k)
        assert uk.min() >= -0.5 - 1e-9 and uk.max() <= 0.5 + 1e-9
        xn = gb.get("x", k + 1)
        pred = np.einsum("bij,bj->bi", data["A"], xk) + np.einsum("bij,bj->bi", data["B"], uk) + data["b"]
        assert np.max(np.abs(pred - xn)) <= 1e-7
        xk = xn
    u0_default = gb.get("u", 0)[:64].copy()
    assert int(gb.scalar("tail_switches")) == 1      # the last survivors finished one wave per instance
    # with every level on the same kernels (tail_max = 0) the batch size must not change any instance's bits
    gb.opts_set("tail_max", 0)
    assert gb.solve() == 0
    u0_big = gb.get("u", 0)[:64].copy()
    assert np.allclose(u0_default, u0_big, rtol=0.0, atol=1e-10)
    del gb
    small = {k: v[:64] for k, v in data.items()}
    gs = OcpQpGpuBatch(lqr_dims(N, 8, 3), 64)
    fill_lqr_batch(gs, small, N)
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        gs.opts_set(f, 1e-8)
    gs.opts_set("tail_max", 0)
    assert gs.solve() == 0
    assert np.array_equal(gs.get("u", 0), u0_big)


def test_large_stage_blocks_gpu(gpu_lib):
    """stage blocks up to nu + nx = 64 (no compiled shape list: run-time dims of the wave-per-instance family;
    the largest tile count T8 = 8 and more than 64 KB of dynamic LDS) against the oracle"""
    from acados_amd.generators import lqr_instance_qp, random_lqr_batch
    for nx, nu, N in ((40, 3, 6), (48, 16, 4), (32, 8, 8)):   # 86, 128 and 80 inequality sides at stage 0
        data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=6, seed=3 + nx)
        b = _check_batch_vs_oracle_gpu([lqr_instance_qp(data, i, N) for i in range(6)], 3)
        assert b.kernel_name.startswith(f"wpi-box(nx={nx},nu={nu}")


def test_c3_full_size_properties_gpu(gpu_lib):
    """C3 at the BASELINE size (C2 data, 65,536 instances, partial condensing to N2 = 10): the expanded solution
    satisfies the full-space problem -- status 0, KKT residual norms of the condensed solve <= 1e-8, dynamics round
    trip over all 50 stages, input bounds, x_0 as given -- and equals the full-space solve of the same batch"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch
    N, B = 50, 65536
    data = random_lqr_batch(N=N, batch=B, seed=0)
    sols = []
    for cond_N in (10, 0):
        gb = OcpQpGpuBatch(lqr_dims(N, 8, 3), B)
        fill_lqr_batch(gb, data, N)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            gb.opts_set(f, 1e-8)
        if cond_N:
            gb.opts_set("cond_N", cond_N)
        assert gb.solve() == 0
        assert int(gb.scalar("cond_N_active")) == (cond_N if cond_N else N)
        for n in ("res_stat", "res_eq", "res_ineq", "res_comp"):
            assert gb.info(n).max() <= 1e-8
        # KKT residuals of the ORIGINAL QP at the (expanded) solution, recomputed independently of the solver
        assert gb.res_compute().max() <= (2e-8 if cond_N else KKT_TOL)
        xk = gb.get("x", 0)
        assert np.array_equal(xk, data["x0"])
        us = []
        for k in range(N):
            uk = gb.get("u", k)
            us.append(uk)
            assert uk.min() >= -0.5 - 1e-9 and uk.max() <= 0.5 + 1e-9
            xn = gb.get("x", k + 1)
            pred = np.einsum("bij,bj->bi", data["A"], xk) + np.einsum("bij,bj->bi", data["B"], uk) + data["b"]
            assert np.max(np.abs(pred - xn)) <= 1e-7
            xk = xn
        sols.append(np.stack(us))
        del gb
    assert np.max(np.abs(sols[0] - sols[1])) <= 1e-6


def _check_batch_vs_oracle_gpu(qps, n_check, tol=1e-8, tol_stat=1e-8):
    from acados_amd import OcpQpGpuBatch
    b = OcpQpGpuBatch.from_qps(qps)
    for f in ("tol_eq", "tol_ineq", "tol_comp"):
        b.opts_set(f, 1e-8)
    b.opts_set("tol_stat", tol_stat)
    assert b.solve() == 0
    for i in np.linspace(0, len(qps) - 1, n_check).astype(int):
        o = OracleQp(qps[i])
        assert o.solve(default_opts(tol_stat=tol_stat)) == 0
        compare_with_oracle(lambda k, f: b.get(f, k)[i], o, qps[i], tol)
    assert max(b.info(n).max() for n in ("res_eq", "res_ineq", "res_comp")) <= 1e-8
    assert b.info("res_stat").max() <= tol_stat
    return b


@pytest.mark.parametrize("fam", ["w16r-gen<NX=24,NU=3,NG=4>", "w16r-gen<NX=24,NU=3,NG=4>/rows", "wpi-gen(nx=24,nu=3,ng=4,ns=8"])
def test_c4_chain_soft_constraints_gpu(gpu_lib, monkeypatch, fam):
    """C4 shape: N=40 nx=24 nu=3, hard input bounds, soft state bounds, soft general rows, ns=8 -- on the two-rows-per-lane
    GEN kernels with the factor sweep on 4 x 4 MFMA tiles (kt_factor<24,3,4>, the default) or on register rows ("/rows":
    ky_factor<24,3,4>, ACADOS_AMD_W16T_GEN=0), and on the wave-per-instance GEN kernels"""
    from acados_amd.generators import chain_soft_qp
    monkeypatch.setenv("ACADOS_AMD_W16G", "1" if fam.startswith("w16r") else "0")
    monkeypatch.setenv("ACADOS_AMD_W16T_GEN", "0" if fam.endswith("/rows") else "1")
    rows, fam = fam.endswith("/rows"), fam.split("/")[0]
    # all four tolerances at 1e-8 (what ocp_nlp sets, ocp_nlp_common.c:1281-1293).  Instance 53 is the one
    # that used to stall at res_stat ~1e-5 with mu at 1e-16 until the slack block was eliminated in its
    # cancellation-free form: it is part of the batch on purpose.
    qps = [chain_soft_qp(i, N=40) for i in range(96)]
    b = _check_batch_vs_oracle_gpu(qps, 4, tol=1e-7, tol_stat=1e-8)
    assert b.kernel_name.startswith(fam)
    if fam.startswith("w16r"):
        assert int(b.scalar("w16_tiles")) == (0 if rows else 1)
    o = OracleQp(qps[53])
    assert o.solve(default_opts(tol_stat=1e-8)) == 0
    compare_with_oracle(lambda k, f: b.get(f, k)[53], o, qps[53], 1e-7)
    assert abs(int(b.info("iter")[53]) - o.iter) <= 1 and o.iter < 20


def test_chain_class_with_eight_general_rows_gpu(gpu_lib, monkeypatch):
    """the "ng = 8 chain class" (nx = 24, nu = 3, N = 40: 15 inequality rows per stage, ns = 12) on the two-rows-per-lane GEN kernels
    (w16r-gen<24,3,8>) against the oracle, the same iteration counts as the wave-per-instance GEN kernels it used to fall back to,
    and at least 1.8 x their rate on 2,048 instances (round-4 review, item 7)"""
    import time
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import chain_soft_qp
    qps = [chain_soft_qp(i, N=40, ng=8) for i in range(64)]
    monkeypatch.setenv("ACADOS_AMD_W16G", "1")
    b = _check_batch_vs_oracle_gpu(qps, 4, tol=1e-7, tol_stat=1e-8)
    assert b.kernel_name.startswith("w16r-gen<NX=24,NU=3,NG=8>")
    big = qps * 32
    rate, iters = {}, {}
    for fam, env in (("w16r-gen", "1"), ("wpi-gen", "0")):
        monkeypatch.setenv("ACADOS_AMD_W16G", env)
        g = OcpQpGpuBatch.from_qps(big)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            g.opts_set(f, 1e-8)
        assert g.kernel_name.startswith(fam) and g.solve() == 0
        t0 = time.perf_counter()
        assert g.solve() == 0
        rate[fam], iters[fam] = len(big) / (time.perf_counter() - t0), g.info("iter").copy()
    print("ng = 8 chain class, 2,048 instances:", {k: f"{v:.3e} solves/s" for k, v in rate.items()})
    assert np.array_equal(iters["w16r-gen"], iters["wpi-gen"])
    assert rate["w16r-gen"] >= 1.8 * rate["wpi-gen"], rate


def test_general_rows_and_slacks_at_small_shapes_gpu(gpu_lib, monkeypatch):
    """general rows + slacks at nu + nx <= 16 on the sixteen-lanes GEN kernels (w16r-gen<12,4,4>, <8,3,4> at one row per lane): oracle
    parity, identical iteration counts and the rate against the wave-per-instance GEN kernels they replace (4,096 instances)"""
    import time
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import chain_soft_qp
    for (nx, nu, ng, nsx), want in (((12, 4, 4, 4), "w16r-gen<NX=12,NU=4,NG=4>"), ((8, 3, 4, 2), "w16r-gen<NX=8,NU=3,NG=4>")):
        qps = [chain_soft_qp(i, N=30, nx=nx, nu=nu, ng=ng, nsx=nsx) for i in range(64)]
        monkeypatch.setenv("ACADOS_AMD_W16G", "1")
        b = _check_batch_vs_oracle_gpu(qps, 4, tol=1e-7, tol_stat=1e-8)
        assert b.kernel_name == want
        big = qps * 64
        rate, iters = {}, {}
        for fam, env in (("w16r-gen", "1"), ("wpi-gen", "0")):
            monkeypatch.setenv("ACADOS_AMD_W16G", env)
            g = OcpQpGpuBatch.from_qps(big)
            for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
                g.opts_set(f, 1e-8)
            assert g.kernel_name.startswith(fam) and g.solve() == 0
            t0 = time.perf_counter()
            assert g.solve() == 0
            rate[fam], iters[fam] = len(big) / (time.perf_counter() - t0), g.info("iter").copy()
        print((nx, nu, ng), "4,096 instances:", {k: f"{v:.3e} solves/s" for k, v in rate.items()})
        assert np.array_equal(iters["w16r-gen"], iters["wpi-gen"])
        assert rate["w16r-gen"] >= 1.5 * rate["wpi-gen"], rate


def test_c4_one_instance_per_lane_kernels_gpu(gpu_lib, monkeypatch):
    """the general one-instance-per-lane kernels (what shapes below nu+nx = 13 with general rows / slacks
    run on) on the C4 shape, forced with ACADOS_AMD_WPI=0"""
    from acados_amd.generators import chain_soft_qp
    monkeypatch.setenv("ACADOS_AMD_WPI", "0")
    b = _check_batch_vs_oracle_gpu([chain_soft_qp(i, N=10) for i in range(8)], 2, tol=1e-7, tol_stat=1e-8)
    assert b.kernel_name == "1tpi<NX=24,NU=3,NG=4,NS=8>"


def test_c4_full_size_properties_gpu(gpu_lib):
    """C4 at the BASELINE size (16,384 instances, N=40, nx=24, nu=3, 4 soft state bounds, 4 soft general rows,
    ns=8): size-independent properties over the whole batch -- status 0, the four KKT residual norms <= 1e-8,
    dynamics round trip, hard input bounds, slacks >= 0 and soft rows satisfied up to their slack, x_0 as given --
    plus four instances against the oracle."""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import chain_soft_batch, chain_soft_dims, chain_soft_instance_qp, fill_chain_soft_batch
    N, B = 40, 16384
    data = chain_soft_batch(N=N, batch=B, seed=1)
    gb = OcpQpGpuBatch(chain_soft_dims(N), B)
    fill_chain_soft_batch(gb, data, N)
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        gb.opts_set(f, 1e-8)
    assert gb.solve() == 0 and gb.kernel_name.startswith("w16r-gen<NX=24,NU=3,NG=4>")
    assert np.all(gb.info("status") == 0)
    for n in ("res_stat", "res_eq", "res_ineq", "res_comp"):
        assert gb.info(n).max() <= 1e-8
    assert gb.res_compute().max() <= KKT_TOL     # independent kernel: ocp_qp_res_compute on the (data, solution) in HBM
    assert gb.info("iter").max() <= 50
    xk = gb.get("x", 0)
    assert np.array_equal(xk, data["x0"])
    ixs = 6 * np.arange(4) + 1
    for k in range(N + 1):
        if k < N:
            uk = gb.get("u", k)
            assert np.abs(uk).max() <= 1.0 + 1e-9
            xn = gb.get("x", k + 1)
            pred = np.einsum("bij,bj->bi", data["A"], xk) + np.einsum("bij,bj->bi", data["B"], uk) + data["b"][:, k]
            assert np.max(np.abs(pred - xn)) <= 1e-7
        if k > 0:
            sl, su = gb.get("sl", k), gb.get("su", k)
            assert sl.min() >= -1e-9 and su.min() >= -1e-9
            xs = xk[:, ixs]
            assert np.all(xs >= -0.3 - sl[:, :4] - 1e-7) and np.all(xs <= 0.3 + su[:, :4] + 1e-7)
            g = np.einsum("bij,bj->bi", data["C"][:, k - 1], xk)
            if k < N:
                g = g + np.einsum("bij,bj->bi", data["D"][:, k - 1], gb.get("u", k))
            assert np.all(g >= -0.5 - sl[:, 4:] - 1e-7) and np.all(g <= 0.5 + su[:, 4:] + 1e-7)
        if k < N:
            xk = xn
    for i in (0, 5461, 10922, 16383):
        qp = chain_soft_instance_qp(data, i, N)
        o = OracleQp(qp)
        assert o.solve(default_opts(tol_stat=1e-8)) == 0
        compare_with_oracle(lambda k, f: gb.get(f, k)[i], o, qp, 1e-7)


def test_c5_per_gpu_share_properties_gpu(gpu_lib):
    """C5 at the per-GPU share of the BASELINE size (524,288 instances on 8 GPUs = 65,536 per GPU, split over
    the 9 shape classes): every class as one device batch, size-independent properties over all instances"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import C5_CLASSES, fill_lqr_batch, lqr_dims, random_lqr_batch
    per_class = 65536 // len(C5_CLASSES)
    for (nx, nu, N) in C5_CLASSES:
        data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=per_class, seed=100 + nx + N)
        gb = OcpQpGpuBatch(lqr_dims(N, nx, nu), per_class)
        fill_lqr_batch(gb, data, N)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            gb.opts_set(f, 1e-8)
        assert gb.solve() == 0, (nx, nu, N, gb.kernel_name)
        for n in ("res_stat", "res_eq", "res_ineq", "res_comp"):
            assert gb.info(n).max() <= 1e-8
        # KKT residuals recomputed independently of the solver (ocp_qp_res_compute), every instance of every class
        assert gb.res_compute().max() <= KKT_TOL
        xk = gb.get("x", 0)
        assert np.array_equal(xk, data["x0"])
        for k in range(N):
            uk = gb.get("u", k)
            assert uk.min() >= -0.5 - 1e-9 and uk.max() <= 0.5 + 1e-9
            xn = gb.get("x", k + 1)
            pred = np.einsum("bij,bj->bi", data["A"], xk) + np.einsum("bij,bj->bi", data["B"], uk) + data["b"]
            assert np.max(np.abs(pred - xn)) <= 1e-6 * max(1.0, np.abs(xn).max())
            xk = xn
        del gb


def test_c5_mixed_shape_classes_gpu(gpu_lib):
    """C5: the 9 shape classes (nx in {4,12,24}, N in {20,50,100}) bucketed by dims.signature(),
    plus the multi-phase class; every bucket is one device batch"""
    from acados_amd.generators import C5_CLASSES, lqr_instance_qp, multiphase_qp, random_lqr_batch
    buckets = {}
    for (nx, nu, N) in C5_CLASSES:
        data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=70, seed=nx + N)
        for i in range(70):
            qp = lqr_instance_qp(data, i, N)
            buckets.setdefault(qp.dims.signature(), []).append(qp)
    for i in range(70):
        qp = multiphase_qp(i, N=20)
        buckets.setdefault(qp.dims.signature(), []).append(qp)
    assert len(buckets) == 10
    for sig, qps in buckets.items():
        _check_batch_vs_oracle_gpu(qps, 2)


def test_c3_partial_condensing_gpu(gpu_lib):
    """C3: C2 data (N=50 nx=8 nu=3) with partial condensing to N2=10 (blocks of 5): condense on the
    device, IPM on the condensed QP (nx=8, nu=15), expansion; checked against the full-space oracle.
    Also the RTI split condense_lhs / condense_rhs_and_solve."""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, lqr_instance_qp, random_lqr_batch
    N, B = 50, 256
    data = random_lqr_batch(N=N, batch=B, seed=0)
    for split in (False, True):
        gb = OcpQpGpuBatch(lqr_dims(N, 8, 3), B)
        fill_lqr_batch(gb, data, N)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            gb.opts_set(f, 1e-8)
        gb.opts_set("cond_N", 10)
        if split:
            assert gb.condense_lhs() == 0 and gb.condense_rhs_and_solve() == 0
        else:
            assert gb.solve() == 0
        assert int(gb.scalar("cond_N_active")) == 10
        for i in (0, 100, 255):
            qp = lqr_instance_qp(data, i, N)
            o = OracleQp(qp)
            assert o.solve(default_opts(tol_stat=1e-8)) == 0
            compare_with_oracle(lambda k, f: gb.get(f, k)[i], o, qp, 1e-8, fields=("x", "u", "pi", "lam"))


def test_device_pointer_input_gpu(gpu_lib):
    """torch CUDA tensors handed over as raw device pointers give the same result as host arrays"""
    import torch
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch
    N, B = 10, 256
    data = random_lqr_batch(N=N, batch=B, seed=2)
    res = []
    for xp in (None, lambda a: _to_dev(torch, a)):
        gb = OcpQpGpuBatch(lqr_dims(N, 8, 3), B)
        fill_lqr_batch(gb, data, N, xp=xp)
        assert gb.solve() == 0
        res.append(gb.get("u", 0))
    assert np.array_equal(res[0], res[1])


def _to_dev(torch, a):
    t = torch.from_numpy(a).cuda()
    torch.cuda.synchronize()
    return t


def test_smoke_entry_gpu(gpu_lib):
    import __graft_entry__ as g
    g.smoke()


@pytest.mark.gpu
@pytest.mark.parametrize("w16", ["1", "0"])
def test_solution_sensitivities_gpu(gpu_lib, monkeypatch, w16):
    """a12 on the device: d(x, u)/dp from one rhs-only backward + one forward sweep with the factorisation at the
    solution, against central finite differences of the solver's own solutions (64 instances, N = 20), on the
    sixteen-lanes and the wave-per-instance kernels"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch
    monkeypatch.setenv("ACADOS_AMD_WPI", "1")
    monkeypatch.setenv("ACADOS_AMD_W16", w16)
    N, B, nx, nu = 20, 64, 8, 3
    data = random_lqr_batch(N=N, batch=B, seed=8)

    def build(d):
        gb = OcpQpGpuBatch(lqr_dims(N, nx, nu), B)
        fill_lqr_batch(gb, d, N)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            gb.opts_set(f, 1e-9)
        assert gb.solve() == 0
        return gb

    def xu(g, pre=""):
        return np.concatenate([g.get(pre + "x", k) for k in range(N + 1)] + [g.get(pre + "u", k) for k in range(N)], axis=1)

    ref = build(data)
    assert ref.kernel_name.startswith("w16-box" if w16 == "1" else "wpi-box")
    e = np.zeros((B, nx)); e[:, 2] = 1.0
    h = 1e-4   # finite-difference noise: solution accuracy / h ~ 1e-5; an instance whose active set changes within
               # +-h has no derivative to compare with -- hence the per-instance statistics below
    for key, seeds in (("q", [("seed_q", k, e) for k in range(N + 1)]), ("x0", [("seed_lbx", 0, e), ("seed_ubx", 0, e)])):
        sols = []
        for sg in (+h, -h):
            d = {k: v.copy() for k, v in data.items()}
            d[key][:, 2] += sg
            sols.append(xu(build(d)))
        fd = (sols[0] - sols[1]) / (2 * h)
        for (f, k, v) in seeds:
            ref.sens_set(f, k, v)
        ref.sens_solve()
        se = xu(ref, "sens_")
        err = np.max(np.abs(fd - se), axis=1) / np.maximum(1.0, np.max(np.abs(se), axis=1))
        assert np.mean(err <= 5e-5) >= 0.9 and np.median(err) <= 1e-5, (key, np.sort(err)[-5:])


@pytest.mark.gpu
def test_solution_sensitivities_large_batch_gpu(gpu_lib, monkeypatch):
    """a12 on a batch of the headline kernel family (one instance per lane, 40,000 instances = three slices through the
    wave-per-instance sub-batch): the x0-sensitivities equal those of a small batch of the same leading instances
    computed on the sixteen-lanes kernels directly, and the slices beyond the first answer too"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch
    N, B, nx, nu, Bs = 10, 40000, 8, 3, 256
    data = random_lqr_batch(N=N, batch=B, seed=12)
    e = np.zeros((B, nx)); e[:, 1] = 1.0

    def run(d, nb):
        monkeypatch.setenv("ACADOS_AMD_WPI", "0" if nb == B else "1")
        gb = OcpQpGpuBatch(lqr_dims(N, nx, nu), nb)
        fill_lqr_batch(gb, d, N)
        gb.opts_set("tol_stat", 1e-9); gb.opts_set("tol_comp", 1e-9)
        assert gb.solve() == 0
        gb.sens_set("seed_lbx", 0, e[:nb]); gb.sens_set("seed_ubx", 0, e[:nb])
        gb.sens_solve()
        return gb, np.concatenate([gb.get("sens_x", k) for k in range(N + 1)] + [gb.get("sens_u", k) for k in range(N)], axis=1)

    big, sb = run(data, B)
    assert big.kernel_name.startswith("1tpi")
    head = {k: v[:Bs] for k, v in data.items()}
    tailp = {k: v[B - Bs:] for k, v in data.items()}
    small, ss = run(head, Bs)
    assert small.kernel_name.startswith("w16")
    _, st = run(tailp, Bs)
    # same kernels on the same solutions up to the solvers' own accuracy (two different iteration paths)
    for got, want in ((sb[:Bs], ss), (sb[B - Bs:], st)):
        err = np.max(np.abs(got - want), axis=1) / np.maximum(1.0, np.max(np.abs(want), axis=1))
        assert np.mean(err <= 1e-5) >= 0.95 and np.median(err) <= 1e-6, np.sort(err)[-5:]
    assert np.all(sb[:, 1] == 1.0) and np.max(np.abs(sb[:, nx:])) > 1e-2


@pytest.mark.gpu
def test_partial_condensing_general_rows_gpu(gpu_lib):
    """a5-a7 beyond the box class on the device: the C4 class (soft state bounds + soft general rows, nx = 24)
    condensed N = 40 -> 20 blocks of 2 (state bounds / general rows of the inner stages become general rows of the
    condensed stages, slacks travel along), 2,048 instances: the expanded solution equals the full-space solution
    of the same batch, and the KKT residuals of the ORIGINAL QP evaluated at the expanded solution by a hot-started
    full-space call are at tolerance.  Plus the reference's mass-spring unit-test QP with state bounds at every
    stage, N2 = 5 and 3, against the oracle."""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import chain_soft_batch, chain_soft_dims, fill_chain_soft_batch, mass_spring_qp
    N, B = 40, 2048
    data = chain_soft_batch(N=N, batch=B, seed=3)
    sol = {}
    for cond_N in (N, 20):
        gb = OcpQpGpuBatch(chain_soft_dims(N), B)
        fill_chain_soft_batch(gb, data, N)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            gb.opts_set(f, 1e-8)
        gb.opts_set("cond_N", cond_N)
        assert gb.solve() == 0
        assert int(gb.scalar("cond_N_active")) == cond_N
        sol[cond_N] = {f: np.concatenate([gb.get(f, k) for k in range(N + (f != "pi" and f != "u"))], axis=1)
                       for f in ("x", "u", "pi", "lam", "sl", "su")}
        # KKT residuals of the ORIGINAL QP at the (expanded) solution from the independent residual kernel
        # (ocp_qp_res_compute): every instance, all four norms
        assert gb.res_compute().max() <= (2e-8 if cond_N < N else KKT_TOL)
        if cond_N < N:
            # the expanded point in the ORIGINAL QP: a hot-started full-space call finds its KKT residuals at
            # tolerance straight away (one more iteration allowed: the condensed residual norms are not the same norms)
            gb.opts_set("cond_N", N)
            gb.opts_set("warm_start", 3)
            assert gb.solve() == 0
            assert int(gb.info("iter").max()) <= 1
    # two iterate paths to the same solution, both inside the 1e-8 KKT ball (asserted above by the independent residual
    # kernel -- that is the sharp statement).  What the ball allows: on a weakly active row complementarity 1e-8 leaves
    # lam ~ t ~ sqrt(1e-8) = 1e-4, so the primal points of such instances differ by up to ~1e-4 and their multipliers
    # (not unique in the degenerate limit) by more; everything else agrees to ~1e-6
    for f in ("x", "u", "sl", "su", "pi", "lam"):
        a, c = sol[N][f], sol[20][f]
        err = np.max(np.abs(a - c) / np.maximum(1.0, np.abs(a)), axis=1)
        primal = f in ("x", "u", "sl", "su")
        w = 1.0 if f in ("x", "u") else 10.0
        assert np.median(err) <= 2e-6 * w and np.mean(err <= 2e-5 * w) >= 0.9 and err.max() <= (5e-4 if primal else 2e-2), \
            (f, np.median(err), np.sort(err)[-5:])
    for cn in (5, 3):
        qp = mass_spring_qp(N=15)
        b = OcpQpGpuBatch.from_qps([qp] * 70)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            b.opts_set(f, 1e-8)
        b.opts_set("cond_N", cn)
        assert b.solve() == 0 and int(b.scalar("cond_N_active")) == cn
        o = OracleQp(qp)
        assert o.solve(default_opts(tol_stat=1e-8)) == 0
        compare_with_oracle(lambda k, f: b.get(f, k)[69], o, qp, 2e-7, fields=("x", "u", "pi", "lam", "t"))


@pytest.mark.gpu
@pytest.mark.parametrize("wpi", ["0", "1"])
def test_random_structures_gpu(gpu_lib, monkeypatch, wpi):
    """40 QPs with random STRUCTURE (tests/random_qp.py: per-stage dims, box subsets, one-sided rows, general rows,
    slacks shared between rows, equality-flagged x0, N = 1..6), 70 copies per batch, on the one-instance-per-lane
    and the wave-per-instance kernels against the oracle; with N >= 2 also partially condensed (N2 = ceil(N/2)):
    the expanded point satisfies the ORIGINAL KKT conditions (hot-started full-space call: 0 iterations)"""
    from acados_amd import OcpQpGpuBatch
    from random_qp import random_structure_qp
    monkeypatch.setenv("ACADOS_AMD_WPI", wpi)
    condensed = sharp = 0
    for seed in range(40):
        qp = random_structure_qp(seed)
        o = OracleQp(qp)
        assert o.solve(default_opts(tol_stat=1e-8, iter_max=60)) == 0, seed
        b = OcpQpGpuBatch.from_qps([qp] * 70)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            b.opts_set(f, 1e-8)
        b.opts_set("iter_max", 60)
        assert b.solve() == 0, (seed, b.kernel_name)
        assert abs(int(b.info("iter")[69]) - o.iter) <= 1, (seed, b.kernel_name)
        compare_with_oracle(lambda k, f: b.get(f, k)[69], o, qp, 1e-7, fields=("x", "u", "sl", "su", "pi", "lam", "t"))
        if qp.N >= 2:
            b.opts_set("cond_N", (qp.N + 1) // 2)
            assert b.solve() == 0, seed
            # condensed run: another iterate path inside the same 1e-8 KKT ball of the ORIGINAL QP -- asserted by the
            # independent residual kernel; the direct comparison is sharp (2e-6 primal, 1e-7 on the multipliers of strictly
            # inactive rows) except on instances with a weakly active row (sqrt(tol) = 1e-4): conftest.py
            assert b.res_compute().max() <= 2e-8, seed
            sharp += compare_condensed_with_oracle(lambda k, f: b.get(f, k)[69], o, qp)
            condensed += 1
            b.opts_set("cond_N", qp.N)
            b.opts_set("warm_start", 3)
            assert b.solve() == 0 and int(b.info("iter").max()) == 0, seed
    assert condensed >= 30 and sharp >= condensed - 4, (condensed, sharp)   # all but the weakly active instances compared sharply


@pytest.mark.gpu
def test_condensing_only_boundary_gpu(gpu_lib):
    """ocp_qp_condense / ocp_qp_expand on the device (condensing_interface.h:73-75): the condensed QP read back from
    the GPU is solved by the CPU oracle, that solution is expanded on the GPU and must be the full-space oracle
    solution -- the condensed DATA are pinned, not only the round trip.  Mass-spring with user block sizes, the golden
    shared-slack fixture, random structures; through the batch API (70 copies) and the acados-shaped module."""
    from acados_amd import AcadosOcpQpCondensing, OcpQpGpuBatch
    from acados_amd.generators import mass_spring_qp
    from random_qp import random_structure_qp
    import ctypes
    cases = [(mass_spring_qp(N=15), 4, [3, 4, 4, 4, 0]), (load_qp("casadi_qp_tests/pend_idxs_rev_min_qp0.json"), 3, None)]
    cases += [(random_structure_qp(s), None, None) for s in (0, 4, 7, 13, 22, 26)]
    for qp, cn, blocks in cases:
        cn = (qp.N + 1) // 2 if cn is None else cn
        o = OracleQp(qp)
        assert o.solve(default_opts(tol_stat=1e-8, iter_max=60)) == 0
        b = OcpQpGpuBatch.from_qps([qp] * 70)
        b.opts_set("cond_N", cn)
        if blocks is not None:
            assert b._L.ocp_qp_gpu_batch_opts_set(b._h, b"cond_block_size", (ctypes.c_int * len(blocks))(*blocks)) == 0
        c = b.condense()
        assert c is not None and c.N == cn
        qc = c.to_qp(69)
        oc = OracleQp(qc)
        assert oc.solve(default_opts(tol_stat=1e-8, iter_max=60)) == 0
        for k in range(cn + 1):
            for f in ("x", "u", "sl", "su", "lam", "t") + (("pi",) if k < cn else ()):
                v = oc.get(k, f)
                if v.size:
                    c.set(f, k, np.tile(v, (70, 1)))
        b.expand()
        compare_condensed_with_oracle(lambda k, f: b.get(f, k)[69], o, qp)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            b.opts_set(f, 1e-8)
        b.opts_set("cond_N", qp.N)
        b.opts_set("warm_start", 3)
        assert b.solve() == 0 and int(b.info("iter").max()) == 0   # KKT of the ORIGINAL QP at the expanded point
        mod = AcadosOcpQpCondensing(qp, cn, block_size=blocks)
        qc2 = mod.condense()
        for k in range(cn + 1):
            for f in ("Q", "R", "S", "q", "r", "lbx", "ubx", "lg", "ug", "C", "D"):
                assert np.allclose(np.asarray(getattr(qc2, f)[k]), np.asarray(getattr(qc, f)[k]), rtol=0, atol=1e-12), (f, k)
        get = mod.expand(lambda k, f: oc.get(k, f))
        compare_condensed_with_oracle(get, o, qp)


@pytest.mark.gpu
def test_sixteen_lanes_soft_box_rows_gpu(gpu_lib, monkeypatch):
    """SOFT variants of the sixteen-lanes kernels on the device: 40 random structures without general rows (soft and
    hard box rows mixed, one-sided rows, per-stage dims; a slack shared by several rows falls back to the general
    wave-per-instance kernels), 70 copies each, against the oracle; then the C2 shape with soft bounds on every state,
    4,096 instances: residual norms, slack signs, soft rows satisfied up to their slack, 4 instances vs the oracle"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, lqr_instance_qp, random_lqr_batch
    from random_qp import random_structure_qp
    monkeypatch.setenv("ACADOS_AMD_WPI", "1")
    fams = {}
    for seed in range(40):
        qp = random_structure_qp(seed, allow_general=False)
        o = OracleQp(qp)
        assert o.solve(default_opts(tol_stat=1e-8, iter_max=60)) == 0, seed
        b = OcpQpGpuBatch.from_qps([qp] * 70)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            b.opts_set(f, 1e-8)
        b.opts_set("iter_max", 60)
        assert b.solve() == 0, (seed, b.kernel_name)
        fam = b.kernel_name.split("<")[0].split("(")[0]
        fams[fam] = fams.get(fam, 0) + 1
        assert abs(int(b.info("iter")[69]) - o.iter) <= 1, (seed, b.kernel_name)
        compare_with_oracle(lambda k, f: b.get(f, k)[69], o, qp, 1e-7, fields=("x", "u", "sl", "su", "pi", "lam", "t"))
    assert fams.get("w16-soft", 0) >= 8 and fams.get("wpi-gen", 0) >= 1, fams
    monkeypatch.delenv("ACADOS_AMD_WPI")
    N, nx, nu, B = 50, 8, 3, 4096
    data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=B, seed=0)
    d = lqr_dims(N, nx, nu)
    d.nbx[1:] = nx
    d.nb[:] = d.nbu + d.nbx
    d.ns[1:] = nx
    gb = OcpQpGpuBatch(d, B)
    for k in range(1, N + 1):
        gb.set_int("idxs_rev", k, np.concatenate([-np.ones(int(d.nbu[k]), dtype=int), np.arange(nx)]))
    fill_lqr_batch(gb, data, N)
    for k in range(1, N + 1):
        gb.set("lbx", k, np.full((B, nx), -1.0)); gb.set("ubx", k, np.full((B, nx), 1.0))
        for f, v in (("Zl", 1e2), ("Zu", 1e2), ("zl", 1e1), ("zu", 1e1), ("lls", 0.0), ("lus", 0.0)):
            gb.set(f, k, np.full((B, nx), v))
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        gb.opts_set(f, 1e-8)
    assert gb.solve() == 0 and gb.kernel_name == "w16-soft<NX=8,NU=3>"
    for n_ in ("res_stat", "res_eq", "res_ineq", "res_comp"):
        assert gb.info(n_).max() <= 1e-8
    viol = 0.0
    for k in range(1, N + 1):
        x, sl, su = gb.get("x", k), gb.get("sl", k), gb.get("su", k)
        assert sl.min() >= -1e-9 and su.min() >= -1e-9
        assert np.all(x >= -1.0 - sl - 1e-7) and np.all(x <= 1.0 + su + 1e-7)
        viol = max(viol, float(sl.max()), float(su.max()))
    assert viol > 1e-2   # the soft bounds are really used
    for i in (0, 1365, 2730, 4095):
        qp = lqr_instance_qp(data, i, N)
        for k in range(1, N + 1):
            nuk = nu if k < N else 0
            qp.set("idxb", k, np.arange(nuk + nx))
            qp.set("lbx", k, -np.ones(nx)); qp.set("ubx", k, np.ones(nx))
            qp.set("lbx_mask", k, np.ones(nx)); qp.set("ubx_mask", k, np.ones(nx))
            qp.set("idxs_rev", k, np.concatenate([-np.ones(nuk, dtype=int), np.arange(nx)]))
            for f, v in (("Zl", 1e2), ("Zu", 1e2), ("zl", 1e1), ("zu", 1e1), ("lls", 0.0), ("lus", 0.0)):
                qp.set(f, k, v * np.ones(nx))
        qp.make_consistent()
        o = OracleQp(qp)
        assert o.solve(default_opts(tol_stat=1e-8)) == 0
        compare_with_oracle(lambda k, f: gb.get(f, k)[i], o, qp, 1e-7, fields=("x", "u", "sl", "su", "pi", "lam", "t"))


@pytest.mark.gpu
def test_random_structures_large_dims_gpu(gpu_lib):
    """random structures with nx up to 40 and nu up to 8 (tile counts T8 = 1..6 of the wave-per-instance GEN kernels,
    stages with more than 64 inequality sides) and nx up to 12 / nu up to 4 (the padded sixteen-lanes shapes, hard and
    soft), default dispatch, 70 copies each, against the oracle"""
    from acados_amd import OcpQpGpuBatch
    from random_qp import random_structure_qp
    fams = set()
    for seed, (nxm, num) in [(s, (40, 8)) for s in range(200, 220)] + [(s, (12, 4)) for s in range(100, 130)]:
        qp = random_structure_qp(seed, nx_max=nxm, nu_max=num, allow_general=(seed % 2 == 0))
        o = OracleQp(qp)
        assert o.solve(default_opts(tol_stat=1e-8, iter_max=80)) == 0, seed
        b = OcpQpGpuBatch.from_qps([qp] * 70)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            b.opts_set(f, 1e-8)
        b.opts_set("iter_max", 80)
        assert b.solve() == 0, (seed, b.kernel_name)
        fams.add(b.kernel_name.split("<")[0].split("(")[0])
        assert abs(int(b.info("iter")[69]) - o.iter) <= 1, (seed, b.kernel_name)
        compare_with_oracle(lambda k, f: b.get(f, k)[69], o, qp, 3e-7, fields=("x", "u", "sl", "su", "pi", "lam", "t"))
    assert len(fams) >= 3, fams


@pytest.mark.gpu
def test_concurrent_solvers_from_host_threads_gpu(gpu_lib):
    """the reference's batch idiom calls ocp_qp_solve from OpenMP threads on distinct solver objects
    (acados_solver.in.c:3232-3236): four host threads, each with its own solver, own stream and own QP (one partially
    condensed, one with soft constraints), eight solves each, reproduce the sequential results bit for bit"""
    import threading
    from acados_amd import AcadosOcpQpOptions, AcadosOcpQpSolver
    from acados_amd.generators import mass_spring_qp
    qps = [mass_spring_qp(N=15), load_qp("casadi_qp_tests/pendulum_slack.json"), mass_spring_qp(N=12),
           load_qp("qp_test/last_qp_one_sided_test.json")]
    condN = [5, None, None, None]

    def run(i, out):
        opts = AcadosOcpQpOptions()
        opts.tol_stat = opts.tol_eq = opts.tol_ineq = opts.tol_comp = 1e-8
        if condN[i]:
            opts.cond_N = condN[i]
        s = AcadosOcpQpSolver(qps[i], opts)
        res = []
        for _ in range(8):
            assert s.solve() == 0
            res.append(np.concatenate([s.get(k, "x") for k in range(qps[i].N + 1)] + [s.get(k, "lam", unique_duals=False) for k in range(qps[i].N + 1)]))
        out[i] = res

    seq, par = {}, {}
    for i in range(4):
        run(i, seq)
    th = [threading.Thread(target=run, args=(i, par)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in range(4):
        assert len(par.get(i, [])) == 8
        for a, c in zip(seq[i], par[i]):
            assert np.array_equal(a, c), i


@pytest.mark.gpu
def test_two_rows_per_lane_family_gpu(gpu_lib, monkeypatch):
    """ipm_kernels_w16r.hpp / ipm_kernels_w16t.hpp on the device (LDS-DMA staging, factor sweep on 4 x 4 MFMA tiles, dense
    list of the live instances): the nx=24 nu=6 class and the condensed C3 shape against the oracle, against the factor
    sweep on register rows (ACADOS_AMD_W16T=0) and against the wave-per-instance kernels on the same batch -- equal
    iteration counts, iterates equal to rounding; the dense-list launches (ACADOS_AMD_W16_PERM) change nothing in the
    results.  Ragged batch: 4 k + 3 instances."""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, lqr_instance_qp, random_lqr_batch
    for nx, nu, N, B in ((24, 6, 12, 1027), (8, 15, 6, 515), (20, 5, 8, 259), (12, 3, 10, 2051)):   # (12,3): the one-row family
        data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=B, seed=11 + nx)
        sols = {}
        for tag, env in (("w16r", {"ACADOS_AMD_W16R": "1", "ACADOS_AMD_W16_PERM": "1", "ACADOS_AMD_W16T": "1"}),
                         ("noperm", {"ACADOS_AMD_W16R": "1", "ACADOS_AMD_W16_PERM": "0", "ACADOS_AMD_W16T": "1"}),
                         ("rows", {"ACADOS_AMD_W16R": "1", "ACADOS_AMD_W16_PERM": "1", "ACADOS_AMD_W16T": "0"}),
                         ("wpi", {"ACADOS_AMD_W16R": "0", "ACADOS_AMD_W16_PERM": "1", "ACADOS_AMD_W16T": "1"})):
            for k_, v_ in env.items():
                monkeypatch.setenv(k_, v_)
            gb = OcpQpGpuBatch(lqr_dims(N, nx, nu), B)
            fill_lqr_batch(gb, data, N)
            for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
                gb.opts_set(f, 1e-8)
            assert gb.solve() == 0
            if nx + nu > 16:
                assert gb.kernel_name.startswith("wpi-box(" if tag == "wpi" else "w16r-box<"), gb.kernel_name
            else:   # ACADOS_AMD_W16R does not concern this shape: "wpi" is a second dense-list run
                assert gb.kernel_name.startswith("w16-box<"), gb.kernel_name
            assert gb.res_compute().max() <= KKT_TOL
            if nx + nu > 16 and tag != "wpi":
                assert int(gb.scalar("w16_tiles")) == (0 if tag == "rows" else 1)
            sols[tag] = ([gb.get(f, k) for f in ("x", "u", "lam") for k in range(N + 1)] + [gb.get("pi", k) for k in range(N)],
                         gb.info("iter").copy())
            if tag == "w16r":
                for i in (0, B // 2, B - 1):
                    qp = lqr_instance_qp(data, i, N)
                    o = OracleQp(qp)
                    assert o.solve(default_opts(tol_stat=1e-8)) == 0
                    compare_with_oracle(lambda k, f: gb.get(f, k)[i], o, qp, 1e-8)
            del gb
        assert np.array_equal(sols["w16r"][1], sols["wpi"][1]) and np.array_equal(sols["w16r"][1], sols["noperm"][1])
        assert np.array_equal(sols["w16r"][1], sols["rows"][1])
        for a, b_ in zip(sols["w16r"][0], sols["noperm"][0]):
            assert np.array_equal(a, b_)
        for other in ("wpi", "rows"):
            for a, b_ in zip(sols["w16r"][0], sols[other][0]):
                if a.size:
                    np.testing.assert_allclose(a, b_, rtol=1e-9, atol=1e-9)


@pytest.mark.gpu
def test_condensing_kernel_pairs_agree_gpu(gpu_lib, monkeypatch):
    """C3 shape: condensing on the FP64 matrix pipe (v_mfma_f64_4x4x4_4b_f64, the default) + expansion one instance per lane
    against the same contraction on register rows (DPP broadcasts) and against the run-time-shaped wave-per-instance pair:
    same condensed solve, expanded solutions equal to rounding; batch 4 k + 1 (a group with three instances beyond the batch)"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch
    N, B = 50, 20481
    data = random_lqr_batch(N=N, batch=B, seed=2)
    sols = []
    for z, mf, le, want in (("1", "1", "1", (3, 1)), ("1", "0", "1", (2, 1)), ("0", "1", "0", (0, 0))):
        monkeypatch.setenv("ACADOS_AMD_PCOND_W16", z)
        monkeypatch.setenv("ACADOS_AMD_PCOND_MFMA", mf)
        monkeypatch.setenv("ACADOS_AMD_PCOND_LANE_EXPAND", le)
        gb = OcpQpGpuBatch(lqr_dims(N, 8, 3), B)
        fill_lqr_batch(gb, data, N)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            gb.opts_set(f, 1e-8)
        gb.opts_set("cond_N", 10)
        assert gb.solve() == 0
        assert (int(gb.scalar("pcond_kernel")), int(gb.scalar("pexpand_kernel"))) == want
        assert gb.res_compute().max() <= 2e-8
        sols.append(([gb.get(f, k) for f in ("x", "u", "lam") for k in range(N + 1)] + [gb.get("pi", k) for k in range(N)],
                     gb.info("iter").copy()))
        del gb
    for other in sols[1:]:
        assert np.array_equal(sols[0][1], other[1])
        for a, b_ in zip(sols[0][0], other[0]):
            if a.size:
                np.testing.assert_allclose(a, b_, rtol=1e-7, atol=1e-9)


@pytest.mark.gpu
def test_finished_lanes_ride_along_gpu(gpu_lib, monkeypatch):
    """one-instance-per-lane box sweeps: lanes whose instance has finished ride along while a lane of their wave iterates
    (full-line stores); the instances that finish first in each of the three waves -- the longest riders -- and a few
    others end bit-identical to the same instance solved alone"""
    from conftest import check_finished_lanes_ride_along
    monkeypatch.setenv("ACADOS_AMD_WPI", "0")   # one instance per lane whatever the batch size
    N, B = 20, 192
    it = check_finished_lanes_ride_along(gpu_lib, N, B, 7, alone=list(range(0, B, 17)))
    first = [int(np.argmin(it[w * 64:(w + 1) * 64])) + w * 64 for w in range(3)]
    last = [int(np.argmax(it[w * 64:(w + 1) * 64])) + w * 64 for w in range(3)]
    check_finished_lanes_ride_along(gpu_lib, N, B, 7, alone=first + last)


@pytest.mark.gpu
def test_whole_solve_in_one_launch_gpu(gpu_lib, monkeypatch):
    """small batches: the whole solve in one launch (kx_solve) is the launch-per-sweep loop of the same kernels bit for
    bit -- hard and soft box rows, a single QP, 64 different C2-shaped instances (16 waves, rows of a wave finish at
    different iterations), 250 instances (the largest batches that take this path)"""
    from conftest import check_whole_solve_in_one_launch
    from acados_amd.generators import lqr_instance_qp, mass_spring_qp, random_lqr_batch
    from random_qp import random_structure_qp
    monkeypatch.setenv("ACADOS_AMD_WPI", "1")   # (tests/conftest.py switches the small-batch rule off)
    sets = [[random_structure_qp(seed, allow_general=False)] * (1 + seed % 7) for seed in range(24)]
    data = random_lqr_batch(N=30, nx=8, nu=3, batch=250, seed=11)
    sets.append([lqr_instance_qp(data, i, 30) for i in range(64)])
    sets.append([lqr_instance_qp(data, i, 30) for i in range(250)])
    sets.append([mass_spring_qp(N=20)])
    used = check_whole_solve_in_one_launch(gpu_lib, sets)
    assert used.get("w16-box", 0) >= 4 and used.get("w16-soft", 0) >= 4, used


@pytest.mark.gpu
def test_w16r_repeatability_gpu(gpu_lib):
    """the LDS-DMA / two-rows-per-lane kernels under repetition and co-residency (round-2 review, item 10): C3 at the
    BASELINE size (65,536 instances, N2 = 10, condensed shape nx = 8 nu = 15 on `w16r-box`) solved 20 times from a cold start
    while a SECOND batch (the nx = 24 nu = 6 class, same family, its own stream, driven from another host thread) keeps
    the chip busy with waves of the same kernels: every repeat must give status 0 on every instance, the independently
    recomputed KKT residual of every instance within tolerance, and BIT-IDENTICAL solutions and iteration counts (the
    placement of an instance in the dense list of live instances varies from run to run, its arithmetic must not).
    A development build with two waves of the forward sweep per SIMD fails exactly this test."""
    import threading
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch
    N, B = 50, 65536
    data = random_lqr_batch(N=N, batch=B, seed=3)
    gb = OcpQpGpuBatch(lqr_dims(N, 8, 3), B)
    fill_lqr_batch(gb, data, N)
    side_d = random_lqr_batch(N=20, nx=24, nu=6, batch=4096, seed=9)
    side = OcpQpGpuBatch(lqr_dims(20, 24, 6), 4096)
    fill_lqr_batch(side, side_d, 20)
    for g in (gb, side):
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            g.opts_set(f, 1e-8)
    gb.opts_set("cond_N", 10)
    assert side.solve() == 0
    side_ref = side.get("x", 10).copy()
    stop, side_bad = threading.Event(), []

    def churn():
        while not stop.is_set():
            if side.solve() != 0 or not np.array_equal(side.get("x", 10), side_ref):
                side_bad.append(1)

    # ... and a FOREIGN kernel -- a plain streaming copy of 2 GB on a third stream, small register footprint: its waves take
    # the free slots of the SIMDs the two-rows waves run on (round-3 review: the co-resident waves of the C5 leg are not only
    # this family's).  What LDS-DMA and partial vmcnt waits do under such co-residency: profiles/r04_vmcnt_probe.txt
    import torch
    fstream = torch.cuda.Stream()
    fa = torch.zeros(1 << 28, dtype=torch.float64, device="cuda")
    fb = torch.empty_like(fa)

    def foreign():
        with torch.cuda.stream(fstream):
            while not stop.is_set():
                for _ in range(8):
                    fb.copy_(fa, non_blocking=True)
                fstream.synchronize()

    th = threading.Thread(target=churn)
    th.start()
    tf = threading.Thread(target=foreign)
    tf.start()
    try:
        ref = None
        for rep in range(20):
            assert gb.solve() == 0, rep
            assert (gb.condensed_kernel_name() or "").startswith("w16r-box<NX=8,NU=15>")
            res = gb.res_compute()
            assert res.shape == (B, 4) and float(res.max()) <= 2e-8, (rep, float(res.max()))
            sol = (np.concatenate([gb.get("x", k).ravel() for k in (0, N // 2, N)] + [gb.get("u", k).ravel() for k in (0, N - 1)]),
                   gb.info("iter").copy())
            if ref is None:
                ref = sol
            assert np.array_equal(ref[0], sol[0]) and np.array_equal(ref[1], sol[1]), rep
    finally:
        stop.set()
        tf.join()
        stop.set()
        th.join()
    assert not side_bad


@pytest.mark.gpu
@pytest.mark.parametrize("xbox", [False, True])
def test_small_block_pipeline_gpu(gpu_lib, monkeypatch, xbox):
    """nx = 4, nu = 1 at the per-GPU share of a C5 class (7,281 instances, N = 20) and at 16,384 with bounds on the states:
    the default dispatch puts them on the pipelined one-instance-per-lane kernels (ipm_kernels_box_small.hpp); against the
    phase-ordered kernels of ipm_kernels_box.hpp (same arithmetic in the same order -- bit-identical under the host
    simulation, test_small_block_pipeline_bit_identical_hostsim; on the device hipcc contracts the multiply-adds of the two
    code shapes differently): identical iteration counts, outputs within 1e-11 (measured 7e-14 absolute); iteration counts
    equal to the sixteen-lanes family's; a sample against the oracle at 1e-8"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, lqr_instance_qp, random_lqr_batch
    nx, nu, N = 4, 1, 20
    B = 16384 if xbox else 7281
    monkeypatch.delenv("ACADOS_AMD_WPI_BATCH_MAX", raising=False)      # the default dispatch is what is tested
    data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=B, seed=77)
    out = {}
    for fam in ("small", "phases", "w16"):
        monkeypatch.setenv("ACADOS_AMD_KB_SMALL", "0" if fam == "phases" else "1")
        if fam == "w16":
            monkeypatch.setenv("ACADOS_AMD_WPI", "1")
        d = lqr_dims(N, nx, nu)
        if xbox:
            d.nbx[1:] = nx
            d.nb[:] = d.nbu + d.nbx
        gb = OcpQpGpuBatch(d, B)
        fill_lqr_batch(gb, data, N)
        if xbox:
            for k in range(1, N + 1):
                gb.set("lbx", k, np.full((B, nx), -50.0)); gb.set("ubx", k, np.full((B, nx), 50.0))
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            gb.opts_set(f, 1e-8)
        assert gb.solve() == 0
        assert gb.kernel_name.startswith("w16-box" if fam == "w16" else "1tpi-%s<NX=4,NU=1,XBOX=%d" % ("pipe" if fam == "small" else "box", int(xbox))), gb.kernel_name
        assert gb.res_compute().max() <= KKT_TOL
        out[fam] = ([gb.get(f, k) for f in ("x", "lam", "t") for k in range(N + 1)] + [gb.get(f, k) for f in ("u", "pi") for k in range(N)]
                    + [gb.info("iter")])
        del gb
    assert np.array_equal(out["small"][-1], out["phases"][-1])
    for a, b in zip(out["small"][:-1], out["phases"][:-1]):
        assert a.shape == b.shape and (a.size == 0 or np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) <= 1e-11)
    assert np.max(np.abs(out["small"][-1] - out["w16"][-1])) <= (1 if xbox else 0)
    if not xbox:
        worst = 0.0
        for i in range(0, B, 227):
            o = OracleQp(lqr_instance_qp(data, i, N))
            assert o.solve(default_opts(tol_stat=1e-8)) == 0
            assert out["small"][-1][i] == o.iter
            for k in range(N + 1):
                rx = o.get(k, "x")
                worst = max(worst, np.max(np.abs(out["small"][k][i] - rx) / np.maximum(1.0, np.abs(rx))))
        assert worst <= 1e-8, worst


@pytest.mark.gpu
def test_concurrent_shape_classes_gpu(gpu_lib, monkeypatch):
    """acados_amd/shape_classes.py on the device: three C5 classes (three kernel families) solved from one host thread
    each, the class with the most work on a high-priority HIP stream; every output bit-identical to the same batches
    solved one after the other, over several concurrent repeats"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch
    from acados_amd.shape_classes import ConcurrentClasses
    monkeypatch.delenv("ACADOS_AMD_WPI_BATCH_MAX", raising=False)
    classes = [(4, 1, 50, 7281), (12, 3, 20, 3000), (24, 6, 20, 3000)]
    batches, ref = [], []
    for ci, (nx, nu, N, B) in enumerate(classes):
        data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=B, seed=300 + ci)
        gb = OcpQpGpuBatch(lqr_dims(N, nx, nu), B)
        fill_lqr_batch(gb, data, N)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            gb.opts_set(f, 1e-8)
        assert gb.solve() == 0
        ref.append([gb.get(f, k).copy() for f in ("x", "lam") for k in range(N + 1)] + [gb.info("iter").copy()])
        batches.append(gb)
    assert [b.kernel_name.split("<")[0] for b in batches] == ["1tpi-pipe", "w16-box", "w16r-box"]
    with ConcurrentClasses(batches) as cc:
        for _ in range(3):
            assert cc.solve() == 0
            for gb, r, (nx, nu, N, B) in zip(batches, ref, classes):
                got = [gb.get(f, k) for f in ("x", "lam") for k in range(N + 1)] + [gb.info("iter")]
                for a, b in zip(got, r):
                    assert np.array_equal(a, b)
                assert gb.res_compute().max() <= KKT_TOL


def test_bulk_blob_whole_and_in_chunks_gpu(gpu_lib):
    """_get_bulk_in / _set_bulk / _set_bulk_chunk + _set_bulk_staged through the C-ABI on the device (tests/conftest.py::bulk_chunk_case)"""
    bulk_chunk_case(None)


def test_structure_fuzz_gpu(gpu_lib):
    """a slice of tools/fuzz_parity.py in the GPU tier: 48 random structures (four size classes up to nx = 40) through the default
    dispatch, both forced families and the condensed path with 70 copies each, a hot start and the RTI split per structure, and a
    batch of 1,536 perturbed instances per structure (tail switch, live-instance permutation, redo pair) with six instances
    against the oracle on the data read back from the device.  Asserted: none of the disagreements that mean a defect (an
    exception, copies of one QP that differ, a hot start that iterates, an RTI pair that is not the plain solve, an independent
    residual above the tolerance); listed but tolerated: fields a few 1e-7 beyond the comparison's tolerance, a slowest instance two
    iterations from the oracle, instances at MAXITER (limit cycles the oracle has too: DESIGN 3)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    fails, fams = fz.run("gpu", 7000, 7048)
    kinds = {}
    for f in fails:
        kinds.setdefault(fz.kind(f), []).append(f)
    print("structure fuzz, seeds 7000-7047:", {k: len(v) for k, v in kinds.items()}, "families", len(fams))
    bad = {k: v for k, v in kinds.items() if k in ("hot", "rti", "residual", "copies", "error")}
    assert not bad, bad
    assert len(fams) >= 8, fams
    assert len(kinds.get("maxiter", [])) <= 6 and len(kinds.get("tolerance", [])) <= 12, kinds


def test_conditional_corrector_ends_a_limit_cycle_gpu(gpu_lib):
    """the GPU-tier twin of tests/test_host_logic.py::test_conditional_corrector_ends_a_limit_cycle_hostsim: the five instances that
    cycled before the conditional corrector applied HPIPM's test, on the one-instance-per-lane and the sixteen-lanes kernels of
    the device, against the oracle (iteration counts within one, solution at 1e-7); without the conditional corrector they still
    end at MAXITER"""
    limit_cycle_case(None)


def test_device_against_certified_dense_solutions_gpu(gpu_lib, monkeypatch):
    """The device against a reference NEITHER the oracle NOR the kernels have a part in: tests/dense_ref.py::solve_exact -- a dense
    path-following + active-set solve of the stacked QP whose result carries its own optimality certificate (every inactive row
    feasible, every active multiplier >= 0, stationarity ~1e-12) -- on one or two instances of every kernel family the BASELINE
    configurations run on: C2 (one instance per lane, and the same instances on the sixteen-lanes kernels), the condensed C3 path,
    an nx = 12 and an nx = 24 class of C5 (one / two rows per lane, MFMA tile factor), the multi-phase class, the C4 class and the
    ng = 8 chain class (GEN).  The device runs at tight tolerances (complementarity 1e-12): what remains is the distance of an
    interior point to the vertex: measured 2e-12 ... 3e-11 relative on the primal solution, asserted at 1e-9."""
    from dense_ref import solve_exact, split
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import chain_soft_qp, lqr_instance_qp, mass_spring_qp, multiphase_batch, multiphase_instance_qp, random_lqr_batch
    c2 = random_lqr_batch(N=50, nx=8, nu=3, batch=2, seed=0)
    d12 = random_lqr_batch(N=20, nx=12, nu=3, batch=1, seed=203)
    d24 = random_lqr_batch(N=20, nx=24, nu=6, batch=1, seed=206)
    dm = multiphase_batch(N=20, batch=1)
    cases = [("C2 1tpi", [lqr_instance_qp(c2, i, 50) for i in range(2)], {"ACADOS_AMD_WPI": "0"}, 0, "1tpi-box"),
             ("C2 w16", [lqr_instance_qp(c2, i, 50) for i in range(2)], {"ACADOS_AMD_WPI": "1"}, 0, "w16-box"),
             ("C3 condensed", [lqr_instance_qp(c2, i, 50) for i in range(2)], {"ACADOS_AMD_WPI": "0"}, 10, "1tpi-box"),
             ("nx=12", [lqr_instance_qp(d12, 0, 20)], {}, 0, "w16-box<NX=12"),
             ("nx=24", [lqr_instance_qp(d24, 0, 20)], {}, 0, "w16r-box<NX=24"),
             ("multi-phase", [multiphase_instance_qp(dm, 0)], {}, 0, "w16-box<NX=12"),
             ("C4 class", [chain_soft_qp(1, N=10)], {}, 0, "w16r-gen<NX=24,NU=3,NG=4>"),
             ("ng=8 chain class", [chain_soft_qp(2, N=10, ng=8)], {}, 0, "w16r-gen<NX=24,NU=3,NG=8>"),
             # the dense path of full condensing (dense_kernels.hpp): C2 (161 dense columns) and the mass-spring class, whose state
             # bounds behind stage 0 become rows of the state map
             ("C2 full condensing, dense path", [lqr_instance_qp(c2, i, 50) for i in range(2)], {"ACADOS_AMD_WPI": "0"}, -1, "1tpi-box"),
             ("mass-spring N=20 full condensing, dense path", [mass_spring_qp(N=20)], {}, -1, "")]
    worst = {}
    exact = {}      # (the certified solve of a QP is the expensive part: C2's three cases share theirs)
    c2_qps = cases[0][1]
    cases = [(nm, c2_qps if nm.startswith("C2") or nm.startswith("C3") else q, e, c, f) for nm, q, e, c, f in cases]
    for name, qps, env, cond, fam in cases:
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        b = OcpQpGpuBatch.from_qps(qps)
        for f, v in (("tol_stat", 1e-9), ("tol_eq", 1e-11), ("tol_ineq", 1e-11), ("tol_comp", 1e-12)):
            b.opts_set(f, v)
        b.opts_set("iter_max", 100)
        if cond > 0:
            b.opts_set("cond_N", cond)
        if cond < 0:
            b.opts_set("full_dense", 1)
        assert b.solve() == 0, name
        assert b.kernel_name.startswith(fam), (name, b.kernel_name)
        if cond > 0:
            assert int(b.scalar("cond_N_active")) == cond
        for i, qp in enumerate(qps):
            if id(qp) not in exact:
                exact[id(qp)] = solve_exact(qp)
            w, off, info = exact[id(qp)]
            assert info["cert"] <= 1e-10 and info["stationarity"] <= 1e-9, (name, info)
            sol = split(qp, w, off)
            e = 0.0
            for k in range(qp.N + 1):
                for f in ("x", "u", "sl", "su"):
                    ref = sol[f][k]
                    if ref.size:
                        e = max(e, float(np.max(np.abs(b.get(f, k)[i][:ref.size] - ref) / np.maximum(1.0, np.abs(ref)))))
            worst[name] = max(worst.get(name, 0.0), e)
        for k_ in env:
            monkeypatch.delenv(k_)
    print("device vs certified dense solutions (rel. primal):", {k: f"{v:.1e}" for k, v in worst.items()})
    assert max(worst.values()) <= 1e-9, worst


@pytest.mark.gpu
def test_device_against_certified_solutions_on_random_structures_gpu(gpu_lib):
    """conftest.certified_random_structures_case on the device: sixty random structures (general rows, shared slacks, one-sided / masked
    rows, free x0, per-stage dims), every kernel family they are dispatched to, against the certified dense solutions"""
    from conftest import certified_random_structures_case
    worst = certified_random_structures_case(None, range(60))
    print("device vs certified dense solutions, 60 random structures:", {k: f"{v:.1e}" for k, v in worst.items()})
    assert len(worst) >= 2


@pytest.mark.gpu
def test_full_condensing_dense_path_c2_gpu(gpu_lib):
    """FULL CONDENSING of the C2 shape (N = 50, nx = 8, nu = 3: 158 condensed columns + the padded terminal inputs, 300 inequality sides)
    on the dense path (option full_dense, dense_kernels.hpp; VERDICT r05 missing 1): 512 instances, every one converged, the independent
    KKT kernel on the ORIGINAL QP, 16 instances against the oracle; the rate is printed beside the stage-wise one (correctness first)"""
    import time
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, lqr_instance_qp, random_lqr_batch
    N, B = 50, 512
    data = random_lqr_batch(N=N, batch=B, seed=0)
    gb = OcpQpGpuBatch(lqr_dims(N, 8, 3), B)
    fill_lqr_batch(gb, data, N)
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        gb.opts_set(f, 1e-8)
    gb.solve()
    t0 = time.perf_counter(); assert gb.solve() == 0; t_stage = time.perf_counter() - t0
    it_stage = gb.info("iter").copy()
    gb.opts_set("full_dense", 1)
    gb.solve()
    t0 = time.perf_counter(); bad = gb.solve(); t_dense = time.perf_counter() - t0
    assert bad == 0, gb.info("status")
    assert int(gb.scalar("dense_columns")) == 51 * 3 + 8
    assert gb.res_compute().max() <= 1e-8 * (1.0 + 1e-3) + 1e-12
    assert np.abs(gb.info("iter") - it_stage).max() <= 4
    worst = 0.0
    for i in range(0, B, 32):
        o = OracleQp(lqr_instance_qp(data, i, N))
        assert o.solve(default_opts(tol_stat=1e-8, tol_eq=1e-8, tol_ineq=1e-8, tol_comp=1e-8)) == 0
        for k in range(N + 1):
            for f in ("x", "u") if k < N else ("x",):
                ref = o.get(k, f)
                worst = max(worst, float(np.max(np.abs(gb.get(f, k)[i] - ref) / np.maximum(1.0, np.abs(ref)))))
    print(f"full condensing, dense path: C2 shape, {B} instances {t_dense * 1e3:.1f} ms = {B / t_dense:.3g} solves/s (stage-wise on the same batch "
          f"{t_stage * 1e3:.1f} ms = {B / t_stage:.3g}); max rel primal difference to the oracle at the same tolerance {worst:.2e}")
    assert worst <= 1e-4          # two iterate paths inside the 1e-8 ball; the residual assertion above is the sharp one
    gb.opts_set("full_dense", 0)
    assert gb.solve() == 0
