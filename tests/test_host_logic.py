"""CPU tier: host logic + kernel logic through the C-ABI, with the kernel sources compiled by
g++ against the host-simulation shim (tests/hostsim).  The checker is the oracle.  Also checks
that the PRODUCT library (hipcc build) loads and exports every symbol the headers declare --
without calling into it (no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import (GOLDEN_PAIRS, bulk_chunk_case, certified_random_structures_case, limit_cycle_case, INPUT_ONLY, ROOT, compare_condensed_with_oracle, compare_with_oracle, fold_stage0, load_qp,
                      load_sol)
from oracle.oracle import OracleQp, default_opts

ALL_QPS = [p for p, _ in GOLDEN_PAIRS] + INPUT_ONLY


def test_product_library_exports_declared_symbols():
    """every function declared in include/acados_amd/*.h is exported by libacados_amd_qp.so"""
    from acados_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "HIP library not built: run __graft_entry__.build()"
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = set()
    for h in ("ocp_qp_gpu_batch.h", "ocp_qp_interface.h"):
        src = open(os.path.join(ROOT, "include", "acados_amd", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(ocp_qp_[a-z0-9_]+)\s*\(", src))
    assert len(names) > 50
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, f"symbols declared but not exported: {missing}"


def test_c_abi_headers_are_plain_c():
    """include/*.h is what a C maintainer binds: both headers must compile as C11 on their own"""
    import subprocess
    for h in ("ocp_qp_gpu_batch.h", "ocp_qp_interface.h"):
        src = f'#include "acados_amd/{h}"\nint main(void) {{ return 0; }}\n'
        r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-x", "c", "-"],
                           input=src.encode(), capture_output=True)
        assert r.returncode == 0, r.stderr.decode()


def test_product_path_has_no_cpu_fallback():
    """the package loader only knows the HIP library; a missing library raises"""
    from acados_amd import _lib
    src = open(_lib.__file__).read()
    assert "hostsim" not in src and "oracle" not in src
    for mod in ("gpu_batch.py", "ocp_qp_solver.py", "ocp_qp.py", "generators.py", "__init__.py"):
        s = open(os.path.join(ROOT, "acados_amd", mod)).read()
        assert "import oracle" not in s and "from oracle" not in s and "hostsim" not in s


@pytest.mark.parametrize("qp_file", ALL_QPS)
def test_batch_abi_matches_oracle_hostsim(hostsim_lib, qp_file):
    from acados_amd import OcpQpGpuBatch
    qp = load_qp(qp_file)
    o = OracleQp(qp)
    assert o.solve(default_opts(iter_max=100, tol_stat=1e-8)) == 0
    b = OcpQpGpuBatch.from_qps([qp] * 3, _clib=hostsim_lib)
    b.opts_set("tol_stat", 1e-8)
    b.opts_set("iter_max", 100)
    assert b.solve() == 0
    assert np.all(b.info("iter") == o.iter)
    for inst in range(3):
        compare_with_oracle(lambda k, f: b.get(f, k)[inst], o, qp, 1e-9, fields=("x", "u", "sl", "su", "pi", "lam", "t"))
    assert max(b.info(n).max() for n in ("res_stat", "res_eq", "res_ineq", "res_comp")) <= 1e-8


@pytest.mark.parametrize("qp_file,sol_file", GOLDEN_PAIRS)
def test_acados_api_golden_hostsim(hostsim_lib, qp_file, sol_file):
    """the reference's own test (test_ocpqp_solver.py:15-53) read against this backend"""
    from acados_amd import AcadosOcpQpOptions, AcadosOcpQpSolver
    qp, sol = load_qp(qp_file), load_sol(sol_file)
    opts = AcadosOcpQpOptions()
    opts.iter_max = 500
    solver = AcadosOcpQpSolver(qp, opts=opts, _clib=hostsim_lib)
    assert solver.solve() == 0
    tol = 1e-5
    for stage in range(qp.N + 1):
        ref = sol.get(f"lam_{stage}", np.zeros(0))
        if ref.size:
            assert np.allclose(solver.get(stage, "lam"), ref, atol=tol)
        if stage < qp.N:
            assert np.allclose(solver.get(stage, "pi"), sol[f"pi_{stage}"], atol=tol)
    assert solver.get_stats("iter") > 0
    st = solver.get_stats("statistics")
    assert st.shape == (solver.get_stats("iter") + 1, 20) and st[-1, 7] <= 1e-6


def test_mass_spring_and_padding_hostsim(hostsim_lib):
    """C1 (mass_spring_qp.c restated): residual <= 1e-8 like test_qpsolvers.cpp:238-251; also
    exercises dimension padding (nu_N = 0 inside an NU=3 kernel)."""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import mass_spring_qp
    for N in (15, 20):
        qp = mass_spring_qp(N=N)
        o = OracleQp(qp)
        assert o.solve(default_opts(tol_stat=1e-8)) == 0
        b = OcpQpGpuBatch.from_qps([qp], _clib=hostsim_lib)
        b.opts_set("tol_stat", 1e-8)
        assert b.solve() == 0
        assert b.kernel_name == "1tpi-box<NX=8,NU=3,XBOX=1>"
        compare_with_oracle(lambda k, f: b.get(f, k)[0], o, qp, 1e-9)
        assert max(b.info(n).max() for n in ("res_stat", "res_eq", "res_ineq", "res_comp")) <= 1e-8


def test_general_kernels_on_box_qps_hostsim(hostsim_lib, monkeypatch):
    """box-only QPs are normally served by the fast-path kernels; the general kernels must give
    the same answer (ACADOS_AMD_GENERAL_KERNELS=1 is a debugging switch of the library)"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import mass_spring_qp
    qp = mass_spring_qp(N=10)
    fast = OcpQpGpuBatch.from_qps([qp], _clib=hostsim_lib)
    assert fast.solve() == 0 and fast.kernel_name.startswith("1tpi-box")
    monkeypatch.setenv("ACADOS_AMD_GENERAL_KERNELS", "1")
    gen = OcpQpGpuBatch.from_qps([qp], _clib=hostsim_lib)
    assert gen.solve() == 0 and gen.kernel_name == "1tpi<NX=8,NU=3,NG=0,NS=0>"
    for k in range(qp.N + 1):
        for f in ("x", "u", "lam", "t") + (("pi",) if k < qp.N else ()):
            assert np.allclose(fast.get(f, k), gen.get(f, k), rtol=1e-10, atol=1e-12), (f, k)
    assert fast.info("iter")[0] == gen.info("iter")[0]


def test_lqr_batch_and_ragged_hostsim(hostsim_lib):
    """C2-shaped instances at small N; batch size not a multiple of 64; per-instance results"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, lqr_instance_qp, random_lqr_batch
    N, B = 8, 70
    data = random_lqr_batch(N=N, batch=B, seed=5)
    gb = OcpQpGpuBatch(lqr_dims(N, 8, 3), B, _clib=hostsim_lib)
    fill_lqr_batch(gb, data, N)
    gb.opts_set("tol_stat", 1e-8)
    assert gb.solve() == 0
    # the last survivors (<= a quarter of the level) finish on the wave-per-instance kernels
    assert gb.kernel_name.startswith("1tpi") and int(gb.scalar("tail_switches")) == 1
    # ... and their statistics rows are merged back: the table of the slowest of the first 64 instances is
    # complete (mu and the residual norms of every iteration, converged values in the last row)
    it = gb.info("iter")
    slow = int(np.argmax(it[:64]))
    st = gb.stat(slow)
    assert st.shape[0] >= it[slow] + 1 and np.all(st[: it[slow] + 1, 6] > 0.0)
    assert st[it[slow], 7] <= 1e-8 and st[it[slow], 10] <= 1e-8
    x = [gb.get("x", k) for k in range(N + 1)]
    u = [gb.get("u", k) for k in range(N)]
    for i in (0, 1, 63, 64, 69):
        o = OracleQp(lqr_instance_qp(data, i, N))
        assert o.solve(default_opts(tol_stat=1e-8)) == 0
        for k in range(N + 1):
            assert np.allclose(x[k][i], o.get(k, "x"), atol=1e-9)
            if k < N:
                assert np.allclose(u[k][i], o.get(k, "u"), atol=1e-9)
        assert gb.info("iter")[i] == o.iter


def test_error_behaviour_hostsim(hostsim_lib):
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import lqr_dims
    gb = OcpQpGpuBatch(lqr_dims(4, 8, 3), 2, _clib=hostsim_lib)
    with pytest.raises(ValueError):
        gb.opts_set("no_such_option", 1)
    with pytest.raises(ValueError):
        gb.set("no_such_field", 0, np.zeros((2, 3)))
    # nx=40 has no one-instance-per-lane instantiation: served by the wave-per-instance family (nu+nx <= 64)
    assert OcpQpGpuBatch(lqr_dims(4, 40, 3), 2, _clib=hostsim_lib).kernel_name.startswith("wpi-box(nx=40,nu=3")
    with pytest.raises(RuntimeError):
        OcpQpGpuBatch(lqr_dims(4, 70, 3), 2, _clib=hostsim_lib)  # nothing covers nu+nx > 64


def test_maxiter_status_hostsim(hostsim_lib):
    """status codes are acados' (types.h:74-87): iter_max reached -> ACADOS_MAXITER (2)"""
    from acados_amd import OcpQpGpuBatch
    qp = load_qp("casadi_qp_tests/pendulum_qp.json")
    b = OcpQpGpuBatch.from_qps([qp], _clib=hostsim_lib)
    b.opts_set("iter_max", 2)
    b.opts_set("tol_stat", 1e-12)
    assert b.solve() == 1
    assert b.info("status")[0] == 2 and b.info("iter")[0] == 2


def _check_batch_vs_oracle(qps, lib, tol=1e-8):
    from acados_amd import OcpQpGpuBatch
    b = OcpQpGpuBatch.from_qps(qps, _clib=lib)
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        b.opts_set(f, 1e-8)
    assert b.solve() == 0
    for i, qp in enumerate(qps):
        o = OracleQp(qp)
        assert o.solve(default_opts(tol_stat=1e-8)) == 0
        compare_with_oracle(lambda k, f: b.get(f, k)[i], o, qp, tol)
        assert abs(int(b.info("iter")[i]) - o.iter) <= 1
    return b


@pytest.mark.parametrize("wpi", ["0", "1"])
def test_c4_shape_general_constraints_and_slacks_hostsim(hostsim_lib, monkeypatch, wpi):
    """C4 (nx=24, nu=3, soft state bounds + soft general rows, ns=8) at a short horizon, on both kernel
    families (one instance per lane / one wave per instance, the default for this shape)"""
    from acados_amd.generators import chain_soft_qp
    monkeypatch.setenv("ACADOS_AMD_WPI", wpi)
    b = _check_batch_vs_oracle([chain_soft_qp(i, N=6) for i in range(2)], hostsim_lib)
    assert b.kernel_name.startswith("w16r-gen<NX=24,NU=3,NG=4>" if wpi == "1" else "1tpi<NX=24,NU=3,NG=4,NS=8>")
    if wpi == "1":   # the wave-per-instance GEN kernels the class ran on before (and still does where a slack is shared)
        monkeypatch.setenv("ACADOS_AMD_W16G", "0")
        b = _check_batch_vs_oracle([chain_soft_qp(i, N=6) for i in range(2)], hostsim_lib)
        assert b.kernel_name.startswith("wpi-gen(nx=24,nu=3,ng=4,ns=8")


def test_chain_class_with_eight_general_rows_hostsim(hostsim_lib, monkeypatch):
    """the "ng = 8 chain class" of the round-4 review (nx = 24, nu = 3, 3 hard input rows + 4 soft state rows + 8 soft general rows =
    15 inequality rows per stage, ns = 12): served by the two-rows-per-lane GEN kernels (w16r-gen<24,3,8>, factor sweep on register
    rows: the tile sweep of this instantiation spills) instead of the wave-per-instance ones; both against the oracle"""
    from acados_amd.generators import chain_soft_qp
    monkeypatch.setenv("ACADOS_AMD_WPI", "1")
    qps = [chain_soft_qp(i, N=4, ng=8) for i in range(5)]       # ragged: one full workgroup of four instances + one
    b = _check_batch_vs_oracle(qps, hostsim_lib)
    assert b.kernel_name.startswith("w16r-gen<NX=24,NU=3,NG=8>") and int(b.scalar("w16_tiles")) == 0
    it = b.info("iter").copy()
    monkeypatch.setenv("ACADOS_AMD_W16G", "0")
    b2 = _check_batch_vs_oracle(qps, hostsim_lib)
    assert b2.kernel_name.startswith("wpi-gen(nx=24,nu=3,ng=8,ns=12") and np.array_equal(it, b2.info("iter"))


def test_general_rows_and_slacks_at_small_shapes_on_sixteen_lanes_hostsim(hostsim_lib, monkeypatch):
    """general rows + slacks (one slack per row) at nu + nx <= 16 used to run on the wave-per-instance GEN kernels at every batch
    size; the two-rows GEN kernels instantiated at R = 1 row per lane (<12,4,4>, <8,3,4>; factor sweep on 4 x 4 MFMA tiles) take
    them, shapes padded inside the compiled ones included -- same solution as the oracle, same iteration counts as wpi-gen"""
    from acados_amd.generators import chain_soft_qp
    monkeypatch.setenv("ACADOS_AMD_WPI", "1")
    for (nx, nu, ng, nsx), want in (((12, 4, 4, 4), "w16r-gen<NX=12,NU=4,NG=4>"), ((10, 3, 4, 3), "w16r-gen<NX=12,NU=4,NG=4>"),
                                    ((8, 3, 2, 4), "w16r-gen<NX=8,NU=3,NG=4>")):
        qps = [chain_soft_qp(i, N=4, nx=nx, nu=nu, ng=ng, nsx=nsx) for i in range(5)]
        monkeypatch.setenv("ACADOS_AMD_W16G", "1")
        b = _check_batch_vs_oracle(qps, hostsim_lib)
        assert b.kernel_name == want and int(b.scalar("w16_tiles")) == 1, b.kernel_name
        it = b.info("iter").copy()
        monkeypatch.setenv("ACADOS_AMD_W16G", "0")
        b2 = _check_batch_vs_oracle(qps, hostsim_lib)
        assert b2.kernel_name.startswith("wpi-gen(") and np.array_equal(it, b2.info("iter"))
    # the golden shared-slack structure (nx = 4, nu = 1, two general rows on one slack) keeps its own dispatch: a slack shared by
    # several rows is not what these kernels carry
    monkeypatch.setenv("ACADOS_AMD_W16G", "1")
    b = _check_batch_vs_oracle([load_qp("casadi_qp_tests/pend_idxs_rev_min_qp0.json")] * 3, hostsim_lib)
    assert b.kernel_name.startswith("wpi-gen("), b.kernel_name


def test_more_than_64_inequality_sides_hostsim(hostsim_lib):
    """stages with 65..128 inequality sides (two activity words per stage, wave-per-instance kernels only):
    nx + nu = 40 with x0 as equality bounds (80 sides at stage 0), and partial condensing with blocks of 10
    stages (child stage: 30 input rows + 8 state rows = 76 sides); more than 128 sides are refused (NULL handle,
    message) instead of aborting"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import lqr_dims, lqr_instance_qp, random_lqr_batch
    data = random_lqr_batch(N=3, nx=32, nu=8, batch=2, seed=4)
    b = _check_batch_vs_oracle([lqr_instance_qp(data, i, 3) for i in range(2)], hostsim_lib)
    assert b.kernel_name.startswith("wpi-box(nx=32,nu=8")
    data = random_lqr_batch(N=20, batch=2, seed=6)
    qps = [lqr_instance_qp(data, i, 20) for i in range(2)]
    b = OcpQpGpuBatch.from_qps(qps, _clib=hostsim_lib)
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        b.opts_set(f, 1e-8)
    b.opts_set("cond_N", 2)
    assert b.solve() == 0 and int(b.scalar("cond_N_active")) == 2
    for i, qp in enumerate(qps):
        o = OracleQp(qp)
        assert o.solve(default_opts(tol_stat=1e-8)) == 0
        compare_with_oracle(lambda k, f: b.get(f, k)[i], o, qp, 1e-8, fields=("x", "u", "pi", "lam", "t"))
    d = lqr_dims(3, 40, 20)
    d.ng[0] = 8                                                     # 2 * (40 + 20 + 8) = 136 sides at stage 0
    with pytest.raises(RuntimeError):
        OcpQpGpuBatch(d, 2, _clib=hostsim_lib)


def test_tail_switch_with_general_rows_hostsim(hostsim_lib, monkeypatch):
    """a one-instance-per-lane level hands its last survivors to the wave-per-instance kernels; with
    equality-flagged x0 and general rows the multipliers of the fixed variables must come out of the family
    that finished the instance (regression: they were left at the value of the hand-over iteration)"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import chain_soft_qp
    monkeypatch.setenv("ACADOS_AMD_WPI", "0")
    qps = [chain_soft_qp(i, N=5) for i in range(8)]
    b = OcpQpGpuBatch.from_qps(qps, _clib=hostsim_lib)
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        b.opts_set(f, 1e-8)
    assert b.solve() == 0
    assert b.kernel_name.startswith("1tpi<") and int(b.scalar("tail_switches")) == 1
    for i, qp in enumerate(qps):
        o = OracleQp(qp)
        assert o.solve(default_opts(tol_stat=1e-8)) == 0
        compare_with_oracle(lambda k, f: b.get(f, k)[i], o, qp, 1e-7)


def test_wave_per_instance_general_rows_shared_slacks_hostsim(hostsim_lib, monkeypatch):
    """the reference's casadi QP fixtures (general constraints, slacks, one slack shared by several rows,
    masks) forced onto the wave-per-instance kernels: same solution as the oracle"""
    from acados_amd import OcpQpGpuBatch
    monkeypatch.setenv("ACADOS_AMD_WPI", "1")
    for rel in INPUT_ONLY:
        qp = load_qp(rel)
        o = OracleQp(qp)
        assert o.solve(default_opts(tol_stat=1e-8)) == 0
        b = OcpQpGpuBatch.from_qps([qp, qp], _clib=hostsim_lib)
        b.opts_set("tol_stat", 1e-8)
        assert b.solve() == 0 and b.kernel_name.startswith(("wpi-", "w16-"))
        compare_with_oracle(lambda k, f: b.get(f, k)[1], o, qp, 1e-7)
        assert int(b.info("iter")[1]) == o.iter


@pytest.mark.parametrize("fam", ["1tpi", "wpi", "w16"])
def test_c5_shape_classes_and_multiphase_hostsim(hostsim_lib, monkeypatch, fam):
    """C5: shape classes nx in {4,12,24} (nu = ceil(nx/4)) and the multi-phase class whose state
    dimension switches 12 -> 4 mid-horizon (per-stage dims inside one padded kernel shape).
    All three kernel families: one instance per lane (ACADOS_AMD_WPI=0), one wave per instance
    (ACADOS_AMD_WPI=1 ACADOS_AMD_W16=0) and sixteen lanes per instance (ACADOS_AMD_WPI=1; one row per lane for
    nu+nx <= 16, two rows per lane for nx=24 nu=6)."""
    from acados_amd.generators import lqr_instance_qp, multiphase_qp, random_lqr_batch
    monkeypatch.setenv("ACADOS_AMD_WPI", "0" if fam == "1tpi" else "1")
    monkeypatch.setenv("ACADOS_AMD_W16", "1" if fam == "w16" else "0")
    for nx, nu in ((4, 1), (12, 3), (24, 6)):
        data = random_lqr_batch(N=5, nx=nx, nu=nu, batch=2 if fam != "w16" else 5, seed=7)
        nb = 2 if fam != "w16" else 5   # 5 instances: a full 4-instance workgroup and a ragged one
        b = _check_batch_vs_oracle([lqr_instance_qp(data, i, 5) for i in range(nb)], hostsim_lib)
        want = {"1tpi": "1tpi-pipe<" if nx + nu <= 6 else "1tpi-box<", "wpi": "wpi-box(", "w16": "w16-box<" if nx + nu <= 16 else "w16r-box<"}[fam]
        assert b.kernel_name.startswith(want)
    b = _check_batch_vs_oracle([multiphase_qp(i, N=8) for i in range(3)], hostsim_lib)
    assert b.kernel_name.startswith({"1tpi": "1tpi-box<NX=12,NU=3", "wpi": "wpi-box(nx=12,nu=3", "w16": "w16-box<NX=12,NU=3>"}[fam])


def test_wave_per_instance_default_rule_hostsim(hostsim_lib, monkeypatch):
    """without the override: small stage blocks stay on the one-instance-per-lane kernels, nu+nx >= 13
    goes to the wave-per-instance family; a batch that is not a multiple of anything, per-instance
    iteration counts, Riccati getters and hot start on that family"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, lqr_instance_qp, random_lqr_batch
    N, B = 6, 5
    data = random_lqr_batch(N=N, nx=12, nu=3, batch=B, seed=11)
    monkeypatch.setenv("ACADOS_AMD_W16", "0")   # this test is about the run-time-shaped family
    gb = OcpQpGpuBatch(lqr_dims(N, 12, 3), B, _clib=hostsim_lib)
    assert gb.kernel_name.startswith("wpi-box(nx=12,nu=3")
    fill_lqr_batch(gb, data, N)
    gb.opts_set("tol_stat", 1e-8)
    assert gb.solve() == 0
    iters = gb.info("iter").copy()
    for i in range(B):
        qp = lqr_instance_qp(data, i, N)
        o = OracleQp(qp)
        assert o.solve(default_opts(tol_stat=1e-8)) == 0
        compare_with_oracle(lambda k, f: gb.get(f, k)[i], o, qp, 1e-8)
        assert abs(int(iters[i]) - o.iter) <= 1
        if i == 0:
            o.refactor()
            for k in (0, N):
                assert np.allclose(gb.get("ric_L", k)[i], o.get(k, "ric_L"), rtol=1e-6, atol=1e-8)
    small = OcpQpGpuBatch(lqr_dims(N, 8, 3), 2, _clib=hostsim_lib)
    assert small.kernel_name.startswith("1tpi")   # the test session switches the batch-size rule off (conftest)
    # batch-size rule: small stage blocks also go to the wave-per-instance kernels while the batch is small
    monkeypatch.setenv("ACADOS_AMD_WPI_BATCH_MAX", "100")
    assert OcpQpGpuBatch(lqr_dims(N, 8, 3), 100, _clib=hostsim_lib).kernel_name.startswith("wpi-box(nx=8,nu=3")
    monkeypatch.setenv("ACADOS_AMD_W16", "1")   # ... and, where one is compiled, to the 16-lanes-per-instance kernels
    assert OcpQpGpuBatch(lqr_dims(N, 8, 3), 100, _clib=hostsim_lib).kernel_name == "w16-box<NX=8,NU=3>"
    assert OcpQpGpuBatch(lqr_dims(N, 8, 3), 101, _clib=hostsim_lib).kernel_name.startswith("1tpi")
    monkeypatch.setenv("ACADOS_AMD_WPI_BATCH_MAX", "0")
    # hot start from the solution: converged at the first residual evaluation
    gb.opts_set("warm_start", 3)
    assert gb.solve() == 0
    assert int(gb.info("iter").max()) <= 1


def test_partial_condensing_hostsim(hostsim_lib, monkeypatch):
    """a5-a7: condensing N -> N2 blocks, IPM on the condensed QP, expansion; the expanded solution
    must equal the full-space oracle solution (x, u, pi, lam, t).  Covers C3 (N=50 -> 10 blocks of 5),
    uneven block sizes, the reference's golden pendulum QP and the RTI lhs/rhs split."""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import lqr_instance_qp, mass_spring_qp, random_lqr_batch

    def run(qps, cond_N, expect_active, split=False):
        b = OcpQpGpuBatch.from_qps(qps, _clib=hostsim_lib)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            b.opts_set(f, 1e-8)
        b.opts_set("cond_N", cond_N)
        if split:
            assert b.condense_lhs() == 0
            assert b.condense_rhs_and_solve() == 0
        else:
            assert b.solve() == 0
        assert int(b.scalar("cond_N_active")) == expect_active
        for i, qp in enumerate(qps):
            o = OracleQp(qp)
            assert o.solve(default_opts(tol_stat=1e-8)) == 0
            compare_with_oracle(lambda k, f: b.get(f, k)[i], o, qp, 1e-8, fields=("x", "u", "pi", "lam", "t"))
        return b

    data = random_lqr_batch(N=50, batch=2, seed=0)
    qps = [lqr_instance_qp(data, i, 50) for i in range(2)]
    b = run(qps, 10, 10)
    assert b.kernel_name.startswith("1tpi-box<NX=8,NU=3")
    # C3 shape, wave-tiled parent: condensing on 4 x 4 MFMA tiles (km_pcond), expansion one instance per lane
    assert b.scalar("pcond_kernel") == 3 and b.scalar("pexpand_kernel") == 1
    run(qps, 10, 10, split=True)
    monkeypatch.setenv("ACADOS_AMD_PCOND_MFMA", "0")   # the same contraction on register rows with DPP broadcasts (kz_pcond)
    b = run(qps, 10, 10)
    assert b.scalar("pcond_kernel") == 2 and b.scalar("pexpand_kernel") == 1
    run(qps, 10, 10, split=True)
    monkeypatch.delenv("ACADOS_AMD_PCOND_MFMA")
    monkeypatch.setenv("ACADOS_AMD_PCOND_W16", "0")    # the run-time-shaped wave-per-instance pair
    monkeypatch.setenv("ACADOS_AMD_PCOND_LANE_EXPAND", "0")
    b = run(qps, 10, 10)
    assert b.scalar("pcond_kernel") == 0 and b.scalar("pexpand_kernel") == 0
    run(qps, 10, 10, split=True)
    monkeypatch.delenv("ACADOS_AMD_PCOND_W16")
    monkeypatch.delenv("ACADOS_AMD_PCOND_LANE_EXPAND")
    monkeypatch.setenv("ACADOS_AMD_PCOND_1TPI", "1")   # the compiled one-instance-per-lane condensing kernels
    run(qps, 10, 10)
    monkeypatch.delenv("ACADOS_AMD_PCOND_1TPI")
    data = random_lqr_batch(N=10, batch=3, seed=3)
    run([lqr_instance_qp(data, i, 10) for i in range(3)], 3, 3)        # blocks of 4, 3, 3
    # second compiled shape of the sixteen-lanes condensing kernel (nx=4 nu=1, blocks of 4); five instances = one full
    # workgroup of four rows + one row with three rows beyond the batch
    data = random_lqr_batch(N=8, nx=4, nu=1, batch=5, seed=5)
    b = run([lqr_instance_qp(data, i, 8) for i in range(5)], 2, 2)
    assert b.scalar("pcond_kernel") == 3
    monkeypatch.setenv("ACADOS_AMD_PCOND_MFMA", "0")
    b = run([lqr_instance_qp(data, i, 8) for i in range(5)], 2, 2)
    assert b.scalar("pcond_kernel") == 2
    monkeypatch.delenv("ACADOS_AMD_PCOND_MFMA")
    data = random_lqr_batch(N=7, nx=4, nu=1, batch=3, seed=6)     # blocks of 4 and 3: a short block inside the compiled shape
    b = run([lqr_instance_qp(data, i, 7) for i in range(3)], 2, 2)
    assert b.scalar("pcond_kernel") == 3
    run([lqr_instance_qp(data, i, 7) for i in range(3)], 2, 2, split=True)
    monkeypatch.setenv("ACADOS_AMD_PCOND_MFMA", "0")
    b = run([lqr_instance_qp(data, i, 7) for i in range(3)], 2, 2)
    assert b.scalar("pcond_kernel") == 2
    run([lqr_instance_qp(data, i, 7) for i in range(3)], 2, 2, split=True)
    monkeypatch.delenv("ACADOS_AMD_PCOND_MFMA")
    run([load_qp("qp_test/last_qp_nonuniform_pendulum.json")], 3, 3)   # N=7 -> 3, 2, 2 ; x0 equality rows


def test_partial_condensing_general_rows_hostsim(hostsim_lib):
    """a5-a7 beyond the box class: state bounds inside a block become general rows of the condensed stage
    (coefficients = rows of the block's state-transfer matrix, bounds shifted by the free response), general rows are
    carried through the same substitution, slacks (also shared ones, idxs_rev) travel with their rows, one-sided
    bounds keep their activity bits; the expansion recovers pi of the eliminated dynamics including the inequality
    terms.  Mass-spring with N2 in {5, 3} is the reference's own unit-test setting (test_qpsolvers.cpp:117-268)."""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import chain_soft_qp, mass_spring_qp

    # two different iterate paths (condensed / full space) stopped at residuals <= 1e-8: multipliers of nearly
    # active rows (lam ~ 1e-6 and below: 4.5e-7 against 6.8e-8 in the C4-class case, both zero at the tolerance) agree to a few
    # 1e-7, hence 1e-6 here (2e-7 while both paths scaled their steps by 0.995: the paths ended closer together)
    def run(qp, cond_N, split=False, tol=1e-6):
        b = OcpQpGpuBatch.from_qps([qp, qp], _clib=hostsim_lib)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            b.opts_set(f, 1e-8)
        b.opts_set("cond_N", cond_N)
        if split:
            assert b.condense_lhs() == 0
            assert b.condense_rhs_and_solve() == 0
        else:
            assert b.solve() == 0
        assert int(b.scalar("cond_N_active")) == cond_N
        o = OracleQp(qp)
        assert o.solve(default_opts(tol_stat=1e-8)) == 0
        compare_with_oracle(lambda k, f: b.get(f, k)[1], o, qp, tol, fields=("x", "u", "sl", "su", "pi", "lam", "t"))

    run(mass_spring_qp(N=15), 5)
    run(mass_spring_qp(N=15), 3, split=True)
    run(load_qp("qp_test/last_qp_one_sided_test.json"), 5)              # one-sided state bounds inside the blocks
    run(load_qp("casadi_qp_tests/pendulum_slack.json"), 2)              # general rows + slacks, N = 5 -> 3, 2
    run(load_qp("casadi_qp_tests/pend_idxs_rev_min_qp0.json"), 3)       # slack shared by two rows
    run(chain_soft_qp(N=6, seed=2, i=0), 3)                             # C4 class: nx = 24, soft state bounds + general rows


def test_compaction_is_bit_identical_hostsim(hostsim_lib):
    """late IPM iterations continue on a dense sub-batch of the still-active instances (recursive
    compaction); per-instance arithmetic is unchanged, so every output bit must equal the run
    without compaction"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch
    N, B = 12, 150
    data = random_lqr_batch(N=N, batch=B, seed=9)
    runs = []
    for cmin in (1 << 30, 4):
        gb = OcpQpGpuBatch(lqr_dims(N, 8, 3), B, _clib=hostsim_lib)
        fill_lqr_batch(gb, data, N)
        gb.opts_set("tol_stat", 1e-8)
        gb.opts_set("compact_min", cmin)
        gb.opts_set("tail_max", 0)   # keep every level on the same kernels: this test is about bit identity
        assert gb.solve() == 0
        runs.append(gb)
    assert int(runs[0].scalar("compactions")) == 0 and int(runs[1].scalar("compactions")) >= 2
    assert len(set(runs[0].info("iter").tolist())) >= 3     # instances really finish at different iterations
    for f in ("status", "iter", "res_stat", "res_comp", "mu"):
        assert np.array_equal(runs[0].info(f), runs[1].info(f)), f
    for k in range(N + 1):
        for f in ("x", "u", "lam", "t") + (("pi",) if k < N else ()):
            assert np.array_equal(runs[0].get(f, k), runs[1].get(f, k)), (f, k)


def test_finished_lanes_ride_along_hostsim(hostsim_lib, monkeypatch):
    """a finished instance that rides along with its wave (the host build lets EVERY finished lane ride along until the
    batch is done) ends bit-identical to the same instance solved alone"""
    from conftest import check_finished_lanes_ride_along
    monkeypatch.setenv("ACADOS_AMD_WPI", "0")   # one instance per lane whatever the batch size
    it = check_finished_lanes_ride_along(hostsim_lib, N=10, B=40, seed=5, alone=range(0, 40, 3))
    assert it.max() - it.min() >= 2


@pytest.mark.parametrize("fam", ["1tpi", "wpi", "w16"])
def test_terminal_polishing_step_hostsim(hostsim_lib, monkeypatch, fam):
    """option "polish" (opt-in): one more iteration for the converged instances that hold a weakly active row -- same status and
    iteration counts, exit test still passed, the rest of the batch bit-identical, the tail of the distance to the solution gone"""
    from conftest import check_polish
    monkeypatch.setenv("ACADOS_AMD_WPI", "0" if fam == "1tpi" else "1")
    monkeypatch.setenv("ACADOS_AMD_W16", "1" if fam == "w16" else "0")
    # (a batch this small holds no pair above the default ratio 1e-3: the selection threshold is lowered so that it picks some)
    d0, d1, changed = check_polish(hostsim_lib, N=8, B=48 if fam == "1tpi" else 24, seed=21, ratio=3e-6)
    assert changed.sum() >= 2
    # (the reference itself is ~1e-9 from the exact solution: below 2e-8 a distance says nothing)
    assert d1[changed].max() <= 0.1 * d0[changed].max() and np.all(d1 <= np.maximum(d0 * 1.01, 2e-8)), (d0[changed], d1[changed])


def test_whole_solve_in_one_launch_hostsim(hostsim_lib, monkeypatch):
    """kx_solve against the launch-per-sweep loop, bit for bit: random structures without general rows (hard and soft box
    rows, per-stage dims), ragged batches of 1, 5 and 6 instances (a single row, one full workgroup + 1 / + 2)"""
    from conftest import check_whole_solve_in_one_launch
    from acados_amd.generators import lqr_instance_qp, mass_spring_qp, random_lqr_batch
    from random_qp import random_structure_qp
    monkeypatch.setenv("ACADOS_AMD_WPI", "1")
    sets = [[random_structure_qp(seed, allow_general=False)] * (1 + seed % 5) for seed in range(11)]
    data = random_lqr_batch(N=7, nx=8, nu=3, batch=6, seed=2)
    sets.append([lqr_instance_qp(data, i, 7) for i in range(6)])        # different instances: rows finish at different iterations
    sets.append([mass_spring_qp(N=8)])                                   # the single QP of an acados control loop
    used = check_whole_solve_in_one_launch(hostsim_lib, sets)
    assert used.get("w16-box", 0) >= 2 and used.get("w16-soft", 0) >= 3, used


def test_json_wire_format_roundtrip(tmp_path):
    """f3: the dump_last_qp_to_json format is read AND written (zero-padded stage keys, natural-sign
    bounds); a round trip preserves every field bit for bit"""
    from acados_amd import AcadosOcpQp
    from acados_amd.ocp_qp import ALL_FIELDS
    qp = load_qp("casadi_qp_tests/pend_idxs_rev_min_qp0.json")
    f = tmp_path / "qp.json"
    qp.to_json(str(f))
    qp2 = AcadosOcpQp.from_json(str(f))
    assert qp2.N == qp.N and qp2.dims.signature() == qp.dims.signature()
    for name in ALL_FIELDS:
        for a, b in zip(getattr(qp, name), getattr(qp2, name)):
            assert np.array_equal(np.asarray(a), np.asarray(b)), name


def test_hot_start_hostsim(hostsim_lib):
    """f2 / warm_start >= 2 (acados_ocp_options.py:1029-1032) through the plugin's evaluate: pi, lam, t handed over in
    qp_out are the starting point while the primal part is zeroed as the reference does before EVERY solve
    (ocp_qp_hpipm.c:325-336); restarting from the converged multipliers needs fewer iterations than the cold start
    and lands on the same solution.  (The exact restart -- primal kept -- is the batch API's, tested in
    test_wave_per_instance_default_rule_hostsim.)"""
    from acados_amd import AcadosOcpQpOptions, AcadosOcpQpSolver
    qp = load_qp("casadi_qp_tests/pendulum_qp.json")
    opts = AcadosOcpQpOptions()
    opts.tol_stat = opts.tol_eq = opts.tol_ineq = opts.tol_comp = 1e-8
    cold = AcadosOcpQpSolver(qp, opts, _clib=hostsim_lib)
    assert cold.solve() == 0
    it_cold = cold.get_stats("iter")
    x_cold = [cold.get(k, "x") for k in range(qp.N + 1)]
    cold.opts_set("warm_start", 3)          # same solver object: qp_out still holds the solution
    assert cold.solve() == 0
    assert cold.get_stats("iter") < it_cold
    for k in range(qp.N + 1):
        assert np.allclose(cold.get(k, "x"), x_cold[k], atol=1e-7)
    # warm_start 1 is warm_start 0 (acados_ocp_options.py:1029-1031): same iteration count as the cold start
    cold.opts_set("warm_start", 1)
    assert cold.solve() == 0 and cold.get_stats("iter") == it_cold


def test_hot_start_condensed_keeps_xcond_iterate_hostsim(hostsim_lib):
    """warm_start >= 2 with partial condensing: the reference starts the condensed solve from the condensed iterate kept in
    its memory and re-derives it from the caller's qp_out only when `initialize_next_xcond_qp_from_qp_out` is set
    (ocp_qp_xcond_solver.c:554-571).  Told apart with an EDITED qp_out (multipliers overwritten with ones): without the
    flag the edit is not looked at -- the restart from the kept iterate converges in a few iterations; with the flag the
    solve starts from the edited values and needs visibly more."""
    import ctypes as C
    from acados_amd import AcadosOcpQpOptions, AcadosOcpQpSolver
    from acados_amd.generators import mass_spring_qp
    qp = mass_spring_qp(N=15)
    opts = AcadosOcpQpOptions()
    opts.tol_stat = opts.tol_eq = opts.tol_ineq = opts.tol_comp = 1e-8
    opts.cond_N = 5
    s = AcadosOcpQpSolver(qp, opts, _clib=hostsim_lib)
    assert s.solve() == 0
    it_cold = s.get_stats("iter")
    x_ref = [s.get(k, "x") for k in range(qp.N + 1)]

    class _Out(C.Structure):
        _fields_ = [("dim", C.c_void_p)] + [(n, C.POINTER(C.POINTER(C.c_double))) for n in ("ux", "pi", "lam", "t")] + [("misc", C.c_void_p)]

    def spoil():
        o = C.cast(s.c_out, C.POINTER(_Out)).contents
        d = qp.dims
        for k in range(qp.N + 1):
            for e in range(2 * int(d.nbx[k] + d.nbu[k] + d.ng[k] + d.ns[k])):
                o.lam[k][e] = 1.0
                o.t[k][e] = 1.0
            if k < qp.N:
                for e in range(int(d.nx[k + 1])):
                    o.pi[k][e] = 0.0

    s.opts_set("warm_start", 3)
    spoil()
    assert s.solve() == 0
    it_kept = s.get_stats("iter")
    spoil()
    s.opts_set("initialize_next_xcond_qp_from_qp_out", True)
    assert s.solve() == 0
    it_from_out = s.get_stats("iter")
    print("iterations: cold", it_cold, "hot from the kept condensed iterate", it_kept, "hot from the edited qp_out", it_from_out)
    assert it_kept <= 3 and it_kept < it_cold and it_from_out > it_kept + 2
    for k in range(qp.N + 1):
        assert np.allclose(s.get(k, "x"), x_ref[k], atol=1e-6)


@pytest.mark.parametrize("fam", ["1tpi", "wpi"])
@pytest.mark.parametrize("t0_init", [0, 1])
def test_t0_init_schemes_hostsim(hostsim_lib, monkeypatch, fam, t0_init):
    """`t0_init` (acados_ocp_options.py:1128-1143; ocp_qp_hpipm.c routes it to HPIPM's argument of that name): 0 -> lam = t =
    sqrt(mu0), 1 -> lam = mu0, t = 1, both with the primal iterate left at zero (an infeasible start), 2 -> the default from
    the constraint residuals.  Device and oracle run the same scheme: same solution, same iteration count (+-1), and the
    count differs from the default scheme's (the option selects something)."""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import mass_spring_qp
    monkeypatch.setenv("ACADOS_AMD_WPI", "0" if fam == "1tpi" else "1")
    diff = 0
    for qp in (mass_spring_qp(N=8), load_qp("casadi_qp_tests/pendulum_slack.json"), load_qp("qp_test/last_qp_one_sided_test.json")):
        its = {}
        for scheme in (t0_init, 2):
            o = OracleQp(qp)
            assert o.solve(default_opts(tol_stat=1e-8, t0_init=scheme, mu0=4.0)) == 0
            b = OcpQpGpuBatch.from_qps([qp, qp], _clib=hostsim_lib)
            for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
                b.opts_set(f, 1e-8)
            b.opts_set("mu0", 4.0)
            b.opts_set("t0_init", scheme)
            assert b.solve() == 0
            assert abs(int(b.info("iter")[1]) - o.iter) <= 1, (scheme, b.info("iter"), o.iter)
            compare_with_oracle(lambda k, f: b.get(f, k)[1], o, qp, 1e-7, fields=("x", "u", "sl", "su", "pi", "lam"))
            its[scheme] = o.iter
        diff += abs(its[t0_init] - its[2])
    assert diff > 0


def test_riccati_getters_hostsim(hostsim_lib):
    """a11: P, p, K, k, Lr of the last factorisation (ocp_qp_hpipm.c:417-478) against the oracle's factor
    at the same iterate; and the feedback-law property Delta u = K Delta x + k on the Newton step"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import mass_spring_qp
    qp = mass_spring_qp(N=8)
    o = OracleQp(qp)
    assert o.solve(default_opts(tol_stat=1e-8)) == 0
    o.refactor()
    b = OcpQpGpuBatch.from_qps([qp, qp], _clib=hostsim_lib)
    b.opts_set("tol_stat", 1e-8)
    assert b.solve() == 0
    for k in range(qp.N + 1):
        nv = int(qp.dims.nu[k] + qp.dims.nx[k])
        Lo = o.get(k, "ric_L").reshape(nv, nv, order="F")
        Lg = b.get("ric_L", k)[1].reshape(nv, nv, order="F")
        assert np.allclose(Lg, Lo, rtol=1e-6, atol=1e-9), k
        assert np.allclose(b.get("ric_l", k)[1], o.get(k, "ric_l"), rtol=1e-5, atol=1e-9), k
        ric = b.riccati(k)
        nu = int(qp.dims.nu[k])
        Lx = Lo[nu:, nu:]
        assert np.allclose(ric["P"][0], Lx @ Lx.T, rtol=1e-6, atol=1e-9)
        assert np.allclose(ric["P"][0], ric["P"][0].T)
        if nu:
            # [Lr 0; Ls Lx][Lr' Ls'; 0 Lx'] = M  =>  K = -Muu^-1 Mux
            M = Lo @ Lo.T
            assert np.allclose(ric["K"][0], -np.linalg.solve(M[:nu, :nu], M[:nu, nu:]), rtol=1e-6, atol=1e-9)


def test_solver_get_vtable_hostsim(hostsim_lib):
    """the qp_solver_config.solver_get slot (ocp_qp_common.h:73) answers K, k, P, p, Lr"""
    import ctypes as C
    from acados_amd import AcadosOcpQpOptions, AcadosOcpQpSolver
    from acados_amd.generators import mass_spring_qp
    qp = mass_spring_qp(N=6)
    s = AcadosOcpQpSolver(qp, AcadosOcpQpOptions(), _clib=hostsim_lib)
    assert s.solve() == 0
    L = hostsim_lib
    L.ocp_qp_solver_get_ric.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
    nu, nx = 3, 8
    K = np.zeros((nx, nu)); kk = np.zeros(nu); P = np.zeros((nx, nx))
    L.ocp_qp_solver_get_ric(s.c_solver, s.c_in, s.c_out, b"K", 2, K.ctypes.data_as(C.c_void_p), nu, nx)
    L.ocp_qp_solver_get_ric(s.c_solver, s.c_in, s.c_out, b"k", 2, kk.ctypes.data_as(C.c_void_p), nu, 1)
    L.ocp_qp_solver_get_ric(s.c_solver, s.c_in, s.c_out, b"P", 2, P.ctypes.data_as(C.c_void_p), nx, nx)
    Kmat = K.T.copy()          # column-major nu x nx
    assert np.all(np.isfinite(Kmat)) and np.abs(Kmat).max() > 0 and np.allclose(P, P.T) and np.all(np.linalg.eigvalsh(P) > 0)


@pytest.mark.parametrize("fam", ["1tpi", "wpi", "w16"])
def test_edge_cases_hostsim(hostsim_lib, monkeypatch, fam):
    """edge cases: no inequality at all (pure LQR: one Newton step), N = 1, single instance, exact wave
    multiple, infeasible bounds (non-zero acados status, no hang), NaN in the data (ACADOS_NAN_DETECTED) -- on each
    of the three kernel families"""
    from acados_amd import AcadosOcpQp, OcpQpGpuBatch
    from acados_amd.generators import lqr_instance_qp, random_lqr_batch
    monkeypatch.setenv("ACADOS_AMD_WPI", "0" if fam == "1tpi" else "1")
    monkeypatch.setenv("ACADOS_AMD_W16", "1" if fam == "w16" else "0")

    # pure LQR with free initial state: no inequality rows anywhere
    data = random_lqr_batch(N=6, batch=2, seed=1)
    qp = AcadosOcpQp(6)
    for k in range(7):
        qp.set("Q", k, data["Q"][0]); qp.set("q", k, data["q"][0])
        if k < 6:
            qp.set("R", k, data["R"][0]); qp.set("r", k, data["r"][0]); qp.set("S", k, data["S"][0])
            qp.set("A", k, data["A"][0]); qp.set("B", k, data["B"][0]); qp.set("b", k, data["b"][0])
    qp.make_consistent()
    o = OracleQp(qp)
    assert o.solve(default_opts(tol_stat=1e-8)) == 0
    b = OcpQpGpuBatch.from_qps([qp], _clib=hostsim_lib)
    b.opts_set("tol_stat", 1e-8)
    assert b.solve() == 0 and b.info("iter")[0] == 1 == o.iter
    compare_with_oracle(lambda k, f: b.get(f, k)[0], o, qp, 1e-10, fields=("x", "u", "pi"))

    # N = 1 and exactly one wave of instances
    data = random_lqr_batch(N=1, batch=64, seed=2)
    qps = [lqr_instance_qp(data, i, 1) for i in range(64)]
    b = OcpQpGpuBatch.from_qps(qps, _clib=hostsim_lib)
    b.opts_set("tol_stat", 1e-8)
    assert b.solve() == 0
    o = OracleQp(qps[63])
    assert o.solve(default_opts(tol_stat=1e-8)) == 0
    compare_with_oracle(lambda k, f: b.get(f, k)[63], o, qps[63], 1e-9)

    # infeasible: lower input bound above the upper one -> status MAXITER (2) or MINSTEP (3), never 0
    data = random_lqr_batch(N=4, batch=3, seed=3)
    data["lbu"][1] = 1.0
    data["ubu"][1] = -1.0
    qps = [lqr_instance_qp(data, i, 4) for i in range(3)]
    b = OcpQpGpuBatch.from_qps(qps, _clib=hostsim_lib)
    b.opts_set("iter_max", 30)
    assert b.solve() == 1
    st = b.info("status")
    assert st[0] == 0 and st[2] == 0 and st[1] in (2, 3)

    # NaN in the data of one instance -> ACADOS_NAN_DETECTED (1) for that instance only
    data = random_lqr_batch(N=4, batch=3, seed=4)
    data["q"][2, 0] = np.nan
    qps = [lqr_instance_qp(data, i, 4) for i in range(3)]
    b = OcpQpGpuBatch.from_qps(qps, _clib=hostsim_lib)
    assert b.solve() == 1
    assert list(b.info("status")) == [0, 0, 1]


@pytest.mark.parametrize("fam", ["w16", "wpi", "1tpi"])
def test_solution_sensitivities_hostsim(hostsim_lib, monkeypatch, fam):
    """a12: forward sensitivities with the factorisation at the solution (what eval_forw_sens / eval_adj_sens stand
    for, ocp_qp_hpipm.c:481-506): d(x, u)/dp from one rhs-only backward + one forward sweep equals the central finite
    difference of the solver's own solution, for p in the gradient, in the dynamics offset, in x0 (equality-flagged
    bound: the parameter IS the variable) and in an input bound.  A batch on the one-instance-per-lane kernels answers
    through slices handed to a wave-per-instance sub-batch (here: 3 instances in slices of 2)"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch
    monkeypatch.setenv("ACADOS_AMD_WPI", "0" if fam == "1tpi" else "1")
    monkeypatch.setenv("ACADOS_AMD_W16", "1" if fam == "w16" else "0")
    monkeypatch.setenv("ACADOS_AMD_SENS_SLICE", "2")
    N, B, nx, nu = 4, (3 if fam == "1tpi" else 2), 8, 3
    data = random_lqr_batch(N=N, batch=B, seed=4)

    def build(d):
        gb = OcpQpGpuBatch(lqr_dims(N, nx, nu), B, _clib=hostsim_lib)
        fill_lqr_batch(gb, d, N)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            gb.opts_set(f, 1e-10)
        assert gb.solve() == 0
        return gb

    def xu(g, pre=""):
        return np.concatenate([g.get(pre + "x", k) for k in range(N + 1)] + [g.get(pre + "u", k) for k in range(N)], axis=1)

    e3 = np.zeros((B, nx)); e3[:, 3] = 1.0
    eu = np.zeros((B, nu)); eu[:, 1] = 1.0
    cases = {
        "q": (("q", 3), [("seed_q", k, e3) for k in range(N + 1)]),
        "b": (("b", 3), [("seed_b", k, e3) for k in range(N)]),
        "x0": (("x0", 3), [("seed_lbx", 0, e3), ("seed_ubx", 0, e3)]),
        "ubu": (("ubu", 1), [("seed_ubu", k, eu) for k in range(N)]),
    }
    ref = build(data)
    assert ref.kernel_name.startswith({"w16": "w16-box", "wpi": "wpi-box", "1tpi": "1tpi-box"}[fam])
    for name, ((key, idx), seeds) in cases.items():
        sols = []
        for sg in (+1e-6, -1e-6):
            d = {k: v.copy() for k, v in data.items()}
            d[key][:, idx] += sg
            sols.append(xu(build(d)))
        fd = (sols[0] - sols[1]) / 2e-6
        for (f, k, v) in seeds:
            ref.sens_set(f, k, v)
        ref.sens_solve()
        se = xu(ref, "sens_")
        assert np.max(np.abs(fd - se)) <= 1e-6 * max(1.0, np.max(np.abs(se))), name
        if name == "ubu":
            assert np.max(np.abs(se)) > 1e-3   # some upper input bound is active somewhere: the solution moves with it


def test_solution_sensitivities_acados_api_hostsim(hostsim_lib, monkeypatch):
    """the eval_forw_sens / eval_adj_sens slots of the plugin on the reference's golden pendulum QP: sensitivity
    w.r.t. x0 (seed on both sides of the equality-flagged row, as ocp_nlp_common.c:4057-4064 sets it) against finite
    differences; the adjoint slot returns the same for a gradient seed (symmetric KKT matrix)"""
    import copy
    from acados_amd import AcadosOcpQpOptions, AcadosOcpQpSolver
    monkeypatch.setenv("ACADOS_AMD_WPI", "1")
    qp = load_qp("qp_test/last_qp_nonuniform_pendulum.json")
    opts = AcadosOcpQpOptions()
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        setattr(opts, f, 1e-10)

    def solve(q):
        s = AcadosOcpQpSolver(q, opts, _clib=hostsim_lib)
        assert s.solve() == 0
        return s

    def xu(s):
        return np.concatenate([s.get(k, "x") for k in range(qp.N + 1)] + [s.get(k, "u") for k in range(qp.N)])

    s0 = solve(qp)
    i, h = 2, 1e-6
    sols = []
    for sg in (+h, -h):
        q_ = copy.deepcopy(qp)
        lb = np.array(q_.lbx[0], dtype=float)
        lb[i] += sg
        q_.set("lbx", 0, lb); q_.set("ubx", 0, lb)
        sols.append(xu(solve(q_)))
    fd = (sols[0] - sols[1]) / (2 * h)
    e = np.zeros(len(qp.lbx[0])); e[i] = 1.0
    se = s0.eval_solution_sens({("lbx", 0): e, ("ubx", 0): e})
    sv = np.concatenate(se["x"] + se["u"])
    assert np.max(np.abs(fd - sv)) <= 1e-6 * np.max(np.abs(sv))
    assert abs(se["x"][0][i] - 1.0) <= 1e-12                      # d x0_i / d x0_i
    eq = np.zeros(int(qp.dims.nx[3])); eq[1] = 1.0
    f_ = s0.eval_solution_sens({("q", 3): eq})
    a_ = s0.eval_solution_sens({("q", 3): eq}, adjoint=True)
    assert all(np.array_equal(f_["x"][k], a_["x"][k]) for k in range(qp.N + 1))


def test_solution_sensitivities_soft_constraints_hostsim(hostsim_lib, monkeypatch):
    """sensitivities on the general-constraint / slack kernels (C4 class, short horizon): gradient seed and x0 seed
    against finite differences, at the tolerances acados uses (1e-8).  (Pushing the tolerances to 1e-11 drives
    Gamma = lam/t of the active soft rows to 1e17 and the rounding of dlam = -Gamma dt up to the percent level --
    the same amplification the barrier floor exists for; at 1e-8 the agreement is 1e-6 relative.)"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import chain_soft_batch, chain_soft_dims, fill_chain_soft_batch
    monkeypatch.setenv("ACADOS_AMD_WPI", "1")
    N, B = 3, 2
    data = chain_soft_batch(N=N, batch=B, seed=1)

    def build(d, tol=1e-8):
        gb = OcpQpGpuBatch(chain_soft_dims(N), B, _clib=hostsim_lib)
        fill_chain_soft_batch(gb, d, N)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            gb.opts_set(f, tol)
        gb.opts_set("tol_comp_soft_scale", 1.0)   # sensitivities of a soft class are taken at the 1e-8 iterate: three more orders of mu
                                                  # cost three digits of the direction (Gamma = lam / t), the docstring above
        assert gb.solve() == 0
        return gb

    def xus(g, pre=""):
        return np.concatenate([g.get(pre + "x", k) for k in range(N + 1)] + [g.get(pre + "u", k) for k in range(N)]
                              + [g.get(pre + "sl", k) for k in range(1, N + 1)] + [g.get(pre + "su", k) for k in range(1, N + 1)], axis=1)

    ref = build(data)
    assert ref.kernel_name.startswith("w16r-gen<")   # (the wave-per-instance GEN kernels: same test on the GPU tier, ACADOS_AMD_W16G=0)
    e = np.zeros((B, 24)); e[:, 7] = 1.0
    h = 1e-3
    for name in ("q", "x0"):
        sols = []
        for sg in (+h, -h):
            d = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in data.items()}
            if name == "q":
                d["q"][:, 2, 7] += sg
            else:
                d["x0"][:, 7] += sg
            # (the two legs of the difference quotient at 1e-11: at 1e-8 an iterate of this class lies up to 1e-6 from the solution, and
            # the quotient carries that over h = 1e-3 unless both legs happen to stop at the same point of their balls)
            sols.append(xus(build(d, 1e-11)))
        fd = (sols[0] - sols[1]) / (2 * h)
        if name == "q":
            ref.sens_set("seed_q", 2, e)
        else:
            ref.sens_set("seed_lbx", 0, e); ref.sens_set("seed_ubx", 0, e)
        ref.sens_solve()
        se = xus(ref, "sens_")
        assert np.max(np.abs(fd - se)) <= 2e-5 * np.max(np.abs(se)), name


@pytest.mark.parametrize("wpi", ["0", "1"])
def test_random_structures_hostsim(hostsim_lib, monkeypatch, wpi):
    """30 QPs with random STRUCTURE (tests/random_qp.py: per-stage dims, box subsets, one-sided rows, general rows,
    slacks shared between rows, equality-flagged x0, N = 1..6) on the one-instance-per-lane kernels (where a compiled
    shape covers the dims) and on the wave-per-instance family, three copies per batch, against the oracle"""
    from acados_amd import OcpQpGpuBatch
    from random_qp import random_structure_qp
    monkeypatch.setenv("ACADOS_AMD_WPI", wpi)
    fams = set()
    for seed in range(30):
        qp = random_structure_qp(seed)
        o = OracleQp(qp)
        assert o.solve(default_opts(tol_stat=1e-8, iter_max=60)) == 0, seed
        b = OcpQpGpuBatch.from_qps([qp] * 3, _clib=hostsim_lib)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            b.opts_set(f, 1e-8)
        b.opts_set("iter_max", 60)
        assert b.solve() == 0, (seed, b.kernel_name)
        fams.add(b.kernel_name.split("(")[0].split("<")[0])
        assert abs(int(b.info("iter")[2]) - o.iter) <= 1, (seed, b.kernel_name)
        try:
            compare_with_oracle(lambda k, f: b.get(f, k)[2], o, qp, 1e-7, fields=("x", "u", "sl", "su", "pi", "lam", "t"))
        except AssertionError as e:
            raise AssertionError(f"seed {seed} kernel {b.kernel_name}: {e}")
    assert all(f.startswith(("wpi", "w16") if wpi == "1" else "1tpi") for f in fams), fams


def test_conditional_corrector_ends_a_limit_cycle_hostsim(hostsim_lib):
    """Found by the structure fuzz of round 5 (tools/fuzz_parity.py, perturbed batches): with the corrector redone only when its
    step collapsed (alpha < 0.1 alpha_aff, rounds 1-5) the Mehrotra iteration of this instance -- N = 1, nx = 11, all states and
    one input bounded, random structure 7105 with its linear cost terms scaled -- runs into a four-cycle at mu ~ 3e-4 and ends at
    MAXITER, on the device and in the oracle alike.  The test HPIPM applies (the step is taken again with the centering term
    alone when it would more than double the duality measure) ends the cycle: both converge, in the same number of iterations."""
    limit_cycle_case(hostsim_lib)


def test_bulk_blob_whole_and_in_chunks_hostsim(hostsim_lib):
    """_get_bulk_in / _set_bulk / _set_bulk_chunk + _set_bulk_staged / the zero-copy gather through the C-ABI
    (tests/conftest.py::bulk_chunk_case; two of the three structures here, all three on the device)"""
    bulk_chunk_case(hostsim_lib, seeds=(3, 41))


def test_random_structures_partial_condensing_hostsim(hostsim_lib):
    """partial condensing on QPs with random structure (tests/random_qp.py; N2 = ceil(N/2), uneven blocks, per-stage
    dims, one-sided rows, general rows, shared slacks): the expanded point must satisfy the KKT conditions of the
    ORIGINAL QP at tolerance -- a hot-started full-space call converges at its first residual evaluation -- and lie
    close to the oracle's solution (the two iterate paths stop at different points of the 1e-8 complementarity ball)"""
    from acados_amd import OcpQpGpuBatch
    from random_qp import random_structure_qp
    condensed = sharp = 0
    for seed in range(40):
        qp = random_structure_qp(seed)
        if qp.N < 2:
            continue
        o = OracleQp(qp)
        assert o.solve(default_opts(tol_stat=1e-8, iter_max=60)) == 0
        b = OcpQpGpuBatch.from_qps([qp] * 2, _clib=hostsim_lib)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            b.opts_set(f, 1e-8)
        b.opts_set("iter_max", 60)
        b.opts_set("cond_N", (qp.N + 1) // 2)
        assert b.solve() == 0, seed
        condensed += int(b.scalar("cond_N_active")) == (qp.N + 1) // 2
        sharp += compare_condensed_with_oracle(lambda k, f: b.get(f, k)[1], o, qp)
        b.opts_set("cond_N", qp.N)
        b.opts_set("warm_start", 3)
        assert b.solve() == 0 and int(b.info("iter").max()) == 0, seed
        assert max(b.info(n).max() for n in ("res_stat", "res_eq", "res_ineq", "res_comp")) <= 1e-8
    # all but the weakly active instances are compared at 2e-6 (primal) / 1e-7 (multipliers of inactive rows)
    assert condensed >= 30 and sharp >= 28, (condensed, sharp)


def test_condensing_only_boundary_hostsim(hostsim_lib):
    """the condensing-only boundary (ocp_qp_condense / ocp_qp_expand, condensing_interface.h:73-75): the condensed QP
    is read back from the device as a QP of its own, solved by the CPU ORACLE (an independent solver: this pins the
    condensed data -- Hbar, gbar, [Bbar Abar], bbar, the general rows made of inner state bounds, shifted bounds,
    slacks, masks -- not just the round trip), its solution is written into the condensed batch and expanded; the
    result must be the full-space oracle solution.  Random structures + user block sizes (cond_block_size)."""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import lqr_instance_qp, mass_spring_qp, random_lqr_batch
    from random_qp import random_structure_qp

    def roundtrip(qp, cond_N, blocks=None, tol=1e-4):
        o = OracleQp(qp)
        assert o.solve(default_opts(tol_stat=1e-8, iter_max=60)) == 0
        b = OcpQpGpuBatch.from_qps([qp] * 2, _clib=hostsim_lib)
        b.opts_set("cond_N", cond_N)
        if blocks is not None:
            arr = (ctypes.c_int * len(blocks))(*blocks)
            assert b._L.ocp_qp_gpu_batch_opts_set(b._h, b"cond_block_size", arr) == 0
        c = b.condense()
        # a non-zero last block size (the reference's own test, pcond_getters_test.py:200) is one more block with inputs in
        # front of an input-free terminal stage
        n_blk = cond_N + (1 if blocks is not None and blocks[-1] else 0)
        assert c is not None and c.N == n_blk
        qc = c.to_qp(1)
        if blocks is not None:
            assert list(qc.dims.nu[:n_blk]) == [bs * int(qp.dims.nu[0]) for bs in blocks[:n_blk]] and qc.dims.nu[n_blk] == 0
        oc = OracleQp(qc)
        assert oc.solve(default_opts(tol_stat=1e-8, iter_max=60)) == 0
        for k in range(c.N + 1):
            for f in ("x", "u", "sl", "su", "lam", "t") + (("pi",) if k < c.N else ()):
                v = oc.get(k, f)
                if v.size:
                    c.set(f, k, np.tile(v, (2, 1)))
        b.expand()
        compare_condensed_with_oracle(lambda k, f: b.get(f, k)[1], o, qp, loose=tol)
        # ... and the expanded point satisfies the original KKT conditions at tolerance
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            b.opts_set(f, 1e-8)
        b.opts_set("cond_N", qp.N)
        b.opts_set("warm_start", 3)
        assert b.solve() == 0 and int(b.info("iter").max()) == 0

    import ctypes
    data = random_lqr_batch(N=10, batch=1, seed=5)
    roundtrip(lqr_instance_qp(data, 0, 10), 3, blocks=[2, 5, 3, 0])
    roundtrip(lqr_instance_qp(data, 0, 10), 3, blocks=[2, 5, 2, 1])
    data20 = random_lqr_batch(N=20, nx=4, nu=1, batch=1, seed=6)
    roundtrip(lqr_instance_qp(data20, 0, 20), 5, blocks=[6, 5, 4, 2, 2, 1])      # pcond_getters_test.py:200
    roundtrip(mass_spring_qp(N=15), 3, blocks=[5, 5, 3, 2])
    roundtrip(mass_spring_qp(N=15), 5)
    n = 0
    for seed in range(24):
        qp = random_structure_qp(seed)
        if qp.N >= 2:
            roundtrip(qp, (qp.N + 1) // 2)
            n += 1
    assert n >= 18
    # block sizes that do not sum to N are refused (the reference exits, ocp_qp_partial_condensing.c:346-356)
    b = OcpQpGpuBatch.from_qps([mass_spring_qp(N=15)], _clib=hostsim_lib)
    b.opts_set("cond_N", 3)
    assert b._L.ocp_qp_gpu_batch_opts_set(b._h, b"cond_block_size", (ctypes.c_int * 4)(5, 5, 4, 0)) != 0


def test_condensing_module_acados_api_hostsim(hostsim_lib):
    """the condensing module on the acados-shaped containers (ocp_qp_condensing_create / ocp_qp_condense /
    ocp_qp_expand, condensing_interface.h:61-75): xcond dims, the condensed QP in a plain ocp_qp_in solved by the CPU
    oracle, its solution expanded through a plain ocp_qp_out == the full-space oracle solution; user block sizes"""
    from acados_amd import AcadosOcpQpCondensing
    from acados_amd.generators import mass_spring_qp
    from random_qp import random_structure_qp
    cases = [(mass_spring_qp(N=15), 5, None), (mass_spring_qp(N=15), 4, [3, 4, 4, 4, 0]), (mass_spring_qp(N=15), 4, [3, 4, 4, 3, 1]),
             (load_qp("casadi_qp_tests/pend_idxs_rev_min_qp0.json"), 3, None), (load_qp("qp_test/last_qp_one_sided_test.json"), 4, None)]
    cases += [(random_structure_qp(s), None, None) for s in (0, 4, 7, 13, 22)]
    for qp, cn, blocks in cases:
        cn = (qp.N + 1) // 2 if cn is None else cn
        o = OracleQp(qp)
        assert o.solve(default_opts(tol_stat=1e-8, iter_max=60)) == 0
        mod = AcadosOcpQpCondensing(qp, cn, block_size=blocks, _clib=hostsim_lib)
        xd = mod.xcond_dims()
        n_blk = cn + (1 if blocks is not None and blocks[-1] else 0)
        if blocks is not None:
            assert list(xd["nu"][:n_blk]) == [bs * int(qp.dims.nu[0]) for bs in blocks[:n_blk]]
        qc = mod.condense()
        assert qc.N == n_blk and list(qc.dims.nx) == list(xd["nx"])
        oc = OracleQp(qc)
        assert oc.solve(default_opts(tol_stat=1e-8, iter_max=60)) == 0
        get = mod.expand(lambda k, f: oc.get(k, f))
        compare_condensed_with_oracle(get, o, qp)
        # objective values agree: the condensed QP is the same problem (up to the constant the elimination drops,
        # which both solutions share) -- compare the primal solutions' cost in the ORIGINAL QP instead
        x = np.concatenate([get(k, "x") for k in range(qp.N + 1)]); xr = np.concatenate([o.get(k, "x") for k in range(qp.N + 1)])
        assert np.max(np.abs(x - xr)) <= 1e-4 * max(1.0, np.max(np.abs(xr)))


def test_condense_after_structure_change_hostsim(hostsim_lib):
    """one condensing module, two condense calls with the index sets changed in between (idxb of stage 3 reversed): the
    device batch is re-created -- often at the address of the one just destroyed -- and must be told cond_N again (the
    module once compared batch POINTERS and then failed with ACADOS_QP_FAILURE depending on the heap)"""
    import copy
    from acados_amd import AcadosOcpQpCondensing
    from acados_amd.generators import mass_spring_qp
    qp = mass_spring_qp(N=15)
    mod = AcadosOcpQpCondensing(qp, 5, _clib=hostsim_lib)
    qc1 = mod.condense()
    for rep in range(3):   # several re-creations: whatever the allocator does with the addresses
        q2 = copy.deepcopy(qp)
        k = 3 + rep
        order = np.arange(len(qp.idxb[k]))[::-1]
        nbu = int(qp.dims.nbu[k])
        # reversed order inside the u block and inside the x block (acados keeps [u rows; x rows])
        ou, ox = np.arange(nbu)[::-1], nbu + np.arange(len(order) - nbu)[::-1]
        perm = np.concatenate([ou, ox])
        q2.set("idxb", k, np.asarray(qp.idxb[k])[perm])
        for f, sel in (("lbu", ou), ("ubu", ou), ("lbx", ox - nbu), ("ubx", ox - nbu)):
            q2.set(f, k, np.asarray(getattr(qp, f)[k])[sel])
        mod.set_qp(q2)
        qc2 = mod.condense()          # raised "ocp_qp_condense failed" before
        assert qc2.N == qc1.N
        o1, o2 = OracleQp(qc1), OracleQp(qc2)
        assert o1.solve(default_opts(tol_stat=1e-8)) == 0 and o2.solve(default_opts(tol_stat=1e-8)) == 0
        for kk in range(qc1.N + 1):   # the same QP with its rows listed in another order: same condensed solution
            assert np.allclose(o1.get(kk, "x"), o2.get(kk, "x"), rtol=1e-7, atol=1e-9)
            assert np.allclose(o1.get(kk, "u"), o2.get(kk, "u"), rtol=1e-7, atol=1e-9)
        mod.set_qp(qp)
        mod.condense()


def test_cond_block_size_option_acados_api_hostsim(hostsim_lib, capfd):
    """`cond_block_size` through the xcond-solver options (ocp_qp_partial_condensing.c:305-313; the Python driver
    sends it like acados_ocp_qp_solver.py does): user blocks reach the device condensing, same solution"""
    from acados_amd import AcadosOcpQpOptions, AcadosOcpQpSolver
    from acados_amd.generators import mass_spring_qp
    qp = mass_spring_qp(N=15)
    o = OracleQp(qp)
    assert o.solve(default_opts(tol_stat=1e-8)) == 0
    opts = AcadosOcpQpOptions()
    opts.tol_stat = opts.tol_eq = opts.tol_ineq = opts.tol_comp = 1e-8
    opts.cond_N = 4
    for blocks in ([3, 4, 4, 4, 0], [3, 4, 4, 2, 2]):
        opts.cond_block_size = blocks
        s = AcadosOcpQpSolver(qp, opts, _clib=hostsim_lib)
        assert s.solve() == 0
        compare_with_oracle(lambda k, f: s.get(k, f, unique_duals=False), o, qp, 2e-7, fields=("x", "u", "pi", "lam", "t"))
        assert "solving the full-space QP" not in capfd.readouterr().err


def test_sixteen_lanes_covering_shapes_hostsim(hostsim_lib, monkeypatch):
    """box-constrained shapes without a compiled sixteen-lanes instantiation of their own run in the smallest one
    that covers them (dims padded inside the compiled shape): (6,2) and (7,3) in <8,3>, (3,1) in <4,1>, (10,4) in
    <12,4>, (3,3) in <4,4>; ragged batch of 6 = one full workgroup of four instances + two"""
    from acados_amd.generators import lqr_instance_qp, random_lqr_batch
    monkeypatch.setenv("ACADOS_AMD_WPI", "1")
    for (nx, nu), want in (((6, 2), "w16-box<NX=8,NU=3>"), ((7, 3), "w16-box<NX=8,NU=3>"), ((3, 1), "w16-box<NX=4,NU=1>"),
                           ((10, 4), "w16-box<NX=12,NU=4>"), ((3, 3), "w16-box<NX=4,NU=4>")):
        data = random_lqr_batch(N=5, nx=nx, nu=nu, batch=6, seed=20 + nx)
        b = _check_batch_vs_oracle([lqr_instance_qp(data, i, 5) for i in range(6)], hostsim_lib)
        assert b.kernel_name == want, b.kernel_name


def test_sixteen_lanes_two_rows_per_lane_hostsim(hostsim_lib, monkeypatch):
    """17 <= nu+nx <= 32, box rows only: sixteen lanes per instance with TWO rows per lane (ipm_kernels_w16r.hpp), the
    factor sweep on 4 x 4 MFMA tiles (ipm_kernels_w16t.hpp, the default) or on register rows (ACADOS_AMD_W16T=0).
    The compiled shapes <24,6> and <8,15>, shapes padded inside them, state bounds on every stage, fixed initial
    state, a ragged batch (one full workgroup of four instances + one), and agreement with the wave-per-instance
    kernels on the same batch (ACADOS_AMD_W16R=0) to rounding"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import lqr_instance_qp, random_lqr_batch
    for (nx, nu), want in (((24, 6), "w16r-box<NX=24,NU=6>"), ((8, 15), "w16r-box<NX=8,NU=15>"), ((20, 5), "w16r-box<NX=24,NU=6>"),
                           ((6, 12), "w16r-box<NX=8,NU=15>"), ((13, 4), "wpi-box(nx=13,nu=4")):   # (13,4): 17 of 30 -- too much padding
        data = random_lqr_batch(N=4, nx=nx, nu=nu, batch=5, seed=40 + nx)
        b = _check_batch_vs_oracle([lqr_instance_qp(data, i, 4) for i in range(5)], hostsim_lib)
        assert b.kernel_name.startswith(want), b.kernel_name
        if want.startswith("w16r"):
            assert b.scalar("w16_tiles") == 1
            monkeypatch.setenv("ACADOS_AMD_W16T", "0")
            b = _check_batch_vs_oracle([lqr_instance_qp(data, i, 4) for i in range(5)], hostsim_lib)
            assert b.kernel_name.startswith(want) and b.scalar("w16_tiles") == 0
            monkeypatch.delenv("ACADOS_AMD_W16T")
    # the same batch on both families: iterates agree to rounding, iteration counts are equal
    data = random_lqr_batch(N=6, nx=24, nu=6, batch=3, seed=3)
    qps = [lqr_instance_qp(data, i, 6) for i in range(3)]
    sols = {}
    for fam in ("1", "0"):
        monkeypatch.setenv("ACADOS_AMD_W16R", fam)
        g = OcpQpGpuBatch.from_qps(qps, _clib=hostsim_lib)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            g.opts_set(f, 1e-8)
        assert g.solve() == 0
        assert g.kernel_name.startswith("w16r-box<" if fam == "1" else "wpi-box(")
        sols[fam] = ([g.get(f, k) for f in ("x", "u", "lam") for k in range(7)] + [g.get("pi", k) for k in range(6)],
                     np.array(g.info("iter")))
    for x, y in zip(sols["1"][0], sols["0"][0]):
        np.testing.assert_allclose(x, y, rtol=1e-9, atol=1e-9)
    assert np.array_equal(sols["1"][1], sols["0"][1])


def test_two_rows_per_lane_general_rows_and_slacks_hostsim(hostsim_lib, monkeypatch):
    """w16r-gen (C4 class: soft state bounds + soft general rows, one slack per row) against the wave-per-instance GEN
    kernels on the same ragged batch: equal iteration counts, solutions (x, u, slacks, multipliers) equal to 1e-9; a
    structure the family does not cover (a slack shared by two rows) falls back to the wave-per-instance kernels"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import chain_soft_qp
    N, B = 3, 5
    qps = [chain_soft_qp(i, N=N) for i in range(B)]
    sols = {}
    # "1": the default (factor sweep on 4 x 4 MFMA tiles, kt_factor<24,3,4>); "rows": the same family with the factor sweep on
    # register rows (ky_factor<24,3,4>, ACADOS_AMD_W16T_GEN=0); "0": the wave-per-instance GEN kernels
    for fam in ("1", "rows", "0"):
        monkeypatch.setenv("ACADOS_AMD_W16G", "0" if fam == "0" else "1")
        monkeypatch.setenv("ACADOS_AMD_W16T_GEN", "0" if fam == "rows" else "1")
        g = OcpQpGpuBatch.from_qps(qps, _clib=hostsim_lib)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            g.opts_set(f, 1e-8)
        assert g.solve() == 0
        assert g.kernel_name.startswith("wpi-gen(" if fam == "0" else "w16r-gen<NX=24,NU=3,NG=4>")
        if fam != "0":
            assert g.scalar("w16_tiles") == (1 if fam == "1" else 0)
        sols[fam] = ([g.get(f, k) for f in ("x", "u", "lam", "sl", "su") for k in range(N + 1)] + [g.get("pi", k) for k in range(N)],
                     np.array(g.info("iter")))
    monkeypatch.delenv("ACADOS_AMD_W16T_GEN")
    for other in ("rows", "0"):
        assert np.array_equal(sols["1"][1], sols[other][1])
        for a, c in zip(sols["1"][0], sols[other][0]):
            if a.size:
                np.testing.assert_allclose(a, c, rtol=1e-9, atol=1e-9)
    monkeypatch.setenv("ACADOS_AMD_W16G", "1")
    qp = chain_soft_qp(0, N=N)
    rev = np.array(qp.idxs_rev[1]).copy()
    soft = np.flatnonzero(rev >= 0)
    rev[soft[1]] = rev[soft[0]]          # two rows of stage 1 share one slack
    qp.idxs_rev[1] = rev
    g = OcpQpGpuBatch.from_qps([qp], _clib=hostsim_lib)
    g.solve()
    assert g.kernel_name.startswith("wpi-gen("), g.kernel_name


def test_solution_sensitivities_after_partial_condensing_hostsim(hostsim_lib, monkeypatch):
    """sensitivities of a partially condensed solve: computed in the full space at the expanded solution, equal to
    the finite differences of (partially condensed) solves"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch
    monkeypatch.setenv("ACADOS_AMD_WPI", "1")
    N, B, nx, nu = 6, 2, 8, 3
    data = random_lqr_batch(N=N, batch=B, seed=4)

    def build(d):
        gb = OcpQpGpuBatch(lqr_dims(N, nx, nu), B, _clib=hostsim_lib)
        fill_lqr_batch(gb, d, N)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            gb.opts_set(f, 1e-10)
        gb.opts_set("cond_N", 3)
        assert gb.solve() == 0 and int(gb.scalar("cond_N_active")) == 3
        return gb

    def xu(g, pre=""):
        return np.concatenate([g.get(pre + "x", k) for k in range(N + 1)] + [g.get(pre + "u", k) for k in range(N)], axis=1)

    e3 = np.zeros((B, nx)); e3[:, 3] = 1.0
    ref = build(data)
    sols = []
    for sg in (+1e-6, -1e-6):
        d = {k: v.copy() for k, v in data.items()}
        d["x0"][:, 3] += sg
        sols.append(xu(build(d)))
    fd = (sols[0] - sols[1]) / 2e-6
    ref.sens_set("seed_lbx", 0, e3); ref.sens_set("seed_ubx", 0, e3)
    ref.sens_solve()
    se = xu(ref, "sens_")
    assert np.max(np.abs(fd - se)) <= 1e-5 * max(1.0, np.max(np.abs(se)))


def test_sixteen_lanes_soft_box_rows_hostsim(hostsim_lib, monkeypatch):
    """SOFT variants of the sixteen-lanes kernels: soft box rows with one slack per row (slack block eliminated inside
    the lane that owns the row), mixed with hard rows, one-sided rows, per-stage dims; structures with a slack shared
    by several rows fall back to the general wave-per-instance kernels.  40 random structures without general rows
    (tests/random_qp.py) + the C2 shape with soft bounds on every state, against the oracle."""
    from acados_amd import AcadosOcpQp, OcpQpGpuBatch
    from acados_amd.generators import lqr_instance_qp, random_lqr_batch
    from random_qp import random_structure_qp
    monkeypatch.setenv("ACADOS_AMD_WPI", "1")
    names = {}
    for seed in range(40):
        qp = random_structure_qp(seed, allow_general=False)
        o = OracleQp(qp)
        assert o.solve(default_opts(tol_stat=1e-8, iter_max=60)) == 0, seed
        b = OcpQpGpuBatch.from_qps([qp] * 5, _clib=hostsim_lib)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            b.opts_set(f, 1e-8)
        b.opts_set("iter_max", 60)
        assert b.solve() == 0, (seed, b.kernel_name)
        fam = b.kernel_name.split("<")[0].split("(")[0]
        names[fam] = names.get(fam, 0) + 1
        shared = any(np.sum(np.asarray(qp.idxs_rev[k]) == j) > 1 for k in range(qp.N + 1) for j in range(int(qp.dims.ns[k])))
        if int(np.sum(qp.dims.ns)) > 0:
            assert fam == ("wpi-gen" if shared else "w16-soft"), (seed, b.kernel_name, shared)
        assert abs(int(b.info("iter")[4]) - o.iter) <= 1, (seed, b.kernel_name)
        try:
            compare_with_oracle(lambda k, f: b.get(f, k)[4], o, qp, 1e-7, fields=("x", "u", "sl", "su", "pi", "lam", "t"))
        except AssertionError as e:
            raise AssertionError(f"seed {seed} kernel {b.kernel_name}: {e}")
    assert names.get("w16-soft", 0) >= 8 and names.get("wpi-gen", 0) >= 1 and names.get("w16-box", 0) >= 5, names
    # C2 shape, every state bound soft from stage 1 on
    N, nx, nu = 6, 8, 3
    data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=2, seed=3)
    qps = []
    for i in range(2):
        qp = lqr_instance_qp(data, i, N)
        for k in range(1, N + 1):
            nuk = nu if k < N else 0
            qp.set("idxb", k, np.arange(nuk + nx))
            qp.set("lbx", k, -0.5 * np.ones(nx)); qp.set("ubx", k, 0.5 * np.ones(nx))
            qp.set("lbx_mask", k, np.ones(nx)); qp.set("ubx_mask", k, np.ones(nx))
            qp.set("idxs_rev", k, np.concatenate([-np.ones(nuk, dtype=int), np.arange(nx)]))
            for f, v in (("Zl", 1e2), ("Zu", 1e2), ("zl", 1e1), ("zu", 1e1), ("lls", 0.0), ("lus", 0.0)):
                qp.set(f, k, v * np.ones(nx))
        qp.make_consistent()
        qps.append(qp)
    b = _check_batch_vs_oracle(qps, hostsim_lib)
    assert b.kernel_name == "w16-soft<NX=8,NU=3>"
    assert max(float(np.max(b.get("sl", k))) for k in range(1, N + 1)) > 1e-3   # some soft bound is really violated


def test_solution_sensitivities_soft_box_rows_hostsim(hostsim_lib, monkeypatch):
    """sensitivities on the SOFT sixteen-lanes kernels (C2 shape, soft bounds on every state): x0 seed against finite
    differences, slack sensitivities included"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch
    monkeypatch.setenv("ACADOS_AMD_WPI", "1")
    N, B, nx, nu = 4, 2, 8, 3
    data = random_lqr_batch(N=N, batch=B, seed=3)

    def build(dd):
        d = lqr_dims(N, nx, nu)
        d.nbx[1:] = nx
        d.nb[:] = d.nbu + d.nbx
        d.ns[1:] = nx
        gb = OcpQpGpuBatch(d, B, _clib=hostsim_lib)
        for k in range(1, N + 1):
            gb.set_int("idxs_rev", k, np.concatenate([-np.ones(int(d.nbu[k]), dtype=int), np.arange(nx)]))
        fill_lqr_batch(gb, dd, N)
        for k in range(1, N + 1):
            gb.set("lbx", k, np.full((B, nx), -0.5)); gb.set("ubx", k, np.full((B, nx), 0.5))
            for f, v in (("Zl", 1e2), ("Zu", 1e2), ("zl", 1e1), ("zu", 1e1), ("lls", 0.0), ("lus", 0.0)):
                gb.set(f, k, np.full((B, nx), v))
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            gb.opts_set(f, 1e-8)
        gb.opts_set("tol_comp_soft_scale", 1.0)   # sensitivities of a soft class: at the 1e-8 iterate (Gamma = lam / t stays ~1e10)
        assert gb.solve() == 0
        return gb

    def xus(g, pre=""):
        return np.concatenate([g.get(pre + "x", k) for k in range(N + 1)] + [g.get(pre + "u", k) for k in range(N)]
                              + [g.get(pre + "sl", k) for k in range(1, N + 1)] + [g.get(pre + "su", k) for k in range(1, N + 1)], axis=1)

    ref = build(data)
    assert ref.kernel_name == "w16-soft<NX=8,NU=3>"
    e = np.zeros((B, nx)); e[:, 2] = 1.0
    h = 1e-3
    sols = []
    for sg in (+h, -h):
        d = {k: v.copy() for k, v in data.items()}
        d["x0"][:, 2] += sg
        sols.append(xus(build(d)))
    fd = (sols[0] - sols[1]) / (2 * h)
    ref.sens_set("seed_lbx", 0, e); ref.sens_set("seed_ubx", 0, e)
    ref.sens_solve()
    se = xus(ref, "sens_")
    assert np.max(np.abs(se[:, -2 * N * nx:])) > 1e-3          # some slack moves with x0
    assert np.max(np.abs(fd - se)) <= 5e-5 * np.max(np.abs(se))


@pytest.mark.parametrize("wpi", ["0", "1"])
def test_random_structures_larger_dims_hostsim(hostsim_lib, monkeypatch, wpi):
    """random structures with nx up to 12 and nu up to 4 (every second one without general rows): whatever family the
    dispatch picks -- one-instance-per-lane general / box kernels, wave-per-instance GEN, sixteen-lanes box / soft,
    padded into <12,3> / <12,4> -- against the oracle; a shape no compiled one-instance-per-lane set covers runs on the
    wave-per-instance family even when the test override asks for the other one"""
    from acados_amd import OcpQpGpuBatch
    from random_qp import random_structure_qp
    monkeypatch.setenv("ACADOS_AMD_WPI", wpi)
    fams = set()
    for seed in range(100, 130):
        qp = random_structure_qp(seed, nx_max=12, nu_max=4, allow_general=(seed % 2 == 0))
        o = OracleQp(qp)
        assert o.solve(default_opts(tol_stat=1e-8, iter_max=80)) == 0, seed
        b = OcpQpGpuBatch.from_qps([qp] * 3, _clib=hostsim_lib)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            b.opts_set(f, 1e-8)
        b.opts_set("iter_max", 80)
        assert b.solve() == 0, (seed, b.kernel_name)
        fams.add(b.kernel_name.split("<")[0].split("(")[0])
        assert abs(int(b.info("iter")[2]) - o.iter) <= 1, (seed, b.kernel_name)
        try:
            compare_with_oracle(lambda k, f: b.get(f, k)[2], o, qp, 1e-7, fields=("x", "u", "sl", "su", "pi", "lam", "t"))
        except AssertionError as e:
            raise AssertionError(f"seed {seed} kernel {b.kernel_name}: {e}")
    assert len(fams) >= 3, fams


def test_concurrent_solvers_from_host_threads_hostsim(hostsim_lib):
    """the reference's batch idiom calls ocp_qp_solve from OpenMP threads on distinct solver objects
    (acados_solver.in.c:3232-3236): four host threads, each with its own solver and its own QP (one of them partially
    condensed, one with soft constraints), must reproduce the sequential results bit for bit"""
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
        s = AcadosOcpQpSolver(qps[i], opts, _clib=hostsim_lib)
        res = []
        for _ in range(3):
            assert s.solve() == 0
            res.append(np.concatenate([s.get(k, "x") for k in range(qps[i].N + 1)] + [s.get(k, "lam", unique_duals=False) for k in range(qps[i].N + 1)]))
        out[i] = res

    seq = {}
    for i in range(4):
        run(i, seq)
    par = {}
    th = [threading.Thread(target=run, args=(i, par)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in range(4):
        assert len(par[i]) == 3
        for a, c in zip(seq[i], par[i]):
            assert np.array_equal(a, c), i


@pytest.mark.parametrize("fam", ["0", "1"])
def test_degenerate_sizes_hostsim(hostsim_lib, monkeypatch, fam):
    """N = 0 (a single stage, no dynamics, no inputs) and an EMPTY batch (zero instances: solve is a no-op returning 0)"""
    from acados_amd import AcadosOcpQp, OcpQpGpuBatch
    from acados_amd.generators import lqr_dims
    monkeypatch.setenv("ACADOS_AMD_WPI", fam)
    g = np.random.default_rng(5)
    qp = AcadosOcpQp(0)
    M = g.standard_normal((3, 3))
    qp.set("Q", 0, M @ M.T + np.eye(3)); qp.set("q", 0, g.standard_normal(3))
    qp.set("R", 0, np.zeros((0, 0))); qp.set("S", 0, np.zeros((0, 3))); qp.set("r", 0, np.zeros(0))
    qp.set("idxb", 0, np.arange(3)); qp.set("lbx", 0, -0.1 * np.ones(3)); qp.set("ubx", 0, 0.1 * np.ones(3))
    qp.make_consistent()
    o = OracleQp(qp)
    assert o.solve(default_opts(tol_stat=1e-8)) == 0
    b = OcpQpGpuBatch.from_qps([qp] * 2, _clib=hostsim_lib)
    b.opts_set("tol_stat", 1e-8)
    assert b.solve() == 0 and abs(int(b.info("iter")[1]) - o.iter) <= 1
    compare_with_oracle(lambda k, f: b.get(f, k)[1], o, qp, 1e-8, fields=("x", "lam", "t"))
    e = OcpQpGpuBatch(lqr_dims(5, 8, 3), 0, _clib=hostsim_lib)
    assert e.solve() == 0 and e.get("x", 0).shape == (0, 8)


@pytest.mark.parametrize("xbox", [False, True])
def test_small_block_pipeline_bit_identical_hostsim(hostsim_lib, monkeypatch, xbox):
    """nu + nx <= 6: the pipelined one-instance-per-lane kernels (ipm_kernels_box_small.hpp: a stage's loads as one record,
    a ring of records in flight) against the phase-ordered kernels of ipm_kernels_box.hpp they replace -- same arithmetic
    in the same order, so every output and every iteration count must be bit-identical; horizons shorter than, equal to
    and longer than the ring; a ragged batch; a sample against the oracle"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, lqr_instance_qp, random_lqr_batch
    nx, nu, B = 4, 1, 70
    for N in (1, 2, 3, 9):
        data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=B, seed=40 + N)
        out = {}
        for small in ("1", "0"):
            monkeypatch.setenv("ACADOS_AMD_KB_SMALL", small)
            d = lqr_dims(N, nx, nu)
            if xbox:
                d.nbx[1:] = nx
                d.nb[:] = d.nbu + d.nbx
            gb = OcpQpGpuBatch(d, B, _clib=hostsim_lib)
            fill_lqr_batch(gb, data, N)
            if xbox:
                for k in range(1, N + 1):
                    gb.set("lbx", k, np.full((B, nx), -50.0)); gb.set("ubx", k, np.full((B, nx), 50.0))
            gb.opts_set("tol_stat", 1e-8)
            assert gb.solve() == 0
            assert gb.kernel_name.startswith("1tpi-%s<NX=4,NU=1,XBOX=%d" % ("pipe" if small == "1" else "box", int(xbox)))
            out[small] = ([gb.get(f, k) for f in ("x", "lam", "t") for k in range(N + 1)] + [gb.get(f, k) for f in ("u", "pi") for k in range(N)]
                          + [gb.info("iter"), gb.info("res_stat"), gb.info("res_comp")])
            assert gb.res_compute().max() <= 1e-7
        for a, b in zip(out["1"], out["0"]):
            assert np.array_equal(a, b)
        if not xbox:
            for i in (0, 63, 64, 69):
                o = OracleQp(lqr_instance_qp(data, i, N))
                assert o.solve(default_opts(tol_stat=1e-8)) == 0
                assert out["1"][-3][i] == o.iter
                for k in range(N + 1):
                    assert np.allclose(out["1"][k][i], o.get(k, "x"), atol=1e-9)


def test_concurrent_shape_classes_hostsim(hostsim_lib):
    """acados_amd/shape_classes.py: several device batches solved from one host thread each, the class with the most work
    on a high-priority stream; results are those of the classes solved one after the other"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch
    from acados_amd.shape_classes import ConcurrentClasses, estimated_work
    classes = [(4, 1, 6, 9), (8, 3, 5, 7), (4, 1, 12, 5)]
    batches, ref = [], []
    for ci, (nx, nu, N, B) in enumerate(classes):
        data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=B, seed=60 + ci)
        gb = OcpQpGpuBatch(lqr_dims(N, nx, nu), B, _clib=hostsim_lib)
        fill_lqr_batch(gb, data, N)
        gb.opts_set("tol_stat", 1e-8)
        assert gb.solve() == 0
        ref.append([gb.get("x", k).copy() for k in range(N + 1)] + [gb.info("iter").copy()])
        batches.append(gb)
    assert int(np.argmax([estimated_work(b) for b in batches])) == 1
    with ConcurrentClasses(batches) as cc:
        for _ in range(2):
            assert cc.solve() == 0
            for gb, r, (nx, nu, N, B) in zip(batches, ref, classes):
                for k in range(N + 1):
                    assert np.array_equal(gb.get("x", k), r[k])
                assert np.array_equal(gb.info("iter"), r[-1])


def test_device_failure_is_a_status_not_an_exit_hostsim(hostsim_lib, monkeypatch):
    """a HIP error under a solve (injected: the n-th stream synchronisation reports hipErrorLaunchFailure) comes back as -1 from
    the device-batch entry and as ACADOS_QP_FAILURE from the plugin's evaluate -- what ocp_nlp handles (ocp_nlp_sqp.c:720-751) --
    instead of exit(1) taking the process, and the other capsules of an MPC fleet, down (round-3 review); the next call on
    the same objects solves normally"""
    from acados_amd import AcadosOcpQpOptions, AcadosOcpQpSolver, OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, mass_spring_qp, random_lqr_batch
    N, B = 5, 3
    data = random_lqr_batch(N=N, batch=B, seed=3)
    gb = OcpQpGpuBatch(lqr_dims(N, 8, 3), B, _clib=hostsim_lib)
    fill_lqr_batch(gb, data, N)
    assert gb.solve() == 0
    x_ok = gb.get("x", N).copy()
    monkeypatch.setenv("GQP_HOSTSIM_FAIL_SYNC", "2")
    assert gb.solve() == -1
    monkeypatch.setenv("GQP_HOSTSIM_FAIL_SYNC", "")
    assert gb.solve() == 0 and np.array_equal(gb.get("x", N), x_ok)
    # through the 22-slot vtable (acados_c layer)
    qp = mass_spring_qp(N=6)
    opts = AcadosOcpQpOptions()
    s = AcadosOcpQpSolver(qp, opts=opts, _clib=hostsim_lib)
    assert s.solve() == 0
    u_ok = s.get(0, "u").copy()
    monkeypatch.setenv("GQP_HOSTSIM_FAIL_SYNC", "3")
    assert s.solve() == 4            # ACADOS_QP_FAILURE (types.h:74-87)
    monkeypatch.setenv("GQP_HOSTSIM_FAIL_SYNC", "")
    assert s.solve() == 0 and np.allclose(s.get(0, "u"), u_ok, atol=1e-12)


def _full_dense_case(clib, qps, tol=1e-8):
    """FULL CONDENSING of any size (option full_dense; dense_kernels.hpp): every state but x0 condensed, the IPM on the dense problem,
    expansion -- against the oracle (stage-wise Riccati IPM on the original QP) and the independent KKT residual kernel"""
    from acados_amd import OcpQpGpuBatch
    gb = OcpQpGpuBatch.from_qps(qps, _clib=clib)
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        gb.opts_set(f, tol)
    gb.opts_set("full_dense", 1)
    assert gb.solve() == 0, gb.info("status")
    assert gb.res_compute().max() <= tol * (1.0 + 1e-3) + 1e-12, gb.res_compute()
    worst = 0.0
    for i, qp in enumerate(qps):
        o = OracleQp(qp)
        assert o.solve(default_opts(tol_stat=tol, tol_eq=tol, tol_ineq=tol, tol_comp=tol)) == 0
        for k in range(qp.N + 1):
            for f in ("x", "u", "pi", "lam") if k < qp.N else ("x", "lam"):
                ref = o.get(k, f)
                if ref.size:
                    worst = max(worst, float(np.max(np.abs(gb.get(f, k)[i][:ref.size] - ref) / np.maximum(1.0, np.abs(ref)))))
        assert abs(int(gb.info("iter")[i]) - o.iter) <= 3
    return gb, worst


def test_full_condensing_dense_path_hostsim(hostsim_lib):
    from acados_amd.generators import lqr_instance_qp, mass_spring_qp, random_lqr_batch
    N = 4
    data = random_lqr_batch(N=N, batch=3, seed=31)
    gb, worst = _full_dense_case(hostsim_lib, [lqr_instance_qp(data, i, N) for i in range(3)], tol=1e-10)
    assert int(gb.scalar("dense_columns")) == (N + 1) * 3 + 8 and worst <= 1e-7, worst
    # state bounds behind stage 0 become rows of the state map (mass-spring: nb = 11 = 3 inputs + 8 states per stage)
    gb, worst = _full_dense_case(hostsim_lib, [mass_spring_qp(N=3)], tol=1e-10)
    assert worst <= 1e-7, worst


def test_full_condensing_dense_path_random_structures_hostsim(hostsim_lib):
    """the dense path on random STRUCTURES without slacks (per-stage dims, box subsets, one-sided rows through the masks, general rows,
    x0 fixed or free): converged, KKT residuals of the ORIGINAL QP by the independent kernel, primal solution against the oracle"""
    from acados_amd import OcpQpGpuBatch
    from random_qp import random_structure_qp
    done = 0
    for seed in range(14):
        qp = random_structure_qp(seed, allow_slack=False)
        o = OracleQp(qp)
        if o.solve(default_opts(tol_stat=1e-10, tol_eq=1e-10, tol_ineq=1e-10, tol_comp=1e-10)) != 0:
            continue
        gb = OcpQpGpuBatch.from_qps([qp, qp], _clib=hostsim_lib)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            gb.opts_set(f, 1e-10)
        gb.opts_set("full_dense", 1)
        assert gb.solve() == 0, (seed, gb.info("status"))
        assert gb.res_compute().max() <= 1e-10 * (1.0 + 1e-3) + 1e-12, (seed, gb.res_compute())
        for k in range(qp.N + 1):
            for f in ("x", "u") if k < qp.N else ("x",):
                ref = o.get(k, f)
                if ref.size:
                    assert np.allclose(gb.get(f, k)[1][:ref.size], ref, rtol=1e-6, atol=1e-7), (seed, k, f)
        done += 1
    assert done >= 10


def test_kernels_against_certified_solutions_on_random_structures_hostsim(hostsim_lib):
    """conftest.certified_random_structures_case on the host simulation: ten random structures (the GPU tier runs sixty)"""
    worst = certified_random_structures_case(hostsim_lib, (0, 1, 5, 7, 11, 13, 22, 27, 33, 38))
    print("kernels (host simulation) vs certified dense solutions:", {k: f"{v:.1e}" for k, v in worst.items()})
