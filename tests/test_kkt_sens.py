"""a10 / a11 / a12 against references that do NOT come from the solver under test:

  a10  KKT residuals of an arbitrary (qp_in, qp_out): the device kernel behind ocp_qp_gpu_batch_res_compute /
       ocp_qp_res_compute / ocp_qp_inf_norm_residuals (acados/ocp_qp/ocp_qp_common.c:559-667,
       interfaces/acados_c/ocp_qp_interface.c:642-650) against a plain-NumPy restatement (tests/dense_ref.py) on solved
       AND on random iterates, and as the assertion of the reference's unit test (test/ocp_qp/test_qpsolvers.cpp:240-251).
  a11  P p K k Lr (ocp_qp_hpipm.c:417-478) against the oracle's factor.
  a12  eval_forw_sens / eval_adj_sens (ocp_qp_hpipm.c:481-506) against ONE dense solve of the linearised KKT system at
       the ORACLE's solution (tests/dense_ref.py sens_dense) -- every instance, every seed kind.

Every test exists in both tiers: `hostsim` (kernel sources under g++, CPU) and `gpu` (the product library)."""
import numpy as np
import pytest

from conftest import GOLDEN_PAIRS, INPUT_ONLY, load_qp
from dense_ref import kkt_residual_norms, kkt_residuals, sens_dense
from oracle.oracle import OracleQp, default_opts

# tolerance of an INDEPENDENTLY recomputed residual for a solve at tol 1e-8: the IPM judges complementarity by
# |lam t - tau| with the barrier floor tau = 1e-3 tol_comp, the residual kernel reports lam t itself
KKT_TOL = 1e-8 * (1.0 + 1e-3) + 1e-13

ALL_QPS = [p for p, _ in GOLDEN_PAIRS] + INPUT_ONLY
TIERS = [pytest.param("hostsim", id="hostsim"), pytest.param("gpu", id="gpu", marks=pytest.mark.gpu)]


@pytest.fixture
def clib(request):
    return request.getfixturevalue("hostsim_lib" if request.param == "hostsim" else "gpu_lib")


def _getter(gb, i):
    return lambda k, f: gb.get(f, k)[i] if not (f == "pi" and k == gb.N) else np.zeros(0)


def _res_vectors(gb, i, k):
    g = np.concatenate([gb.get("res_g", k)[i], gb.get("res_gs", k)[i]])
    return {"res_g": g, "res_b": gb.get("res_b", k)[i], "res_d": gb.get("res_d", k)[i], "res_m": gb.get("res_m", k)[i]}


@pytest.mark.parametrize("clib", TIERS, indirect=True)
@pytest.mark.parametrize("wpi", ["0", "1"])
@pytest.mark.parametrize("qp_file", ALL_QPS)
def test_res_compute_matches_numpy(clib, monkeypatch, qp_file, wpi):
    """the residual kernel on both HBM layouts: (i) at the solver's solution the independently recomputed norms are at
    tolerance and equal the NumPy ones; (ii) on a RANDOM iterate (nothing the solver produced) every residual vector
    equals the NumPy restatement element by element"""
    from acados_amd import OcpQpGpuBatch
    monkeypatch.setenv("ACADOS_AMD_WPI", wpi)
    qp = load_qp(qp_file)
    gb = OcpQpGpuBatch.from_qps([qp, qp, qp], _clib=clib)
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        gb.opts_set(f, 1e-8)
    assert gb.solve() == 0
    nrm = gb.res_compute()
    assert nrm.shape == (3, 4) and np.all(nrm <= KKT_TOL), nrm
    ref = kkt_residual_norms(qp, _getter(gb, 1))
    assert np.allclose(nrm[1], ref, rtol=1e-6, atol=1e-11), (nrm[1], ref)   # residuals of O(1) terms: absolute rounding ~1e-12
    # the solver's own by-product norms say the same (they are what the status decision used)
    own = np.array([gb.info(n)[1] for n in ("res_stat", "res_eq", "res_ineq", "res_comp")])
    assert np.all(own <= 1e-8)
    # (ii) random iterate: positive lam / t, anything else
    rng = np.random.default_rng(5)
    d = qp.dims
    for k in range(qp.N + 1):
        nct = 2 * int(d.nb[k] + d.ng[k] + d.ns[k])
        for f, n in (("u", int(d.nu[k])), ("x", int(d.nx[k])), ("sl", int(d.ns[k])), ("su", int(d.ns[k])),
                     ("pi", int(d.nx[k + 1]) if k < qp.N else 0)):
            if n:
                gb.set(f, k, rng.standard_normal((3, n)))
        if nct:
            gb.set("lam", k, rng.uniform(0.1, 2.0, (3, nct)))
            gb.set("t", k, rng.uniform(0.1, 2.0, (3, nct)))
    nrm = gb.res_compute()
    get = _getter(gb, 2)
    want = kkt_residuals(qp, get)
    for k in range(qp.N + 1):
        got = _res_vectors(gb, 2, k)
        for name in ("res_g", "res_b", "res_d", "res_m"):
            assert np.allclose(got[name], want[name][k], rtol=1e-12, atol=1e-12), (name, k, got[name], want[name][k])
    assert np.allclose(nrm[2], kkt_residual_norms(qp, get), rtol=1e-12)


@pytest.mark.parametrize("clib", TIERS, indirect=True)
@pytest.mark.parametrize("N2", [15, 5, 3])
def test_inf_norm_residuals_mass_spring(clib, N2):
    """the assertion of the reference's unit test (test/ocp_qp/test_qpsolvers.cpp:238-251): mass-spring N = 15, partial
    condensing N2 in {15, 5, 3}, status 0 and max KKT residual <= 1e-8 -- the residual evaluated by
    ocp_qp_inf_norm_residuals on (qp_in, qp_out), not read from the solver"""
    from acados_amd import AcadosOcpQpOptions, AcadosOcpQpSolver
    from acados_amd.generators import mass_spring_qp
    qp = mass_spring_qp(N=15)
    opts = AcadosOcpQpOptions()
    opts.tol_stat = opts.tol_eq = opts.tol_ineq = opts.tol_comp = 1e-8
    opts.cond_N = N2
    s = AcadosOcpQpSolver(qp, opts, _clib=clib)
    assert s.solve() == 0
    res = s.inf_norm_residuals()
    assert res.shape == (4,) and np.max(res) <= KKT_TOL, res
    ref = kkt_residual_norms(qp, lambda k, f: s.get(k, f, unique_duals=False) if not (f == "pi" and k == qp.N) else np.zeros(0))
    assert np.allclose(res, ref, rtol=1e-6, atol=1e-11)


@pytest.mark.gpu
def test_riccati_getters_gpu(gpu_lib, monkeypatch):
    """a11 on the device (both kernel families): ric_L / ric_l and P p K k Lr of the last factorisation against the
    oracle's factor at the same iterate, and through the solver_get slot of the vtable"""
    import ctypes as C
    from acados_amd import AcadosOcpQpOptions, AcadosOcpQpSolver, OcpQpGpuBatch
    from acados_amd.generators import mass_spring_qp
    qp = mass_spring_qp(N=8)
    o = OracleQp(qp)
    assert o.solve(default_opts(tol_stat=1e-8)) == 0
    o.refactor()
    for wpi in ("0", "1"):
        monkeypatch.setenv("ACADOS_AMD_WPI", wpi)
        b = OcpQpGpuBatch.from_qps([qp] * 5, _clib=gpu_lib)
        b.opts_set("tol_stat", 1e-8)
        assert b.solve() == 0
        for k in range(qp.N + 1):
            nu, nv = int(qp.dims.nu[k]), int(qp.dims.nu[k] + qp.dims.nx[k])
            Lo = o.get(k, "ric_L").reshape(nv, nv, order="F")
            Lg = b.get("ric_L", k)[3].reshape(nv, nv, order="F")
            assert np.allclose(Lg, Lo, rtol=1e-6, atol=1e-9), (wpi, k)
            assert np.allclose(b.get("ric_l", k)[3], o.get(k, "ric_l"), rtol=1e-5, atol=1e-9), (wpi, k)
            ric = b.riccati(k)
            Lx = Lo[nu:, nu:]
            assert np.allclose(ric["P"][4], Lx @ Lx.T, rtol=1e-6, atol=1e-9)
            assert np.allclose(ric["p"][4], Lx @ o.get(k, "ric_l")[nu:], rtol=1e-5, atol=1e-9)
            if nu:
                M = Lo @ Lo.T   # [Lr 0; Ls Lx][Lr' Ls'; 0 Lx'] = M  =>  K = -Muu^-1 Mux, k = -Lr^-T lr
                # (an input pinned at a bound carries Gamma = lam/t ~ 1e11 on the diagonal: its gain is ~1e-10 and moves
                # with t from iterate to iterate -- absolute tolerance 1e-7 on the gains)
                assert np.allclose(ric["K"][4], -np.linalg.solve(M[:nu, :nu], M[:nu, nu:]), rtol=1e-5, atol=1e-7)
                assert np.allclose(ric["k"][4], -np.linalg.solve(Lo[:nu, :nu].T, o.get(k, "ric_l")[:nu]), rtol=1e-5, atol=1e-7)
                assert np.allclose(ric["Lr"][4], Lo[:nu, :nu], rtol=1e-6, atol=1e-9)
    # the solver_get slot (ocp_qp_common.h:73; ocp_nlp_ddp.c:373-377 reads K, k through it)
    opts = AcadosOcpQpOptions()
    opts.tol_stat = opts.tol_eq = opts.tol_ineq = 1e-8     # tol_comp stays at the oracle's default (1e-8 there, set below)
    opts.tol_comp = 1e-8
    s = AcadosOcpQpSolver(qp, opts, _clib=gpu_lib)
    assert s.solve() == 0
    L = gpu_lib
    L.ocp_qp_solver_get_ric.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
    nu, nx, k = 3, 8, 2
    K = np.zeros((nx, nu)); kk = np.zeros(nu); P = np.zeros((nx, nx)); p = np.zeros(nx); Lr = np.zeros((nu, nu))
    for name, arr, s1, s2 in (("K", K, nu, nx), ("k", kk, nu, 1), ("P", P, nx, nx), ("p", p, nx, 1), ("Lr", Lr, nu, nu)):
        L.ocp_qp_solver_get_ric(s.c_solver, s.c_in, s.c_out, name.encode(), k, arr.ctypes.data_as(C.c_void_p), s1, s2)
    Lo = o.get(k, "ric_L").reshape(nu + nx, nu + nx, order="F")
    M = Lo @ Lo.T
    assert np.allclose(K.T, -np.linalg.solve(M[:nu, :nu], M[:nu, nu:]), rtol=1e-5, atol=1e-7)    # column-major nu x nx
    assert np.allclose(P.T, Lo[nu:, nu:] @ Lo[nu:, nu:].T, rtol=1e-6, atol=1e-9)
    assert np.allclose(p, Lo[nu:, nu:] @ o.get(k, "ric_l")[nu:], rtol=1e-5, atol=1e-9)
    assert np.allclose(Lr.T, Lo[:nu, :nu], rtol=1e-6, atol=1e-9)
    assert np.allclose(kk, -np.linalg.solve(Lo[:nu, :nu].T, o.get(k, "ric_l")[:nu]), rtol=1e-5, atol=1e-7)


def _check_sens(gb, qps, seeds_dev, seeds_dense, tol, fields=("x", "u", "pi", "sl", "su", "lam", "t"), tol_own=1e-8, tol_mult=1e-4,
                tol_mult_own=2e-6, tol_solve=1e-9, soft_scale=None):
    """device sensitivities of EVERY instance against the dense linearised-KKT solve (tests/dense_ref.py), twice:
    at the ORACLE's solution of the same QP -- a reference the solver under test had no part in: primal and pi at `tol`,
    lam / t at `tol_mult` (d lam of a nearly active side depends on how far the complementarity products of the two
    iterates -- both inside the 1e-9 ball -- differ); and at the device's own final iterate, where the dense solve and
    the two sweeps with the stored factor must agree to rounding (`tol_own`, every field)."""
    for (f, k, v) in seeds_dev:
        gb.sens_set(f, k, v)
    gb.sens_solve()
    worst = 0.0
    for i, qp in enumerate(qps):
        o = OracleQp(qp)
        assert o.solve(default_opts(tol_stat=tol_solve, tol_eq=tol_solve, tol_ineq=tol_solve, tol_comp=tol_solve), soft_scale=soft_scale) == 0
        sd = {key: val[i] for key, val in seeds_dense.items()}
        for which, ref in (("oracle", sens_dense(qp, o.get, sd)), ("own", sens_dense(qp, _getter(gb, i), sd))):
            scale = max(1.0, max(np.max(np.abs(ref(k, f))) for k in range(qp.N + 1) for f in ("x", "u") if ref(k, f).size))
            for k in range(qp.N + 1):
                for f in fields:
                    if f == "pi" and k == qp.N:
                        continue
                    want = ref(k, f)
                    if want.size == 0:
                        continue
                    got = gb.get("sens_" + f, k)[i]
                    if f in ("lam", "t"):
                        # sides that do not take part (masked, equality-flagged) carry no sensitivity in the reference
                        sel = np.array([(k, e) in ref.active for e in range(want.size)])
                        got, want = got[sel], want[sel]
                        if want.size == 0:
                            continue
                    err = np.max(np.abs(got - want)) / max(scale, np.max(np.abs(want)))
                    # (dense LU on a matrix holding t ~ 1e-10 next to lam ~ 1: its own d lam is good to ~1e-6)
                    lim = (max(tol_own, tol_mult_own) if f in ("lam", "t") else tol_own) if which == "own" else (tol_mult if f in ("lam", "t") else tol)
                    assert err <= lim, (which, i, k, f, err, got, want)
                    if which == "oracle" and f not in ("lam", "t"):
                        worst = max(worst, err)
    return worst


@pytest.mark.parametrize("clib", TIERS, indirect=True)
@pytest.mark.parametrize("fam", ["w16", "wpi", "1tpi"])
def test_sensitivities_box_vs_dense(clib, request, monkeypatch, fam):
    """a12, box-constrained class (C2 shape): seeds in q, r, b, x0 (equality-flagged row: both sides, as
    ocp_nlp_common.c:4057-4064 sets them) and an input bound, on the three kernel families, EVERY instance against the
    dense linearised-KKT solve at the oracle's solution at 1e-6"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, lqr_instance_qp, random_lqr_batch
    gpu = "gpu" in request.node.callspec.id.split("-")
    monkeypatch.setenv("ACADOS_AMD_WPI", "0" if fam == "1tpi" else "1")
    monkeypatch.setenv("ACADOS_AMD_W16", "1" if fam == "w16" else "0")
    monkeypatch.setenv("ACADOS_AMD_SENS_SLICE", "5")
    N, nx, nu = (10, 8, 3) if gpu else (4, 8, 3)
    B = 24 if gpu else (7 if fam == "1tpi" else 3)
    data = random_lqr_batch(N=N, batch=B, seed=21)
    gb = OcpQpGpuBatch(lqr_dims(N, nx, nu), B, _clib=clib)
    fill_lqr_batch(gb, data, N)
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        gb.opts_set(f, 1e-9)
    assert gb.solve() == 0
    assert gb.kernel_name.startswith({"w16": "w16-box", "wpi": "wpi-box", "1tpi": "1tpi-box"}[fam])
    qps = [lqr_instance_qp(data, i, N) for i in range(B)]
    rng = np.random.default_rng(3)
    ex, eu = rng.standard_normal((B, nx)), rng.standard_normal((B, nu))
    cases = {
        "q": ([("seed_q", k, ex) for k in range(N + 1)], {("q", k): ex for k in range(N + 1)}),
        "r": ([("seed_r", 1, eu)], {("r", 1): eu}),
        "b": ([("seed_b", k, ex) for k in range(N)], {("b", k): ex for k in range(N)}),
        "x0": ([("seed_lbx", 0, ex), ("seed_ubx", 0, ex)], {("lbx", 0): ex, ("ubx", 0): ex}),
        "ubu": ([("seed_ubu", k, eu) for k in range(N)], {("ubu", k): eu for k in range(N)}),
        "lbu": ([("seed_lbu", 0, eu)], {("lbu", 0): eu}),
    }
    for name, (sdev, sdense) in cases.items():
        print("seed case", name)
        # a seed that moves an ACTIVE bound: d t = d(bound) - d u is the difference of two O(1) numbers that agree to
        # ~12 digits (t ~ 1e-12 at tol_comp 1e-9), so d lam = -(lam/t) d t carries 1e-5 .. 1e-3 relative rounding -- in the
        # sweeps as in any IPM-linearised solve; the primal sensitivities are not affected
        bound = name in ("ubu", "lbu")
        worst = _check_sens(gb, qps, sdev, sdense, 1e-6, fields=("x", "u", "pi", "lam", "t"), tol_mult_own=1e-2 if bound else 2e-6,
                            tol_mult=1e-2 if bound else 1e-4)
        assert worst <= 1e-6, (name, worst)


@pytest.mark.parametrize("fam", ["w16r-gen<", "wpi-gen("])
@pytest.mark.parametrize("clib", TIERS, indirect=True)
def test_sensitivities_soft_and_general_rows_vs_dense(clib, request, monkeypatch, fam):
    """a12 on the general-constraint / slack kernels (C4 class: soft state bounds + soft general rows): seeds in q, b,
    x0 and in a general-row bound; x, u, slacks, pi and the multipliers of every instance against the dense solve at the
    oracle's solution (extended-precision LU).  Tolerance 2e-4 relative at the acados tolerances (1e-8): the stage matrix
    H + sum Gamma a a' of an active GENERAL soft row has condition number ~ Gamma = lam/t ~ 1e10, and a direction out of
    its Cholesky factor carries Gamma * eps of rounding -- measured 1e-10 at tol 1e-5, 2e-5 at 1e-8, 5e-4 at 1e-10
    against the same dense solve at the device's own iterate; the IPM iteration itself is self-correcting, a
    one-shot sensitivity solve is not (same structure in any Riccati-based IPM)"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import chain_soft_batch, chain_soft_dims, chain_soft_instance_qp, fill_chain_soft_batch
    gpu = "gpu" in request.node.callspec.id.split("-")
    monkeypatch.setenv("ACADOS_AMD_WPI", "1")
    monkeypatch.setenv("ACADOS_AMD_W16G", "1" if fam.startswith("w16r") else "0")   # both families that carry general rows + slacks
    N, B = (4, 5) if gpu else (3, 2)
    data = chain_soft_batch(N=N, batch=B, seed=1)
    gb = OcpQpGpuBatch(chain_soft_dims(N), B, _clib=clib)
    fill_chain_soft_batch(gb, data, N)
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        gb.opts_set(f, 1e-8)
    gb.opts_set("tol_comp_soft_scale", 1.0)   # the docstring's Gamma * eps: a one-shot sensitivity solve is taken at the 1e-8 iterate
    assert gb.solve() == 0
    assert gb.kernel_name.startswith(fam)
    qps = [chain_soft_instance_qp(data, i, N) for i in range(B)]
    rng = np.random.default_rng(9)
    ex, eg = rng.standard_normal((B, 24)), rng.standard_normal((B, 4))
    cases = {
        "q": ([("seed_q", 2, ex)], {("q", 2): ex}),
        "b": ([("seed_b", 1, ex)], {("b", 1): ex}),
        "x0": ([("seed_lbx", 0, ex), ("seed_ubx", 0, ex)], {("lbx", 0): ex, ("ubx", 0): ex}),
        "ug": ([("seed_ug", 2, eg)], {("ug", 2): eg}),
        "lg": ([("seed_lg", 1, eg)], {("lg", 1): eg}),
    }
    for name, (sdev, sdense) in cases.items():
        print("seed case", name)
        worst = _check_sens(gb, qps, sdev, sdense, 2e-4, tol_own=2e-4, tol_mult_own=1e-3 if name in ("ug", "lg") else 1e-4,
                            tol_mult=1e-3 if name in ("ug", "lg") else 5e-4, tol_solve=1e-8, soft_scale=1.0)
        assert worst <= 2e-4, (name, worst)


@pytest.mark.parametrize("clib", TIERS, indirect=True)
@pytest.mark.parametrize("pf", ["0", "2"])
def test_mfma_blocked_cholesky_factor_kernel(clib, monkeypatch, pf):
    """kw_factor_m (ipm_kernels_wpi_mfma.hpp): W = [B A]' Lx+, M += W W' and the rank-4 trailing updates of a blocked
    Cholesky on v_mfma_f64_16x16x4_f64, forced on for every shape in its range (17 <= nu + nx <= 32): box class
    nx = 24 nu = 6 (two column tiles of the x-block), nx = 14 nu = 4 (one), the condensed C3 shape nx = 8 nu = 15, and the
    C4 class with general rows + slacks -- against the oracle, and bit-for-bit status / iteration agreement with the
    register-tile kernel it stands in for"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import chain_soft_qp, lqr_instance_qp, random_lqr_batch
    monkeypatch.setenv("ACADOS_AMD_WPI", "1")
    monkeypatch.setenv("ACADOS_AMD_W16R", "0")   # the box shapes of this range default to the two-rows-per-lane family
    monkeypatch.setenv("ACADOS_AMD_WPI_MFMA_PF", pf)
    cases = []
    for nx, nu, N in ((24, 6, 5), (14, 4, 6), (8, 15, 4), (17, 15, 3)):
        data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=3, seed=40 + nx)
        cases.append([lqr_instance_qp(data, i, N) for i in range(3)])
    cases.append([chain_soft_qp(i, N=5) for i in range(3)])
    for qps in cases:
        res = {}
        for mf in ("1", "0"):
            monkeypatch.setenv("ACADOS_AMD_WPI_MFMA", mf)
            b = OcpQpGpuBatch.from_qps(qps, _clib=clib)
            for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
                b.opts_set(f, 1e-8)
            assert b.solve() == 0, b.kernel_name
            assert ("mfma" in b.kernel_name) == (mf == "1"), b.kernel_name
            assert b.res_compute().max() <= KKT_TOL
            res[mf] = (b.info("iter").copy(), [b.get("x", k) for k in range(qps[0].N + 1)], b.get("ric_L", 1), b)
        assert np.array_equal(res["1"][0], res["0"][0])
        for xa, xb in zip(res["1"][1], res["0"][1]):
            assert np.allclose(xa, xb, rtol=1e-9, atol=1e-10)
        # the factor itself, compared through the matrix it factors (entries of L next to Gamma ~ 1e10 pivots carry the
        # conditioning of the stage matrix, L L' does not)
        nv = int(qps[0].dims.nu[1] + qps[0].dims.nx[1])
        for i in range(len(qps)):
            La, Lb = res["1"][2][i].reshape(nv, nv, order="F"), res["0"][2][i].reshape(nv, nv, order="F")
            Ma, Mb = La @ La.T, Lb @ Lb.T
            assert np.max(np.abs(Ma - Mb)) <= 1e-7 * np.max(np.abs(Mb))   # (Gamma = lam / t of the two final iterates agrees to ~1e-8)
        for i, qp in enumerate(qps):
            o = OracleQp(qp)
            assert o.solve(default_opts(tol_stat=1e-8)) == 0
            for k in range(qp.N + 1):
                assert np.allclose(res["1"][1][k][i], o.get(k, "x"), rtol=1e-7, atol=1e-8)
