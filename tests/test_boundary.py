"""The drop-in boundary where the reference binds it, both tiers (hostsim / gpu):

  - the vtable layouts: 17-slot qp_solver_config, 20-slot ocp_qp_xcond_config, 22-slot ocp_qp_xcond_solver_config + the
    two sub-vtables, every slot filled (ocp_qp_common.h:60-107, ocp_qp_xcond_solver.h:81-107)
  - RTI split condense_lhs / condense_rhs_and_solve (ocp_qp_xcond_solver.c:591-669) on a class with general rows + slacks
  - hot start of a FRESH batch from a perturbed, non-converged iterate (root and, with partial condensing, the iterate
    handed to the condensed QP: condense_qp_out, ocp_qp_xcond_solver.c:554-565)
  - the factorize-only contract of ocp_nlp_common.c:3946-3971: warm_start 3, iter_max 0, update_fact_exit 1,
    t0_min / lam0_min, then sensitivities
  - FULL_CONDENSING_GPU_IPM: one block where nx + N nu <= 64, the dense path (dense_kernels.hpp) beyond
  - the 20-slot condensing module composed the way ocp_qp_xcond_solve composes it (condensing -> inner evaluate on the
    condensed QP -> expansion), i.e. the DEVICE condensing reached through reference-shaped slots
"""
import ctypes as C
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, compare_with_oracle, load_qp
from oracle.oracle import OracleQp, default_opts

TIERS = [pytest.param("hostsim", id="hostsim"), pytest.param("gpu", id="gpu", marks=pytest.mark.gpu)]


@pytest.fixture
def clib(request):
    return request.getfixturevalue("hostsim_lib" if request.param == "hostsim" else "gpu_lib")


@pytest.mark.parametrize("clib", TIERS, indirect=True)
def test_vtable_layouts(clib):
    """the three config structs as arrays of pointers: sizes and no empty slot"""
    L = clib
    vp = C.c_void_p
    L.ocp_qp_xcond_solver_config_create_from_name.restype = vp
    L.ocp_qp_xcond_solver_config_create_from_name.argtypes = [C.c_char_p]
    L.ocp_qp_xcond_solver_config_calculate_size.restype = C.c_size_t
    for name in (b"PARTIAL_CONDENSING_GPU_IPM", b"PARTIAL_CONDENSING_HPIPM", b"FULL_CONDENSING_GPU_IPM"):
        cfg = L.ocp_qp_xcond_solver_config_create_from_name(name)
        assert cfg
        outer = (vp * 24).from_address(cfg)
        assert all(outer[q] for q in range(24)), name            # 22 function pointers + qp_solver + xcond
        inner = (vp * 17).from_address(outer[22])                # qp_solver_config: 17 slots (ocp_qp_common.h:60-79)
        xcond = (vp * 20).from_address(outer[23])                # ocp_qp_xcond_config: 20 slots (:84-107)
        assert all(inner[q] for q in range(17)) and all(xcond[q] for q in range(20)), name
        L.ocp_qp_xcond_solver_config_free(vp(cfg))
    assert L.ocp_qp_xcond_solver_config_calculate_size() >= 8 * (24 + 17 + 20)
    assert not L.ocp_qp_xcond_solver_config_create_from_name(b"PARTIAL_CONDENSING_OSQP")
    # the slots are the exported functions, in the reference's order
    cfg = L.ocp_qp_xcond_solver_config_create_from_name(b"PARTIAL_CONDENSING_GPU_IPM")
    outer = (vp * 24).from_address(cfg)
    addr = lambda n: C.cast(getattr(L, n), vp).value
    assert outer[16] == addr("ocp_qp_gpu_xcond_solve") and outer[17] == addr("ocp_qp_gpu_xcond_condense_lhs")
    assert outer[18] == addr("ocp_qp_gpu_xcond_condense_rhs_and_solve") and outer[21] == addr("ocp_qp_gpu_xcond_solver_terminate")
    inner = (vp * 17).from_address(outer[22])
    assert inner[11] == addr("ocp_qp_gpu_ipm") and inner[16] == addr("ocp_qp_gpu_ipm_terminate")
    xcond = (vp * 20).from_address(outer[23])
    assert xcond[13] == addr("ocp_qp_gpu_pcond_condensing") and xcond[14] == addr("ocp_qp_gpu_pcond_condense_rhs")
    assert xcond[16] == addr("ocp_qp_gpu_pcond_condense_lhs") and xcond[17] == addr("ocp_qp_gpu_pcond_condense_qp_out")
    assert xcond[18] == addr("ocp_qp_gpu_pcond_expansion")
    L.ocp_qp_xcond_solver_config_free(vp(cfg))


def _soft_qp(i, N):
    from acados_amd.generators import chain_soft_qp
    return chain_soft_qp(i, N=N)


@pytest.mark.parametrize("clib", TIERS, indirect=True)
def test_rti_split_general_rows(clib, monkeypatch):
    """RTI split on the C4 class (soft state bounds + soft general rows): condense_lhs with the matrices, THEN new
    vectors (gradient, dynamics offset, x0, a general bound), condense_rhs_and_solve -- equals the oracle's solution of
    the QP with the new vectors; through the batch API and through the two slots of the xcond-solver vtable"""
    import copy
    from acados_amd import AcadosOcpQpOptions, AcadosOcpQpSolver, OcpQpGpuBatch
    monkeypatch.setenv("ACADOS_AMD_WPI", "1")
    N, N2 = 6, 3
    rng = np.random.default_rng(2)
    qp = _soft_qp(0, N)
    qp2 = copy.deepcopy(qp)
    for k in range(N + 1):
        qp2.set("q", k, qp.q[k] + 0.3 * rng.standard_normal(qp.q[k].shape))
        if k < N:
            qp2.set("b", k, qp.b[k] + 0.02 * rng.standard_normal(qp.b[k].shape))
    x0 = qp.lbx[0] + 0.1 * rng.standard_normal(qp.lbx[0].shape)
    qp2.set("lbx", 0, x0); qp2.set("ubx", 0, x0)
    qp2.set("ug", 2, qp.ug[2] - 0.1)
    o = OracleQp(qp2)
    assert o.solve(default_opts(tol_stat=1e-8)) == 0
    # batch API
    b = OcpQpGpuBatch.from_qps([qp] * 3, _clib=clib)
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        b.opts_set(f, 1e-8)
    b.opts_set("cond_N", N2)
    assert b.condense_lhs() == 0
    for k in range(N + 1):
        b.set("q", k, np.tile(qp2.q[k], (3, 1)))
        if k < N:
            b.set("b", k, np.tile(qp2.b[k], (3, 1)))
    b.set("lbx", 0, np.tile(x0, (3, 1))); b.set("ubx", 0, np.tile(x0, (3, 1)))
    b.set("ug", 2, np.tile(qp2.ug[2], (3, 1)))
    assert b.condense_rhs_and_solve() == 0 and int(b.scalar("cond_N_active")) == N2
    assert b.res_compute().max() <= 2e-8
    compare_with_oracle(lambda k, f: b.get(f, k)[2], o, qp2, 1e-5, fields=("x", "u", "sl", "su", "pi"))
    # the two slots of the outer vtable (what ocp_nlp_sqp_rti.c:509, 1115 call)
    opts = AcadosOcpQpOptions()
    opts.tol_stat = opts.tol_eq = opts.tol_ineq = opts.tol_comp = 1e-8
    opts.cond_N = N2
    s = AcadosOcpQpSolver(qp, opts, _clib=clib)
    assert s.condense_lhs() == 0
    for k in range(N + 1):
        s.set(k, "q", qp2.q[k])
        if k < N:
            s.set(k, "b", qp2.b[k])
    s.set(0, "lbx", x0); s.set(0, "ubx", x0)
    s.set(2, "ug", qp2.ug[2])
    assert s.condense_rhs_and_solve() == 0
    assert s.inf_norm_residuals().max() <= 2e-8
    compare_with_oracle(lambda k, f: s.get(k, f), o, qp2, 1e-5, fields=("x", "u", "sl", "su", "pi"))
    assert s.get_stats("time_qp_xcond") > 0.0


@pytest.mark.parametrize("clib", TIERS, indirect=True)
@pytest.mark.parametrize("fam", ["1tpi", "w16", "wpi-gen"])
def test_hot_start_fresh_batch_from_perturbed_iterate(clib, monkeypatch, fam):
    """warm_start 3 on a batch that has NEVER been solved, from a perturbed (non-converged) iterate: converges to the
    oracle's solution in fewer iterations than the cold start; with partial condensing the root's iterate is what the
    condensed QP starts from (condense_qp_out).  (A stale per-instance step length of 0 used to read as MINSTEP.)"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import lqr_instance_qp, random_lqr_batch
    monkeypatch.setenv("ACADOS_AMD_WPI", "0" if fam == "1tpi" else "1")
    monkeypatch.setenv("ACADOS_AMD_W16", "1" if fam == "w16" else "0")
    if fam == "wpi-gen":
        N = 4
        qps = [_soft_qp(i, N) for i in range(3)]
    else:
        N = 6
        data = random_lqr_batch(N=N, batch=3, seed=31)
        qps = [lqr_instance_qp(data, i, N) for i in range(3)]
    oracles = []
    for qp in qps:
        o = OracleQp(qp)
        assert o.solve(default_opts(tol_stat=1e-8)) == 0
        oracles.append(o)
    rng = np.random.default_rng(1)
    for cond_N in (N, 2):
        cold = OcpQpGpuBatch.from_qps(qps, _clib=clib)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            cold.opts_set(f, 1e-8)
        cold.opts_set("cond_N", cond_N)
        assert cold.solve() == 0
        it_cold = cold.info("iter").copy()
        b = OcpQpGpuBatch.from_qps(qps, _clib=clib)     # fresh: never solved
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            b.opts_set(f, 1e-8)
        b.opts_set("cond_N", cond_N)
        d = qps[0].dims
        for k in range(N + 1):
            for f in ("x", "u", "sl", "su", "pi", "lam", "t"):
                if f == "pi" and k == N:
                    continue
                ref = np.stack([o.get(k, f) for o in oracles])
                if ref.shape[1] == 0:
                    continue
                if f in ("lam", "t"):
                    val = np.maximum(ref, 1e-3) * rng.uniform(0.5, 2.0, ref.shape)     # interior, not converged
                elif f == "x" and k == 0:
                    val = ref                                                             # x0 is data
                else:
                    val = ref + 1e-2 * rng.standard_normal(ref.shape)
                b.set(f, k, val)
        b.opts_set("warm_start", 3)
        assert b.solve() == 0, (fam, cond_N, b.info("status"))
        assert int(b.scalar("cond_N_active")) == cond_N
        assert np.all(b.info("iter") >= 1) and np.all(b.info("iter") < it_cold), (b.info("iter"), it_cold)
        assert b.res_compute().max() <= 2e-8
        for i, (qp, o) in enumerate(zip(qps, oracles)):
            compare_with_oracle(lambda k, f: b.get(f, k)[i], o, qp, 1e-5, fields=("x", "u", "sl", "su"))


@pytest.mark.parametrize("clib", TIERS, indirect=True)
def test_factorize_only_contract(clib, monkeypatch):
    """ocp_nlp_common_setup_qp_matrices_and_factorize (ocp_nlp_common.c:3914-3985): initialize_next_xcond_qp_from_qp_out,
    warm_start 3, iter_max 0, update_fact_exit 1, t0_min / lam0_min read (opts_get), raised and restored; the QP call
    returns SUCCESS or MAXITER (both tolerated, :3974), the factorisation at the handed-over point is in place and the
    sensitivity slots answer from it -- equal to the dense linearised-KKT solve at that (clipped) point"""
    from acados_amd import AcadosOcpQpOptions, AcadosOcpQpSolver
    from dense_ref import sens_dense
    monkeypatch.setenv("ACADOS_AMD_WPI", "1")
    qp = load_qp("qp_test/last_qp_nonuniform_pendulum.json")
    opts = AcadosOcpQpOptions()
    opts.tol_stat = opts.tol_eq = opts.tol_ineq = opts.tol_comp = 1e-8
    s = AcadosOcpQpSolver(qp, opts, _clib=clib)
    assert s.solve() == 0
    sol = {(k, f): s.get(k, f, unique_duals=False) for k in range(qp.N + 1) for f in ("x", "u", "lam", "t")}
    L = clib
    L.ocp_qp_xcond_solver_opts_set.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p]
    cfg = (C.c_void_p * 24).from_address(s.c_config.value)
    opts_get = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p)(cfg[9])       # slot 10: opts_get
    t0_bkp, l0_bkp = C.c_double(-1), C.c_double(-1)
    opts_get(s.c_config, s.c_opts, b"t0_min", C.byref(t0_bkp))
    opts_get(s.c_config, s.c_opts, b"lam0_min", C.byref(l0_bkp))
    assert t0_bkp.value == 1e-16 and l0_bkp.value == 1e-16
    flag = C.c_bool(True)
    L.ocp_qp_xcond_solver_opts_set(s.c_config, s.c_opts, b"initialize_next_xcond_qp_from_qp_out", C.byref(flag))
    s.opts_set("warm_start", 3)
    s.opts_set("iter_max", 0)
    s.opts_set("update_fact_exit", 1)
    clip = 1e-4
    s.opts_set("t0_min", clip)
    s.opts_set("lam0_min", clip)
    status = s.solve()
    assert status in (0, 2) and s.get_stats("iter") == 0
    # no iteration was taken: the multipliers are the handed-over ones, clipped from below at t0_min / lam0_min
    for k in range(qp.N + 1):
        lam, t = s.get(k, "lam", unique_duals=False), s.get(k, "t", unique_duals=False)
        act = np.concatenate([qp.lbu_mask[k], qp.lbx_mask[k], qp.lg_mask[k], qp.ubu_mask[k], qp.ubx_mask[k], qp.ug_mask[k],
                              qp.lls_mask[k], qp.lus_mask[k]]) != 0
        nbu, nb, nbg = int(qp.dims.nbu[k]), int(qp.dims.nb[k]), int(qp.dims.nb[k] + qp.dims.ng[k])
        for e in qp.idxe[k]:
            act[int(e)] = act[nbg + int(e)] = False
        assert np.allclose(lam[act], np.maximum(sol[(k, "lam")][act], clip), rtol=1e-12, atol=0)
        assert np.allclose(t[act], np.maximum(sol[(k, "t")][act], clip), rtol=1e-12, atol=0)
    # sensitivities from the factorisation at that point
    e = np.zeros(len(qp.lbx[0])); e[1] = 1.0
    se = s.eval_solution_sens({("lbx", 0): e, ("ubx", 0): e})
    get = lambda k, f: s.get(k, f, unique_duals=False) if not (f == "pi" and k == qp.N) else np.zeros(0)
    ref = sens_dense(qp, get, {("lbx", 0): e, ("ubx", 0): e})
    for k in range(qp.N + 1):
        assert np.allclose(se["x"][k], ref(k, "x"), rtol=1e-7, atol=1e-9), k
        if k < qp.N:
            assert np.allclose(se["u"][k], ref(k, "u"), rtol=1e-7, atol=1e-9), k
    # restore (ocp_nlp_common.c:3962-3966) and solve again
    s.opts_set("warm_start", 0)
    s.opts_set("iter_max", 50)
    s.opts_set("t0_min", t0_bkp.value)
    s.opts_set("lam0_min", l0_bkp.value)
    assert s.solve() == 0
    for k in range(qp.N + 1):
        assert np.allclose(s.get(k, "x"), sol[(k, "x")], atol=1e-7)


@pytest.mark.parametrize("clib", TIERS, indirect=True)
def test_full_condensing(clib, monkeypatch, capfd):
    """FULL_CONDENSING_GPU_IPM (f4 tail; the reference's module is ocp_qp_full_condensing.c:468-556): every stage in ONE
    block through the device condensing kernels where nx + N nu <= 64, against the full-space oracle -- box class and
    a class whose state bounds / general rows become rows of the single condensed stage; beyond one condensed stage: the dense path"""
    from acados_amd import AcadosOcpQpCondensing, AcadosOcpQpOptions, AcadosOcpQpSolver
    from acados_amd.generators import mass_spring_qp
    for qp in (load_qp("qp_test/last_qp_nonuniform_pendulum.json"), mass_spring_qp(N=4), _soft_qp(0, 2)):
        o = OracleQp(qp)
        assert o.solve(default_opts(tol_stat=1e-8)) == 0
        opts = AcadosOcpQpOptions()
        opts.qp_solver = "FULL_CONDENSING_GPU_IPM"
        opts.tol_stat = opts.tol_eq = opts.tol_ineq = opts.tol_comp = 1e-8
        s = AcadosOcpQpSolver(qp, opts, _clib=clib)
        assert s.solve() == 0
        assert s.inf_norm_residuals().max() <= 2e-8
        compare_with_oracle(lambda k, f: s.get(k, f), o, qp, 1e-5, fields=("x", "u", "sl", "su", "pi"))
        # the condensed QP has ONE stage with all inputs (+ the terminal stage)
        c = AcadosOcpQpCondensing(qp, 1, full=True, _clib=clib)
        xd = c.xcond_dims()
        assert c.cond_N == 1 and len(xd["nu"]) == 2 and xd["nu"][0] >= int(np.sum(qp.dims.nu)) and xd["nu"][1] == 0
    # beyond one condensed STAGE (64 variables / 128 sides): the dense path (dense_kernels.hpp) -- every state but x0 condensed into ONE
    # dense problem inside the solve, dense Cholesky; mass-spring N = 19: 8 + 19 * 3 = 65 > 64 condensed columns (68 dense ones with the padded terminal inputs), 19 * 8 state bounds as dense rows
    qp = mass_spring_qp(N=19)
    o = OracleQp(qp)
    assert o.solve(default_opts(tol_stat=1e-8)) == 0
    opts = AcadosOcpQpOptions()
    opts.qp_solver = "FULL_CONDENSING_GPU_IPM"
    opts.tol_stat = opts.tol_eq = opts.tol_ineq = opts.tol_comp = 1e-8
    s = AcadosOcpQpSolver(qp, opts, _clib=clib)
    assert s.solve() == 0
    assert s.inf_norm_residuals().max() <= 1e-8 * (1 + 1e-3) + 1e-12
    compare_with_oracle(lambda k, f: s.get(k, f), o, qp, 1e-5, fields=("x", "u", "pi"))
    assert "exceeds the 64 variables" not in capfd.readouterr().out
    # slacks are not carried by the dense path: a loud failure status, not a wrong answer
    s2 = AcadosOcpQpSolver(_soft_qp(0, 20), opts, _clib=clib)     # chain class: 24 + 20 * 3 = 84 columns, soft rows
    assert s2.solve() != 0


@pytest.mark.parametrize("clib", TIERS, indirect=True)
def test_condensing_slots_composed_like_xcond_solve(clib, monkeypatch):
    """ocp_qp_xcond_solve restated on the SLOTS (ocp_qp_xcond_solver.c:529-587): xcond->condensing(qp_in, xcond_qp_in) ->
    [condense_qp_out] -> qp_solver->evaluate(xcond_qp_in, xcond_qp_out) -> xcond->expansion(xcond_qp_out, qp_out), every
    call through the function pointers of the two sub-vtables, with the module memory answering "xcond_qp_in" /
    "xcond_qp_out" / "qp_out_info": the device condensing is reachable exactly where acados would call HPIPM's.  Also
    condense_lhs + condense_rhs (RTI) through the same slots."""
    from acados_amd.generators import mass_spring_qp
    from acados_amd.ocp_qp_condensing import _bind
    from acados_amd.ocp_qp_solver import AcadosOcpQpSolver
    from acados_amd import AcadosOcpQpOptions
    L = _bind(clib)
    vp, ci, cp = C.c_void_p, C.c_int, C.c_char_p
    qp = mass_spring_qp(N=15)
    o = OracleQp(qp)
    assert o.solve(default_opts(tol_stat=1e-8)) == 0
    N, N2 = qp.N, 5
    # the acados-shaped containers of the original QP (filled through the Python driver of the outer level)
    opts = AcadosOcpQpOptions()
    opts.tol_stat = opts.tol_eq = opts.tol_ineq = opts.tol_comp = 1e-8
    drv = AcadosOcpQpSolver(qp, opts, _clib=clib)
    qp_in, qp_out = drv.c_in, drv.c_out
    # sub-vtables of a config
    L.ocp_qp_xcond_solver_config_create_from_name.restype = vp
    cfg = L.ocp_qp_xcond_solver_config_create_from_name(b"PARTIAL_CONDENSING_GPU_IPM")
    outer = (vp * 24).from_address(cfg)
    inner, xc = (vp * 17).from_address(outer[22]), (vp * 20).from_address(outer[23])
    F = lambda res, *args: C.CFUNCTYPE(res, *args)
    sz = C.c_size_t
    x_dims_size, x_dims_assign = F(sz, vp, ci)(xc[0]), F(vp, vp, ci, vp)(xc[1])
    x_dims_set, x_dims_get = F(None, vp, vp, ci, cp, C.POINTER(ci))(xc[2]), F(None, vp, vp, cp, vp)(xc[3])
    x_opts_size, x_opts_assign, x_opts_init = F(sz, vp)(xc[4]), F(vp, vp, vp)(xc[5]), F(None, vp, vp)(xc[6])
    x_opts_update, x_opts_set = F(None, vp, vp)(xc[7]), F(None, vp, cp, vp)(xc[8])
    x_mem_size, x_mem_assign, x_mem_get = F(sz, vp, vp)(xc[9]), F(vp, vp, vp, vp)(xc[10]), F(None, vp, vp, cp, vp)(xc[11])
    condensing, condense_rhs = F(ci, vp, vp, vp, vp, vp)(xc[13]), F(ci, vp, vp, vp, vp, vp)(xc[14])
    condense_lhs = F(ci, vp, vp, vp, vp, vp)(xc[16])
    condense_qp_out = F(ci, vp, vp, vp, vp, vp, vp, vp)(xc[17])
    expansion = F(ci, vp, vp, vp, vp, vp)(xc[18])
    q_opts_size, q_opts_assign, q_opts_init = F(sz, vp, vp)(inner[1]), F(vp, vp, vp, vp)(inner[2]), F(None, vp, vp, vp)(inner[3])
    q_opts_set = F(None, vp, vp, cp, vp)(inner[5])
    q_mem_size, q_mem_assign = F(sz, vp, vp, vp)(inner[7]), F(vp, vp, vp, vp, vp)(inner[8])
    q_evaluate, q_terminate = F(ci, vp, vp, vp, vp, vp, vp)(inner[11]), F(None, vp, vp, vp)(inner[16])
    keep = []

    def block(n):
        buf = (C.c_char * int(n + 16))()
        keep.append(buf)
        return C.cast(buf, vp)

    xcfg, qcfg = vp(outer[23]), vp(outer[22])
    xdims = vp(x_dims_assign(xcfg, N, block(x_dims_size(xcfg, N))))
    for k in range(N + 1):
        for name in ("nx", "nu", "nbx", "nbu", "ng", "ns", "nbxe"):
            v = ci(int(getattr(qp.dims, name)[k]))
            x_dims_set(xcfg, xdims, k, name.encode(), C.byref(v))
    xopts = vp(x_opts_assign(xdims, block(x_opts_size(xdims))))
    x_opts_init(xdims, xopts)
    v = ci(N2)
    x_opts_set(xopts, b"N", C.byref(v))
    x_opts_update(xdims, xopts)
    xmem = vp(x_mem_assign(xdims, xopts, block(x_mem_size(xdims, xopts))))
    xcond_dims, xcond_in, xcond_out, info = vp(), vp(), vp(), vp()
    x_dims_get(xcfg, xdims, b"xcond_dims", C.byref(xcond_dims))
    x_mem_get(xcfg, xmem, b"xcond_qp_in", C.byref(xcond_in))
    x_mem_get(xcfg, xmem, b"xcond_qp_out", C.byref(xcond_out))
    x_mem_get(xcfg, xmem, b"qp_out_info", C.byref(info))
    qopts = vp(q_opts_assign(qcfg, xcond_dims, block(q_opts_size(qcfg, xcond_dims))))
    q_opts_init(qcfg, xcond_dims, qopts)
    for f in (b"tol_stat", b"tol_eq", b"tol_ineq", b"tol_comp"):
        d = C.c_double(1e-8)
        q_opts_set(qcfg, qopts, f, C.byref(d))
    qmem = vp(q_mem_assign(qcfg, xcond_dims, qopts, block(q_mem_size(qcfg, xcond_dims, qopts))))

    def check():
        get = lambda k, f: drv.get(k, f, unique_duals=False) if not (f == "pi" and k == N) else np.zeros(0)
        compare_with_oracle(get, o, qp, 2e-6, fields=("x", "u", "pi", "lam", "t"))
        assert drv.inf_norm_residuals().max() <= 2e-8

    # ---- ocp_qp_xcond_solve, line by line ----
    assert condensing(qp_in, xcond_in, xopts, xmem, None) == 0
    assert q_evaluate(qcfg, xcond_in, xcond_out, qopts, qmem, None) == 0
    assert expansion(xcond_out, qp_out, xopts, xmem, None) == 0
    it_cold = C.cast(info.value + 32, C.POINTER(ci))[0]      # qp_info.num_iter of the condensed solve ("qp_out_info")
    assert it_cold >= 3
    check()
    # ---- with initialize_next_xcond_qp_from_qp_out: the full-space solution condensed into the guess, hot start ----
    assert condensing(qp_in, xcond_in, xopts, xmem, None) == 0
    assert condense_qp_out(qp_in, xcond_in, qp_out, xcond_out, xopts, xmem, None) == 0
    w = ci(3)
    q_opts_set(qcfg, qopts, b"warm_start", C.byref(w))
    assert q_evaluate(qcfg, xcond_in, xcond_out, qopts, qmem, None) == 0
    assert expansion(xcond_out, qp_out, xopts, xmem, None) == 0
    assert C.cast(info.value + 32, C.POINTER(ci))[0] < it_cold
    check()
    # ---- RTI: condense_lhs, then condense_rhs + solve + expansion (ocp_qp_xcond_solver.c:591-669) ----
    w = ci(0)
    q_opts_set(qcfg, qopts, b"warm_start", C.byref(w))
    assert condense_lhs(qp_in, xcond_in, xopts, xmem, None) == 0
    assert condense_rhs(qp_in, xcond_in, xopts, xmem, None) == 0
    assert q_evaluate(qcfg, xcond_in, xcond_out, qopts, qmem, None) == 0
    assert expansion(xcond_out, qp_out, xopts, xmem, None) == 0
    check()
    # ---- sensitivities through the condensing, line by line (ocp_qp_xcond_solver.c:680-697): condense_rhs_seed ->
    # qp_solver->eval_forw_sens on the condensed QP -> expand_sol_seed; against ONE dense solve of the linearised KKT system
    # of the ORIGINAL QP at the oracle's solution (tests/dense_ref.py) ----
    from dense_ref import sens_dense
    condense_rhs_seed, expand_sol_seed = F(ci, vp, vp, vp, vp, vp, vp)(xc[15]), F(ci, vp, vp, vp, vp, vp)(xc[19])
    q_eval_forw_sens = F(None, vp, vp, vp, vp, vp, vp, vp)(inner[14])
    xcond_seed = vp()
    x_mem_get(xcfg, xmem, b"xcond_seed", C.byref(xcond_seed))

    class _Dims(C.Structure):
        _fields_ = [("N", ci)] + [(n_, C.POINTER(ci)) for n_ in ("nx", "nu", "nb", "nbx", "nbu", "ng", "ns", "nbxe", "nbue", "nge")]

    class _Seed(C.Structure):
        _fields_ = [("dim", C.POINTER(_Dims))] + [(n_, C.POINTER(C.POINTER(C.c_double))) for n_ in ("seed_g", "seed_b", "seed_d", "seed_m")]

    class _In(C.Structure):
        _fields_ = [("dim", C.POINTER(_Dims))]

    L.ocp_qp_seed_create.restype, L.ocp_qp_seed_create.argtypes = vp, [vp]
    L.ocp_qp_out_create.restype, L.ocp_qp_out_create.argtypes = vp, [vp]
    dim_ptr = C.cast(qp_in, C.POINTER(_In)).contents.dim
    seed = vp(L.ocp_qp_seed_create(dim_ptr))
    sens_out = vp(L.ocp_qp_out_create(dim_ptr))
    sd = C.cast(seed, C.POINTER(_Seed)).contents
    rng = np.random.default_rng(5)
    d = qp.dims
    seeds = {}
    for k in range(N + 1):
        nu, nx, nbu, nbx = int(d.nu[k]), int(d.nx[k]), int(d.nbu[k]), int(d.nbx[k])
        seeds[("r", k)], seeds[("q", k)] = rng.standard_normal(nu), rng.standard_normal(nx)
        if k < N:
            seeds[("b", k)] = rng.standard_normal(int(d.nx[k + 1]))
        seeds[("lbu", k)], seeds[("ubu", k)] = 0.1 * rng.standard_normal(nbu), 0.1 * rng.standard_normal(nbu)
        if k == 0:
            e = rng.standard_normal(nbx)
            seeds[("lbx", 0)], seeds[("ubx", 0)] = e, e          # x0: both sides, ocp_nlp_common.c:4057-4066
        else:
            seeds[("lbx", k)], seeds[("ubx", k)] = 0.1 * rng.standard_normal(nbx), 0.1 * rng.standard_normal(nbx)
        nb = nbu + nbx
        g = np.concatenate([seeds[("r", k)], seeds[("q", k)]])
        for e_, v_ in enumerate(g):
            sd.seed_g[k][e_] = v_
        if k < N:
            for e_, v_ in enumerate(seeds[("b", k)]):
                sd.seed_b[k][e_] = v_
        dd = np.concatenate([seeds[("lbu", k)], seeds[("lbx", k)], seeds[("ubu", k)], seeds[("ubx", k)]])   # natural sign (this library's containers)
        assert dd.size == 2 * nb
        for e_, v_ in enumerate(dd):
            sd.seed_d[k][e_] = v_
    assert condensing(qp_in, xcond_in, xopts, xmem, None) == 0
    assert q_evaluate(qcfg, xcond_in, xcond_out, qopts, qmem, None) == 0      # factorisation at the condensed solution
    assert expansion(xcond_out, qp_out, xopts, xmem, None) == 0
    assert condense_rhs_seed(qp_in, seed, xcond_seed, xopts, xmem, None) == 0
    q_eval_forw_sens(qcfg, xcond_in, xcond_seed, xcond_out, qopts, qmem, None)
    assert expand_sol_seed(xcond_out, sens_out, xopts, xmem, None) == 0
    ref = sens_dense(qp, o.get, seeds)
    scale = max(1.0, max(np.max(np.abs(ref(k, f))) for k in range(N + 1) for f in ("x", "u") if ref(k, f).size))
    worst = 0.0
    for k in range(N + 1):
        for f in ("x", "u", "pi"):
            if f == "pi" and k == N:
                continue
            got = drv.get(k, f, unique_duals=False, _c_out=sens_out)
            err = np.max(np.abs(got - ref(k, f))) / scale if got.size else 0.0
            assert err <= 1e-5, (k, f, err, got, ref(k, f))
            worst = max(worst, err)
        got, want = drv.get(k, "lam", unique_duals=False, _c_out=sens_out), ref(k, "lam")
        sel = np.array([(k, e_) in ref.active for e_ in range(want.size)], dtype=bool)
        if sel.any():
            assert np.max(np.abs(got[sel] - want[sel])) <= 1e-3 * max(scale, np.max(np.abs(want[sel]))), (k, got[sel], want[sel])
    print("sensitivities through condense_rhs_seed / expand_sol_seed vs dense KKT solve:", worst)
    # the evaluate after it still solves the ORIGINAL data (the seed pass restored the vectors)
    assert condensing(qp_in, xcond_in, xopts, xmem, None) == 0
    assert q_evaluate(qcfg, xcond_in, xcond_out, qopts, qmem, None) == 0
    assert expansion(xcond_out, qp_out, xopts, xmem, None) == 0
    check()
    L.ocp_qp_seed_free(seed)
    L.ocp_qp_out_free(sens_out)
    q_terminate(qcfg, qmem, None)
    L.ocp_qp_gpu_pcond_memory_release.argtypes = [vp]
    L.ocp_qp_gpu_pcond_memory_release(xmem)
    L.ocp_qp_xcond_solver_config_free(vp(cfg))
