"""f1, the stretch of the round-3 review: the REFERENCE'S OWN ocp_qp_xcond_solver.c (the 22-slot solver `ocp_nlp` holds),
ocp_qp_common.c (containers, ocp_qp_compute_t, ocp_qp_res_compute), utils/mem.c and utils/timing.c, compiled UNMODIFIED from
/root/reference, drive BOTH slots of this repository's plugin -- integration/ocp_qp_gpu_ipm.c as config->qp_solver and
integration/ocp_qp_gpu_pcond.c, the device condensing behind acados' own types, as config->xcond: dims and opts routing
("cond_" strings to the condensing module, the rest to the inner solver), memory carving in the reference's one block,
ocp_qp_xcond_solve (:529-587), the RTI pair condense_lhs / condense_rhs_and_solve (:591-669), memory_get, qp_info.  HPIPM and
BLASFEO -- empty submodules in the reference tree -- are the stand-ins of tests/mock_hpipm.  Settings: the reference's unit test
(test/ocp_qp/test_qpsolvers.cpp:117-268: mass-spring N = 15, N2 in {15, 5, 3}), user block sizes with a non-zero last entry
(pcond_getters_test.py:200), the slack fixtures; the copy stand-in tests/mock_acados/copy_xcond.c (N2 = N) stays as a cross-check.
The solution is compared with the oracle, the plugin's t with the reference's own ocp_qp_compute_t, the reference's residual
entry must report <= 1e-8, forward sensitivities through the reference's eval_forw_sens against a dense KKT solve, and a
1,024-capsule C3-shaped batch (N = 50 -> N2 = 10) goes through ocp_qp_gpu_xcond_solver_acados_evaluate_batch (fused on the device).

CPU tier: linked against the host-simulation library, built here.  GPU tier: /root/reference does not exist on the GPU box --
the binary built in the build container against the product library (integration/Makefile) is run."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_qp
from oracle.oracle import OracleQp, default_opts
from test_mock_acados import MOCK, REFERENCE, _check_sens_vs_dense, _perturbed, _seeds, _split_bin, _write_qp

TIERS = [pytest.param("hostsim", id="hostsim"), pytest.param("gpu", id="gpu", marks=pytest.mark.gpu)]
PREBUILT = os.path.join(ROOT, "integration", "_ref_build", "ref_xcond_driver")
REF_SOURCES = ["acados/ocp_qp/ocp_qp_xcond_solver.c", "acados/ocp_qp/ocp_qp_common.c", "acados/utils/mem.c", "acados/utils/timing.c"]


@pytest.fixture
def clib(request):
    return request.getfixturevalue("hostsim_lib" if request.param == "hostsim" else "gpu_lib")


_BUILT = {}


def _build(libpath, tmp_path):
    if libpath not in _BUILT:
        _BUILT[libpath] = _build_once(libpath, tmp_path)
    return _BUILT[libpath]


def _build_once(libpath, tmp_path):
    libdir, libname = os.path.dirname(libpath), os.path.basename(libpath)
    if not os.path.isdir(os.path.join(REFERENCE, "acados", "ocp_qp")):
        if libname == "libacados_amd_qp.so" and os.path.exists(PREBUILT):
            return PREBUILT
        pytest.skip("no reference tree and no prebuilt driver for this library")
    import tempfile
    exe = os.path.join(tempfile.mkdtemp(prefix="ref_xcond_"), "ref_xcond_driver")
    cmd = ["gcc", "-std=gnu11", "-O2", "-fopenmp", "-Wall", "-Wno-unused-parameter", "-I", REFERENCE, "-I", os.path.join(ROOT, "tests", "mock_hpipm"),
           "-I", os.path.join(ROOT, "include"), "-I", MOCK,
           "-I", os.path.join(ROOT, "integration"),
           os.path.join(MOCK, "ref_xcond_driver.c"), os.path.join(MOCK, "copy_xcond.c"), os.path.join(ROOT, "integration", "ocp_qp_gpu_ipm.c"),
           os.path.join(ROOT, "integration", "ocp_qp_gpu_pcond.c"),
           os.path.join(ROOT, "tests", "mock_hpipm", "mock_hpipm.c")] + [os.path.join(REFERENCE, f) for f in REF_SOURCES] + \
          ["-o", exe, "-L", libdir, "-l:" + libname, "-Wl,-rpath," + libdir, "-lm", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def _run(exe, qp, tmp_path, flags):
    qp_file, sol_file = str(tmp_path / "qp.txt"), str(tmp_path / "sol.txt")
    _write_qp(qp, qp_file)
    r = subprocess.run([exe, qp_file, sol_file] + flags, capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout, r.stderr)
    lines = open(sol_file).read().splitlines()
    h = lines[0].split()
    head = {h[i]: int(h[i + 1]) for i in range(0, len(h), 2)}
    c = lines[1].split()
    checks = {"t_diff": float(c[2]), "rti_diff": float(c[4]), "res": [float(v) for v in c[6:10]]}
    sol, sens, cur = {}, {}, None
    cur = sol
    for ln in lines[2:]:
        p = ln.split()
        if p[0] == "sens":
            cur = sens
            continue
        cur[(p[0], int(p[1]))] = np.array([float(x) for x in p[2:]])
    return head, checks, sol, sens, r.stderr


def _check_solution(qp, head, checks, sol):
    # status / iterations as the reference's layers report them: evaluate's return, memory_get, qp_info copied from the module's info
    assert head["status"] == 0 and head["status_mem"] == 0 and head["iter"] == head["iter_info"] >= 1 and head["t_computed"] == 1
    # the RTI pair (only the vectors of qp_in differ between the two calls) reproduces the plain solve
    assert head["rti_status"] == 0 and checks["rti_diff"] <= 1e-9
    # the reference's ocp_qp_compute_t (ocp_qp_common.c:874-921) on the plugin's primal solution reproduces the plugin's t
    assert checks["t_diff"] <= 1e-12, checks
    # the reference's residual entry on (qp_in, qp_out) of the plugin
    assert max(checks["res"]) <= 1e-8 * (1 + 1e-3) + 1e-13, checks
    o = OracleQp(qp)
    assert o.solve(default_opts(tol_stat=1e-8)) == 0 and abs(head["iter"] - o.iter) <= 1
    for k in range(qp.N + 1):
        ref = np.concatenate([o.get(k, "u"), o.get(k, "x"), o.get(k, "sl"), o.get(k, "su")])
        assert np.allclose(sol[("ux", k)], ref, rtol=1e-7, atol=1e-8), (k, sol[("ux", k)], ref)
        if k < qp.N:
            assert np.allclose(sol[("pi", k)], o.get(k, "pi"), rtol=1e-6, atol=1e-7)
        assert np.allclose(sol[("lam", k)], o.get(k, "lam"), rtol=1e-5, atol=1e-6)
        assert np.allclose(sol[("t", k)], o.get(k, "t"), rtol=1e-5, atol=1e-6)
    return o


@pytest.mark.parametrize("clib", TIERS, indirect=True)
@pytest.mark.parametrize("xcond", ["gpu", "copy"])
@pytest.mark.parametrize("qp_name", ["mass_spring", "casadi_qp_tests/pendulum_slack.json", "casadi_qp_tests/pend_idxs_rev_min_qp0.json",
                                     "qp_test/last_qp_one_sided_test.json"])
def test_reference_xcond_solver_drives_the_plugin(clib, tmp_path, qp_name, xcond):
    """N2 = N (the reference's default): the device module hands the QP through, the copy stand-in copies it"""
    from acados_amd.generators import mass_spring_qp
    qp = mass_spring_qp(N=15) if qp_name == "mass_spring" else load_qp(qp_name)
    exe = _build(clib._name, tmp_path)
    head, checks, sol, _, _ = _run(exe, qp, tmp_path, ["--xcond", xcond])
    assert head["xcond_N"] == qp.N
    _check_solution(qp, head, checks, sol)


# (QP, N2, block sizes or None, condensed stages expected): the reference's unit-test settings N2 in {5, 3} on mass-spring N = 15
# (test/ocp_qp/test_qpsolvers.cpp:117-268), user blocks incl. a non-zero last one, the slack fixtures (general rows + shared slacks
# in the condensed stages), the one-sided golden QP
CONDENSED = [("mass_spring", 5, None, 5), ("mass_spring", 3, None, 3), ("mass_spring", 3, [5, 5, 3, 2], 4), ("mass_spring", 4, [3, 4, 4, 4, 0], 4),
             ("casadi_qp_tests/pendulum_slack.json", 4, None, 4), ("casadi_qp_tests/pend_idxs_rev_min_qp0.json", 3, None, 3),
             ("qp_test/last_qp_one_sided_test.json", 5, None, 5), ("qp_test/last_qp_nonuniform_pendulum.json", 3, None, 3)]


@pytest.mark.parametrize("clib", TIERS, indirect=True)
@pytest.mark.parametrize("qp_name,N2,blocks,n_stages", CONDENSED)
def test_reference_xcond_solver_drives_both_slots_condensed(clib, tmp_path, qp_name, N2, blocks, n_stages):
    """the reference's ocp_qp_xcond_solve / condense_lhs + condense_rhs_and_solve / warm start / eval_forw_sens around the DEVICE
    condensing module on acados' types (ocp_qp_gpu_pcond.c) and the QP solver (ocp_qp_gpu_ipm.c), N2 < N"""
    from acados_amd.generators import mass_spring_qp
    qp = mass_spring_qp(N=15) if qp_name == "mass_spring" else load_qp(qp_name)
    exe = _build(clib._name, tmp_path)
    flags = ["--cond-N", str(N2), "--sens"] + (["--block-size", ",".join(str(b) for b in blocks)] if blocks else [])
    head, checks, sol, sens, err = _run(exe, qp, tmp_path, flags)
    assert "solving the full-space QP" not in err, err
    # the condensed QP the reference's layers saw: N2 stages (one more when the last user block is not 0), inputs stacked per block
    assert head["xcond_N"] == n_stages and head["xcond_N"] < qp.N
    bs0 = blocks[0] if blocks else qp.N // N2 + (1 if qp.N % N2 else 0)
    assert head["xcond_nu0"] >= bs0 * int(qp.dims.nu[0])
    # x0 is eliminated before the device condenses, as HPIPM's d_ocp_qp_reduce_eq_dof does (ocp_qp_partial_condensing.c:542): stage 0 of
    # the condensed QP carries the states the equality-flagged bounds leave free, and no box rows on the ones they fix (round 6)
    assert head["xcond_nx0"] == int(qp.dims.nx[0]) - int(qp.dims.nbxe[0]), head
    _check_solution(qp, head, checks, sol)
    # warm start from the solution through the module's condense_qp_out: converges (to the same point, checked in the driver) in
    # no more iterations than the cold solve
    assert 0 <= head["warm_iter"] <= head["iter"], head
    # forward sensitivities through the reference's eval_forw_sens: condense_rhs_seed -> inner eval_forw_sens -> expand_sol_seed
    soft = int(np.sum(qp.dims.ns)) > 0
    _check_sens_vs_dense(qp, sol, sens, _seeds(qp, 0), 2e-4 if soft else 1e-6, 1e-2)


def _run_batch(exe, qp, tmp_path, n, flags, reps=1, default_dispatch=False):
    qp_file, out = str(tmp_path / "qp.txt"), str(tmp_path / "batch.bin")
    _write_qp(qp, qp_file)
    env = dict(os.environ, OMP_NUM_THREADS=str(min(16, os.cpu_count() or 1)))
    if default_dispatch:
        env.pop("ACADOS_AMD_WPI_BATCH_MAX", None)
    r = subprocess.run([exe, "batch", str(n), qp_file, out] + flags + [str(reps)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    lines = r.stdout.strip().splitlines()
    h = lines[0].split()
    info = {h[i]: float(h[i + 1]) for i in range(1, len(h) - 1, 2)}
    per = [tuple(int(x) for x in ln.split()) for ln in lines[1:1 + n]]
    return info, per, np.fromfile(out), r.stderr


@pytest.mark.parametrize("clib", TIERS, indirect=True)
def test_batch_through_the_reference_solver_objects_fused_on_device(clib, tmp_path):
    """9 capsules, each holding the reference's 22-slot solver around the two plugin slots; ONE call of
    ocp_qp_gpu_xcond_solver_acados_evaluate_batch: cond_N / cond_block_size are taken from the condensing module's options and the
    ORIGINAL QPs go to the device, where condensing, IPM and expansion run back to back; every capsule against the oracle, status /
    iter through the reference's memory_get, and the fused result equals the reference's per-capsule ocp_qp_xcond_solve"""
    from acados_amd.generators import mass_spring_qp
    qp = mass_spring_qp(N=15)
    exe = _build(clib._name, tmp_path)
    for flags, n_stages in ((["--cond-N", "5"], 5), (["--cond-N", "3", "--block-size", "5,5,3,2"], 4)):
        n = 9
        info, per, raw, err = _run_batch(exe, qp, tmp_path, n, flags)
        assert "solving the full-space QP" not in err, err
        assert info["status"] == 0 and info["orchestrated_status"] == 0 and info["xcond_N"] == n_stages and info["cond_N_active"] == n_stages
        assert info["fused_vs_orchestrated"] <= 1e-9
        assert info["res_max"] <= 1e-8 * (1 + 1e-3) + 1e-13          # the reference's residual entry, every capsule
        per_inst = raw.size // n
        assert per_inst * n == raw.size
        for i in range(n):
            qi = _perturbed(qp, i)
            sol, used = _split_bin(qi, raw[i * per_inst:])
            assert used == per_inst
            o = OracleQp(qi)
            assert o.solve(default_opts(tol_stat=1e-8)) == 0
            assert per[i][0] == i and per[i][1] == 0 and per[i][2] == per[i][3] and abs(per[i][2] - o.iter) <= 1 and per[i][4] == 1
            for k in range(qi.N + 1):
                ref = np.concatenate([o.get(k, "u"), o.get(k, "x")])
                assert np.allclose(sol[("ux", k)], ref, rtol=1e-7, atol=1e-8), (i, k)
                assert np.allclose(sol[("lam", k)], o.get(k, "lam"), rtol=1e-5, atol=1e-6)
                if k < qi.N:
                    assert np.allclose(sol[("pi", k)], o.get(k, "pi"), rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
def test_batch_1024_c3_shaped_through_the_reference_solver_objects(gpu_lib, tmp_path):
    """1,024 capsules of the C3 shape (N = 50, nx = 8, nu = 3, input box, x0; cond_N = 10) in acados structs, each with the
    reference's solver object; one fused batch call (km_pcond + kt_factor + k_pexpand on the device); all converged, 30 of them against
    the oracle, the fused result equals the reference's per-capsule orchestration, time per call reported"""
    from acados_amd.generators import lqr_instance_qp, random_lqr_batch
    N, n = 50, 1024
    data = random_lqr_batch(N=N, batch=1, seed=5)
    qp = lqr_instance_qp(data, 0, N)
    exe = _build(gpu_lib._name, tmp_path)
    info, per, raw, err = _run_batch(exe, qp, tmp_path, n, ["--cond-N", "10"], reps=5, default_dispatch=True)
    print("1,024 C3-shaped capsules, reference solver objects, fused batch call:", info)
    assert "solving the full-space QP" not in err, err
    assert info["status"] == 0 and all(st == 0 for _, st, _, _, _ in per) and info["orchestrated_status"] == 0
    assert info["xcond_N"] == 10 and info["xcond_nu0"] == 15 and info["cond_N_active"] == 10
    assert info["fused_vs_orchestrated"] <= 1e-8
    assert info["res_max"] <= 1e-8 * (1 + 1e-3) + 1e-13              # the reference's residual entry on all 1,024 capsules
    per_inst = raw.size // n
    assert per_inst * n == raw.size
    for i in list(range(0, n, 37)) + [n - 1]:
        qi = _perturbed(qp, i)
        sol, used = _split_bin(qi, raw[i * per_inst:])
        o = OracleQp(qi)
        assert o.solve(default_opts(tol_stat=1e-8)) == 0
        for k in range(N + 1):
            ref = np.concatenate([o.get(k, "u"), o.get(k, "x")])
            assert np.allclose(sol[("ux", k)], ref, rtol=1e-6, atol=1e-7), (i, k)
    assert info["ms_per_call"] <= 25.0, info
