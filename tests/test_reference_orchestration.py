"""f1, the stretch of the round-3 review: the REFERENCE'S OWN ocp_qp_xcond_solver.c (the 22-slot solver `ocp_nlp` holds),
ocp_qp_common.c (containers, ocp_qp_compute_t, ocp_qp_res_compute), utils/mem.c and utils/timing.c, compiled UNMODIFIED from
/root/reference, drive this repository's plugin (integration/ocp_qp_gpu_ipm.c as config->qp_solver): dims and opts routing
("cond_" strings to the condensing module, the rest to the inner solver), memory carving in the reference's one block,
ocp_qp_xcond_solve (:529-587), the RTI pair condense_lhs / condense_rhs_and_solve (:591-669), memory_get, qp_info.  HPIPM and
BLASFEO -- empty submodules in the reference tree -- are the stand-ins of tests/mock_hpipm; the condensing module is
tests/mock_acados/copy_xcond.c (N2 = N, the reference's default).  The solution is compared with the oracle, the plugin's t
with the reference's own ocp_qp_compute_t, and the reference's residual entry must report <= 1e-8.

CPU tier: linked against the host-simulation library, built here.  GPU tier: /root/reference does not exist on the GPU box --
the binary built in the build container against the product library (integration/Makefile) is run."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_qp
from oracle.oracle import OracleQp, default_opts
from test_mock_acados import MOCK, REFERENCE, _write_qp

TIERS = [pytest.param("hostsim", id="hostsim"), pytest.param("gpu", id="gpu", marks=pytest.mark.gpu)]
PREBUILT = os.path.join(ROOT, "integration", "_ref_build", "ref_xcond_driver")
REF_SOURCES = ["acados/ocp_qp/ocp_qp_xcond_solver.c", "acados/ocp_qp/ocp_qp_common.c", "acados/utils/mem.c", "acados/utils/timing.c"]


@pytest.fixture
def clib(request):
    return request.getfixturevalue("hostsim_lib" if request.param == "hostsim" else "gpu_lib")


def _build(libpath, tmp_path):
    libdir, libname = os.path.dirname(libpath), os.path.basename(libpath)
    if not os.path.isdir(os.path.join(REFERENCE, "acados", "ocp_qp")):
        if libname == "libacados_amd_qp.so" and os.path.exists(PREBUILT):
            return PREBUILT
        pytest.skip("no reference tree and no prebuilt driver for this library")
    exe = str(tmp_path / "ref_xcond_driver")
    cmd = ["gcc", "-std=gnu11", "-O2", "-fopenmp", "-Wall", "-Wno-unused-parameter", "-I", REFERENCE, "-I", os.path.join(ROOT, "tests", "mock_hpipm"),
           "-I", os.path.join(ROOT, "include"), "-I", MOCK,
           os.path.join(MOCK, "ref_xcond_driver.c"), os.path.join(MOCK, "copy_xcond.c"), os.path.join(ROOT, "integration", "ocp_qp_gpu_ipm.c"),
           os.path.join(ROOT, "tests", "mock_hpipm", "mock_hpipm.c")] + [os.path.join(REFERENCE, f) for f in REF_SOURCES] + \
          ["-o", exe, "-L", libdir, "-l:" + libname, "-Wl,-rpath," + libdir, "-lm", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.parametrize("clib", TIERS, indirect=True)
@pytest.mark.parametrize("qp_name", ["mass_spring", "casadi_qp_tests/pendulum_slack.json", "casadi_qp_tests/pend_idxs_rev_min_qp0.json",
                                     "qp_test/last_qp_one_sided_test.json"])
def test_reference_xcond_solver_drives_the_plugin(clib, tmp_path, qp_name):
    from acados_amd.generators import mass_spring_qp
    qp = mass_spring_qp(N=15) if qp_name == "mass_spring" else load_qp(qp_name)
    exe = _build(clib._name, tmp_path)
    qp_file, sol_file = str(tmp_path / "qp.txt"), str(tmp_path / "sol.txt")
    _write_qp(qp, qp_file)
    r = subprocess.run([exe, qp_file, sol_file], capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout, r.stderr)
    lines = open(sol_file).read().splitlines()
    h = lines[0].split()
    head = {h[i]: int(h[i + 1]) for i in range(0, len(h), 2)}
    c = lines[1].split()
    t_diff, rti_diff, res = float(c[2]), float(c[4]), [float(v) for v in c[6:10]]
    sol = {}
    for ln in lines[2:]:
        p = ln.split()
        sol[(p[0], int(p[1]))] = np.array([float(x) for x in p[3:]])
    # status / iterations as the reference's layers report them: evaluate's return, memory_get, qp_info copied from the module's info
    assert head["status"] == 0 and head["status_mem"] == 0 and head["iter"] == head["iter_info"] >= 1 and head["t_computed"] == 1
    assert head["rti_status"] == 0 and rti_diff <= 1e-9
    # the reference's ocp_qp_compute_t (ocp_qp_common.c:874-921) on the plugin's primal solution reproduces the plugin's t
    assert t_diff <= 1e-12, t_diff
    # the reference's residual entry on (qp_in, qp_out) of the plugin
    assert max(res) <= 1e-8 * (1 + 1e-3) + 1e-13, res
    o = OracleQp(qp)
    assert o.solve(default_opts(tol_stat=1e-8)) == 0 and abs(head["iter"] - o.iter) <= 1
    for k in range(qp.N + 1):
        ref = np.concatenate([o.get(k, "u"), o.get(k, "x"), o.get(k, "sl"), o.get(k, "su")])
        assert np.allclose(sol[("ux", k)], ref, rtol=1e-7, atol=1e-8), (k, sol[("ux", k)], ref)
        if k < qp.N:
            assert np.allclose(sol[("pi", k)], o.get(k, "pi"), rtol=1e-6, atol=1e-7)
        assert np.allclose(sol[("lam", k)], o.get(k, "lam"), rtol=1e-5, atol=1e-6)
        assert np.allclose(sol[("t", k)], o.get(k, "t"), rtol=1e-5, atol=1e-6)
