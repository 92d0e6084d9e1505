"""The lock-step batched RTI loop RUN end to end (second half).

n capsules of a linear MPC problem (mass-spring, N stages, nx = 8, nu = 3, u-box, x0 as equality bounds).  Every capsule is an
`ocp_nlp` of the REFERENCE's own code -- interfaces/acados_c/ocp_nlp_interface.c, ocp_nlp_common.c and ocp_nlp_sqp_rti.c with
integration/acados.patch applied, ocp_nlp_cost_ls.c, ocp_nlp_dynamics_disc.c, ocp_nlp_constraints_bgh.c, ocp_nlp_qpscaling.c ... --
compiled from /root/reference against the HPIPM / BLASFEO stand-ins (tests/mock_hpipm: plain-loop linear algebra), with the QP
solver `PARTIAL_CONDENSING_GPU_IPM` created from the plan.  tests/mock_acados/lockstep_driver.c does what a generated solver does and
calls the two batch functions cut verbatim from the PATCHED template:

  * `_acados_batch_solve_gpu_qp`: option "batch_qp_phase" 1 (every capsule's RTI step up to its QP solve, on host threads), ONE call
    of ocp_qp_gpu_xcond_solver_acados_evaluate_batch for all n QPs (condensing + IPM + expansion on the device), phase 2;
  * `_acados_batch_solve`: the reference's per-capsule OpenMP loop (acados_solver.in.c:3222-3243) on TWINS of a subset of the capsules.

Five RTI steps in closed loop (x0 := x1 of the step).  Asserted: the option string reaches SQP_RTI (the old name `qp_batch_phase` was
routed to the QP solver and ended in exit(1)); lock-step == per-capsule loop on the twins; the first step of sampled capsules equals
the oracle's solution of the MPC QP written down independently in Python.

CPU tier: hostsim library, a handful of capsules.  GPU tier: the binary integration/Makefile prebuilt (1,024 capsules)."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from conftest import ROOT
from oracle.oracle import OracleQp, default_opts

REFERENCE = "/root/reference"
PREBUILT = os.path.join(ROOT, "integration", "_ref_build", "lockstep_driver")
TIERS = [pytest.param("hostsim", id="hostsim"), pytest.param("gpu", id="gpu", marks=pytest.mark.gpu)]
_BUILT = {}


@pytest.fixture
def clib(request):
    return request.getfixturevalue("hostsim_lib" if request.param == "hostsim" else "gpu_lib")


def _exe(libpath):
    if libpath in _BUILT:
        return _BUILT[libpath]
    if not os.path.isdir(os.path.join(REFERENCE, "acados", "ocp_nlp")):
        if os.path.basename(libpath) == "libacados_amd_qp.so" and os.path.exists(PREBUILT):
            _BUILT[libpath] = PREBUILT
            return PREBUILT
        pytest.skip("no reference tree and no prebuilt driver for this library")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from lockstep_build import build
    exe = os.path.join(tempfile.mkdtemp(prefix="lockstep_"), "lockstep_driver")
    _BUILT[libpath] = build(libpath, exe, REFERENCE)
    return exe


def _model(N, n, seed):
    from acados_amd.generators import mass_spring_system
    A, B = mass_spring_system(0.5, 8, 3)
    g = np.random.default_rng(seed)
    return dict(N=N, nx=8, nu=3, n=n, A=A, B=B, Q=np.ones(8), R=2.0 * np.ones(3), QN=np.ones(8), umax=0.5, x0=g.uniform(-2.5, 2.5, (n, 8)))


def _write(m, path):
    with open(path, "w") as f:
        f.write(f"{m['N']} {m['nx']} {m['nu']} {m['n']}\n")
        for a in (m["A"].ravel(order="F"), m["B"].ravel(order="F"), m["Q"], m["R"], m["QN"], [m["umax"]], m["x0"].ravel()):
            f.write(" ".join(repr(float(v)) for v in a) + "\n")


def _mpc_qp(m, x0):
    """the QP of one RTI step from a zero initial guess, written down independently: min 1/2 sum x'Qx + u'Ru, x+ = Ax + Bu, |u| <= umax"""
    from acados_amd import AcadosOcpQp
    N, nx, nu = m["N"], m["nx"], m["nu"]
    qp = AcadosOcpQp(N)
    for k in range(N + 1):
        last = k == N
        nuk = 0 if last else nu
        qp.set("Q", k, np.diag(m["QN"] if last else m["Q"])); qp.set("q", k, np.zeros(nx))
        qp.set("R", k, np.diag(m["R"])[:nuk, :nuk]); qp.set("r", k, np.zeros(nuk)); qp.set("S", k, np.zeros((nuk, nx)))
        if not last:
            qp.set("A", k, m["A"]); qp.set("B", k, m["B"]); qp.set("b", k, np.zeros(nx))
        qp.set("lbu", k, -m["umax"] * np.ones(nuk)); qp.set("ubu", k, m["umax"] * np.ones(nuk))
        if k == 0:
            qp.set("lbx", k, x0); qp.set("ubx", k, x0)
            qp.set("idxb", k, np.concatenate([np.arange(nuk), nuk + np.arange(nx)])); qp.set("idxe", k, nuk + np.arange(nx))
        else:
            qp.set("lbx", k, np.zeros(0)); qp.set("ubx", k, np.zeros(0)); qp.set("idxb", k, np.arange(nuk))
    qp.make_consistent()
    return qp


def _run(exe, m, tmp_path, twins, iters, cond_N, threads=4, split=0):
    mf, of = str(tmp_path / "model.txt"), str(tmp_path / "out.txt")
    _write(m, mf)
    r = subprocess.run([exe, mf, of, str(twins), str(iters), str(cond_N), str(threads), str(split)], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    rows = {"lock": {}, "twin": {}}
    times = None
    for ln in open(of).read().splitlines():
        p = ln.split()
        if p[0] == "time":
            times = (float(p[2]), float(p[4]), int(p[6]))
            continue
        iu, ix = p.index("u0"), p.index("x1")
        rows[p[0]][(int(p[1]), int(p[2]))] = dict(status=int(p[4]), qp_status=int(p[6]), qp_iter=int(p[8]),
                                                  u0=np.array([float(v) for v in p[iu + 1:ix]]), x1=np.array([float(v) for v in p[ix + 1:]]))
    return rows, times, r.stdout + r.stderr


@pytest.mark.parametrize("clib", TIERS, indirect=True)
@pytest.mark.parametrize("split", [0, 1], ids=["one-call", "preparation+feedback"])
@pytest.mark.parametrize("cond_N", [5, 0], ids=["condensed", "full-space"])
def test_lock_step_rti_loop_end_to_end(clib, tmp_path, cond_N, split, request):
    """split = 1: every step as the two halves of real-time iteration through the same generated function -- rti_phase PREPARATION (per
    capsule on host threads, then ocp_qp_gpu_xcond_solver_acados_condense_lhs_batch: all matrices to the device, condensed there), the
    new x0, rti_phase FEEDBACK (ocp_qp_gpu_xcond_solver_acados_condense_rhs_and_solve_batch: only the QPs' vector members travel)"""
    gpu = "gpu" in request.node.callspec.id
    N, iters = 20, 5
    # (GPU tier: the 1,024-capsule run the review asked for on the condensed one-call variant; 128 capsules on the other three -- creating a
    #  capsule is 50 ms of reference code, most of such a test's time)
    n, twins = ((1024, 24) if cond_N and not split else (128, 8)) if gpu else (6, 3)
    m = _model(N, n, seed=7)
    exe = _exe(clib._name)
    rows, times, log = _run(exe, m, tmp_path, twins, iters, cond_N if cond_N else N, split=split)
    assert "wrong field" not in log and "batch_phase" not in log, log
    assert len(rows["lock"]) == n * iters and len(rows["twin"]) == twins * iters
    # every capsule of every step: NLP status SUCCESS, QP converged
    for key, r in rows["lock"].items():
        assert r["status"] == 0 and r["qp_status"] == 0 and 1 <= r["qp_iter"] <= 30, (key, r)
    # lock-step == the reference's per-capsule loop on the twins, every step of the closed loop
    worst = 0.0
    for (it, i), t in rows["twin"].items():
        lk = rows["lock"][(it, i)]
        assert t["status"] == 0 and t["qp_status"] == 0
        worst = max(worst, np.abs(lk["u0"] - t["u0"]).max(), np.abs(lk["x1"] - t["x1"]).max())
        assert abs(lk["qp_iter"] - t["qp_iter"]) <= 1, (it, i, lk["qp_iter"], t["qp_iter"])
    assert worst <= 2e-7, worst            # (different batch sizes run on different kernel families: rounding + the 1e-8 exit ball)
    # step 0 against the oracle on the MPC QP written down in Python (zero initial guess: the RTI step IS the QP solution)
    for i in sorted({0, n // 2, n - 1}):
        o = OracleQp(_mpc_qp(m, m["x0"][i]))
        assert o.solve(default_opts(tol_stat=1e-8, tol_eq=1e-8, tol_ineq=1e-8, tol_comp=1e-8)) == 0
        lk = rows["lock"][(0, i)]
        assert np.allclose(lk["u0"], o.get(0, "u"), atol=2e-6) and np.allclose(lk["x1"], o.get(1, "x"), atol=2e-6), (i, lk["u0"], o.get(0, "u"))
        assert np.all(np.abs(lk["u0"]) <= m["umax"] + 1e-9)
    # what the last batch call sent per QP: the whole input blob (N (nx (nx + nu + 1) + ...) doubles), or -- feedback half of a split
    # step -- the vector members only
    full_blob = N * (8 * 8 + 8 * 3 + 8) + (N + 1) * (8 * 8 + 8 + 3) + N * (9 + 3 * 8 + 3)
    assert (times[2] < 0.3 * full_blob) if split else (times[2] > 0.9 * full_blob), (times[2], full_blob)
    # the closed loop moves: later steps solve different QPs
    assert np.abs(rows["lock"][(iters - 1, 0)]["x1"] - rows["lock"][(0, 0)]["x1"]).max() > 1e-3
    print(f"lock-step ({'preparation + feedback' if split else 'one call per step'}): {n} capsules x {iters} RTI steps {times[0] * 1e3:.1f} ms; per-capsule loop on {twins} twins {times[1] * 1e3:.1f} ms; "
          f"max |lock - twin| {worst:.2e}")


def test_batch_qp_phase_rejected_outside_the_feedback_step(hostsim_lib, tmp_path):
    """the patched ocp_nlp_sqp_rti refuses batch_qp_phase != 0 with rti_phase PREPARATION or AS-RTI: read off the
    patched source the driver is built from"""
    if not os.path.isdir(os.path.join(REFERENCE, "acados", "ocp_nlp")):
        pytest.skip("no reference tree")
    sys.path.insert(0, os.path.join(ROOT, "integration"))
    from patched_copy import patched_copy
    pat = patched_copy(REFERENCE, str(tmp_path / "patched"))
    rti = open(os.path.join(pat, "acados/ocp_nlp/ocp_nlp_sqp_rti.c")).read()
    i = rti.index("int ocp_nlp_sqp_rti(void *config_")
    guard = rti.index("opts->batch_qp_phase != 0 && (opts->as_rti_level != STANDARD_RTI || rti_phase == PREPARATION)", i)
    assert guard < rti.index("ocp_nlp_sqp_rti_feedback_step(config, dims, nlp_in, nlp_out, opts, mem, work);", i)
    assert "exit(1);" in rti[guard:guard + 400]
