import ctypes
import json
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


# The library picks the kernel family by stage-block size AND batch size (small batches go to the
# wave-per-instance kernels).  The tests run small batches but mean to cover both families, so the batch rule is
# switched off here; tests that want it (or a specific family) set ACADOS_AMD_WPI_BATCH_MAX / ACADOS_AMD_WPI.
os.environ.setdefault("ACADOS_AMD_WPI_BATCH_MAX", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu():
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def hostsim_lib():
    """g++ build of the kernel sources against the host-simulation shim (test infra only)."""
    from hostsim.build import build
    from acados_amd import _lib
    return _lib.bind(ctypes.CDLL(build()))


@pytest.fixture(scope="session")
def gpu_lib():
    """the product library (hipcc, gfx950); tests using it are marked gpu"""
    from acados_amd import _lib
    return _lib.lib()


def load_qp(rel):
    from acados_amd import AcadosOcpQp
    return AcadosOcpQp.from_json(os.path.join(GOLDEN, rel))


def load_sol(rel):
    sol = json.load(open(os.path.join(GOLDEN, rel)))
    return {re.sub(r"_0*(\d+)$", lambda m: "_" + m.group(1), k): np.asarray(v, dtype=float).ravel() for k, v in sol.items()}


GOLDEN_PAIRS = [("qp_test/last_qp_nonuniform_pendulum.json", "qp_test/sqp_sol_nonuniform_pendulum.json"),
                ("qp_test/last_qp_one_sided_test.json", "qp_test/sqp_sol_one_sided_test.json")]
INPUT_ONLY = ["casadi_qp_tests/pendulum_qp.json", "casadi_qp_tests/pendulum_slack.json",
              "casadi_qp_tests/pend_idxs_rev_min_qp0.json"]


def fold_stage0(lam, hard):
    """unique-dual fold of acados_ocp_qp_solver.py:388-397"""
    lam = lam.copy()
    u = lam[hard:2 * hard] - lam[:hard]
    lam[:hard] = np.maximum(0.0, -u)
    lam[hard:2 * hard] = np.maximum(0.0, u)
    return lam


def compare_with_oracle(get_fn, oracle, qp, tol, fields=("x", "u", "sl", "su", "pi", "lam")):
    """max abs deviation of get_fn(stage, field) from the oracle's solution"""
    worst = 0.0
    for k in range(qp.N + 1):
        for f in fields:
            if f == "pi" and k == qp.N:
                continue
            ref = oracle.get(k, f)
            if ref.size == 0:
                continue
            got = np.asarray(get_fn(k, f))
            err = np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref)))
            assert err <= tol, f"{f} at stage {k}: {err} > {tol}\n got {got}\n ref {ref}"
            worst = max(worst, err)
    return worst
