"""CPU tier, build container only (needs the reference tree): integration/acados.patch -- the registration of
PARTIAL_CONDENSING_GPU_IPM -- is what integration/make_patch.py generates, applies to the reference's files
(`patch -p1 --dry-run`), and the PATCHED interfaces/acados_c/ocp_qp_interface.c compiles with -DACADOS_WITH_GPU_IPM against the
reference's own headers (only hpipm/include/*.h and blasfeo*.h, empty submodules there, are stand-ins), as does the adapter
with the header the patch adds in front of it (prototypes agree)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

REF = "/root/reference"
FILES = ["interfaces/acados_c/ocp_qp_interface.h", "interfaces/acados_c/ocp_qp_interface.c", "CMakeLists.txt", "acados/CMakeLists.txt",
         "interfaces/acados_template/acados_template/acados_ocp_options.py",
         "interfaces/acados_template/acados_template/c_templates_tera/CMakeLists.in.txt",
         "interfaces/acados_template/acados_template/c_templates_tera/Makefile.in"]
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "acados", "ocp_qp")), reason="reference tree not present (GPU box)")


def _copy(tmp_path):
    root = tmp_path / "acados"
    for f in FILES:
        os.makedirs(root / os.path.dirname(f), exist_ok=True)
        shutil.copy(os.path.join(REF, f), root / f)
    return root


def test_patch_is_current_and_applies(tmp_path):
    patch = os.path.join(ROOT, "integration", "acados.patch")
    committed = open(patch).read()
    subprocess.check_call(["python", os.path.join(ROOT, "integration", "make_patch.py"), REF], stdout=subprocess.DEVNULL)
    assert open(patch).read() == committed, "integration/acados.patch is stale: run integration/make_patch.py"
    root = _copy(tmp_path)
    r = subprocess.run(["patch", "-p1", "--dry-run", "-i", patch], cwd=root, capture_output=True, text=True)
    assert r.returncode == 0 and "FAILED" not in r.stdout and "fuzz" not in r.stdout, r.stdout + r.stderr
    assert r.stdout.count("checking file") == len(FILES) + 1


def test_patched_registration_compiles_against_reference_headers(tmp_path):
    root = _copy(tmp_path)
    subprocess.check_call(["patch", "-p1", "-s", "-i", os.path.join(ROOT, "integration", "acados.patch")], cwd=root)
    shutil.copy(os.path.join(ROOT, "integration", "ocp_qp_gpu_ipm.c"), root / "acados" / "ocp_qp" / "ocp_qp_gpu_ipm.c")
    inc = ["-I", str(root), "-I", str(root / "interfaces"), "-I", REF, "-I", os.path.join(REF, "interfaces"),
           "-I", os.path.join(ROOT, "tests", "mock_hpipm"), "-I", os.path.join(ROOT, "include")]
    base = ["gcc", "-std=gnu11", "-fsyntax-only", "-Wall", "-Wno-unused-parameter", "-DACADOS_WITH_GPU_IPM"] + inc
    # the reference's registration file with the new enum value, case and name
    r = subprocess.run(base + [str(root / "interfaces" / "acados_c" / "ocp_qp_interface.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    src = open(root / "interfaces" / "acados_c" / "ocp_qp_interface.c").read()
    assert "case PARTIAL_CONDENSING_GPU_IPM:" in src and "ocp_qp_gpu_ipm_acados_config_initialize_default(solver_config->qp_solver);" in src
    # without the define the tree is what it was (the enum keeps its slot: PARTIAL_CONDENSING_GPU_IPM_NOT_AVAILABLE)
    r = subprocess.run([a for a in base if a != "-DACADOS_WITH_GPU_IPM"] + [str(root / "interfaces" / "acados_c" / "ocp_qp_interface.c")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # the adapter behind the header the patch adds: conflicting prototypes would not compile
    r = subprocess.run(base + ["-fopenmp", "-include", "acados/ocp_qp/ocp_qp_gpu_ipm.h", str(root / "acados" / "ocp_qp" / "ocp_qp_gpu_ipm.c")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
