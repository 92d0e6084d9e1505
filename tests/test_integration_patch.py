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
         "interfaces/acados_template/acados_template/c_templates_tera/Makefile.in",
         "acados/ocp_qp/ocp_qp_xcond_solver.c", "acados/ocp_nlp/ocp_nlp_common.h", "acados/ocp_nlp/ocp_nlp_common.c", "acados/ocp_nlp/ocp_nlp_sqp_rti.h", "acados/ocp_nlp/ocp_nlp_sqp_rti.c",
         "interfaces/acados_template/acados_template/c_templates_tera/acados_solver.in.h",
         "interfaces/acados_template/acados_template/c_templates_tera/acados_solver.in.c"]
PLUGIN_FILES = ["ocp_qp_gpu_ipm.c", "ocp_qp_gpu_pcond.c", "ocp_qp_gpu_segments.h"]
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
    for f in PLUGIN_FILES:
        shutil.copy(os.path.join(ROOT, "integration", f), root / "acados" / "ocp_qp" / f)
    inc = ["-I", str(root), "-I", str(root / "interfaces"), "-I", REF, "-I", os.path.join(REF, "interfaces"),
           "-I", os.path.join(ROOT, "tests", "mock_hpipm"), "-I", os.path.join(ROOT, "include")]
    base = ["gcc", "-std=gnu11", "-fsyntax-only", "-Wall", "-Wno-unused-parameter", "-DACADOS_WITH_GPU_IPM"] + inc
    # the reference's registration file with the new enum value, case and name
    r = subprocess.run(base + [str(root / "interfaces" / "acados_c" / "ocp_qp_interface.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    src = open(root / "interfaces" / "acados_c" / "ocp_qp_interface.c").read()
    assert "case PARTIAL_CONDENSING_GPU_IPM:" in src and "ocp_qp_gpu_ipm_acados_config_initialize_default(solver_config->qp_solver);" in src
    # the DEVICE condensing module is what the case registers in the xcond slot (round-4 review: it was HPIPM's CPU module)
    case = src[src.index("case PARTIAL_CONDENSING_GPU_IPM:"):]
    case = case[:case.index("break;")]
    assert "ocp_qp_gpu_pcond_acados_config_initialize_default(solver_config->xcond);" in case and "ocp_qp_partial_condensing_config" not in case
    # without the define the tree is what it was (the enum keeps its slot: PARTIAL_CONDENSING_GPU_IPM_NOT_AVAILABLE)
    r = subprocess.run([a for a in base if a != "-DACADOS_WITH_GPU_IPM"] + [str(root / "interfaces" / "acados_c" / "ocp_qp_interface.c")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # the adapter and the condensing module behind the header the patch adds: conflicting prototypes would not compile
    for f in ("ocp_qp_gpu_ipm.c", "ocp_qp_gpu_pcond.c"):
        r = subprocess.run(base + ["-fopenmp", "-Wno-unused-function", "-include", "acados/ocp_qp/ocp_qp_gpu_ipm.h", str(root / "acados" / "ocp_qp" / f)],
                           capture_output=True, text=True)
        assert r.returncode == 0, (f, r.stderr)
    # the reference's files the lock-step batch touches compile with and without the define: the outer solver's terminate (releases
    # the condensing module's device batch), ocp_nlp_common.c (batch_qp_phase: return in front of / resume behind the QP solve) and
    # ocp_nlp_sqp_rti.c (the feedback step split at the QP solve)
    for f in ("acados/ocp_qp/ocp_qp_xcond_solver.c", "acados/ocp_nlp/ocp_nlp_common.c", "acados/ocp_nlp/ocp_nlp_sqp_rti.c"):
        for cmd in (base, [a for a in base if a != "-DACADOS_WITH_GPU_IPM"]):
            r = subprocess.run(cmd + [str(root / f)], capture_output=True, text=True)
            assert r.returncode == 0, (f, r.stderr)
    rti = open(root / "acados" / "ocp_nlp" / "ocp_nlp_sqp_rti.c").read()
    assert "batch_qp_resume: ;" in rti and "qp_batch_phase" not in rti
    # the option string has no `qp_` prefix (ocp_nlp_opts_set hands every `qp_*` string to the QP solver, which exits on an unknown field:
    # ocp_nlp_common.c:1337-1349) and lives in SQP_RTI's own opts; the field of ocp_nlp_opts is set around the feedback step's call only
    assert '!strcmp(field, "batch_qp_phase")' in rti and "nlp_opts->batch_qp_phase = opts->batch_qp_phase;" in rti and "nlp_opts->batch_qp_phase = 0;" in rti
    common = open(root / "acados" / "ocp_nlp" / "ocp_nlp_common.c").read()
    assert '"batch_qp_phase"' not in common and common.count("nlp_opts->batch_qp_phase") == 3
    tpl = open(root / "interfaces/acados_template/acados_template/c_templates_tera/acados_solver.in.c").read()
    assert tpl.count('"batch_qp_phase"') == 3 and '"qp_batch_phase"' not in tpl


def test_generated_lock_step_batch_call_compiles(tmp_path):
    """the `_acados_batch_solve_gpu_qp` hunk of acados_solver.in.c (the explicit lock-step batch call next to the per-capsule loop of
    :3222-3243): the function is cut out of the PATCHED template, its Tera placeholders are filled in by hand ({{ name }} -> mpc) and it
    is compiled against the reference's own headers with the capsule struct restated from acados_solver.in.h"""
    root = _copy(tmp_path)
    subprocess.check_call(["patch", "-p1", "-s", "-i", os.path.join(ROOT, "integration", "acados.patch")], cwd=root)
    tpl = open(root / "interfaces/acados_template/acados_template/c_templates_tera/acados_solver.in.c").read()
    hdr = open(root / "interfaces/acados_template/acados_template/c_templates_tera/acados_solver.in.h").read()
    assert "_acados_batch_solve_gpu_qp(" in hdr
    i = tpl.index("void {{ name }}_acados_batch_solve_gpu_qp(")
    guard = tpl.rindex("{%- if", 0, i)
    assert 'solver_options.qp_solver == "PARTIAL_CONDENSING_GPU_IPM"' in tpl[guard:i] and 'solver_options.nlp_solver_type == "SQP_RTI"' in tpl[guard:i]
    fn = tpl[guard:tpl.index("{%- endif %}", i)]
    fn = fn[fn.index("\n") + 1:].replace("{{ name }}", "mpc")          # the guard line itself is Tera
    assert "{{" not in fn and "{%" not in fn
    src = tmp_path / "batch_fn.c"
    src.write_text('''#include <stdlib.h>
#include <omp.h>
#include "acados_c/ocp_nlp_interface.h"
typedef struct mpc_solver_capsule
{
    ocp_nlp_in *nlp_in; ocp_nlp_out *nlp_out; ocp_nlp_out *sens_out; ocp_nlp_solver *nlp_solver; void *nlp_opts; ocp_nlp_plan_t *nlp_solver_plan;
    ocp_nlp_config *nlp_config; ocp_nlp_dims *nlp_dims;
} mpc_solver_capsule;
''' + fn)
    inc = ["-I", str(root), "-I", str(root / "interfaces"), "-I", REF, "-I", os.path.join(REF, "interfaces"),
           "-I", os.path.join(ROOT, "tests", "mock_hpipm"), "-I", os.path.join(ROOT, "include")]
    r = subprocess.run(["gcc", "-std=gnu11", "-fsyntax-only", "-fopenmp", "-Wall", "-Wno-unused-parameter", "-DACADOS_WITH_GPU_IPM"] + inc + [str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
