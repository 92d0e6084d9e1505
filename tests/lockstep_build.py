#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- builds tests/mock_acados/lockstep_driver.c: the reference's OWN ocp_nlp stack (SQP_RTI with
integration/acados.patch applied, LINEAR_LS, DISCRETE_MODEL, BGH, NO_REGULARIZE, FIXED_STEP, the acados_c layer) compiled from
/root/reference against the HPIPM / BLASFEO stand-ins of tests/mock_hpipm, around this repository's two plugin files, linked against
the library given.  The two batch functions of the generated solver are cut out of the PATCHED template
(c_templates_tera/acados_solver.in.c) into lockstep_batch_fns.inc with `{{ name }}` -> mpc.  Nothing of the reference is stored in
this repository; the binary for the GPU tier is built by integration/Makefile (_ref_build/lockstep_driver) where the reference exists.

    python tests/lockstep_build.py <libacados_amd_qp.so | hostsim library> <output binary> [reference tree]
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SOURCES = ["acados/ocp_nlp/ocp_nlp_cost_common.c", "acados/ocp_nlp/ocp_nlp_cost_ls.c", "acados/ocp_nlp/ocp_nlp_dynamics_common.c",
               "acados/ocp_nlp/ocp_nlp_dynamics_disc.c", "acados/ocp_nlp/ocp_nlp_constraints_common.c", "acados/ocp_nlp/ocp_nlp_constraints_bgh.c",
               "acados/ocp_nlp/ocp_nlp_reg_common.c", "acados/ocp_nlp/ocp_nlp_reg_noreg.c", "acados/ocp_nlp/ocp_nlp_globalization_common.c",
               "acados/ocp_nlp/ocp_nlp_globalization_fixed_step.c", "acados/ocp_nlp/ocp_nlp_qpscaling.c", "acados/ocp_qp/ocp_qp_common.c",
               "acados/utils/mem.c", "acados/utils/timing.c", "acados/utils/math.c", "acados/utils/external_function_generic.c",
               "acados/sim/sim_common.c", "interfaces/acados_c/ocp_nlp_interface.c"]
PATCHED_SOURCES = ["acados/ocp_nlp/ocp_nlp_common.c", "acados/ocp_nlp/ocp_nlp_sqp_rti.c", "acados/ocp_qp/ocp_qp_xcond_solver.c",
                   "interfaces/acados_c/ocp_qp_interface.c", "acados/ocp_qp/ocp_qp_gpu_ipm.c", "acados/ocp_qp/ocp_qp_gpu_pcond.c"]


def batch_functions(template):
    """`_acados_batch_solve` (acados_solver.in.c:3222-3243) and the `_acados_batch_solve_gpu_qp` the patch adds, verbatim, `{{ name }}` -> mpc"""
    out = []
    for name in ("void {{ name }}_acados_batch_solve(", "void {{ name }}_acados_batch_solve_gpu_qp("):
        i = template.index(name)
        j = template.index("\n}\n", i) + 3
        fn = template[i:j].replace("{{ name }}", "mpc")
        assert "{{" not in fn and "{%" not in fn, fn
        out.append(fn)
    return "\n".join(out)


def build(libpath, exe, ref="/root/reference"):
    sys.path.insert(0, os.path.join(ROOT, "integration"))
    from patched_copy import patched_copy
    libdir, libname = os.path.dirname(os.path.abspath(libpath)), os.path.basename(libpath)
    with tempfile.TemporaryDirectory(prefix="lockstep_") as tmp:
        pat = patched_copy(ref, os.path.join(tmp, "patched"))
        tpl = open(os.path.join(pat, "interfaces/acados_template/acados_template/c_templates_tera/acados_solver.in.c")).read()
        open(os.path.join(tmp, "lockstep_batch_fns.inc"), "w").write(batch_functions(tpl))
        mock = os.path.join(ROOT, "tests", "mock_acados")
        cmd = ["gcc", "-std=gnu11", "-O2", "-fopenmp", "-w", "-DACADOS_WITH_GPU_IPM", "-DACADOS_WITH_OPENMP", "-I", tmp, "-I", pat, "-I", os.path.join(pat, "interfaces"),
               "-I", ref, "-I", os.path.join(ref, "interfaces"), "-I", os.path.join(ROOT, "tests", "mock_hpipm"), "-I", os.path.join(ROOT, "include"),
               "-I", mock, os.path.join(mock, "lockstep_driver.c"), os.path.join(mock, "nlp_stubs.c"),
               os.path.join(ROOT, "tests", "mock_hpipm", "mock_hpipm.c"), os.path.join(ROOT, "tests", "mock_hpipm", "mock_blasfeo_nlp.c")] + \
              [os.path.join(pat, f) for f in PATCHED_SOURCES] + [os.path.join(ref, f) for f in REF_SOURCES] + \
              ["-o", exe, "-L", libdir, "-l:" + libname, "-Wl,-rpath," + ("$ORIGIN/../../acados_amd/csrc" if libname == "libacados_amd_qp.so" else libdir),
               "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined", "-lm", "-lpthread"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-4000:])
    return exe


if __name__ == "__main__":
    build(sys.argv[1], sys.argv[2], *(sys.argv[3:4]))
    print("built", sys.argv[2])
