#!/usr/bin/env python3
"""Generates integration/acados.patch: the registration of PARTIAL_CONDENSING_GPU_IPM in an acados checkout.
The edits are made on a scratch copy of the eight reference files they touch and `diff -u` writes the patch, so the patch
applies by construction (`patch -p1 --dry-run` is re-checked in tests/test_integration_patch.py).  Nothing of the reference
is stored in this repository besides the context lines a unified diff carries.

    python integration/make_patch.py [/root/reference]
"""
import os
import shutil
import subprocess
import sys
import tempfile

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

EDITS = {
    # 1. the enum value, among the partial-condensing solvers (they have to come before the full-condensing ones: :58-59)
    "interfaces/acados_c/ocp_qp_interface.h": [
        ("///   PARTIAL_CONDENSING_CLARABEL\n", "///   PARTIAL_CONDENSING_CLARABEL\n///   PARTIAL_CONDENSING_GPU_IPM\n"),
        ("#ifdef ACADOS_WITH_QPDUNES\n    PARTIAL_CONDENSING_QPDUNES,\n",
         "#ifdef ACADOS_WITH_GPU_IPM\n    PARTIAL_CONDENSING_GPU_IPM,\n#else\n    PARTIAL_CONDENSING_GPU_IPM_NOT_AVAILABLE,\n#endif\n"
         "#ifdef ACADOS_WITH_QPDUNES\n    PARTIAL_CONDENSING_QPDUNES,\n"),
    ],
    # 2. plan -> config, name -> plan
    "interfaces/acados_c/ocp_qp_interface.c": [
        ('#ifdef ACADOS_WITH_CLARABEL\n#include "acados/ocp_qp/ocp_qp_clarabel.h"\n#endif\n',
         '#ifdef ACADOS_WITH_CLARABEL\n#include "acados/ocp_qp/ocp_qp_clarabel.h"\n#endif\n\n'
         '#ifdef ACADOS_WITH_GPU_IPM\n#include "acados/ocp_qp/ocp_qp_gpu_ipm.h"\n#endif\n'),
        ("#ifdef ACADOS_WITH_QPDUNES\n        case PARTIAL_CONDENSING_QPDUNES:\n",
         "#ifdef ACADOS_WITH_GPU_IPM\n        case PARTIAL_CONDENSING_GPU_IPM:\n"
         "            ocp_qp_xcond_solver_config_initialize_default(solver_config);\n"
         "            ocp_qp_gpu_ipm_acados_config_initialize_default(solver_config->qp_solver);\n"
         "            ocp_qp_partial_condensing_config_initialize_default(solver_config->xcond);\n"
         "            break;\n#endif\n"
         "#ifdef ACADOS_WITH_QPDUNES\n        case PARTIAL_CONDENSING_QPDUNES:\n"),
        ('#ifdef ACADOS_WITH_QPDUNES\n    else if (!strcmp(solver_name, "PARTIAL_CONDENSING_QPDUNES"))\n',
         '#ifdef ACADOS_WITH_GPU_IPM\n    else if (!strcmp(solver_name, "PARTIAL_CONDENSING_GPU_IPM"))\n    {\n'
         '        plan.qp_solver = PARTIAL_CONDENSING_GPU_IPM;\n    }\n#endif\n'
         '#ifdef ACADOS_WITH_QPDUNES\n    else if (!strcmp(solver_name, "PARTIAL_CONDENSING_QPDUNES"))\n'),
    ],
    # 3. build options
    "CMakeLists.txt": [
        ('option(ACADOS_WITH_CLARABEL "Clarabel solver" OFF)\n',
         'option(ACADOS_WITH_CLARABEL "Clarabel solver" OFF)\n'
         'option(ACADOS_WITH_GPU_IPM "batched OCP-QP IPM on AMD MI355X (libacados_amd_qp.so, set ACADOS_AMD_QP_DIR)" OFF)\n'),
        ("if(${ACADOS_WITH_CLARABEL})\n    set(LINK_FLAG_CLARABEL -lclarabel_c)\nendif()\n",
         "if(${ACADOS_WITH_CLARABEL})\n    set(LINK_FLAG_CLARABEL -lclarabel_c)\nendif()\n"
         "if(${ACADOS_WITH_GPU_IPM})\n    set(LINK_FLAG_GPU_IPM -lacados_amd_qp)\nendif()\n"),
    ],
    "acados/CMakeLists.txt": [
        ('if(NOT ACADOS_WITH_CLARABEL)\n    list(REMOVE_ITEM ACADOS_SRC "${PROJECT_SOURCE_DIR}/acados/ocp_qp/ocp_qp_clarabel.c")\nendif()\n',
         'if(NOT ACADOS_WITH_CLARABEL)\n    list(REMOVE_ITEM ACADOS_SRC "${PROJECT_SOURCE_DIR}/acados/ocp_qp/ocp_qp_clarabel.c")\nendif()\n\n'
         'if(NOT ACADOS_WITH_GPU_IPM)\n    list(REMOVE_ITEM ACADOS_SRC "${PROJECT_SOURCE_DIR}/acados/ocp_qp/ocp_qp_gpu_ipm.c")\nendif()\n'),
        ("    target_compile_definitions(acados PUBLIC ACADOS_WITH_CLARABEL)\nendif()\n",
         "    target_compile_definitions(acados PUBLIC ACADOS_WITH_CLARABEL)\nendif()\n\n"
         "if(ACADOS_WITH_GPU_IPM)\n"
         "    # the C-ABI library of the MI355X backend (hipcc product) and its headers (include/acados_amd/*.h)\n"
         "    find_library(ACADOS_AMD_QP_LIB acados_amd_qp HINTS ${ACADOS_AMD_QP_DIR}/acados_amd/csrc REQUIRED)\n"
         "    target_include_directories(acados PRIVATE ${ACADOS_AMD_QP_DIR}/include)\n"
         "    target_link_libraries(acados PUBLIC ${ACADOS_AMD_QP_LIB})\n\n"
         "    target_compile_definitions(acados PUBLIC ACADOS_WITH_GPU_IPM)\nendif()\n"),
        ("                                        ACADOS_WITH_CLARABEL\n", "                                        ACADOS_WITH_CLARABEL\n"
         "                                        ACADOS_WITH_GPU_IPM\n"),
    ],
    # 4. Python allow-lists (the option docs and the two validators)
    "interfaces/acados_template/acados_template/acados_ocp_options.py": [
        ("'PARTIAL_CONDENSING_QPDUNES', 'PARTIAL_CONDENSING_OSQP', 'PARTIAL_CONDENSING_CLARABEL', \\\n",
         "'PARTIAL_CONDENSING_QPDUNES', 'PARTIAL_CONDENSING_OSQP', 'PARTIAL_CONDENSING_CLARABEL', 'PARTIAL_CONDENSING_GPU_IPM', \\\n"),
    ],
    # 5. generated-code build files: link flag + define when the solver is selected
    "interfaces/acados_template/acados_template/c_templates_tera/CMakeLists.in.txt": [
        ('{%- if qp_solver == "PARTIAL_CONDENSING_CLARABEL" -%}\n    -DACADOS_WITH_CLARABEL\n',
         '{%- if qp_solver == "PARTIAL_CONDENSING_GPU_IPM" -%}\n    -DACADOS_WITH_GPU_IPM\n{%- endif -%}\n'
         '{%- if qp_solver == "PARTIAL_CONDENSING_CLARABEL" -%}\n    -DACADOS_WITH_CLARABEL\n'),
    ],
    "interfaces/acados_template/acados_template/c_templates_tera/Makefile.in": [
        ('{%- if qp_solver == "PARTIAL_CONDENSING_CLARABEL" %}\nCPPFLAGS += -DACADOS_WITH_CLARABEL\n',
         '{%- if qp_solver == "PARTIAL_CONDENSING_GPU_IPM" %}\nCPPFLAGS += -DACADOS_WITH_GPU_IPM\nLDLIBS += -lacados_amd_qp\n{%- endif %}\n'
         '{%- if qp_solver == "PARTIAL_CONDENSING_CLARABEL" %}\nCPPFLAGS += -DACADOS_WITH_CLARABEL\n'),
    ],
}

HEADER = '''/*
 * acados/ocp_qp/ocp_qp_gpu_ipm.h -- inner QP plugin "PARTIAL_CONDENSING_GPU_IPM": the batched OCP-QP interior-point solver on
 * AMD MI355X behind acados' qp_solver_config (acados/ocp_qp/ocp_qp_common.h:60-79).  Implementation:
 * acados/ocp_qp/ocp_qp_gpu_ipm.c (= integration/ocp_qp_gpu_ipm.c of the acados_amd repository), which calls the C-ABI of
 * libacados_amd_qp.so (include/acados_amd/ocp_qp_gpu_batch.h).
 */
#ifndef ACADOS_OCP_QP_OCP_QP_GPU_IPM_H_
#define ACADOS_OCP_QP_OCP_QP_GPU_IPM_H_

#ifdef __cplusplus
extern "C" {
#endif

#include "acados/ocp_qp/ocp_qp_common.h"
#include "acados/utils/types.h"

/* fills the 17 slots of qp_solver_config (the counterpart of ocp_qp_hpipm_config_initialize_default, ocp_qp_hpipm.c:517-540) */
void ocp_qp_gpu_ipm_acados_config_initialize_default(void *config);

/* batch extension: n capsules' QPs as ONE device batch per structure class (replaces the per-capsule loop of
 * acados_solver.in.c:3222-3243); mem[i] = the qp solver memory of capsule i */
int ocp_qp_gpu_ipm_acados_evaluate_batch(void *config, int n, void **qp_in, void **qp_out, void *opts, void **mem, void *work);
void ocp_qp_gpu_ipm_acados_eval_sens_batch(void *config, int n, void **qp_in, void **seed, void **sens_qp_out, void *opts, void **mem,
                                           void *work);

/* rendezvous: the n `evaluate` calls of an UNMODIFIED _acados_batch_solve loop become one device batch (option "rendezvous") */
typedef struct ocp_qp_gpu_ipm_rendezvous_ ocp_qp_gpu_ipm_rendezvous;
ocp_qp_gpu_ipm_rendezvous *ocp_qp_gpu_ipm_acados_rendezvous_create(int n_capsules);
void ocp_qp_gpu_ipm_acados_rendezvous_reset(ocp_qp_gpu_ipm_rendezvous *r);
void ocp_qp_gpu_ipm_acados_rendezvous_leave(ocp_qp_gpu_ipm_rendezvous *r);
void ocp_qp_gpu_ipm_acados_rendezvous_destroy(ocp_qp_gpu_ipm_rendezvous *r);

#ifdef __cplusplus
}
#endif

#endif  // ACADOS_OCP_QP_OCP_QP_GPU_IPM_H_
'''


def main():
    with tempfile.TemporaryDirectory() as tmp:
        a, b = os.path.join(tmp, "a"), os.path.join(tmp, "b")
        for rel in EDITS:
            for root in (a, b):
                os.makedirs(os.path.dirname(os.path.join(root, rel)), exist_ok=True)
                shutil.copy(os.path.join(REF, rel), os.path.join(root, rel))
            txt = open(os.path.join(b, rel)).read()
            for old, new in EDITS[rel]:
                if old not in txt:
                    raise SystemExit(f"{rel}: anchor not found: {old[:60]!r}")
                txt = txt.replace(old, new)      # every occurrence (the Python validators appear twice)
            open(os.path.join(b, rel), "w").write(txt)
        os.makedirs(os.path.join(b, "acados", "ocp_qp"), exist_ok=True)
        open(os.path.join(b, "acados", "ocp_qp", "ocp_qp_gpu_ipm.h"), "w").write(HEADER)
        r = subprocess.run(["diff", "-ruN", "a", "b"], cwd=tmp, capture_output=True, text=True)
        assert r.returncode in (0, 1), r.stderr
        import re
        patch = re.sub(r"^(---|\+\+\+) (\S+)\t.*$", r"\1 \2", r.stdout, flags=re.M)      # no timestamps: the file is reproducible
        head = ("# Registration of PARTIAL_CONDENSING_GPU_IPM in an acados checkout (INTEGRATION.md 3).  Apply from the acados root:\n"
                "#     patch -p1 < acados.patch && cp <acados_amd>/integration/ocp_qp_gpu_ipm.c acados/ocp_qp/\n"
                "#     cmake -DACADOS_WITH_GPU_IPM=ON -DACADOS_AMD_QP_DIR=<acados_amd> ..\n"
                "# Generated by integration/make_patch.py against the reference tree; checked by tests/test_integration_patch.py.\n")
        open(os.path.join(HERE, "acados.patch"), "w").write(head + patch)
        print(f"wrote {os.path.join(HERE, 'acados.patch')}: {patch.count(chr(10))} lines, {len(EDITS) + 1} files")


if __name__ == "__main__":
    main()
