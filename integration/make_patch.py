#!/usr/bin/env python3
"""Generates integration/acados.patch: the registration of PARTIAL_CONDENSING_GPU_IPM in an acados checkout.
The edits are made on a scratch copy of the reference files they touch and `diff -u` writes the patch, so the patch
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

# the lock-step batch call of the generated solver (next to _acados_batch_solve, acados_solver.in.c:3222-3243)
BATCH_GPU_QP = '''{%- if solver_options.qp_solver == "PARTIAL_CONDENSING_GPU_IPM" and solver_options.nlp_solver_type == "SQP_RTI" %}
#include "acados/ocp_nlp/ocp_nlp_sqp_rti.h"
#include "acados/ocp_qp/ocp_qp_gpu_ipm.h"
/*
 * One RTI step of EVERY capsule with ONE device batch for the QPs (ACADOS_WITH_GPU_IPM).  What runs is decided by the capsules' own
 * rti_phase option (acados_ocp_options.py: 0 preparation + feedback, 1 preparation, 2 feedback; the capsules share their options, the
 * batch runs with capsule 0's):
 *   0 / 2   host threads linearise and set up each capsule's QP (phase 1: ocp_nlp_solve returns in front of the QP solve), the N_batch
 *           QPs -- acados structs, as each capsule's NLP solver scaled them -- go to the GPU in one call (partial condensing, IPM and
 *           expansion fused on the device), host threads finish the step (phase 2: dual correction, globalisation, update of the iterate).
 *           rti_phase 2 (FEEDBACK) sends only the VECTOR members of the QPs (gradient, offsets, bounds): the matrices are resident
 *           since the preparation call;
 *   1       every capsule's preparation step on host threads (its condense_lhs slot has nothing to do: option qp_cond_batch_owned),
 *           then the matrices of all N_batch QPs to the GPU and the matrix part of the condensing there.
 * Unlike the one-thread-per-capsule loop above, num_threads_in_batch_solve is only the width of the host phases; the batch may hold
 * thousands of capsules.
 */
void {{ name }}_acados_batch_solve_gpu_qp({{ name }}_solver_capsule ** capsules, int * status_out, int N_batch, int num_threads_in_batch_solve)
{
    int num_threads_bkp;
    if (num_threads_in_batch_solve > 1)
    {
        num_threads_bkp = omp_get_num_threads();
        omp_set_num_threads(num_threads_in_batch_solve);
    }

    ocp_nlp_sqp_rti_opts *opts0 = capsules[0]->nlp_opts;
    const int rti_phase = opts0->rti_phase;
    int phase = rti_phase == PREPARATION ? 0 : 1;
    int batch_owned = 1;
    #pragma omp parallel for
    for (int i = 0; i < N_batch; i++)
    {
        ocp_nlp_solver_opts_set(capsules[i]->nlp_config, capsules[i]->nlp_opts, "qp_cond_batch_owned", &batch_owned);
        ocp_nlp_solver_opts_set(capsules[i]->nlp_config, capsules[i]->nlp_opts, "batch_qp_phase", &phase);
        status_out[i] = ocp_nlp_solve(capsules[i]->nlp_solver, capsules[i]->nlp_in, capsules[i]->nlp_out);
    }

    ocp_qp_in **qp_in = malloc(N_batch * sizeof(ocp_qp_in *));
    ocp_qp_out **qp_out = malloc(N_batch * sizeof(ocp_qp_out *));
    void **qp_mem = malloc(N_batch * sizeof(void *));
    for (int i = 0; i < N_batch; i++)
    {
        ocp_nlp_memory *nlp_mem;
        ocp_nlp_get(capsules[i]->nlp_solver, "nlp_mem", &nlp_mem);
        qp_in[i] = nlp_mem->scaled_qp_in;
        qp_out[i] = nlp_mem->scaled_qp_out;
        qp_mem[i] = nlp_mem->qp_solver_mem;
    }
    if (rti_phase == PREPARATION)
    {
        ocp_qp_gpu_xcond_solver_acados_condense_lhs_batch(capsules[0]->nlp_config->qp_solver, capsules[0]->nlp_dims->qp_solver, N_batch,
                                                          qp_in, opts0->nlp_opts->qp_solver_opts, qp_mem, NULL);
    }
    else if (rti_phase == FEEDBACK)
    {
        ocp_qp_gpu_xcond_solver_acados_condense_rhs_and_solve_batch(capsules[0]->nlp_config->qp_solver, capsules[0]->nlp_dims->qp_solver, N_batch,
                                                                    qp_in, qp_out, opts0->nlp_opts->qp_solver_opts, qp_mem, NULL);
    }
    else
    {
        ocp_qp_gpu_xcond_solver_acados_evaluate_batch(capsules[0]->nlp_config->qp_solver, capsules[0]->nlp_dims->qp_solver, N_batch,
                                                      qp_in, qp_out, opts0->nlp_opts->qp_solver_opts, qp_mem, NULL);
    }
    free(qp_in);
    free(qp_out);
    free(qp_mem);

    if (rti_phase != PREPARATION)
    {
        phase = 2;
        #pragma omp parallel for
        for (int i = 0; i < N_batch; i++)
        {
            ocp_nlp_solver_opts_set(capsules[i]->nlp_config, capsules[i]->nlp_opts, "batch_qp_phase", &phase);
            status_out[i] = ocp_nlp_solve(capsules[i]->nlp_solver, capsules[i]->nlp_in, capsules[i]->nlp_out);
            int phase_off = 0;
            ocp_nlp_solver_opts_set(capsules[i]->nlp_config, capsules[i]->nlp_opts, "batch_qp_phase", &phase_off);
        }
    }
    batch_owned = 0;
    for (int i = 0; i < N_batch; i++)
        ocp_nlp_solver_opts_set(capsules[i]->nlp_config, capsules[i]->nlp_opts, "qp_cond_batch_owned", &batch_owned);

    if (num_threads_in_batch_solve > 1)
    {
        omp_set_num_threads( num_threads_bkp );
    }
    return;
}
{%- endif %}


'''

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
         "            // partial condensing ON THE DEVICE (km_pcond / k_pexpand of libacados_amd_qp.so) behind ocp_qp_xcond_config\n"
         "            ocp_qp_gpu_pcond_acados_config_initialize_default(solver_config->xcond);\n"
         "            break;\n#endif\n"
         "#ifdef ACADOS_WITH_QPDUNES\n        case PARTIAL_CONDENSING_QPDUNES:\n"),
        ('#ifdef ACADOS_WITH_QPDUNES\n    else if (!strcmp(solver_name, "PARTIAL_CONDENSING_QPDUNES"))\n',
         '#ifdef ACADOS_WITH_GPU_IPM\n    else if (!strcmp(solver_name, "PARTIAL_CONDENSING_GPU_IPM"))\n    {\n'
         '        plan.qp_solver = PARTIAL_CONDENSING_GPU_IPM;\n    }\n#endif\n'
         '#ifdef ACADOS_WITH_QPDUNES\n    else if (!strcmp(solver_name, "PARTIAL_CONDENSING_QPDUNES"))\n'),
    ],
    # 2b. the condensing module's device batch goes with the solver (the xcond vtable has no terminate slot, ocp_qp_common.h:84-107)
    "acados/ocp_qp/ocp_qp_xcond_solver.c": [
        ('#include "acados/utils/types.h"\n',
         '#include "acados/utils/types.h"\n\n#ifdef ACADOS_WITH_GPU_IPM\n#include "acados/ocp_qp/ocp_qp_gpu_ipm.h"\n#endif\n'),
        ("    qp_solver->terminate(config->qp_solver, memory->solver_memory, work->qp_solver_work);\n",
         "    qp_solver->terminate(config->qp_solver, memory->solver_memory, work->qp_solver_work);\n"
         "#ifdef ACADOS_WITH_GPU_IPM\n"
         "    if (ocp_qp_gpu_pcond_acados_is_module(config->xcond)) ocp_qp_gpu_pcond_acados_memory_release(memory->xcond_memory);\n"
         "#endif\n"),
    ],
    # 2c. lock-step batch (acados_solver.in.c:3222-3243): an SQP-RTI feedback step that stops in front of the QP solve
    # (phase 1) and resumes behind it (phase 2), so that the QPs of ALL capsules are solved by ONE device batch in between
    "acados/ocp_nlp/ocp_nlp_common.h": [
        ("    int ext_qp_res;\n",
         "    int ext_qp_res;\n"
         "    // TRANSIENT, not an option string: SQP_RTI's feedback step sets it around ITS call of ocp_nlp_solve_qp_and_correct_dual only\n"
         "    // (1: return in front of the QP solve; 2: resume behind it, the QP was solved in a device batch); 0 for every other caller\n"
         "    int batch_qp_phase;\n"),
    ],
    "acados/ocp_nlp/ocp_nlp_common.c": [
        ("    opts->ext_qp_res = 0;\n", "    opts->ext_qp_res = 0;\n    opts->batch_qp_phase = 0;\n"),
        ("    // solve qp\n    acados_tic(&timer);\n    if (precondensed_lhs)\n    {\n",
         "    // lock-step batch on a GPU QP solver: phase 1 stops here -- the caller sends every capsule's (scaled) QP to the device in\n"
         "    // ONE batch (ocp_qp_gpu_xcond_solver_acados_evaluate_batch) -- phase 2 picks the result up from the QP solver's memory\n"
         "    if (nlp_opts->batch_qp_phase == 1)\n    {\n        return ACADOS_SUCCESS;\n    }\n\n"
         "    // solve qp\n    acados_tic(&timer);\n"
         "    if (nlp_opts->batch_qp_phase == 2)\n    {\n"
         '        qp_solver->memory_get(qp_solver, qp_mem, "status", &qp_status);\n    }\n'
         "    else if (precondensed_lhs)\n    {\n"),
        # phase 2: the condensing ran inside the device batch -- the module's own timer still holds an earlier per-capsule call
        ('    qp_solver->memory_get(qp_solver, qp_mem, "time_qp_xcond", &tmp_time);\n    nlp_timings->time_qp_xcond += tmp_time;\n\n'
         "    // evaluate QP residual externally\n",
         "    if (nlp_opts->batch_qp_phase != 2)\n    {\n"
         '        qp_solver->memory_get(qp_solver, qp_mem, "time_qp_xcond", &tmp_time);\n        nlp_timings->time_qp_xcond += tmp_time;\n    }\n\n'
         "    // evaluate QP residual externally\n"),
    ],
    # the option lives in SQP_RTI's OWN opts ("batch_qp_phase": no `qp_` prefix -- ocp_nlp_opts_set routes every `qp_*` string to the
    # QP solver, ocp_nlp_common.c:1337-1349, as acados' own `ext_qp_res` shows) and only the feedback step's QP solve is split
    "acados/ocp_nlp/ocp_nlp_sqp_rti.h": [
        ("    int rti_log_only_available_residuals;\n",
         "    int rti_log_only_available_residuals;\n"
         "    int batch_qp_phase;  // 0: as ever; 1: the feedback step returns in front of its QP solve; 2: it resumes behind it (lock-step batch)\n"),
    ],
    "acados/ocp_nlp/ocp_nlp_sqp_rti.c": [
        ("    opts->rti_log_only_available_residuals = 0;\n", "    opts->rti_log_only_available_residuals = 0;\n    opts->batch_qp_phase = 0;\n"),
        ('        else if (!strcmp(field, "as_rti_level"))\n',
         '        else if (!strcmp(field, "batch_qp_phase"))\n        {\n            int* batch_qp_phase = (int *) value;\n'
         '            if (*batch_qp_phase < 0 || *batch_qp_phase > 2)\n            {\n'
         '                printf("\\nerror: ocp_nlp_sqp_rti_opts_set: invalid value for batch_qp_phase field.\\n");\n'
         '                printf("possible values are: 0, 1, 2, got %d.\\n", *batch_qp_phase);\n                exit(1);\n            }\n'
         '            opts->batch_qp_phase = *batch_qp_phase;\n        }\n'
         '        else if (!strcmp(field, "as_rti_level"))\n'),
        ("    int qp_iter = 0;\n    int qp_status, globalization_status;\n\n    // update QP rhs for SQP (step prim var, abs dual var)\n",
         "    int qp_iter = 0;\n    int qp_status, globalization_status;\n\n"
         "    // lock-step batch (opts->batch_qp_phase): phase 2 resumes behind the QP solve, everything in front of it ran in phase 1\n"
         "    if (opts->batch_qp_phase == 2)\n    {\n        goto batch_qp_resume;\n    }\n\n"
         "    // update QP rhs for SQP (step prim var, abs dual var)\n"),
        ("    // solve QP\n    bool precondensed_lhs = true;\n",
         "batch_qp_resume: ;\n    // solve QP\n    bool precondensed_lhs = true;\n"),
        ("    qp_status = ocp_nlp_solve_qp_and_correct_dual(config, dims, nlp_opts, nlp_mem, nlp_work, precondensed_lhs, NULL, NULL, NULL, NULL, NULL);\n\n"
         "    qp_info *qp_info_;\n",
         "    nlp_opts->batch_qp_phase = opts->batch_qp_phase;   // this call only: no other caller of the function ever sees it set\n"
         "    qp_status = ocp_nlp_solve_qp_and_correct_dual(config, dims, nlp_opts, nlp_mem, nlp_work, precondensed_lhs, NULL, NULL, NULL, NULL, NULL);\n"
         "    nlp_opts->batch_qp_phase = 0;\n"
         "    if (opts->batch_qp_phase == 1)\n    {\n"
         "        // the QP is set up (vectors, regularisation, warm-start option): it is solved with the other capsules' QPs\n"
         "        return;\n    }\n\n"
         "    qp_info *qp_info_;\n"),
        ("    int rti_phase = opts->rti_phase;\n\n    if (rti_phase == FEEDBACK)\n",
         "    int rti_phase = opts->rti_phase;\n\n"
         "    if (opts->batch_qp_phase != 0 && (opts->as_rti_level != STANDARD_RTI || rti_phase == PREPARATION))\n    {\n"
         '        printf("ocp_nlp_sqp_rti: batch_qp_phase != 0 splits the FEEDBACK step of standard RTI; it is not supported with AS-RTI or rti_phase == PREPARATION.\\n\\n");\n'
         "        exit(1);\n    }\n\n"
         "    if (rti_phase == FEEDBACK || opts->batch_qp_phase == 2)\n"),
    ],
    "interfaces/acados_template/acados_template/c_templates_tera/acados_solver.in.h": [
        ("ACADOS_SYMBOL_EXPORT void {{ name }}_acados_batch_solve({{ name }}_solver_capsule ** capsules, int * status_out, int N_batch, int num_threads_in_batch_solve);\n",
         "ACADOS_SYMBOL_EXPORT void {{ name }}_acados_batch_solve({{ name }}_solver_capsule ** capsules, int * status_out, int N_batch, int num_threads_in_batch_solve);\n"
         '{%- if solver_options.qp_solver == "PARTIAL_CONDENSING_GPU_IPM" and solver_options.nlp_solver_type == "SQP_RTI" %}\n'
         "// lock-step RTI over the batch, the QPs of all capsules in ONE device batch per call\n"
         "ACADOS_SYMBOL_EXPORT void {{ name }}_acados_batch_solve_gpu_qp({{ name }}_solver_capsule ** capsules, int * status_out, int N_batch, int num_threads_in_batch_solve);\n"
         "{%- endif %}\n"),
    ],
    "interfaces/acados_template/acados_template/c_templates_tera/acados_solver.in.c": [
        ("void {{ name }}_acados_batch_setup_qp_matrices_and_factorize(", BATCH_GPU_QP + "void {{ name }}_acados_batch_setup_qp_matrices_and_factorize("),
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
         'if(NOT ACADOS_WITH_GPU_IPM)\n    list(REMOVE_ITEM ACADOS_SRC "${PROJECT_SOURCE_DIR}/acados/ocp_qp/ocp_qp_gpu_ipm.c")\n'
         '    list(REMOVE_ITEM ACADOS_SRC "${PROJECT_SOURCE_DIR}/acados/ocp_qp/ocp_qp_gpu_pcond.c")\nendif()\n'),
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
#include "acados/ocp_qp/ocp_qp_xcond_solver.h"
#include "acados/utils/types.h"

/* fills the 17 slots of qp_solver_config (the counterpart of ocp_qp_hpipm_config_initialize_default, ocp_qp_hpipm.c:517-540) */
void ocp_qp_gpu_ipm_acados_config_initialize_default(void *config);

/* batch extension: n capsules' QPs as ONE device batch per structure class (replaces the per-capsule loop of
 * acados_solver.in.c:3222-3243); mem[i] = the qp solver memory of capsule i */
int ocp_qp_gpu_ipm_acados_evaluate_batch(void *config, int n, void **qp_in, void **qp_out, void *opts, void **mem, void *work);
int ocp_qp_gpu_ipm_acados_condense_lhs_batch(void *config, int n, void **qp_in, void *opts, void **mem, void *work);
int ocp_qp_gpu_ipm_acados_condense_rhs_and_solve_batch(void *config, int n, void **qp_in, void **qp_out, void *opts, void **mem, void *work);
void ocp_qp_gpu_ipm_acados_eval_sens_batch(void *config, int n, void **qp_in, void **seed, void **sens_qp_out, void *opts, void **mem,
                                           void *work);

/* partial condensing on the device behind ocp_qp_xcond_config (acados/ocp_qp/ocp_qp_gpu_pcond.c = integration/ocp_qp_gpu_pcond.c):
 * fills the 20 slots (the counterpart of ocp_qp_partial_condensing_config_initialize_default, ocp_qp_partial_condensing.c:720-750) */
void ocp_qp_gpu_pcond_acados_config_initialize_default(void *config);
/* the module's device batch (the vtable has no terminate slot; ocp_qp_xcond_solver_terminate calls this) */
void ocp_qp_gpu_pcond_acados_memory_release(void *mem);
int ocp_qp_gpu_pcond_acados_is_module(const void *xcond_config);

/* batch route at the level of the 22-slot solver `ocp_nlp` holds: n capsules' ORIGINAL QPs, condensing options taken from the
 * condensing module's opts, condensing + IPM + expansion fused on the device; mem[i] = capsule i's ocp_qp_xcond_solver_memory */
int ocp_qp_gpu_xcond_solver_acados_evaluate_batch(void *config, ocp_qp_xcond_solver_dims *dims, int n, ocp_qp_in **qp_in, ocp_qp_out **qp_out,
                                                  void *opts, void **mem, void *work);
/* ... and the two halves of an RTI step (the batch counterparts of condense_lhs / condense_rhs_and_solve, ocp_qp_xcond_solver.c:591-669):
 * the feedback half reads and sends only the vector members of every qp_in, the matrices stay on the device in between */
int ocp_qp_gpu_xcond_solver_acados_condense_lhs_batch(void *config, ocp_qp_xcond_solver_dims *dims, int n, ocp_qp_in **qp_in, void *opts, void **mem,
                                                      void *work);
int ocp_qp_gpu_xcond_solver_acados_condense_rhs_and_solve_batch(void *config, ocp_qp_xcond_solver_dims *dims, int n, ocp_qp_in **qp_in,
                                                                ocp_qp_out **qp_out, void *opts, void **mem, void *work);

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
                "#     patch -p1 < acados.patch && cp <acados_amd>/integration/ocp_qp_gpu_{ipm.c,pcond.c,segments.h} acados/ocp_qp/\n"
                "#     cmake -DACADOS_WITH_GPU_IPM=ON -DACADOS_AMD_QP_DIR=<acados_amd> ..\n"
                "# Generated by integration/make_patch.py against the reference tree; checked by tests/test_integration_patch.py.\n")
        open(os.path.join(HERE, "acados.patch"), "w").write(head + patch)
        print(f"wrote {os.path.join(HERE, 'acados.patch')}: {patch.count(chr(10))} lines, {len(EDITS) + 1} files")


if __name__ == "__main__":
    main()
