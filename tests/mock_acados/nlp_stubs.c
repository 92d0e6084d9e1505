/*
 * TEST INFRASTRUCTURE ONLY -- symbols the reference's interfaces/acados_c/ocp_nlp_interface.c / ocp_qp_interface.c /
 * acados/ocp_nlp/ocp_nlp_common.c reference beside the path under test (tests/test_lockstep_rti.py: SQP_RTI + LINEAR_LS +
 * DISCRETE_MODEL + BGH + NO_REGULARIZE + FIXED_STEP + PARTIAL_CONDENSING_GPU_IPM): the initialisers of the module variants the
 * plan does not select, HPIPM's string-keyed getters behind `ocp_nlp_get_at_stage("qp_*")`, the QP printers.  The ones that would
 * compute something abort; the printers print nothing.
 */
#include <stdio.h>
#include <stdlib.h>

#define STUB(name) void name(void) { printf("nlp_stubs: %s is not on the path under test\n", #name); exit(3); }
STUB(ocp_qp_hpipm_config_initialize_default)
STUB(dense_qp_hpipm_config_initialize_default)
STUB(ocp_qp_partial_condensing_config_initialize_default)
STUB(ocp_qp_full_condensing_config_initialize_default)
STUB(dense_qp_dims_get)
STUB(d_dense_qp_get_H)
STUB(d_ocp_qp_set)
STUB(d_ocp_qp_sol_get_x) STUB(d_ocp_qp_sol_get_u) STUB(d_ocp_qp_sol_get_pi) STUB(d_ocp_qp_sol_get_sl) STUB(d_ocp_qp_sol_get_su)
STUB(d_ocp_qp_get_A) STUB(d_ocp_qp_get_B) STUB(d_ocp_qp_get_C) STUB(d_ocp_qp_get_D) STUB(d_ocp_qp_get_Q) STUB(d_ocp_qp_get_R) STUB(d_ocp_qp_get_S)
STUB(d_ocp_qp_get_Zl) STUB(d_ocp_qp_get_Zu) STUB(d_ocp_qp_get_b) STUB(d_ocp_qp_get_idxb) STUB(d_ocp_qp_get_idxe) STUB(d_ocp_qp_get_idxs)
STUB(d_ocp_qp_get_idxs_rev) STUB(d_ocp_qp_get_lbu) STUB(d_ocp_qp_get_lbu_mask) STUB(d_ocp_qp_get_lbx) STUB(d_ocp_qp_get_lbx_mask) STUB(d_ocp_qp_get_lg)
STUB(d_ocp_qp_get_lg_mask) STUB(d_ocp_qp_get_lls) STUB(d_ocp_qp_get_lls_mask) STUB(d_ocp_qp_get_lus) STUB(d_ocp_qp_get_lus_mask) STUB(d_ocp_qp_get_q)
STUB(d_ocp_qp_get_r) STUB(d_ocp_qp_get_ubu) STUB(d_ocp_qp_get_ubu_mask) STUB(d_ocp_qp_get_ubx) STUB(d_ocp_qp_get_ubx_mask) STUB(d_ocp_qp_get_ug)
STUB(d_ocp_qp_get_ug_mask) STUB(d_ocp_qp_get_zl) STUB(d_ocp_qp_get_zu)
STUB(d_ocp_qp_seed_set_zero)
STUB(ocp_nlp_constraints_bgp_config_initialize_default)
STUB(ocp_nlp_cost_conl_config_initialize_default) STUB(ocp_nlp_cost_external_config_initialize_default) STUB(ocp_nlp_cost_nls_config_initialize_default)
STUB(ocp_nlp_ddp_config_initialize_default) STUB(ocp_nlp_sqp_config_initialize_default) STUB(ocp_nlp_sqp_wfqp_config_initialize_default)
STUB(ocp_nlp_dynamics_cont_config_initialize_default)
STUB(ocp_nlp_globalization_funnel_config_initialize_default) STUB(ocp_nlp_globalization_merit_backtracking_config_initialize_default)
STUB(ocp_nlp_globalization_merit_backtracking_ddp_needs_qp_objective_value)
STUB(ocp_nlp_globalization_merit_backtracking_find_acceptable_iterate_for_ddp)
STUB(ocp_nlp_reg_convexify_config_initialize_default) STUB(ocp_nlp_reg_glm_config_initialize_default) STUB(ocp_nlp_reg_mirror_config_initialize_default)
STUB(ocp_nlp_reg_project_config_initialize_default) STUB(ocp_nlp_reg_project_reduc_hess_config_initialize_default)
STUB(sim_erk_config_initialize_default) STUB(sim_gnsf_config_initialize_default) STUB(sim_irk_config_initialize_default)
STUB(sim_lifted_irk_config_initialize_default)

void print_ocp_qp_in(void *qp_in) { (void) qp_in; }
void print_ocp_qp_out(void *qp_out) { (void) qp_out; }
void print_ocp_qp_in_to_file(void *file, void *qp_in) { (void) file; (void) qp_in; }
void print_ocp_qp_out_to_file(void *file, void *qp_out) { (void) file; (void) qp_out; }
const char *status_to_string(int status) { static char buf[32]; snprintf(buf, sizeof(buf), "status %d", status); return buf; }
