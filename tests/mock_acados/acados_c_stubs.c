/*
 * TEST INFRASTRUCTURE ONLY -- symbols the reference's interfaces/acados_c/ocp_qp_interface.c references beside the path under test:
 * the initialisers of the reference's other QP solvers / condensing modules (HPIPM-backed: absent here) and the HPIPM string-keyed
 * setters / getters of ocp_qp_in_set / ocp_qp_out_get.  None of them is on the path of PARTIAL_CONDENSING_GPU_IPM: they abort.
 */
#include <stdio.h>
#include <stdlib.h>

#define STUB(name) void name(void) { printf("acados_c_stubs: %s is not on the path under test\n", #name); exit(3); }
STUB(ocp_qp_hpipm_config_initialize_default)
STUB(dense_qp_hpipm_config_initialize_default)
STUB(ocp_qp_partial_condensing_config_initialize_default)
STUB(ocp_qp_full_condensing_config_initialize_default)
STUB(dense_qp_dims_get)
STUB(d_ocp_qp_set)
STUB(d_ocp_qp_sol_get_x)
STUB(d_ocp_qp_sol_get_u)
STUB(d_ocp_qp_sol_get_pi)
STUB(d_ocp_qp_sol_get_sl)
STUB(d_ocp_qp_sol_get_su)
