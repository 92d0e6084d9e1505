/* TEST INFRASTRUCTURE ONLY -- stand-in for HPIPM's hpipm/include/hpipm_d_dense_qp_sol.h (empty submodule in /root/reference): nothing on the
 * OCP-QP plugin path reads these structs; the reference headers that include this file only name them through pointers */
#ifndef STANDIN_HPIPM_D_DENSE_QP_SOL_H_
#define STANDIN_HPIPM_D_DENSE_QP_SOL_H_
#include "hpipm_d_ocp_qp.h"
#endif
