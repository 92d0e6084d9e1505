/* TEST INFRASTRUCTURE ONLY -- stand-in for HPIPM's hpipm/include/hpipm_d_dense_qp_res.h (empty submodule in /root/reference): nothing on the
 * OCP-QP plugin path reads these structs; the reference headers that include this file only name them through pointers */
#ifndef STANDIN_HPIPM_D_DENSE_QP_RES_H_
#define STANDIN_HPIPM_D_DENSE_QP_RES_H_
#include "hpipm_d_ocp_qp.h"
#endif
