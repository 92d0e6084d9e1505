/* TEST INFRASTRUCTURE ONLY -- stand-in for HPIPM's hpipm/include/hpipm_d_ocp_qp_utils.h (empty submodule in /root/reference) */
#ifndef STANDIN_HPIPM_D_OCP_QP_UTILS_H_
#define STANDIN_HPIPM_D_OCP_QP_UTILS_H_
#include "hpipm_d_ocp_qp.h"
#endif
