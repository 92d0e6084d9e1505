/* TEST INFRASTRUCTURE ONLY -- stand-in for HPIPM's hpipm/include/hpipm_d_ocp_qp_kkt.h (empty submodule in /root/reference) */
#ifndef STANDIN_HPIPM_D_OCP_QP_KKT_H_
#define STANDIN_HPIPM_D_OCP_QP_KKT_H_
#include "hpipm_d_ocp_qp.h"
#endif
