/* TEST INFRASTRUCTURE ONLY -- stand-in for HPIPM's hpipm/include/hpipm_common.h (giaf/hpipm is an empty submodule in /root/reference):
 * the structs acados' ocp_qp layer touches, restated in tests/mock_acados/include/hpipm_d_ocp_qp.h.  With this directory and
 * /root/reference on the include path, acados/ocp_qp/ocp_qp_common.h, acados/utils/types.h ... are the reference's OWN files. */
#ifndef STANDIN_HPIPM_COMMON_H_
#define STANDIN_HPIPM_COMMON_H_
#include "../../mock_hpipm.h"


#endif
