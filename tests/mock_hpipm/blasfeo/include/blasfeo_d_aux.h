/* TEST INFRASTRUCTURE ONLY -- stand-in for BLASFEO's blasfeo/include/blasfeo_d_aux.h (giaf/blasfeo is an empty submodule in
 * /root/reference): panel-major dmat / dvec and the accessors an adapter uses, tests/mock_acados/include/blasfeo_d_aux.h */
#ifndef MOCK_BLASFEO_INCLUDE_BLASFEO_D_AUX_H_
#define MOCK_BLASFEO_INCLUDE_BLASFEO_D_AUX_H_
#include "../../../mock_acados/include/blasfeo_d_aux.h"
#endif
