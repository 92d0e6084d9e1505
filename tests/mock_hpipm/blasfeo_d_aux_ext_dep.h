/* TEST INFRASTRUCTURE ONLY -- stand-in for BLASFEO's blasfeo_d_aux_ext_dep.h as acados' utility headers include it (flat name) */
#ifndef STANDIN_FLAT_BLASFEO_D_AUX_EXT_DEP_H_
#define STANDIN_FLAT_BLASFEO_D_AUX_EXT_DEP_H_
#include "blasfeo/include/blasfeo_d_aux.h"
#include "mock_hpipm.h"
#endif
