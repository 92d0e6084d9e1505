/* TEST INFRASTRUCTURE ONLY -- stand-in for BLASFEO's blasfeo_i_aux_ext_dep.h as acados/utils/print.c includes it (flat name) */
#ifndef STANDIN_FLAT_BLASFEO_I_AUX_EXT_DEP_H_
#define STANDIN_FLAT_BLASFEO_I_AUX_EXT_DEP_H_
#include "mock_hpipm.h"
#endif
