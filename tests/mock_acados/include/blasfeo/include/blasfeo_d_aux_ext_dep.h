/* TEST INFRASTRUCTURE ONLY -- stand-in for BLASFEO's blasfeo/include/blasfeo_d_aux_ext_dep.h (blasfeo_allocate_dmat / _dvec and their
 * free counterparts: inline in the stand-in blasfeo_d_aux.h) */
#ifndef MOCK_BLASFEO_INCLUDE_BLASFEO_D_AUX_EXT_DEP_H_
#define MOCK_BLASFEO_INCLUDE_BLASFEO_D_AUX_EXT_DEP_H_
#include "blasfeo_d_aux.h"
#endif
