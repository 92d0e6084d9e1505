/* TEST INFRASTRUCTURE ONLY -- stand-in for BLASFEO's blasfeo/include/blasfeo_d_aux_ext_dep.h (giaf/blasfeo is an empty submodule in
 * /root/reference): blasfeo_allocate_dmat / _dvec are inline in the stand-in blasfeo_d_aux.h */
#ifndef STANDIN_BLASFEO_INCLUDE_BLASFEO_D_AUX_EXT_DEP_H_
#define STANDIN_BLASFEO_INCLUDE_BLASFEO_D_AUX_EXT_DEP_H_
#include "blasfeo/include/blasfeo_d_aux.h"
#endif
