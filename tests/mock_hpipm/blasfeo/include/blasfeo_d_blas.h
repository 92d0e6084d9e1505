/* TEST INFRASTRUCTURE ONLY -- stand-in for BLASFEO's blasfeo/include/blasfeo_d_blas.h as interfaces/acados_c/ocp_nlp_interface.c includes it
 * (giaf/blasfeo is an empty submodule in /root/reference): tests/mock_hpipm/mock_hpipm.h */
#ifndef STANDIN_BLASFEO_INCLUDE_BLASFEO_D_BLAS_H_
#define STANDIN_BLASFEO_INCLUDE_BLASFEO_D_BLAS_H_
#include "blasfeo/include/blasfeo_d_aux.h"
#include "mock_hpipm.h"
#endif
