/*
 * TEST INFRASTRUCTURE ONLY -- declarations of the HPIPM / BLASFEO FUNCTIONS the reference's acados/ocp_qp/ocp_qp_common.c,
 * ocp_qp_xcond_solver.c and acados/utils/mem.c link against (symbol list: SURVEY.md 8a), implemented in mock_hpipm.c so that
 * those reference sources can be compiled UNMODIFIED and drive this repository's plugin (tests/test_reference_orchestration.py).
 * giaf/hpipm and giaf/blasfeo are empty submodules in /root/reference: signatures are restated from the call sites.
 */
#ifndef MOCK_HPIPM_FUNCS_H_
#define MOCK_HPIPM_FUNCS_H_

#include <stddef.h>

#include "../mock_acados/include/hpipm_d_ocp_qp.h"

typedef size_t hpipm_size_t;

/* residual containers (ocp_qp_common.c:504-662 reads res_g / res_b / res_d / res_m as CONTIGUOUS vectors and `dim`) */
struct d_ocp_qp_res
{
    struct d_ocp_qp_dim *dim;
    struct blasfeo_dvec *res_g, *res_b, *res_d, *res_m;
    double res_max[4];
    hpipm_size_t memsize;
};
struct d_ocp_qp_res_ws
{
    struct blasfeo_dvec *tmp_nbgM, *tmp_nsM;
    hpipm_size_t memsize;
};

hpipm_size_t d_ocp_qp_dim_memsize(int N);
void d_ocp_qp_dim_create(int N, struct d_ocp_qp_dim *dim, void *mem);
void d_ocp_qp_dim_set(char *field, int stage, int value, struct d_ocp_qp_dim *dim);
void d_ocp_qp_dim_get(struct d_ocp_qp_dim *dim, char *field, int stage, int *value);
void d_ocp_qp_dim_copy_all(struct d_ocp_qp_dim *src, struct d_ocp_qp_dim *dst);
hpipm_size_t d_ocp_qp_memsize(struct d_ocp_qp_dim *dim);
void d_ocp_qp_create(struct d_ocp_qp_dim *dim, struct d_ocp_qp *qp, void *mem);
hpipm_size_t d_ocp_qp_sol_memsize(struct d_ocp_qp_dim *dim);
void d_ocp_qp_sol_create(struct d_ocp_qp_dim *dim, struct d_ocp_qp_sol *sol, void *mem);
void d_ocp_qp_sol_copy_all(struct d_ocp_qp_sol *src, struct d_ocp_qp_sol *dst);
hpipm_size_t d_ocp_qp_seed_memsize(struct d_ocp_qp_dim *dim);
void d_ocp_qp_seed_create(struct d_ocp_qp_dim *dim, struct d_ocp_qp_seed *seed, void *mem);
hpipm_size_t d_ocp_qp_res_memsize(struct d_ocp_qp_dim *dim);
void d_ocp_qp_res_create(struct d_ocp_qp_dim *dim, struct d_ocp_qp_res *res, void *mem);
hpipm_size_t d_ocp_qp_res_ws_memsize(struct d_ocp_qp_dim *dim);
void d_ocp_qp_res_ws_create(struct d_ocp_qp_dim *dim, struct d_ocp_qp_res_ws *ws, void *mem);
void d_ocp_qp_res_compute(struct d_ocp_qp *qp, struct d_ocp_qp_sol *sol, struct d_ocp_qp_res *res, struct d_ocp_qp_res_ws *ws);

/* BLASFEO routines beyond the inline accessors of blasfeo_d_aux.h */
hpipm_size_t blasfeo_memsize_dmat(int m, int n);
hpipm_size_t blasfeo_memsize_dvec(int m);
void blasfeo_create_dmat(int m, int n, struct blasfeo_dmat *sA, void *mem);
void blasfeo_create_dvec(int m, struct blasfeo_dvec *sa, void *mem);
void blasfeo_dvecnrm_inf(int m, struct blasfeo_dvec *sx, int xi, double *ptr_norm);
void blasfeo_daxpy(int m, double alpha, struct blasfeo_dvec *sx, int xi, struct blasfeo_dvec *sy, int yi, struct blasfeo_dvec *sz, int zi);
void blasfeo_daxpby(int m, double alpha, struct blasfeo_dvec *sx, int xi, double beta, struct blasfeo_dvec *sy, int yi, struct blasfeo_dvec *sz, int zi);
double blasfeo_ddot(int m, struct blasfeo_dvec *sx, int xi, struct blasfeo_dvec *sy, int yi);
void blasfeo_dvecsc(int m, double alpha, struct blasfeo_dvec *sx, int xi);
void blasfeo_dvecad(int m, double alpha, struct blasfeo_dvec *sx, int xi, struct blasfeo_dvec *sy, int yi);
void blasfeo_dveccp(int m, struct blasfeo_dvec *sx, int xi, struct blasfeo_dvec *sy, int yi);
void blasfeo_dgecp(int m, int n, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dmat *sB, int bi, int bj);
void blasfeo_dvecex_sp(int m, double alpha, int *idx, struct blasfeo_dvec *sx, int xi, struct blasfeo_dvec *sz, int zi);
void blasfeo_dvecad_sp(int m, double alpha, struct blasfeo_dvec *sx, int xi, int *idx, struct blasfeo_dvec *sz, int zi);
void blasfeo_dgemv_t(int m, int n, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dvec *sx, int xi, double beta,
                     struct blasfeo_dvec *sy, int yi, struct blasfeo_dvec *sz, int zi);
void blasfeo_ddiain(int kmax, double alpha, struct blasfeo_dvec *sx, int xi, struct blasfeo_dmat *sA, int ai, int aj);
void blasfeo_dgese(int m, int n, double alpha, struct blasfeo_dmat *sA, int ai, int aj);
void blasfeo_dvecmul(int m, struct blasfeo_dvec *sx, int xi, struct blasfeo_dvec *sy, int yi, struct blasfeo_dvec *sz, int zi);
void blasfeo_dveccpsc(int m, double alpha, struct blasfeo_dvec *sx, int xi, struct blasfeo_dvec *sy, int yi);
void blasfeo_dgemv_n(int m, int n, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dvec *sx, int xi, double beta,
                     struct blasfeo_dvec *sy, int yi, struct blasfeo_dvec *sz, int zi);

/* BLASFEO routines the reference's ocp_nlp layer links against (ocp_nlp_common.c, ocp_nlp_cost_ls.c, ocp_nlp_dynamics_disc.c,
 * ocp_nlp_constraints_bgh.c, ocp_nlp_qpscaling.c): implemented in mock_blasfeo_nlp.c for the lock-step RTI run of
 * tests/test_lockstep_rti.py (signatures restated from the call sites) */
void blasfeo_dcolex(int kmax, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dvec *sx, int xi);
void blasfeo_ddiare(int kmax, double alpha, struct blasfeo_dmat *sA, int ai, int aj);
void blasfeo_ddiaex(int kmax, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dvec *sx, int xi);
void blasfeo_dgead(int m, int n, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dmat *sB, int bi, int bj);
void blasfeo_dgecpsc(int m, int n, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dmat *sB, int bi, int bj);
void blasfeo_dgetr(int m, int n, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dmat *sB, int bi, int bj);
void blasfeo_dtrtr_l(int m, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dmat *sB, int bi, int bj);
void blasfeo_dgemm_nn(int m, int n, int k, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dmat *sB, int bi, int bj, double beta,
                      struct blasfeo_dmat *sC, int ci, int cj, struct blasfeo_dmat *sD, int di, int dj);
void blasfeo_dgemm_nt(int m, int n, int k, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dmat *sB, int bi, int bj, double beta,
                      struct blasfeo_dmat *sC, int ci, int cj, struct blasfeo_dmat *sD, int di, int dj);
void blasfeo_dgemm_tn(int m, int n, int k, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dmat *sB, int bi, int bj, double beta,
                      struct blasfeo_dmat *sC, int ci, int cj, struct blasfeo_dmat *sD, int di, int dj);
void blasfeo_dgemm_nd(int m, int n, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dvec *sB, int bi, double beta,
                      struct blasfeo_dmat *sC, int ci, int cj, struct blasfeo_dmat *sD, int di, int dj);
void blasfeo_dsyrk_ln(int m, int k, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dmat *sB, int bi, int bj, double beta,
                      struct blasfeo_dmat *sC, int ci, int cj, struct blasfeo_dmat *sD, int di, int dj);
void blasfeo_dsyrk_ln_mn(int m, int n, int k, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dmat *sB, int bi, int bj, double beta,
                         struct blasfeo_dmat *sC, int ci, int cj, struct blasfeo_dmat *sD, int di, int dj);
void blasfeo_dpotrf_l(int m, struct blasfeo_dmat *sC, int ci, int cj, struct blasfeo_dmat *sD, int di, int dj);
void blasfeo_dtrmm_rlnn(int m, int n, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dmat *sB, int bi, int bj,
                        struct blasfeo_dmat *sD, int di, int dj);
void blasfeo_dtrmv_ltn(int m, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dvec *sx, int xi, struct blasfeo_dvec *sz, int zi);
void blasfeo_dtrmv_lnn(int m, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dvec *sx, int xi, struct blasfeo_dvec *sz, int zi);
void blasfeo_dsymv_l(int m, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dvec *sx, int xi, double beta, struct blasfeo_dvec *sy,
                     int yi, struct blasfeo_dvec *sz, int zi);
void blasfeo_dsymv_l_mn(int m, int n, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dvec *sx, int xi, double beta,
                        struct blasfeo_dvec *sy, int yi, struct blasfeo_dvec *sz, int zi);
void blasfeo_dvecmulacc(int m, struct blasfeo_dvec *sx, int xi, struct blasfeo_dvec *sy, int yi, struct blasfeo_dvec *sz, int zi);
void blasfeo_dvecpe(int kmax, int *ipiv, struct blasfeo_dvec *sx, int xi);
void blasfeo_dvecpei(int kmax, int *ipiv, struct blasfeo_dvec *sx, int xi);
void blasfeo_drowin(int kmax, double alpha, struct blasfeo_dvec *sx, int xi, struct blasfeo_dmat *sA, int ai, int aj);
void blasfeo_drowex(int kmax, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dvec *sx, int xi);
void blasfeo_dcolin(int kmax, struct blasfeo_dvec *sx, int xi, struct blasfeo_dmat *sA, int ai, int aj);

#endif
