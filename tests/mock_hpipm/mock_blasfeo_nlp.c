/*
 * TEST INFRASTRUCTURE ONLY -- the BLASFEO routines the reference's ocp_nlp layer calls (ocp_nlp_common.c, ocp_nlp_cost_ls.c,
 * ocp_nlp_dynamics_disc.c, ocp_nlp_constraints_bgh.c, ocp_nlp_qpscaling.c, ocp_nlp_reg_*.c), as plain loops over the stand-in
 * containers of tests/mock_acados/include/blasfeo_d_aux.h, so that the reference's OWN (patched) SQP-RTI loop links and RUNS around
 * this repository's plugin (tests/test_lockstep_rti.py).  giaf/blasfeo is an empty submodule in /root/reference: semantics are
 * BLASFEO's documented ones (D = beta C + alpha op(A) op(B), `l` = only the lower triangle is read / written), signatures
 * restated from the call sites.
 */
#include <math.h>

#include "mock_hpipm.h"

#define EL(s, i, j) BLASFEO_DMATEL(s, i, j)

void blasfeo_dcolex(int kmax, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dvec *sx, int xi)
{ for (int i = 0; i < kmax; i++) sx->pa[xi + i] = EL(sA, ai + i, aj); }
void blasfeo_dcolin(int kmax, struct blasfeo_dvec *sx, int xi, struct blasfeo_dmat *sA, int ai, int aj)
{ for (int i = 0; i < kmax; i++) EL(sA, ai + i, aj) = sx->pa[xi + i]; }
void blasfeo_drowin(int kmax, double alpha, struct blasfeo_dvec *sx, int xi, struct blasfeo_dmat *sA, int ai, int aj)
{ for (int j = 0; j < kmax; j++) EL(sA, ai, aj + j) = alpha * sx->pa[xi + j]; }
void blasfeo_drowex(int kmax, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dvec *sx, int xi)
{ for (int j = 0; j < kmax; j++) sx->pa[xi + j] = alpha * EL(sA, ai, aj + j); }
void blasfeo_ddiare(int kmax, double alpha, struct blasfeo_dmat *sA, int ai, int aj)
{ for (int i = 0; i < kmax; i++) EL(sA, ai + i, aj + i) += alpha; }
void blasfeo_ddiaex(int kmax, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dvec *sx, int xi)
{ for (int i = 0; i < kmax; i++) sx->pa[xi + i] = alpha * EL(sA, ai + i, aj + i); }
void blasfeo_dgead(int m, int n, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dmat *sB, int bi, int bj)
{ for (int j = 0; j < n; j++) for (int i = 0; i < m; i++) EL(sB, bi + i, bj + j) += alpha * EL(sA, ai + i, aj + j); }
void blasfeo_dgecpsc(int m, int n, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dmat *sB, int bi, int bj)
{ for (int j = 0; j < n; j++) for (int i = 0; i < m; i++) EL(sB, bi + i, bj + j) = alpha * EL(sA, ai + i, aj + j); }
void blasfeo_dgetr(int m, int n, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dmat *sB, int bi, int bj)
{ for (int j = 0; j < n; j++) for (int i = 0; i < m; i++) EL(sB, bi + j, bj + i) = EL(sA, ai + i, aj + j); }
/* lower triangle of A -> upper triangle of B (B = A', reading only the lower part; in place it mirrors the matrix) */
void blasfeo_dtrtr_l(int m, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dmat *sB, int bi, int bj)
{ for (int j = 0; j < m; j++) for (int i = j; i < m; i++) EL(sB, bi + j, bj + i) = EL(sA, ai + i, aj + j); }

void blasfeo_dgemm_nn(int m, int n, int k, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dmat *sB, int bi, int bj, double beta,
                      struct blasfeo_dmat *sC, int ci, int cj, struct blasfeo_dmat *sD, int di, int dj)
{
    for (int j = 0; j < n; j++)
        for (int i = 0; i < m; i++)
        {
            double a = 0.0;
            for (int l = 0; l < k; l++) a += EL(sA, ai + i, aj + l) * EL(sB, bi + l, bj + j);
            EL(sD, di + i, dj + j) = (beta != 0.0 ? beta * EL(sC, ci + i, cj + j) : 0.0) + alpha * a;
        }
}
void blasfeo_dgemm_nt(int m, int n, int k, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dmat *sB, int bi, int bj, double beta,
                      struct blasfeo_dmat *sC, int ci, int cj, struct blasfeo_dmat *sD, int di, int dj)
{
    for (int j = 0; j < n; j++)
        for (int i = 0; i < m; i++)
        {
            double a = 0.0;
            for (int l = 0; l < k; l++) a += EL(sA, ai + i, aj + l) * EL(sB, bi + j, bj + l);
            EL(sD, di + i, dj + j) = (beta != 0.0 ? beta * EL(sC, ci + i, cj + j) : 0.0) + alpha * a;
        }
}
void blasfeo_dgemm_tn(int m, int n, int k, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dmat *sB, int bi, int bj, double beta,
                      struct blasfeo_dmat *sC, int ci, int cj, struct blasfeo_dmat *sD, int di, int dj)
{
    for (int j = 0; j < n; j++)
        for (int i = 0; i < m; i++)
        {
            double a = 0.0;
            for (int l = 0; l < k; l++) a += EL(sA, ai + l, aj + i) * EL(sB, bi + l, bj + j);
            EL(sD, di + i, dj + j) = (beta != 0.0 ? beta * EL(sC, ci + i, cj + j) : 0.0) + alpha * a;
        }
}
/* D = beta C + alpha A diag(b) */
void blasfeo_dgemm_nd(int m, int n, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dvec *sB, int bi, double beta,
                      struct blasfeo_dmat *sC, int ci, int cj, struct blasfeo_dmat *sD, int di, int dj)
{
    for (int j = 0; j < n; j++)
        for (int i = 0; i < m; i++)
        {
            const double a = alpha * EL(sA, ai + i, aj + j) * sB->pa[bi + j];
            EL(sD, di + i, dj + j) = (beta != 0.0 ? beta * EL(sC, ci + i, cj + j) : 0.0) + a;
        }
}
/* lower triangle of D = beta C + alpha A B' (m x m result, A and B m x k) */
void blasfeo_dsyrk_ln_mn(int m, int n, int k, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dmat *sB, int bi, int bj, double beta,
                         struct blasfeo_dmat *sC, int ci, int cj, struct blasfeo_dmat *sD, int di, int dj)
{
    for (int j = 0; j < n; j++)
        for (int i = j; i < m; i++)
        {
            double a = 0.0;
            for (int l = 0; l < k; l++) a += EL(sA, ai + i, aj + l) * EL(sB, bi + j, bj + l);
            EL(sD, di + i, dj + j) = (beta != 0.0 ? beta * EL(sC, ci + i, cj + j) : 0.0) + alpha * a;
        }
}
void blasfeo_dsyrk_ln(int m, int k, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dmat *sB, int bi, int bj, double beta,
                      struct blasfeo_dmat *sC, int ci, int cj, struct blasfeo_dmat *sD, int di, int dj)
{ blasfeo_dsyrk_ln_mn(m, m, k, alpha, sA, ai, aj, sB, bi, bj, beta, sC, ci, cj, sD, di, dj); }
/* D = lower Cholesky factor of C (lower triangle of C is read) */
void blasfeo_dpotrf_l(int m, struct blasfeo_dmat *sC, int ci, int cj, struct blasfeo_dmat *sD, int di, int dj)
{
    for (int j = 0; j < m; j++)
    {
        double d = EL(sC, ci + j, cj + j);
        for (int l = 0; l < j; l++) d -= EL(sD, di + j, dj + l) * EL(sD, di + j, dj + l);
        const double r = d > 0.0 ? sqrt(d) : 0.0, inv = r > 0.0 ? 1.0 / r : 0.0;
        EL(sD, di + j, dj + j) = r;
        for (int i = j + 1; i < m; i++)
        {
            double a = EL(sC, ci + i, cj + j);
            for (int l = 0; l < j; l++) a -= EL(sD, di + i, dj + l) * EL(sD, di + j, dj + l);
            EL(sD, di + i, dj + j) = a * inv;
        }
    }
}
/* D = alpha B A, A n x n lower triangular, not transposed, non-unit */
void blasfeo_dtrmm_rlnn(int m, int n, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dmat *sB, int bi, int bj,
                        struct blasfeo_dmat *sD, int di, int dj)
{
    for (int j = 0; j < n; j++)       /* column j of the result needs columns l >= j of B: ascending j is safe in place */
        for (int i = 0; i < m; i++)
        {
            double a = 0.0;
            for (int l = j; l < n; l++) a += EL(sB, bi + i, bj + l) * EL(sA, ai + l, aj + j);
            EL(sD, di + i, dj + j) = alpha * a;
        }
}
/* z = A' x, A m x m lower triangular */
void blasfeo_dtrmv_ltn(int m, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dvec *sx, int xi, struct blasfeo_dvec *sz, int zi)
{
    for (int j = 0; j < m; j++)       /* z_j needs x_i, i >= j: ascending j is safe in place */
    {
        double a = 0.0;
        for (int i = j; i < m; i++) a += EL(sA, ai + i, aj + j) * sx->pa[xi + i];
        sz->pa[zi + j] = a;
    }
}
/* z = A x, A m x m lower triangular */
void blasfeo_dtrmv_lnn(int m, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dvec *sx, int xi, struct blasfeo_dvec *sz, int zi)
{
    for (int i = m - 1; i >= 0; i--)  /* z_i needs x_j, j <= i: descending i is safe in place */
    {
        double a = 0.0;
        for (int j = 0; j <= i; j++) a += EL(sA, ai + i, aj + j) * sx->pa[xi + j];
        sz->pa[zi + i] = a;
    }
}
/* z = beta y + alpha A x, A m x n with its leading n x n block symmetric and stored in the lower triangle (m >= n) */
void blasfeo_dsymv_l_mn(int m, int n, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dvec *sx, int xi, double beta,
                        struct blasfeo_dvec *sy, int yi, struct blasfeo_dvec *sz, int zi)
{
    double tmp[512];
    if (m > 512) m = 512;
    for (int i = 0; i < m; i++)
    {
        double a = 0.0;
        for (int j = 0; j < n; j++) a += (i >= j ? EL(sA, ai + i, aj + j) : EL(sA, ai + j, aj + i)) * sx->pa[xi + j];
        tmp[i] = (beta != 0.0 ? beta * sy->pa[yi + i] : 0.0) + alpha * a;
    }
    for (int i = 0; i < m; i++) sz->pa[zi + i] = tmp[i];
}
void blasfeo_dsymv_l(int m, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dvec *sx, int xi, double beta, struct blasfeo_dvec *sy,
                     int yi, struct blasfeo_dvec *sz, int zi)
{ blasfeo_dsymv_l_mn(m, m, alpha, sA, ai, aj, sx, xi, beta, sy, yi, sz, zi); }
void blasfeo_dvecmulacc(int m, struct blasfeo_dvec *sx, int xi, struct blasfeo_dvec *sy, int yi, struct blasfeo_dvec *sz, int zi)
{ for (int i = 0; i < m; i++) sz->pa[zi + i] += sx->pa[xi + i] * sy->pa[yi + i]; }
void blasfeo_dvecpe(int kmax, int *ipiv, struct blasfeo_dvec *sx, int xi)
{ for (int i = 0; i < kmax; i++) if (ipiv[i] != i) { const double t = sx->pa[xi + ipiv[i]]; sx->pa[xi + ipiv[i]] = sx->pa[xi + i]; sx->pa[xi + i] = t; } }
void blasfeo_dvecpei(int kmax, int *ipiv, struct blasfeo_dvec *sx, int xi)
{ for (int i = kmax - 1; i >= 0; i--) if (ipiv[i] != i) { const double t = sx->pa[xi + ipiv[i]]; sx->pa[xi + ipiv[i]] = sx->pa[xi + i]; sx->pa[xi + i] = t; } }
