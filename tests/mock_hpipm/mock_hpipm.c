/*
 * TEST INFRASTRUCTURE ONLY -- the HPIPM / BLASFEO functions behind acados' ocp_qp containers, implemented so that the
 * reference's OWN sources acados/ocp_qp/ocp_qp_common.c, acados/ocp_qp/ocp_qp_xcond_solver.c, acados/utils/mem.c and
 * acados/utils/timing.c compile and link UNMODIFIED and drive this repository's plugin (tests/test_reference_orchestration.py).
 * Memory layout follows the conventions the reference relies on: containers carved from one caller-provided block, vectors of
 * one kind stored contiguously (ocp_qp_common.c:637-662 takes norms over res_g[0..N] as ONE vector), panel-major matrices.
 * d_ocp_qp_res_compute is the KKT residual of acados_ocp_qp.py:24-45 / ocp_qp_clarabel.c:493-683 in HPIPM's sign convention
 * (restated; the arithmetic of HPIPM itself is not available).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mock_hpipm.h"

static char *al64(char *p) { return (char *) (((uintptr_t) p + 63) & ~(uintptr_t) 63); }

/* ------------------------------------------------------------------ BLASFEO */
/* symmetric in (m, n) and zero-friendly like BLASFEO's own (pm * cn + a diagonal's worth): the reference sizes some blocks as (a, b) and
 * creates them as (b, a) with one of the two zero (ocp_nlp_cost_ls.c:174 / 212, Vz) */
hpipm_size_t blasfeo_memsize_dmat(int m, int n)
{
    const int pm = (m + MOCK_PS - 1) / MOCK_PS * MOCK_PS, cn = (n + MOCK_PS - 1) / MOCK_PS * MOCK_PS;
    return sizeof(double) * (size_t) (pm * cn + pm + cn + MOCK_PS);
}
hpipm_size_t blasfeo_memsize_dvec(int m) { return sizeof(double) * (size_t) ((m + MOCK_PS - 1) / MOCK_PS * MOCK_PS + MOCK_PS); }
void blasfeo_create_dmat(int m, int n, struct blasfeo_dmat *sA, void *mem)
{
    sA->m = m; sA->n = n;
    sA->pm = (m + MOCK_PS - 1) / MOCK_PS * MOCK_PS;
    sA->cn = (n + MOCK_PS - 1) / MOCK_PS * MOCK_PS;
    sA->memsize = (int) blasfeo_memsize_dmat(m, n);
    sA->mem = (double *) mem; sA->pA = sA->mem; sA->dA = sA->pA + sA->pm * sA->cn; sA->use_dA = 0;
}
void blasfeo_create_dvec(int m, struct blasfeo_dvec *sa, void *mem)
{
    sa->m = m; sa->pm = (m + MOCK_PS - 1) / MOCK_PS * MOCK_PS; sa->memsize = (int) blasfeo_memsize_dvec(m);
    sa->mem = (double *) mem; sa->pa = sa->mem;
}
void blasfeo_dvecnrm_inf(int m, struct blasfeo_dvec *sx, int xi, double *ptr_norm)
{
    double v = 0.0;
    for (int i = 0; i < m; i++) { const double a = sx->pa[xi + i] < 0 ? -sx->pa[xi + i] : sx->pa[xi + i]; if (a > v || a != a) v = a; }
    *ptr_norm = v;
}
void blasfeo_daxpy(int m, double alpha, struct blasfeo_dvec *sx, int xi, struct blasfeo_dvec *sy, int yi, struct blasfeo_dvec *sz, int zi)
{ for (int i = 0; i < m; i++) sz->pa[zi + i] = sy->pa[yi + i] + alpha * sx->pa[xi + i]; }
void blasfeo_daxpby(int m, double alpha, struct blasfeo_dvec *sx, int xi, double beta, struct blasfeo_dvec *sy, int yi, struct blasfeo_dvec *sz, int zi)
{ for (int i = 0; i < m; i++) sz->pa[zi + i] = beta * sy->pa[yi + i] + alpha * sx->pa[xi + i]; }
double blasfeo_ddot(int m, struct blasfeo_dvec *sx, int xi, struct blasfeo_dvec *sy, int yi)
{ double a = 0.0; for (int i = 0; i < m; i++) a += sx->pa[xi + i] * sy->pa[yi + i]; return a; }
void blasfeo_dvecsc(int m, double alpha, struct blasfeo_dvec *sx, int xi) { for (int i = 0; i < m; i++) sx->pa[xi + i] *= alpha; }
void blasfeo_dvecad(int m, double alpha, struct blasfeo_dvec *sx, int xi, struct blasfeo_dvec *sy, int yi)
{ for (int i = 0; i < m; i++) sy->pa[yi + i] += alpha * sx->pa[xi + i]; }
void blasfeo_dveccp(int m, struct blasfeo_dvec *sx, int xi, struct blasfeo_dvec *sy, int yi) { for (int i = 0; i < m; i++) sy->pa[yi + i] = sx->pa[xi + i]; }
void blasfeo_dveccpsc(int m, double alpha, struct blasfeo_dvec *sx, int xi, struct blasfeo_dvec *sy, int yi)
{ for (int i = 0; i < m; i++) sy->pa[yi + i] = alpha * sx->pa[xi + i]; }
void blasfeo_dvecmul(int m, struct blasfeo_dvec *sx, int xi, struct blasfeo_dvec *sy, int yi, struct blasfeo_dvec *sz, int zi)
{ for (int i = 0; i < m; i++) sz->pa[zi + i] = sx->pa[xi + i] * sy->pa[yi + i]; }
void blasfeo_dgecp(int m, int n, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dmat *sB, int bi, int bj)
{ for (int j = 0; j < n; j++) for (int i = 0; i < m; i++) BLASFEO_DMATEL(sB, bi + i, bj + j) = BLASFEO_DMATEL(sA, ai + i, aj + j); }
void blasfeo_dgese(int m, int n, double alpha, struct blasfeo_dmat *sA, int ai, int aj)
{ for (int j = 0; j < n; j++) for (int i = 0; i < m; i++) BLASFEO_DMATEL(sA, ai + i, aj + j) = alpha; }
void blasfeo_ddiain(int kmax, double alpha, struct blasfeo_dvec *sx, int xi, struct blasfeo_dmat *sA, int ai, int aj)
{ for (int i = 0; i < kmax; i++) BLASFEO_DMATEL(sA, ai + i, aj + i) = alpha * sx->pa[xi + i]; }
void blasfeo_dvecex_sp(int m, double alpha, int *idx, struct blasfeo_dvec *sx, int xi, struct blasfeo_dvec *sz, int zi)
{ for (int i = 0; i < m; i++) sz->pa[zi + i] = alpha * sx->pa[xi + idx[i]]; }
void blasfeo_dvecad_sp(int m, double alpha, struct blasfeo_dvec *sx, int xi, int *idx, struct blasfeo_dvec *sz, int zi)
{ for (int i = 0; i < m; i++) sz->pa[zi + idx[i]] += alpha * sx->pa[xi + i]; }
/* z = beta y + alpha A' x, A the m x n sub-block at (ai, aj) */
void blasfeo_dgemv_t(int m, int n, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dvec *sx, int xi, double beta,
                     struct blasfeo_dvec *sy, int yi, struct blasfeo_dvec *sz, int zi)
{
    for (int j = 0; j < n; j++)
    {
        double a = 0.0;
        for (int i = 0; i < m; i++) a += BLASFEO_DMATEL(sA, ai + i, aj + j) * sx->pa[xi + i];
        sz->pa[zi + j] = beta * sy->pa[yi + j] + alpha * a;
    }
}
void blasfeo_dgemv_n(int m, int n, double alpha, struct blasfeo_dmat *sA, int ai, int aj, struct blasfeo_dvec *sx, int xi, double beta,
                     struct blasfeo_dvec *sy, int yi, struct blasfeo_dvec *sz, int zi)
{
    for (int i = 0; i < m; i++)
    {
        double a = 0.0;
        for (int j = 0; j < n; j++) a += BLASFEO_DMATEL(sA, ai + i, aj + j) * sx->pa[xi + j];
        sz->pa[zi + i] = beta * sy->pa[yi + i] + alpha * a;
    }
}

/* ------------------------------------------------------------------ d_ocp_qp_dim */
hpipm_size_t d_ocp_qp_dim_memsize(int N) { return 10 * sizeof(int) * (size_t) (N + 1) + 64; }
void d_ocp_qp_dim_create(int N, struct d_ocp_qp_dim *dim, void *mem)
{
    memset(mem, 0, d_ocp_qp_dim_memsize(N));
    int *p = (int *) mem;
    int **f[10] = {&dim->nx, &dim->nu, &dim->nb, &dim->nbx, &dim->nbu, &dim->ng, &dim->ns, &dim->nbxe, &dim->nbue, &dim->nge};
    for (int q = 0; q < 10; q++) { *f[q] = p; p += N + 1; }
    dim->N = N;
}
static int *dim_field(struct d_ocp_qp_dim *dim, const char *field)
{
    if (!strcmp(field, "nx")) return dim->nx;
    if (!strcmp(field, "nu")) return dim->nu;
    if (!strcmp(field, "nb")) return dim->nb;
    if (!strcmp(field, "nbx")) return dim->nbx;
    if (!strcmp(field, "nbu")) return dim->nbu;
    if (!strcmp(field, "ng")) return dim->ng;
    if (!strcmp(field, "ns")) return dim->ns;
    if (!strcmp(field, "nbxe")) return dim->nbxe;
    if (!strcmp(field, "nbue")) return dim->nbue;
    if (!strcmp(field, "nge")) return dim->nge;
    if (!strcmp(field, "nsbx") || !strcmp(field, "nsbu") || !strcmp(field, "nsg")) return NULL; /* not kept by this stand-in */
    printf("mock d_ocp_qp_dim: unknown field %s\n", field);
    exit(1);
}
void d_ocp_qp_dim_set(char *field, int stage, int value, struct d_ocp_qp_dim *dim)
{
    int *a = dim_field(dim, field);
    if (!a) return;
    a[stage] = value;
    if (!strcmp(field, "nbx") || !strcmp(field, "nbu")) dim->nb[stage] = dim->nbx[stage] + dim->nbu[stage];
}
void d_ocp_qp_dim_get(struct d_ocp_qp_dim *dim, char *field, int stage, int *value)
{
    int *a = dim_field(dim, field);
    *value = a ? a[stage] : 0;
}
void d_ocp_qp_dim_copy_all(struct d_ocp_qp_dim *src, struct d_ocp_qp_dim *dst)
{
    int *s[10] = {src->nx, src->nu, src->nb, src->nbx, src->nbu, src->ng, src->ns, src->nbxe, src->nbue, src->nge};
    int *d[10] = {dst->nx, dst->nu, dst->nb, dst->nbx, dst->nbu, dst->ng, dst->ns, dst->nbxe, dst->nbue, dst->nge};
    for (int q = 0; q < 10; q++) memcpy(d[q], s[q], sizeof(int) * (size_t) (src->N + 1));
    dst->N = src->N;
}

/* ------------------------------------------------------------------ containers: struct arrays first, data after, each kind contiguous */
static int nct_of(const struct d_ocp_qp_dim *d, int k) { return 2 * (d->nb[k] + d->ng[k] + d->ns[k]); }
static int nx1_of(const struct d_ocp_qp_dim *d, int k) { return k < d->N ? d->nx[k + 1] : 0; }

hpipm_size_t d_ocp_qp_memsize(struct d_ocp_qp_dim *dim)
{
    const int N = dim->N;
    hpipm_size_t s = 3 * sizeof(struct blasfeo_dmat) * (N + 1) + 6 * sizeof(struct blasfeo_dvec) * (N + 1) + 3 * sizeof(int *) * (N + 1) + sizeof(int) * (N + 1);
    for (int k = 0; k <= N; k++)
    {
        const int nu = dim->nu[k], nx = dim->nx[k], nb = dim->nb[k], ng = dim->ng[k], ns = dim->ns[k], nct = nct_of(dim, k);
        s += blasfeo_memsize_dmat(nu + nx + 1, nx1_of(dim, k)) + blasfeo_memsize_dmat(nu + nx + 1, nu + nx) + blasfeo_memsize_dmat(nu + nx, ng);
        s += blasfeo_memsize_dvec(nx1_of(dim, k)) + blasfeo_memsize_dvec(nu + nx + 2 * ns) + 3 * blasfeo_memsize_dvec(nct) + blasfeo_memsize_dvec(2 * ns);
        s += sizeof(int) * (size_t) (2 * nb + ng + nb + ng + 4);
    }
    return s + 3 * 64 + 64;
}
void d_ocp_qp_create(struct d_ocp_qp_dim *dim, struct d_ocp_qp *qp, void *mem)
{
    const int N = dim->N;
    memset(mem, 0, d_ocp_qp_memsize(dim));
    char *c = (char *) mem;
    qp->dim = dim;
    qp->BAbt = (struct blasfeo_dmat *) c; c += sizeof(struct blasfeo_dmat) * (N + 1);
    qp->RSQrq = (struct blasfeo_dmat *) c; c += sizeof(struct blasfeo_dmat) * (N + 1);
    qp->DCt = (struct blasfeo_dmat *) c; c += sizeof(struct blasfeo_dmat) * (N + 1);
    struct blasfeo_dvec **v[6] = {&qp->b, &qp->rqz, &qp->d, &qp->d_mask, &qp->m, &qp->Z};
    for (int q = 0; q < 6; q++) { *v[q] = (struct blasfeo_dvec *) c; c += sizeof(struct blasfeo_dvec) * (N + 1); }
    qp->idxb = (int **) c; c += sizeof(int *) * (N + 1);
    qp->idxs_rev = (int **) c; c += sizeof(int *) * (N + 1);
    qp->idxe = (int **) c; c += sizeof(int *) * (N + 1);
    qp->diag_H_flag = (int *) c; c += sizeof(int) * (N + 1);
    for (int k = 0; k <= N; k++)
    {
        const int nb = dim->nb[k], ng = dim->ng[k];
        qp->idxb[k] = (int *) c; c += sizeof(int) * (nb + 1);
        qp->idxs_rev[k] = (int *) c; c += sizeof(int) * (nb + ng + 1);
        qp->idxe[k] = (int *) c; c += sizeof(int) * (nb + ng + 1);
        for (int e = 0; e < nb + ng; e++) qp->idxs_rev[k][e] = -1;
    }
    c = al64(c);
    for (int k = 0; k <= N; k++)
    {
        const int nu = dim->nu[k], nx = dim->nx[k], ng = dim->ng[k];
        blasfeo_create_dmat(nu + nx + 1, nx1_of(dim, k), qp->BAbt + k, c); c += qp->BAbt[k].memsize;
        blasfeo_create_dmat(nu + nx + 1, nu + nx, qp->RSQrq + k, c); c += qp->RSQrq[k].memsize;
        blasfeo_create_dmat(nu + nx, ng, qp->DCt + k, c); c += qp->DCt[k].memsize;
    }
    /* every kind of vector contiguous over the stages, like HPIPM */
    for (int q = 0; q < 6; q++)
    {
        c = al64(c);
        for (int k = 0; k <= N; k++)
        {
            const int nu = dim->nu[k], nx = dim->nx[k], ns = dim->ns[k];
            const int len = q == 0 ? nx1_of(dim, k) : q == 1 ? nu + nx + 2 * ns : q == 5 ? 2 * ns : nct_of(dim, k);
            blasfeo_create_dvec(len, (*v[q]) + k, c);
            c += sizeof(double) * (size_t) len; /* unpadded: contiguous */
        }
        c += 64;
    }
    for (int k = 0; k <= N; k++) blasfeo_dvecse(nct_of(dim, k), 1.0, qp->d_mask + k, 0);
}

static hpipm_size_t vec_kind_bytes(struct d_ocp_qp_dim *dim, int kind)
{
    hpipm_size_t s = 0;
    for (int k = 0; k <= dim->N; k++)
        s += sizeof(double) * (size_t) (kind == 0 ? dim->nu[k] + dim->nx[k] + 2 * dim->ns[k] : kind == 1 ? nx1_of(dim, k) : nct_of(dim, k));
    return s + 128;
}
static char *carve_kind(struct d_ocp_qp_dim *dim, int kind, struct blasfeo_dvec *arr, char *c)
{
    c = al64(c);
    for (int k = 0; k <= dim->N; k++)
    {
        const int len = kind == 0 ? dim->nu[k] + dim->nx[k] + 2 * dim->ns[k] : kind == 1 ? nx1_of(dim, k) : nct_of(dim, k);
        blasfeo_create_dvec(len, arr + k, c);
        c += sizeof(double) * (size_t) len;
    }
    return c + 64;
}

hpipm_size_t d_ocp_qp_sol_memsize(struct d_ocp_qp_dim *dim)
{ return 4 * sizeof(struct blasfeo_dvec) * (dim->N + 1) + vec_kind_bytes(dim, 0) + vec_kind_bytes(dim, 1) + 2 * vec_kind_bytes(dim, 2) + 64; }
void d_ocp_qp_sol_create(struct d_ocp_qp_dim *dim, struct d_ocp_qp_sol *sol, void *mem)
{
    memset(mem, 0, d_ocp_qp_sol_memsize(dim));
    char *c = (char *) mem;
    sol->dim = dim;
    sol->ux = (struct blasfeo_dvec *) c; c += sizeof(struct blasfeo_dvec) * (dim->N + 1);
    sol->pi = (struct blasfeo_dvec *) c; c += sizeof(struct blasfeo_dvec) * (dim->N + 1);
    sol->lam = (struct blasfeo_dvec *) c; c += sizeof(struct blasfeo_dvec) * (dim->N + 1);
    sol->t = (struct blasfeo_dvec *) c; c += sizeof(struct blasfeo_dvec) * (dim->N + 1);
    c = carve_kind(dim, 0, sol->ux, c); c = carve_kind(dim, 1, sol->pi, c); c = carve_kind(dim, 2, sol->lam, c); c = carve_kind(dim, 2, sol->t, c);
}
void d_ocp_qp_sol_copy_all(struct d_ocp_qp_sol *src, struct d_ocp_qp_sol *dst)
{
    struct d_ocp_qp_dim *d = src->dim;
    for (int k = 0; k <= d->N; k++)
    {
        blasfeo_dveccp(d->nu[k] + d->nx[k] + 2 * d->ns[k], src->ux + k, 0, dst->ux + k, 0);
        blasfeo_dveccp(nx1_of(d, k), src->pi + k, 0, dst->pi + k, 0);
        blasfeo_dveccp(nct_of(d, k), src->lam + k, 0, dst->lam + k, 0);
        blasfeo_dveccp(nct_of(d, k), src->t + k, 0, dst->t + k, 0);
    }
}
hpipm_size_t d_ocp_qp_seed_memsize(struct d_ocp_qp_dim *dim)
{ return 4 * sizeof(struct blasfeo_dvec) * (dim->N + 1) + vec_kind_bytes(dim, 0) + vec_kind_bytes(dim, 1) + 2 * vec_kind_bytes(dim, 2) + 64; }
void d_ocp_qp_seed_create(struct d_ocp_qp_dim *dim, struct d_ocp_qp_seed *seed, void *mem)
{
    memset(mem, 0, d_ocp_qp_seed_memsize(dim));
    char *c = (char *) mem;
    seed->dim = dim;
    seed->seed_g = (struct blasfeo_dvec *) c; c += sizeof(struct blasfeo_dvec) * (dim->N + 1);
    seed->seed_b = (struct blasfeo_dvec *) c; c += sizeof(struct blasfeo_dvec) * (dim->N + 1);
    seed->seed_d = (struct blasfeo_dvec *) c; c += sizeof(struct blasfeo_dvec) * (dim->N + 1);
    seed->seed_m = (struct blasfeo_dvec *) c; c += sizeof(struct blasfeo_dvec) * (dim->N + 1);
    c = carve_kind(dim, 0, seed->seed_g, c); c = carve_kind(dim, 1, seed->seed_b, c); c = carve_kind(dim, 2, seed->seed_d, c); c = carve_kind(dim, 2, seed->seed_m, c);
}
hpipm_size_t d_ocp_qp_res_memsize(struct d_ocp_qp_dim *dim)
{ return 4 * sizeof(struct blasfeo_dvec) * (dim->N + 1) + vec_kind_bytes(dim, 0) + vec_kind_bytes(dim, 1) + 2 * vec_kind_bytes(dim, 2) + 64; }
void d_ocp_qp_res_create(struct d_ocp_qp_dim *dim, struct d_ocp_qp_res *res, void *mem)
{
    memset(mem, 0, d_ocp_qp_res_memsize(dim));
    char *c = (char *) mem;
    res->dim = dim;
    res->res_g = (struct blasfeo_dvec *) c; c += sizeof(struct blasfeo_dvec) * (dim->N + 1);
    res->res_b = (struct blasfeo_dvec *) c; c += sizeof(struct blasfeo_dvec) * (dim->N + 1);
    res->res_d = (struct blasfeo_dvec *) c; c += sizeof(struct blasfeo_dvec) * (dim->N + 1);
    res->res_m = (struct blasfeo_dvec *) c; c += sizeof(struct blasfeo_dvec) * (dim->N + 1);
    c = carve_kind(dim, 0, res->res_g, c); c = carve_kind(dim, 1, res->res_b, c); c = carve_kind(dim, 2, res->res_d, c); c = carve_kind(dim, 2, res->res_m, c);
    res->memsize = d_ocp_qp_res_memsize(dim);
}
hpipm_size_t d_ocp_qp_res_ws_memsize(struct d_ocp_qp_dim *dim) { return 64; }
void d_ocp_qp_res_ws_create(struct d_ocp_qp_dim *dim, struct d_ocp_qp_res_ws *ws, void *mem) { memset(ws, 0, sizeof(*ws)); ws->memsize = 64; }

/* KKT residuals of (qp, sol) with t taken from sol (acados recomputes it first, ocp_qp_common.c:562-566), masked sides dropped:
 *   res_g = H v + g + [B A]' pi_k - [0; pi_{k-1}] - J'(lam_l - lam_u) ; slack rows Z s + z - lam_s - lam of the rows that use the slack
 *   res_b = A x + B u + b - x+ ;  res_d = d - (J v + s) + t per side (upper sides with HPIPM's negated bound) ;  res_m = lam t */
void d_ocp_qp_res_compute(struct d_ocp_qp *qp, struct d_ocp_qp_sol *sol, struct d_ocp_qp_res *res, struct d_ocp_qp_res_ws *ws)
{
    struct d_ocp_qp_dim *dm = qp->dim;
    const int N = dm->N;
    for (int k = 0; k <= N; k++)
    {
        const int nu = dm->nu[k], nx = dm->nx[k], nv = nu + nx, nb = dm->nb[k], ng = dm->ng[k], ns = dm->ns[k], nbg = nb + ng, nx1 = nx1_of(dm, k);
        double *rg = res->res_g[k].pa, *rd = res->res_d[k].pa, *rm = res->res_m[k].pa;
        const double *ux = sol->ux[k].pa, *lam = sol->lam[k].pa, *t = sol->t[k].pa, *d = qp->d[k].pa, *mk = qp->d_mask[k].pa;
        for (int i = 0; i < nv; i++)
        {
            double a = qp->rqz[k].pa[i];
            for (int j = 0; j < nv; j++) a += (i >= j ? BLASFEO_DMATEL(qp->RSQrq + k, i, j) : BLASFEO_DMATEL(qp->RSQrq + k, j, i)) * ux[j];
            for (int j = 0; j < nx1; j++) a += BLASFEO_DMATEL(qp->BAbt + k, i, j) * sol->pi[k].pa[j];
            if (k > 0 && i >= nu) a -= sol->pi[k - 1].pa[i - nu];
            rg[i] = a;
        }
        for (int j = 0; j < 2 * ns; j++) rg[nv + j] = qp->Z[k].pa[j] * ux[nv + j] + qp->rqz[k].pa[nv + j] - lam[2 * nbg + j] * mk[2 * nbg + j];
        for (int r = 0; r < nbg; r++)
        {
            const double ll = lam[r] * mk[r], lu = lam[nbg + r] * mk[nbg + r];
            double c = 0.0;
            if (r < nb) { c = ux[qp->idxb[k][r]]; rg[qp->idxb[k][r]] -= ll - lu; }
            else
                for (int i = 0; i < nv; i++) { const double a = BLASFEO_DMATEL(qp->DCt + k, i, r - nb); c += a * ux[i]; rg[i] -= a * (ll - lu); }
            const int sj = qp->idxs_rev[k][r];
            const double sl = sj >= 0 ? ux[nv + sj] : 0.0, su = sj >= 0 ? ux[nv + ns + sj] : 0.0;
            if (sj >= 0) { rg[nv + sj] -= ll; rg[nv + ns + sj] -= lu; }
            rd[r] = mk[r] != 0.0 ? d[r] - (c + sl) + t[r] : 0.0;                       /* lower: c + sl - t = lb      */
            rd[nbg + r] = mk[nbg + r] != 0.0 ? d[nbg + r] + (c - su) + t[nbg + r] : 0.0; /* upper: d = -ub: -ub + c - su + t */
            rm[r] = mk[r] != 0.0 ? lam[r] * t[r] : 0.0;
            rm[nbg + r] = mk[nbg + r] != 0.0 ? lam[nbg + r] * t[nbg + r] : 0.0;
        }
        for (int j = 0; j < 2 * ns; j++)
        {
            const int e = 2 * nbg + j;
            rd[e] = mk[e] != 0.0 ? d[e] - ux[nv + j] + t[e] : 0.0;
            rm[e] = mk[e] != 0.0 ? lam[e] * t[e] : 0.0;
        }
        for (int j = 0; j < nx1; j++)
        {
            double a = qp->b[k].pa[j] - sol->ux[k + 1].pa[dm->nu[k + 1] + j];
            for (int i = 0; i < nv; i++) a += BLASFEO_DMATEL(qp->BAbt + k, i, j) * ux[i];
            res->res_b[k].pa[j] = a;
        }
    }
}
