/*
 * ocp_qp_oracle.c -- CPU oracle (plain C, FP64) for the batched OCP-QP hot path.
 * TEST INFRASTRUCTURE ONLY -- see ocp_qp_oracle.h for scope, citations and the
 * parity-pinning statement.
 *
 * Problem (acados_ocp_qp.py:24-45), per stage k = 0..N, v=[u;x], slacks sl,su:
 *   min  sum 1/2 v'[R S;S' Q]v + [r;q]'v + 1/2 sl'Zl sl + zl'sl + 1/2 su'Zu su + zu'su
 *   s.t. x+ = A x + B u + b
 *        lb <= v[idxb] + sl[idxs_rev]        v[idxb] - su[idxs_rev] <= ub
 *        lg <= C x + D u + sl[idxs_rev]      C x + D u - su[idxs_rev] <= ug
 *        sl >= lls, su >= lus ;  masks switch single sides off ; idxe marks
 *        box rows that are equalities (variable fixed; acados always eliminates
 *        them before the IPM: ocp_qp_partial_condensing.c:542).
 * lam/t ordering per stage: [lb(nb) lg(ng) ub(nb) ug(ng) ls(ns) us(ns)]
 * (ocp_qp_common.c:874-921, print.c:391-405).
 */
#include "ocp_qp_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define STAT_M 20

typedef struct
{
    int nx, nu, nx1, nb, nbu, nbx, ng, ns, nbxe;
    int n, nbg, nct;
    /* data (column-major matrices) */
    double *A, *B, *b, *Q, *S, *R, *q, *r;
    int *idxb;
    double *lb, *ub, *lb_mask, *ub_mask;
    double *C, *D, *lg, *ug, *lg_mask, *ug_mask;
    double *Zl, *Zu, *zl, *zu, *lls, *lus, *lls_mask, *lus_mask;
    int *idxs_rev, *idxe;
    /* solution */
    double *ux, *pi, *lam, *t;
    /* work */
    int *act;      /* nct: 1 if the inequality row takes part in the IPM */
    int *fixed;    /* n: 1 if variable fixed by an equality-flagged bound */
    double *fixval;
    double *rg, *rb, *rd, *rm;
    double *dux, *dpi, *dlam, *dt;
    double *Gam, *rho;       /* nct */
    double *Dl, *Du, *rsl, *rsu; /* ns */
    double *gme, *El, *Eu, *Xl, *Xu; /* nbg: effective Gamma, exclusive sums of the slack elimination */
    double *tmp2;
    double *L;               /* n x n col-major lower Cholesky factor */
    double *l;               /* n: L^{-1} m */
    double *W;               /* n x nx1 col-major: [B A]' * Lx+ */
    double *gt;              /* n: condensed gradient */
    double *c, *dc;          /* nbg */
    double *rmc;             /* nct: corrected complementarity rhs */
    double *tmp;             /* max(n, nx1) scratch */
} stg;

struct oqp
{
    int N;
    stg *s;
    int iter;
    int status;
    double *stat;
    int stat_rows;
    int n_act;
};

/* ---------------------------------------------------------------- utils */

static double *dz(int n) { return (double *) calloc(n > 0 ? n : 1, sizeof(double)); }
static int *iz(int n) { return (int *) calloc(n > 0 ? n : 1, sizeof(int)); }

void oqp_opts_default(oqp_opts *o)
{
    /* mode BALANCE + acados overrides, ocp_qp_hpipm.c:101-113 */
    o->mu0 = 1e0;
    o->tol_stat = 1e-6;
    o->tol_eq = 1e-8;
    o->tol_ineq = 1e-8;
    o->tol_comp = 1e-8;
    o->alpha_min = 1e-8;
    o->tau_min = 0.0;
    o->lam_min = 1e-16;
    o->t_min = 1e-16;
    o->reg_prim = 1e-15;
    o->iter_max = 50;
    o->pred_corr = 1;
    o->cond_pred_corr = 1;
    o->warm_start = 0;
    o->print_level = 0;
    o->t0_init = 2;
}

oqp *oqp_create(int N, const int *nx, const int *nu, const int *nbx, const int *nbu,
                const int *ng, const int *ns)
{
    oqp *qp = (oqp *) calloc(1, sizeof(oqp));
    qp->N = N;
    qp->s = (stg *) calloc(N + 1, sizeof(stg));
    for (int k = 0; k <= N; k++)
    {
        stg *s = qp->s + k;
        s->nx = nx[k]; s->nu = nu[k]; s->nx1 = k < N ? nx[k + 1] : 0;
        s->nbx = nbx[k]; s->nbu = nbu[k]; s->nb = nbx[k] + nbu[k];
        s->ng = ng[k]; s->ns = ns[k]; s->nbxe = 0;
        int n = s->n = s->nu + s->nx, nbg = s->nbg = s->nb + s->ng;
        int nct = s->nct = 2 * nbg + 2 * s->ns, nss = s->ns, nx1 = s->nx1;
        s->A = dz(nx1 * s->nx); s->B = dz(nx1 * s->nu); s->b = dz(nx1);
        s->Q = dz(s->nx * s->nx); s->S = dz(s->nu * s->nx); s->R = dz(s->nu * s->nu);
        s->q = dz(s->nx); s->r = dz(s->nu);
        s->idxb = iz(s->nb);
        for (int i = 0; i < s->nbu; i++) s->idxb[i] = i;
        for (int i = 0; i < s->nbx; i++) s->idxb[s->nbu + i] = s->nu + i;
        s->lb = dz(s->nb); s->ub = dz(s->nb); s->lb_mask = dz(s->nb); s->ub_mask = dz(s->nb);
        s->C = dz(s->ng * s->nx); s->D = dz(s->ng * s->nu); s->lg = dz(s->ng); s->ug = dz(s->ng);
        s->lg_mask = dz(s->ng); s->ug_mask = dz(s->ng);
        s->Zl = dz(nss); s->Zu = dz(nss); s->zl = dz(nss); s->zu = dz(nss);
        s->lls = dz(nss); s->lus = dz(nss); s->lls_mask = dz(nss); s->lus_mask = dz(nss);
        for (int i = 0; i < s->nb; i++) s->lb_mask[i] = s->ub_mask[i] = 1.0;
        for (int i = 0; i < s->ng; i++) s->lg_mask[i] = s->ug_mask[i] = 1.0;
        for (int i = 0; i < nss; i++) s->lls_mask[i] = s->lus_mask[i] = 1.0;
        s->idxs_rev = iz(nbg);
        for (int i = 0; i < nbg; i++) s->idxs_rev[i] = -1;
        s->idxe = iz(s->nb);
        s->ux = dz(n + 2 * nss); s->pi = dz(nx1); s->lam = dz(nct); s->t = dz(nct);
        s->act = iz(nct); s->fixed = iz(n); s->fixval = dz(n);
        s->rg = dz(n + 2 * nss); s->rb = dz(nx1); s->rd = dz(nct); s->rm = dz(nct);
        s->dux = dz(n + 2 * nss); s->dpi = dz(nx1); s->dlam = dz(nct); s->dt = dz(nct);
        s->Gam = dz(nct); s->rho = dz(nct);
        s->Dl = dz(nss); s->Du = dz(nss); s->rsl = dz(nss); s->rsu = dz(nss);
        s->gme = dz(nbg); s->El = dz(nbg); s->Eu = dz(nbg); s->Xl = dz(nbg); s->Xu = dz(nbg);
        s->tmp2 = dz(n + 1);
        s->L = dz(n * n); s->l = dz(n); s->W = dz(n * nx1); s->gt = dz(n);
        s->c = dz(nbg); s->dc = dz(nbg); s->rmc = dz(nct);
        s->tmp = dz((n > nx1 ? n : nx1) + 1);
    }
    return qp;
}

void oqp_free(oqp *qp)
{
    if (!qp) return;
    for (int k = 0; k <= qp->N; k++)
    {
        stg *s = qp->s + k;
        void *p[] = {s->A, s->B, s->b, s->Q, s->S, s->R, s->q, s->r, s->idxb, s->lb, s->ub,
                     s->lb_mask, s->ub_mask, s->C, s->D, s->lg, s->ug, s->lg_mask, s->ug_mask,
                     s->Zl, s->Zu, s->zl, s->zu, s->lls, s->lus, s->lls_mask, s->lus_mask,
                     s->idxs_rev, s->idxe, s->ux, s->pi, s->lam, s->t, s->act, s->fixed,
                     s->fixval, s->rg, s->rb, s->rd, s->rm, s->dux, s->dpi, s->dlam, s->dt,
                     s->Gam, s->rho, s->Dl, s->Du, s->rsl, s->rsu, s->gme, s->El, s->Eu, s->Xl, s->Xu, s->tmp2, s->L, s->l,
                     s->W, s->gt, s->c, s->dc, s->tmp, s->rmc};
        for (unsigned i = 0; i < sizeof(p) / sizeof(p[0]); i++) free(p[i]);
    }
    free(qp->s);
    free(qp->stat);
    free(qp);
}

void oqp_set_nbxe(oqp *qp, int stage, int nbxe) { qp->s[stage].nbxe = nbxe; }

#define CPD(dst, cnt) do { memcpy((dst), value, sizeof(double) * (size_t)(cnt)); return 0; } while (0)
#define CPI(dst, cnt) do { memcpy((dst), value, sizeof(int) * (size_t)(cnt)); return 0; } while (0)

int oqp_set(oqp *qp, const char *f, int k, const void *value)
{
    stg *s = qp->s + k;
    if (!strcmp(f, "A")) CPD(s->A, s->nx1 * s->nx);
    if (!strcmp(f, "B")) CPD(s->B, s->nx1 * s->nu);
    if (!strcmp(f, "b")) CPD(s->b, s->nx1);
    if (!strcmp(f, "Q")) CPD(s->Q, s->nx * s->nx);
    if (!strcmp(f, "S")) CPD(s->S, s->nu * s->nx);
    if (!strcmp(f, "R")) CPD(s->R, s->nu * s->nu);
    if (!strcmp(f, "q")) CPD(s->q, s->nx);
    if (!strcmp(f, "r")) CPD(s->r, s->nu);
    if (!strcmp(f, "idxb")) CPI(s->idxb, s->nb);
    if (!strcmp(f, "idxbu")) CPI(s->idxb, s->nbu);
    if (!strcmp(f, "idxbx"))
    {
        const int *v = (const int *) value;
        for (int i = 0; i < s->nbx; i++) s->idxb[s->nbu + i] = s->nu + v[i];
        return 0;
    }
    if (!strcmp(f, "lbu")) CPD(s->lb, s->nbu);
    if (!strcmp(f, "ubu")) CPD(s->ub, s->nbu);
    if (!strcmp(f, "lbx")) CPD(s->lb + s->nbu, s->nbx);
    if (!strcmp(f, "ubx")) CPD(s->ub + s->nbu, s->nbx);
    if (!strcmp(f, "lbu_mask")) CPD(s->lb_mask, s->nbu);
    if (!strcmp(f, "ubu_mask")) CPD(s->ub_mask, s->nbu);
    if (!strcmp(f, "lbx_mask")) CPD(s->lb_mask + s->nbu, s->nbx);
    if (!strcmp(f, "ubx_mask")) CPD(s->ub_mask + s->nbu, s->nbx);
    if (!strcmp(f, "C")) CPD(s->C, s->ng * s->nx);
    if (!strcmp(f, "D")) CPD(s->D, s->ng * s->nu);
    if (!strcmp(f, "lg")) CPD(s->lg, s->ng);
    if (!strcmp(f, "ug")) CPD(s->ug, s->ng);
    if (!strcmp(f, "lg_mask")) CPD(s->lg_mask, s->ng);
    if (!strcmp(f, "ug_mask")) CPD(s->ug_mask, s->ng);
    if (!strcmp(f, "Zl")) CPD(s->Zl, s->ns);
    if (!strcmp(f, "Zu")) CPD(s->Zu, s->ns);
    if (!strcmp(f, "zl")) CPD(s->zl, s->ns);
    if (!strcmp(f, "zu")) CPD(s->zu, s->ns);
    if (!strcmp(f, "lls")) CPD(s->lls, s->ns);
    if (!strcmp(f, "lus")) CPD(s->lus, s->ns);
    if (!strcmp(f, "lls_mask")) CPD(s->lls_mask, s->ns);
    if (!strcmp(f, "lus_mask")) CPD(s->lus_mask, s->ns);
    if (!strcmp(f, "idxs_rev")) CPI(s->idxs_rev, s->nbg);
    if (!strcmp(f, "idxe")) CPI(s->idxe, s->nbxe);
    /* warm start values */
    if (!strcmp(f, "ux")) CPD(s->ux, s->n + 2 * s->ns);
    if (!strcmp(f, "pi")) CPD(s->pi, s->nx1);
    if (!strcmp(f, "lam")) CPD(s->lam, s->nct);
    if (!strcmp(f, "t")) CPD(s->t, s->nct);
    fprintf(stderr, "oqp_set: unknown field %s\n", f);
    return -1;
}

int oqp_get(const oqp *qp, const char *f, int k, double *v)
{
    const stg *s = qp->s + k;
    if (!strcmp(f, "u")) { memcpy(v, s->ux, sizeof(double) * s->nu); return 0; }
    if (!strcmp(f, "x")) { memcpy(v, s->ux + s->nu, sizeof(double) * s->nx); return 0; }
    if (!strcmp(f, "sl")) { memcpy(v, s->ux + s->n, sizeof(double) * s->ns); return 0; }
    if (!strcmp(f, "su")) { memcpy(v, s->ux + s->n + s->ns, sizeof(double) * s->ns); return 0; }
    if (!strcmp(f, "pi")) { memcpy(v, s->pi, sizeof(double) * s->nx1); return 0; }
    if (!strcmp(f, "lam")) { memcpy(v, s->lam, sizeof(double) * s->nct); return 0; }
    if (!strcmp(f, "t")) { memcpy(v, s->t, sizeof(double) * s->nct); return 0; }
    if (!strcmp(f, "ric_L"))
    {
        for (int c = 0; c < s->n; c++)
            for (int r = 0; r < s->n; r++) v[r + s->n * c] = r >= c ? s->L[r + s->n * c] : 0.0;
        return 0;
    }
    if (!strcmp(f, "ric_l")) { memcpy(v, s->l, sizeof(double) * s->n); return 0; }
    fprintf(stderr, "oqp_get: unknown field %s\n", f);
    return -1;
}

/* factor of the Newton system at the CURRENT iterate (what the device holds after its final
 * residual/factor launch); afterwards "ric_L" (n x n col-major, lower) and "ric_l" can be read */
void oqp_refactor(oqp *qp, const oqp_opts *o);

int oqp_get_iter(const oqp *qp) { return qp->iter; }
const double *oqp_get_stat(const oqp *qp) { return qp->stat; }

/* ------------------------------------------------------- constraint helpers */

/* c = [v[idxb]; C x + D u] */
static void stage_cv(const stg *s, const double *v, double *c)
{
    const double *u = v, *x = v + s->nu;
    for (int i = 0; i < s->nb; i++) c[i] = v[s->idxb[i]];
    for (int j = 0; j < s->ng; j++)
    {
        double a = 0.0;
        for (int i = 0; i < s->nu; i++) a += s->D[j + s->ng * i] * u[i];
        for (int i = 0; i < s->nx; i++) a += s->C[j + s->ng * i] * x[i];
        c[s->nb + j] = a;
    }
}

/* g[0:n] += sign * J' * nu  with J the (nbg x n) constraint Jacobian */
static void stage_jt_add(const stg *s, const double *nu_, double sign, double *g)
{
    for (int i = 0; i < s->nb; i++) g[s->idxb[i]] += sign * nu_[i];
    for (int j = 0; j < s->ng; j++)
    {
        double a = sign * nu_[s->nb + j];
        if (a == 0.0) continue;
        for (int i = 0; i < s->nu; i++) g[i] += s->D[j + s->ng * i] * a;
        for (int i = 0; i < s->nx; i++) g[s->nu + i] += s->C[j + s->ng * i] * a;
    }
}

/* row i of the constraint Jacobian into a (length n) */
static void stage_jrow(const stg *s, int i, double *a)
{
    memset(a, 0, sizeof(double) * s->n);
    if (i < s->nb) { a[s->idxb[i]] = 1.0; return; }
    int j = i - s->nb;
    for (int c = 0; c < s->nu; c++) a[c] = s->D[j + s->ng * c];
    for (int c = 0; c < s->nx; c++) a[s->nu + c] = s->C[j + s->ng * c];
}

static void setup_active(oqp *qp)
{
    qp->n_act = 0;
    for (int k = 0; k <= qp->N; k++)
    {
        stg *s = qp->s + k;
        int nbg = s->nbg;
        memset(s->fixed, 0, sizeof(int) * s->n);
        for (int i = 0; i < s->nb; i++)
        {
            s->act[i] = s->lb_mask[i] != 0.0;
            s->act[nbg + i] = s->ub_mask[i] != 0.0;
        }
        for (int j = 0; j < s->ng; j++)
        {
            s->act[s->nb + j] = s->lg_mask[j] != 0.0;
            s->act[nbg + s->nb + j] = s->ug_mask[j] != 0.0;
        }
        for (int j = 0; j < s->ns; j++)
        {
            s->act[2 * nbg + j] = s->lls_mask[j] != 0.0;
            s->act[2 * nbg + s->ns + j] = s->lus_mask[j] != 0.0;
        }
        for (int e = 0; e < s->nbxe; e++)
        {
            int i = s->idxe[e];
            s->act[i] = 0; s->act[nbg + i] = 0;
            s->fixed[s->idxb[i]] = 1;
            s->fixval[s->idxb[i]] = s->lb[i];
        }
        for (int i = 0; i < s->nct; i++) qp->n_act += s->act[i];
    }
}

/* value of inequality row idx at (v, slacks): the quantity that must equal t */
static void stage_ineq_val(const stg *s, const double *ux, const double *c, double *val)
{
    int nbg = s->nbg, n = s->n, ns = s->ns;
    for (int i = 0; i < nbg; i++)
    {
        double lo = i < s->nb ? s->lb[i] : s->lg[i - s->nb];
        double up = i < s->nb ? s->ub[i] : s->ug[i - s->nb];
        double sl = 0.0, su = 0.0;
        int j = s->idxs_rev[i];
        if (j >= 0) { sl = ux[n + j]; su = ux[n + ns + j]; }
        val[i] = c[i] + sl - lo;
        val[nbg + i] = up - c[i] + su;
    }
    for (int j = 0; j < ns; j++)
    {
        val[2 * nbg + j] = ux[n + j] - s->lls[j];
        val[2 * nbg + ns + j] = ux[n + ns + j] - s->lus[j];
    }
}

/* ---------------------------------------------------------- residuals */

static void compute_res(oqp *qp, double tau, double *mu, double nrm[4])
{
    double sum = 0.0;
    nrm[0] = nrm[1] = nrm[2] = nrm[3] = 0.0;
    for (int k = 0; k <= qp->N; k++)
    {
        stg *s = qp->s + k;
        int nu = s->nu, nx = s->nx, n = s->n, ns = s->ns, nbg = s->nbg, nx1 = s->nx1;
        const double *u = s->ux, *x = s->ux + nu;
        double *rg = s->rg;
        /* H v + g */
        for (int i = 0; i < nu; i++)
        {
            double a = s->r[i];
            for (int j = 0; j < nu; j++) a += s->R[i + nu * j] * u[j];
            for (int j = 0; j < nx; j++) a += s->S[i + nu * j] * x[j];
            rg[i] = a;
        }
        for (int i = 0; i < nx; i++)
        {
            double a = s->q[i];
            for (int j = 0; j < nu; j++) a += s->S[j + nu * i] * u[j];
            for (int j = 0; j < nx; j++) a += s->Q[i + nx * j] * x[j];
            rg[nu + i] = a;
        }
        /* [B A]' pi_{k+1} - [0; pi_k] */
        for (int i = 0; i < nu; i++)
            for (int j = 0; j < nx1; j++) rg[i] += s->B[j + nx1 * i] * s->pi[j];
        for (int i = 0; i < nx; i++)
            for (int j = 0; j < nx1; j++) rg[nu + i] += s->A[j + nx1 * i] * s->pi[j];
        if (k > 0)
            for (int i = 0; i < nx; i++) rg[nu + i] -= qp->s[k - 1].pi[i];
        /* - J'(lam_l - lam_u) */
        for (int i = 0; i < nbg; i++)
            s->dc[i] = (s->act[i] ? s->lam[i] : 0.0) - (s->act[nbg + i] ? s->lam[nbg + i] : 0.0);
        stage_jt_add(s, s->dc, -1.0, rg);
        /* slack stationarity */
        for (int j = 0; j < ns; j++)
        {
            rg[n + j] = s->Zl[j] * s->ux[n + j] + s->zl[j] - (s->act[2 * nbg + j] ? s->lam[2 * nbg + j] : 0.0);
            rg[n + ns + j] = s->Zu[j] * s->ux[n + ns + j] + s->zu[j]
                             - (s->act[2 * nbg + ns + j] ? s->lam[2 * nbg + ns + j] : 0.0);
        }
        for (int i = 0; i < nbg; i++)
        {
            int j = s->idxs_rev[i];
            if (j < 0) continue;
            if (s->act[i]) rg[n + j] -= s->lam[i];
            if (s->act[nbg + i]) rg[n + ns + j] -= s->lam[nbg + i];
        }
        for (int i = 0; i < n; i++) if (s->fixed[i]) rg[i] = 0.0;
        for (int i = 0; i < n + 2 * ns; i++) if (fabs(rg[i]) > nrm[0] || rg[i] != rg[i]) nrm[0] = fabs(rg[i]);
        /* dynamics */
        if (k < qp->N)
        {
            const double *xn = qp->s[k + 1].ux + qp->s[k + 1].nu;
            for (int i = 0; i < nx1; i++)
            {
                double a = s->b[i] - xn[i];
                for (int j = 0; j < nu; j++) a += s->B[i + nx1 * j] * u[j];
                for (int j = 0; j < nx; j++) a += s->A[i + nx1 * j] * x[j];
                s->rb[i] = a;
                if (fabs(a) > nrm[1] || a != a) nrm[1] = fabs(a);
            }
        }
        /* inequalities + complementarity */
        stage_cv(s, s->ux, s->c);
        stage_ineq_val(s, s->ux, s->c, s->rd);
        for (int i = 0; i < s->nct; i++)
        {
            if (!s->act[i]) { s->rd[i] = 0.0; s->rm[i] = 0.0; continue; }
            s->rd[i] -= s->t[i];
            s->rm[i] = s->lam[i] * s->t[i] - tau;
            sum += s->lam[i] * s->t[i];
            if (fabs(s->rd[i]) > nrm[2] || s->rd[i] != s->rd[i]) nrm[2] = fabs(s->rd[i]);
            if (fabs(s->rm[i]) > nrm[3] || s->rm[i] != s->rm[i]) nrm[3] = fabs(s->rm[i]);
        }
    }
    *mu = qp->n_act > 0 ? sum / qp->n_act : 0.0;
}

void oqp_res_nrm_inf(oqp *qp, double res[4])
{
    double mu;
    setup_active(qp);
    compute_res(qp, 0.0, &mu, res);
}

void oqp_compute_t(oqp *qp)
{
    /* ocp_qp_common.c:874-921 */
    for (int k = 0; k <= qp->N; k++)
    {
        stg *s = qp->s + k;
        stage_cv(s, s->ux, s->c);
        stage_ineq_val(s, s->ux, s->c, s->t);
    }
}

/* ------------------------------------------------------------ KKT step */

/* Hessian/gradient condensation of the inequality part (Gamma, rho) for stage s.
 * rm_eff is the complementarity rhs actually used (affine or corrected). */
static void stage_condense(stg *s, const double *rm_eff, int do_mat, double reg)
{
    int nu = s->nu, nx = s->nx, n = s->n, ns = s->ns, nbg = s->nbg;
    double *M = s->L;
    for (int i = 0; i < s->nct; i++)
    {
        if (!s->act[i]) { s->Gam[i] = 0.0; s->rho[i] = 0.0; continue; }
        s->Gam[i] = s->lam[i] / s->t[i];
        s->rho[i] = (rm_eff[i] + s->lam[i] * s->rd[i]) / s->t[i];
    }
    /* slack elimination.  D = Z + Gamma_s + sum Gamma_row, r~ = slack stationarity + rho_s + sum rho_row.
     * The Schur complement of the slack block is accumulated in its CANCELLATION-FREE form: for a row i
     * coupled to slack j,  Gamma_i - Gamma_i^2/D = Gamma_i E_i/D  with  E_i = D - Gamma_i  summed without
     * Gamma_i (an active soft row has Gamma_i >> Z, and Gamma_i - Gamma_i^2/D would lose every digit of the
     * Z + Gamma_s it should leave behind); likewise rho_i - Gamma_i r~/D = (rho_i E_i - Gamma_i (r~ - rho_i))/D.
     * Rows sharing one slack add the cross terms -Gamma_i Gamma_k/D a_i a_k'. */
    for (int j = 0; j < ns; j++)
    {
        s->Dl[j] = s->Zl[j] + s->Gam[2 * nbg + j];
        s->Du[j] = s->Zu[j] + s->Gam[2 * nbg + ns + j];
        s->rsl[j] = s->rg[n + j] + s->rho[2 * nbg + j];
        s->rsu[j] = s->rg[n + ns + j] + s->rho[2 * nbg + ns + j];
    }
    for (int i = 0; i < nbg; i++)
    {
        int j = s->idxs_rev[i];
        if (j < 0) continue;
        s->Dl[j] += s->Gam[i];
        s->Du[j] += s->Gam[nbg + i];
        s->rsl[j] += s->rho[i];
        s->rsu[j] += s->rho[nbg + i];
    }
    /* per soft row: exclusive sums E (-> El, Eu) and r~ - rho (-> Xl, Xu); effective Gamma and nu */
    for (int i = 0; i < nbg; i++)
    {
        int j = s->idxs_rev[i];
        if (j < 0)
        {
            s->gme[i] = s->Gam[i] + s->Gam[nbg + i];
            s->dc[i] = s->rho[i] - s->rho[nbg + i];
            continue;
        }
        double El = s->Zl[j] + s->Gam[2 * nbg + j], Eu = s->Zu[j] + s->Gam[2 * nbg + ns + j];
        double Xl = s->rg[n + j] + s->rho[2 * nbg + j], Xu = s->rg[n + ns + j] + s->rho[2 * nbg + ns + j];
        for (int k = 0; k < nbg; k++)
            if (k != i && s->idxs_rev[k] == j)
            {
                El += s->Gam[k]; Eu += s->Gam[nbg + k];
                Xl += s->rho[k]; Xu += s->rho[nbg + k];
            }
        s->El[i] = El; s->Eu[i] = Eu; s->Xl[i] = Xl; s->Xu[i] = Xu;
        double il = s->Dl[j] != 0.0 ? 1.0 / s->Dl[j] : 0.0, iu = s->Du[j] != 0.0 ? 1.0 / s->Du[j] : 0.0;
        s->gme[i] = s->Gam[i] * El * il + s->Gam[nbg + i] * Eu * iu;
        s->dc[i] = (s->rho[i] * El - s->Gam[i] * Xl) * il - (s->rho[nbg + i] * Eu - s->Gam[nbg + i] * Xu) * iu;
    }
    /* gradient: gt = rg + J' nu_eff */
    memcpy(s->gt, s->rg, sizeof(double) * n);
    stage_jt_add(s, s->dc, 1.0, s->gt);
    if (!do_mat) return;
    /* Hessian (full symmetric storage, col-major n x n) */
    for (int j = 0; j < nu; j++)
    {
        for (int i = 0; i < nu; i++) M[i + n * j] = s->R[i + nu * j];
        for (int i = 0; i < nx; i++) M[nu + i + n * j] = s->S[j + nu * i];
    }
    for (int j = 0; j < nx; j++)
    {
        for (int i = 0; i < nu; i++) M[i + n * (nu + j)] = s->S[i + nu * j];
        for (int i = 0; i < nx; i++) M[nu + i + n * (nu + j)] = s->Q[i + nx * j];
    }
    for (int i = 0; i < n; i++) M[i + n * i] += reg;
    for (int i = 0; i < s->nb; i++) M[s->idxb[i] * (n + 1)] += s->gme[i];
    for (int g = 0; g < s->ng; g++)
    {
        double gm = s->gme[s->nb + g];
        if (gm == 0.0) continue;
        stage_jrow(s, s->nb + g, s->tmp);
        for (int c = 0; c < n; c++)
            for (int r = 0; r < n; r++) M[r + n * c] += gm * s->tmp[r] * s->tmp[c];
    }
    /* rows sharing a slack: cross terms */
    for (int i = 0; i < nbg; i++)
    {
        int j = s->idxs_rev[i];
        if (j < 0) continue;
        double il = s->Dl[j] != 0.0 ? 1.0 / s->Dl[j] : 0.0, iu = s->Du[j] != 0.0 ? 1.0 / s->Du[j] : 0.0;
        for (int k = 0; k < nbg; k++)
        {
            if (k == i || s->idxs_rev[k] != j) continue;
            double cf = s->Gam[i] * s->Gam[k] * il + s->Gam[nbg + i] * s->Gam[nbg + k] * iu;
            if (cf == 0.0) continue;
            stage_jrow(s, i, s->tmp);
            stage_jrow(s, k, s->tmp2);
            for (int c = 0; c < n; c++)
                for (int r = 0; r < n; r++) M[r + n * c] -= cf * s->tmp[r] * s->tmp2[c];
        }
    }
}

/* in-place lower Cholesky (col-major n x n); non-positive pivots are zeroed as
 * BLASFEO's reference dpotrf does */
static void chol_lower(int n, double *M)
{
    for (int j = 0; j < n; j++)
    {
        double d = M[j + n * j];
        for (int p = 0; p < j; p++) d -= M[j + n * p] * M[j + n * p];
        double inv;
        if (d > 0.0) { d = sqrt(d); inv = 1.0 / d; } else { d = 0.0; inv = 0.0; }
        M[j + n * j] = d;
        for (int i = j + 1; i < n; i++)
        {
            double a = M[i + n * j];
            for (int p = 0; p < j; p++) a -= M[i + n * p] * M[j + n * p];
            M[i + n * j] = a * inv;
        }
    }
}

/* backward sweep: (optionally) factorise, always propagate the rhs row */
static void riccati_backward(oqp *qp, int do_factor)
{
    for (int k = qp->N; k >= 0; k--)
    {
        stg *s = qp->s + k;
        int nu = s->nu, nx = s->nx, n = s->n, nx1 = s->nx1;
        double *M = s->L, *m = s->l;
        memcpy(m, s->gt, sizeof(double) * n);
        if (k < qp->N)
        {
            stg *sn = qp->s + k + 1;
            int nn = sn->n, nun = sn->nu;
            const double *Ln = sn->L; /* Lx+ = Ln[nun:, nun:] */
            if (do_factor)
            {
                /* W = [B A]' * Lx+   (n x nx1) */
                for (int c = 0; c < nx1; c++)
                    for (int r = 0; r < n; r++)
                    {
                        double a = 0.0;
                        const double *col = r < nu ? s->B + nx1 * r : s->A + nx1 * (r - nu);
                        for (int i = c; i < nx1; i++) a += col[i] * Ln[(nun + i) + nn * (nun + c)];
                        s->W[r + n * c] = a;
                    }
                for (int c = 0; c < n; c++)
                    for (int r = 0; r < n; r++)
                    {
                        double a = 0.0;
                        for (int i = 0; i < nx1; i++) a += s->W[r + n * i] * s->W[c + n * i];
                        M[r + n * c] += a;
                    }
            }
            /* w0 = Lx+' rb + lx+ ;  m += W w0 */
            for (int c = 0; c < nx1; c++)
            {
                double a = sn->l[nun + c];
                for (int i = c; i < nx1; i++) a += Ln[(nun + i) + nn * (nun + c)] * s->rb[i];
                s->tmp[c] = a;
            }
            for (int r = 0; r < n; r++)
            {
                double a = 0.0;
                for (int c = 0; c < nx1; c++) a += s->W[r + n * c] * s->tmp[c];
                m[r] += a;
            }
        }
        for (int i = 0; i < n; i++) if (s->fixed[i]) m[i] = 0.0;
        if (do_factor)
        {
            for (int i = 0; i < n; i++)
                if (s->fixed[i])
                {
                    for (int j = 0; j < n; j++) M[i + n * j] = M[j + n * i] = 0.0;
                    M[i + n * i] = 1.0;
                }
            chol_lower(n, M);
        }
        /* l = L^{-1} m */
        for (int i = 0; i < n; i++)
        {
            double a = m[i];
            for (int p = 0; p < i; p++) a -= M[i + n * p] * m[p];
            double d = M[i + n * i];
            m[i] = d != 0.0 ? a / d : 0.0;
        }
        (void) nx;
    }
}

static void riccati_forward(oqp *qp)
{
    for (int k = 0; k <= qp->N; k++)
    {
        stg *s = qp->s + k;
        int nu = s->nu, nx = s->nx, n = s->n, nx1 = s->nx1;
        const double *L = s->L, *l = s->l;
        double *dv = s->dux;
        int top = (k == 0) ? n : nu; /* stage 0: all of [u;x] is free (fixed entries solve to 0) */
        /* rhs for the transposed solve: -(l + Ls' dx) for the rows being solved */
        for (int i = 0; i < top; i++)
        {
            double a = -l[i];
            for (int p = top; p < n; p++) a -= L[p + n * i] * dv[p];
            s->tmp[i] = a;
        }
        for (int i = top - 1; i >= 0; i--)
        {
            double a = s->tmp[i];
            for (int p = i + 1; p < top; p++) a -= L[p + n * i] * dv[p];
            double d = L[i + n * i];
            dv[i] = d != 0.0 ? a / d : 0.0;
        }
        if (k < qp->N)
        {
            stg *sn = qp->s + k + 1;
            int nn = sn->n, nun = sn->nu;
            double *dxn = sn->dux + nun;
            for (int i = 0; i < nx1; i++)
            {
                double a = s->rb[i];
                for (int j = 0; j < nu; j++) a += s->B[i + nx1 * j] * dv[j];
                for (int j = 0; j < nx; j++) a += s->A[i + nx1 * j] * dv[nu + j];
                dxn[i] = a;
            }
            /* dpi = Lx+ (Lx+' dx+ + lx+) */
            for (int c = 0; c < nx1; c++)
            {
                double a = sn->l[nun + c];
                for (int i = c; i < nx1; i++) a += sn->L[(nun + i) + nn * (nun + c)] * dxn[i];
                s->tmp[c] = a;
            }
            for (int i = 0; i < nx1; i++)
            {
                double a = 0.0;
                for (int c = 0; c <= i; c++) a += sn->L[(nun + i) + nn * (nun + c)] * s->tmp[c];
                s->dpi[i] = a;
            }
        }
    }
}

/* recover slack / multiplier / t steps from dv */
static void expand_step(oqp *qp, const double *const *rm_eff)
{
    for (int k = 0; k <= qp->N; k++)
    {
        stg *s = qp->s + k;
        int n = s->n, ns = s->ns, nbg = s->nbg;
        const double *rm = rm_eff[k];
        stage_cv(s, s->dux, s->dc);
        for (int j = 0; j < ns; j++)
        {
            double al = 0.0, au = 0.0;
            for (int i = 0; i < nbg; i++)
                if (s->idxs_rev[i] == j) { al += s->Gam[i] * s->dc[i]; au += s->Gam[nbg + i] * s->dc[i]; }
            s->dux[n + j] = s->Dl[j] != 0.0 ? (-s->rsl[j] - al) / s->Dl[j] : 0.0;
            s->dux[n + ns + j] = s->Du[j] != 0.0 ? (-s->rsu[j] + au) / s->Du[j] : 0.0;
        }
        for (int i = 0; i < nbg; i++)
        {
            int j = s->idxs_rev[i];
            if (j < 0)
            {
                s->dt[i] = s->dc[i] + s->rd[i];
                s->dt[nbg + i] = -s->dc[i] + s->rd[nbg + i];
                continue;
            }
            /* dc + ds in the cancellation-free form (E dc - r~ - sum_{k != i} Gamma_k dc_k)/D */
            double al = 0.0, au = 0.0;
            for (int k = 0; k < nbg; k++)
                if (k != i && s->idxs_rev[k] == j) { al += s->Gam[k] * s->dc[k]; au += s->Gam[nbg + k] * s->dc[k]; }
            double il = s->Dl[j] != 0.0 ? 1.0 / s->Dl[j] : 0.0, iu = s->Du[j] != 0.0 ? 1.0 / s->Du[j] : 0.0;
            s->dt[i] = (s->El[i] * s->dc[i] - s->rsl[j] - al) * il + s->rd[i];
            s->dt[nbg + i] = (-s->Eu[i] * s->dc[i] - s->rsu[j] + au) * iu + s->rd[nbg + i];
        }
        for (int j = 0; j < ns; j++)
        {
            s->dt[2 * nbg + j] = s->dux[n + j] + s->rd[2 * nbg + j];
            s->dt[2 * nbg + ns + j] = s->dux[n + ns + j] + s->rd[2 * nbg + ns + j];
        }
        for (int i = 0; i < s->nct; i++)
        {
            if (!s->act[i]) { s->dt[i] = 0.0; s->dlam[i] = 0.0; continue; }
            s->dlam[i] = -(rm[i] + s->lam[i] * s->dt[i]) / s->t[i];
        }
    }
}

static double step_length(const oqp *qp)
{
    double alpha = 1.0;
    for (int k = 0; k <= qp->N; k++)
    {
        const stg *s = qp->s + k;
        for (int i = 0; i < s->nct; i++)
        {
            if (!s->act[i]) continue;
            if (s->dlam[i] < 0.0 && -s->lam[i] > alpha * s->dlam[i]) alpha = -s->lam[i] / s->dlam[i];
            if (s->dt[i] < 0.0 && -s->t[i] > alpha * s->dt[i]) alpha = -s->t[i] / s->dt[i];
        }
    }
    return alpha;
}

static void init_var(oqp *qp, const oqp_opts *o)
{
    const double thr0 = 1e-1;
    /* t0_init (acados_ocp_options.py:1128-1143): 0 lam = t = sqrt(mu0); 1 lam = mu0, t = 1 -- primal iterate and slacks
     * stay at zero; 2 (default) slacks from the constraint residuals, clipped at 0.1, lam = mu0 / t */
    const int heur = o->t0_init != 0 && o->t0_init != 1;
    const double t_c = o->t0_init == 0 ? sqrt(o->mu0) : 1.0, l_c = o->t0_init == 0 ? sqrt(o->mu0) : o->mu0;
    for (int k = 0; k <= qp->N; k++)
    {
        stg *s = qp->s + k;
        int n = s->n, ns = s->ns, nbg = s->nbg;
        /* cold start zeroes the primal iterate (ocp_qp_hpipm.c:333-336) */
        memset(s->ux, 0, sizeof(double) * (n + 2 * ns));
        memset(s->pi, 0, sizeof(double) * s->nx1);
        for (int i = 0; i < n; i++) if (s->fixed[i]) s->ux[i] = s->fixval[i];
        /* hard box rows: move the variable inside its bounds */
        for (int i = 0; i < s->nb; i++)
        {
            int iv = s->idxb[i];
            if (s->fixed[iv] || s->idxs_rev[i] >= 0 || !heur) continue;
            int al = s->act[i], au = s->act[nbg + i];
            double tl = s->ux[iv] - s->lb[i], tu = s->ub[i] - s->ux[iv];
            if (al && au)
            {
                if (tl < thr0)
                {
                    if (tu < thr0) s->ux[iv] = 0.5 * (s->lb[i] + s->ub[i]);
                    else s->ux[iv] = s->lb[i] + thr0;
                }
                else if (tu < thr0) s->ux[iv] = s->ub[i] - thr0;
            }
            else if (al) { if (tl < thr0) s->ux[iv] = s->lb[i] + thr0; }
            else if (au) { if (tu < thr0) s->ux[iv] = s->ub[i] - thr0; }
        }
        stage_cv(s, s->ux, s->c);
        /* slacks: large enough that every soft row and the slack bound start interior */
        for (int j = 0; j < ns; j++)
        {
            s->ux[n + j] = (heur && s->act[2 * nbg + j]) ? s->lls[j] + thr0 : 0.0;
            s->ux[n + ns + j] = (heur && s->act[2 * nbg + ns + j]) ? s->lus[j] + thr0 : 0.0;
        }
        for (int i = 0; i < nbg; i++)
        {
            int j = s->idxs_rev[i];
            if (j < 0 || !heur) continue;
            double lo = i < s->nb ? s->lb[i] : s->lg[i - s->nb];
            double up = i < s->nb ? s->ub[i] : s->ug[i - s->nb];
            if (s->act[i] && lo - s->c[i] + thr0 > s->ux[n + j]) s->ux[n + j] = lo - s->c[i] + thr0;
            if (s->act[nbg + i] && s->c[i] - up + thr0 > s->ux[n + ns + j]) s->ux[n + ns + j] = s->c[i] - up + thr0;
        }
        stage_ineq_val(s, s->ux, s->c, s->t);
        for (int i = 0; i < s->nct; i++)
        {
            if (!s->act[i]) { s->lam[i] = 0.0; continue; }
            if (s->t[i] < thr0) s->t[i] = thr0;
            s->lam[i] = o->mu0 / s->t[i];
            if (!heur) { s->t[i] = t_c; s->lam[i] = l_c; }
        }
    }
}

/* multipliers of equality-flagged bounds from stationarity; t of inactive rows */
static void finalize_sol(oqp *qp)
{
    for (int k = 0; k <= qp->N; k++)
    {
        stg *s = qp->s + k;
        int nu = s->nu, nx = s->nx, nbg = s->nbg, nx1 = s->nx1;
        if (s->nbxe > 0)
        {
            const double *u = s->ux, *x = s->ux + nu;
            for (int e = 0; e < s->nbxe; e++)
            {
                int ib = s->idxe[e], iv = s->idxb[ib];
                double a;
                if (iv < nu)
                {
                    int i = iv;
                    a = s->r[i];
                    for (int j = 0; j < nu; j++) a += s->R[i + nu * j] * u[j];
                    for (int j = 0; j < nx; j++) a += s->S[i + nu * j] * x[j];
                    for (int j = 0; j < nx1; j++) a += s->B[j + nx1 * i] * s->pi[j];
                }
                else
                {
                    int i = iv - nu;
                    a = s->q[i];
                    for (int j = 0; j < nu; j++) a += s->S[j + nu * i] * u[j];
                    for (int j = 0; j < nx; j++) a += s->Q[i + nx * j] * x[j];
                    for (int j = 0; j < nx1; j++) a += s->A[j + nx1 * i] * s->pi[j];
                    if (k > 0) a -= qp->s[k - 1].pi[i];
                }
                /* other constraints touching this variable */
                for (int i = 0; i < nbg; i++)
                {
                    if (i == ib) continue;
                    double nuv = (s->act[i] ? s->lam[i] : 0.0) - (s->act[nbg + i] ? s->lam[nbg + i] : 0.0);
                    if (nuv == 0.0) continue;
                    stage_jrow(s, i, s->tmp);
                    a -= s->tmp[iv] * nuv;
                }
                s->lam[ib] = a > 0.0 ? a : 0.0;
                s->lam[nbg + ib] = a < 0.0 ? -a : 0.0;
            }
        }
        /* rows that did not take part: natural slack value, zero multiplier */
        stage_cv(s, s->ux, s->c);
        stage_ineq_val(s, s->ux, s->c, s->rd);
        for (int i = 0; i < s->nct; i++)
            if (!s->act[i])
            {
                s->t[i] = s->rd[i];
                int is_eq = 0;
                for (int e = 0; e < s->nbxe; e++)
                    if (s->idxe[e] == i || s->idxe[e] + nbg == i) is_eq = 1;
                if (!is_eq) s->lam[i] = 0.0;
            }
    }
}

/* Barrier floor.  Once the complementarity products are far below tol_comp, pushing mu further only
 * inflates Gamma = lam/t (1e16 and beyond) and with it the rounding error of dlam = -(rm + lam dt)/t: the
 * stationarity residual, already converged, grows again and a few instances in ten thousand never come back
 * (C4: 3 of 16,384 ended in MAXITER at res_stat 1e-7..1e-5 with mu = 5e-16).  The complementarity target
 * is therefore kept at max(tau_min, 1e-3 tol_comp) -- three orders below anything the tolerance can see. */
static double tau_eff(const oqp_opts *o)
{
    const double f = 1e-3 * o->tol_comp;
    return o->tau_min > f ? o->tau_min : f;
}

int oqp_solve(oqp *qp, const oqp_opts *o)
{
    int N = qp->N;
    double mu, nrm[4];
    /* no allocation inside the solve (acados rule, ocp_qp_interface.c:550-563) -- and none that
     * could serialise the OpenMP batch loop on the allocator */
    const double *rm_ptr_buf[1024];
    double *rmc_buf[1024];
    const double **rm_ptr = N + 1 <= 1024 ? rm_ptr_buf : (const double **) malloc(sizeof(double *) * (N + 1));
    double **rmc = N + 1 <= 1024 ? rmc_buf : (double **) malloc(sizeof(double *) * (N + 1));
    for (int k = 0; k <= N; k++) rmc[k] = qp->s[k].rmc;

    setup_active(qp);
    if (o->warm_start < 2) init_var(qp, o);
    compute_res(qp, tau_eff(o), &mu, nrm);

    int it = 0, status = OQP_MAXITER;
    double alpha = 1.0;
    if (qp->stat_rows < o->iter_max + 2)
    {
        free(qp->stat);
        qp->stat_rows = o->iter_max + 2;
        qp->stat = dz(STAT_M * qp->stat_rows);
    }
    double *st = qp->stat;
    memset(st, 0, sizeof(double) * STAT_M);
    st[6] = mu; st[7] = nrm[0]; st[8] = nrm[1]; st[9] = nrm[2]; st[10] = nrm[3];

    for (;;)
    {
        if (nrm[0] != nrm[0] || nrm[1] != nrm[1] || nrm[2] != nrm[2] || nrm[3] != nrm[3] || mu != mu)
        { status = OQP_NAN_DETECTED; break; }
        if (nrm[0] <= o->tol_stat && nrm[1] <= o->tol_eq && nrm[2] <= o->tol_ineq && nrm[3] <= o->tol_comp)
        { status = OQP_SUCCESS; break; }
        if (it >= o->iter_max) { status = OQP_MAXITER; break; }
        if (alpha <= o->alpha_min) { status = OQP_MINSTEP; break; }

        /* affine (predictor) direction */
        for (int k = 0; k <= N; k++) { rm_ptr[k] = qp->s[k].rm; stage_condense(qp->s + k, qp->s[k].rm, 1, o->reg_prim); }
        riccati_backward(qp, 1);
        riccati_forward(qp);
        expand_step(qp, rm_ptr);
        alpha = step_length(qp);
        double alpha_aff = alpha, mu_aff = 0.0, sigma = 0.0;
        if (o->pred_corr && qp->n_act > 0)
        {
            for (int k = 0; k <= N; k++)
            {
                stg *s = qp->s + k;
                for (int i = 0; i < s->nct; i++)
                    if (s->act[i]) mu_aff += (s->lam[i] + alpha * s->dlam[i]) * (s->t[i] + alpha * s->dt[i]);
            }
            mu_aff /= qp->n_act;
            sigma = mu > 0.0 ? mu_aff / mu : 0.0;
            sigma = sigma * sigma * sigma;
            for (int k = 0; k <= N; k++)
            {
                stg *s = qp->s + k;
                for (int i = 0; i < s->nct; i++)
                    rmc[k][i] = s->act[i] ? s->rm[i] + s->dlam[i] * s->dt[i] - sigma * mu : 0.0;
                rm_ptr[k] = rmc[k];
                stage_condense(s, rmc[k], 0, o->reg_prim);
            }
            riccati_backward(qp, 0);
            riccati_forward(qp);
            expand_step(qp, rm_ptr);
            alpha = step_length(qp);
            /* conditional corrector (HPIPM's cond_pred_corr, on in the SPEED / BALANCE / ROBUST modes acados uses; HPIPM's sources
             * are absent from the reference tree -- upstream knowledge of d_ocp_qp_ipm_solve): the duality measure at the end of
             * the predictor-corrector step is evaluated, and a step that would more than double it is computed again from the
             * centering term alone.  (Rounds 1-5 of this restatement asked `alpha < 0.1 alpha_aff` instead: identical on every
             * golden vector and on the BASELINE configurations -- neither test ever fires there -- but on random structures with
             * perturbed costs it left limit cycles of the Mehrotra iteration standing: 10 of 15,360 instances at MAXITER against
             * 0 with this test, which also never needs more iterations; tools/fuzz_parity.py, profiles/NOTES.md round 5) */
            double mu_pc = 0.0;
            if (o->cond_pred_corr)
            {
                for (int k = 0; k <= N; k++)
                {
                    stg *s = qp->s + k;
                    for (int i = 0; i < s->nct; i++)
                        if (s->act[i]) mu_pc += (s->lam[i] + alpha * s->dlam[i]) * (s->t[i] + alpha * s->dt[i]);
                }
                if (qp->n_act > 0) mu_pc /= qp->n_act; /* (as the device kernels guard it; this block only runs with n_act > 0) */
            }
            if (o->cond_pred_corr && mu_pc > 2.0 * mu)
            {
                /* drop the second-order term, keep centering */
                for (int k = 0; k <= N; k++)
                {
                    stg *s = qp->s + k;
                    for (int i = 0; i < s->nct; i++) rmc[k][i] = s->act[i] ? s->rm[i] - sigma * mu : 0.0;
                    stage_condense(s, rmc[k], 0, o->reg_prim);
                }
                riccati_backward(qp, 0);
                riccati_forward(qp);
                expand_step(qp, rm_ptr);
                alpha = step_length(qp);
            }
        }
        /* no inequality rows: the Newton step solves the QP, take it fully (HPIPM solves the
         * unconstrained KKT system once in that case) */
        /* step to the boundary as HPIPM's update scales it (upstream knowledge of UPDATE_VAR_QP; the constant 0.995 of older versions
         * until the end of round 5: same solutions, 2 - 5 % more iterations): close to 0.99 alpha for short steps, the full step at 1 */
        double a = qp->n_act > 0 ? (alpha < 1.0 ? alpha * ((1.0 - alpha) * 0.99 + alpha * 0.9999999) : alpha) : 1.0;
        for (int k = 0; k <= N; k++)
        {
            stg *s = qp->s + k;
            for (int i = 0; i < s->n + 2 * s->ns; i++) s->ux[i] += a * s->dux[i];
            for (int i = 0; i < s->nx1; i++) s->pi[i] += a * s->dpi[i];
            for (int i = 0; i < s->nct; i++)
            {
                if (!s->act[i]) continue;
                s->lam[i] += a * s->dlam[i];
                s->t[i] += a * s->dt[i];
                if (s->lam[i] < o->lam_min) s->lam[i] = o->lam_min;
                if (s->t[i] < o->t_min) s->t[i] = o->t_min;
            }
        }
        compute_res(qp, tau_eff(o), &mu, nrm);
        it++;
        if (it < qp->stat_rows)
        {
            st = qp->stat + STAT_M * it;
            memset(st, 0, sizeof(double) * STAT_M);
            st[0] = alpha_aff; st[1] = alpha_aff; st[2] = mu_aff; st[3] = sigma;
            st[4] = alpha; st[5] = alpha; st[6] = mu;
            st[7] = nrm[0]; st[8] = nrm[1]; st[9] = nrm[2]; st[10] = nrm[3];
        }
        if (o->print_level > 0)
            printf("it %3d a_aff %.3e sig %.3e a %.3e mu %.3e res %.3e %.3e %.3e %.3e\n", it, alpha_aff, sigma,
                   alpha, mu, nrm[0], nrm[1], nrm[2], nrm[3]);
    }
    finalize_sol(qp);
    qp->iter = it;
    qp->status = status;
    if (N + 1 > 1024) { free(rmc); free((void *) rm_ptr); }
    return status;
}

void oqp_refactor(oqp *qp, const oqp_opts *o)
{
    double mu, nrm[4];
    setup_active(qp);
    compute_res(qp, tau_eff(o), &mu, nrm);
    for (int k = 0; k <= qp->N; k++) stage_condense(qp->s + k, qp->s[k].rm, 1, o->reg_prim);
    riccati_backward(qp, 1);
}

/* a second handle with the same data (bench.py cpu_baseline: enough independent solves per thread without building
 * every instance through the Python setters) */
oqp *oqp_clone(const oqp *src)
{
    int N = src->N;
    int *nx = iz(N + 1), *nu = iz(N + 1), *nbx = iz(N + 1), *nbu = iz(N + 1), *ng = iz(N + 1), *ns = iz(N + 1);
    for (int k = 0; k <= N; k++)
    {
        const stg *s = src->s + k;
        nx[k] = s->nx; nu[k] = s->nu; nbx[k] = s->nbx; nbu[k] = s->nbu; ng[k] = s->ng; ns[k] = s->ns;
    }
    oqp *qp = oqp_create(N, nx, nu, nbx, nbu, ng, ns);
    free(nx); free(nu); free(nbx); free(nbu); free(ng); free(ns);
    for (int k = 0; k <= N; k++)
    {
        const stg *s = src->s + k;
        stg *d = qp->s + k;
        d->nbxe = s->nbxe;
#define CP(f, cnt) memcpy(d->f, s->f, sizeof(*d->f) * (size_t) (cnt))
        CP(A, s->nx1 * s->nx); CP(B, s->nx1 * s->nu); CP(b, s->nx1); CP(Q, s->nx * s->nx); CP(S, s->nu * s->nx);
        CP(R, s->nu * s->nu); CP(q, s->nx); CP(r, s->nu); CP(idxb, s->nb); CP(lb, s->nb); CP(ub, s->nb);
        CP(lb_mask, s->nb); CP(ub_mask, s->nb); CP(C, s->ng * s->nx); CP(D, s->ng * s->nu); CP(lg, s->ng); CP(ug, s->ng);
        CP(lg_mask, s->ng); CP(ug_mask, s->ng); CP(Zl, s->ns); CP(Zu, s->ns); CP(zl, s->ns); CP(zu, s->ns);
        CP(lls, s->ns); CP(lus, s->ns); CP(lls_mask, s->ns); CP(lus_mask, s->ns); CP(idxs_rev, s->nbg); CP(idxe, s->nb);
#undef CP
    }
    return qp;
}

int oqp_solve_batch(oqp **qps, int n, const oqp_opts *opts, int *status, int nthreads)
{
    int bad = 0;
    (void) nthreads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) reduction(+ : bad) schedule(static)
#endif
    for (int i = 0; i < n; i++)
    {
        status[i] = oqp_solve(qps[i], opts);
        bad += status[i] != 0;
    }
    return bad;
}
