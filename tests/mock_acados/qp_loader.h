/*
 * TEST INFRASTRUCTURE ONLY -- reads a QP written by tests/test_mock_acados.py into (mock) HPIPM / BLASFEO storage the
 * way acados' setters pack it: BAbt = [B'; A'; b'], RSQrq lower triangle + [r' q'] row, DCt = [D'; C'],
 * d = [lb; lg; -ub; -ug; ls; us], rqz = [r; q; zl; zu] (print.c:220-429, ocp_qp_common.c:897-906), panel-major, and
 * POISONS what a plugin must not read (last rows of BAbt / RSQrq, strict upper triangle of RSQrq).
 *
 * qp.txt: "N", then per stage "dims k nx nu nbx nbu ng ns nbxe", then lines "<field> <k> <n> v0 v1 ..." with column-major
 * matrices and natural-sign bounds.
 */
#ifndef MOCK_QP_LOADER_H_
#define MOCK_QP_LOADER_H_

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "acados/ocp_qp/ocp_qp_common.h"

typedef struct
{
    struct d_ocp_qp_dim dim;
    struct d_ocp_qp qp;
    struct d_ocp_qp_sol sol, sens; /* solution, and a second container for sensitivities */
    struct d_ocp_qp_seed seed;
    qp_info info, sens_info;
} mock_capsule;

static double *mock_readvec(FILE *f, int n)
{
    double *v = (double *) calloc(n > 0 ? n : 1, sizeof(double));
    for (int i = 0; i < n; i++) if (fscanf(f, "%lf", v + i) != 1) { fprintf(stderr, "driver: short vector\n"); exit(2); }
    return v;
}

static void mock_alloc_sol(struct d_ocp_qp_dim *dim, struct d_ocp_qp_sol *sol)
{
    const int N = dim->N;
    sol->dim = dim;
    sol->ux = calloc(N + 1, sizeof(struct blasfeo_dvec)); sol->pi = calloc(N + 1, sizeof(struct blasfeo_dvec));
    sol->lam = calloc(N + 1, sizeof(struct blasfeo_dvec)); sol->t = calloc(N + 1, sizeof(struct blasfeo_dvec));
    for (int s = 0; s <= N; s++)
    {
        const int nx1 = s < N ? dim->nx[s + 1] : 0, nct = 2 * (dim->nb[s] + dim->ng[s] + dim->ns[s]);
        blasfeo_allocate_dvec(dim->nu[s] + dim->nx[s] + 2 * dim->ns[s], sol->ux + s); blasfeo_allocate_dvec(nx1, sol->pi + s);
        blasfeo_allocate_dvec(nct, sol->lam + s); blasfeo_allocate_dvec(nct, sol->t + s);
    }
}

static mock_capsule *mock_load_qp(const char *path)
{
    mock_capsule *c = (mock_capsule *) calloc(1, sizeof(mock_capsule));
    struct d_ocp_qp_dim *dim = &c->dim;
    struct d_ocp_qp *qp = &c->qp;
    int N;
    FILE *f = fopen(path, "r");
    if (!f || fscanf(f, "%d", &N) != 1) { fprintf(stderr, "driver: cannot read %s\n", path); exit(2); }
    int *arr[10];
    for (int q = 0; q < 10; q++) arr[q] = (int *) calloc(N + 1, sizeof(int));
    dim->nx = arr[0]; dim->nu = arr[1]; dim->nb = arr[2]; dim->nbx = arr[3]; dim->nbu = arr[4]; dim->ng = arr[5]; dim->ns = arr[6];
    dim->nbxe = arr[7]; dim->nbue = arr[8]; dim->nge = arr[9]; dim->N = N;
    qp->dim = dim;
    qp->BAbt = calloc(N + 1, sizeof(struct blasfeo_dmat)); qp->RSQrq = calloc(N + 1, sizeof(struct blasfeo_dmat));
    qp->DCt = calloc(N + 1, sizeof(struct blasfeo_dmat));
    qp->b = calloc(N + 1, sizeof(struct blasfeo_dvec)); qp->rqz = calloc(N + 1, sizeof(struct blasfeo_dvec));
    qp->d = calloc(N + 1, sizeof(struct blasfeo_dvec)); qp->d_mask = calloc(N + 1, sizeof(struct blasfeo_dvec));
    qp->m = calloc(N + 1, sizeof(struct blasfeo_dvec)); qp->Z = calloc(N + 1, sizeof(struct blasfeo_dvec));
    qp->idxb = calloc(N + 1, sizeof(int *)); qp->idxs_rev = calloc(N + 1, sizeof(int *)); qp->idxe = calloc(N + 1, sizeof(int *));
    qp->diag_H_flag = calloc(N + 1, sizeof(int));

    char field[64];
    int k, n;
    int allocated = 0;
    while (fscanf(f, "%63s %d", field, &k) == 2)
    {
        if (!strcmp(field, "dims"))
        {
            if (fscanf(f, "%d %d %d %d %d %d %d", &dim->nx[k], &dim->nu[k], &dim->nbx[k], &dim->nbu[k], &dim->ng[k], &dim->ns[k], &dim->nbxe[k]) != 7) exit(2);
            dim->nb[k] = dim->nbx[k] + dim->nbu[k];
            continue;
        }
        if (!allocated)
        {
            for (int s = 0; s <= N; s++)
            {
                const int nu = dim->nu[s], nx = dim->nx[s], nx1 = s < N ? dim->nx[s + 1] : 0, nb = dim->nb[s], ng = dim->ng[s], ns = dim->ns[s];
                const int nct = 2 * (nb + ng + ns);
                blasfeo_allocate_dmat(nu + nx + 1, nx1, qp->BAbt + s);
                blasfeo_allocate_dmat(nu + nx + 1, nu + nx, qp->RSQrq + s);
                blasfeo_allocate_dmat(nu + nx, ng, qp->DCt + s);
                blasfeo_allocate_dvec(nx1, qp->b + s); blasfeo_allocate_dvec(nu + nx + 2 * ns, qp->rqz + s);
                blasfeo_allocate_dvec(nct, qp->d + s); blasfeo_allocate_dvec(nct, qp->d_mask + s); blasfeo_allocate_dvec(nct, qp->m + s);
                blasfeo_allocate_dvec(2 * ns, qp->Z + s);
                blasfeo_dvecse(nct, 1.0, qp->d_mask + s, 0);
                qp->idxb[s] = calloc(nb + 1, sizeof(int)); qp->idxs_rev[s] = calloc(nb + ng + 1, sizeof(int)); qp->idxe[s] = calloc(nb + 1, sizeof(int));
                for (int e = 0; e < dim->nbu[s]; e++) qp->idxb[s][e] = e;
                for (int e = 0; e < dim->nbx[s]; e++) qp->idxb[s][dim->nbu[s] + e] = nu + e;
                for (int e = 0; e < nb + ng; e++) qp->idxs_rev[s][e] = -1;
                /* what a plugin must NOT read: last rows of BAbt / RSQrq and the strict upper triangle of RSQrq */
                for (int cc = 0; cc < nx1; cc++) BLASFEO_DMATEL(qp->BAbt + s, nu + nx, cc) = 1e30;
                for (int cc = 0; cc < nu + nx; cc++)
                {
                    BLASFEO_DMATEL(qp->RSQrq + s, nu + nx, cc) = -1e30;
                    for (int r = 0; r < cc; r++) BLASFEO_DMATEL(qp->RSQrq + s, r, cc) = 7e77;
                }
            }
            allocated = 1;
        }
        if (fscanf(f, "%d", &n) != 1) exit(2);
        const int nu = dim->nu[k], nx = dim->nx[k], nx1 = k < N ? dim->nx[k + 1] : 0, nbu = dim->nbu[k], nb = dim->nb[k], ng = dim->ng[k], ns = dim->ns[k];
        if (!strcmp(field, "idxb") || !strcmp(field, "idxs_rev") || !strcmp(field, "idxe"))
        {
            int *dst = !strcmp(field, "idxb") ? qp->idxb[k] : !strcmp(field, "idxs_rev") ? qp->idxs_rev[k] : qp->idxe[k];
            for (int e = 0; e < n; e++) if (fscanf(f, "%d", dst + e) != 1) exit(2);
            continue;
        }
        double *v = mock_readvec(f, n);
        /* d_ocp_qp_set_* semantics */
        if (!strcmp(field, "A")) blasfeo_pack_tran_dmat(nx1, nx, v, nx1, qp->BAbt + k, nu, 0);
        else if (!strcmp(field, "B")) blasfeo_pack_tran_dmat(nx1, nu, v, nx1, qp->BAbt + k, 0, 0);
        else if (!strcmp(field, "b")) blasfeo_pack_dvec(nx1, v, 1, qp->b + k, 0);
        else if (!strcmp(field, "Q")) { for (int cc = 0; cc < nx; cc++) for (int r = cc; r < nx; r++) BLASFEO_DMATEL(qp->RSQrq + k, nu + r, nu + cc) = v[r + nx * cc]; }
        else if (!strcmp(field, "R")) { for (int cc = 0; cc < nu; cc++) for (int r = cc; r < nu; r++) BLASFEO_DMATEL(qp->RSQrq + k, r, cc) = v[r + nu * cc]; }
        else if (!strcmp(field, "S")) blasfeo_pack_tran_dmat(nu, nx, v, nu, qp->RSQrq + k, nu, 0); /* S (nu x nx) stored as S' in the lower-left block */
        else if (!strcmp(field, "r")) blasfeo_pack_dvec(nu, v, 1, qp->rqz + k, 0);
        else if (!strcmp(field, "q")) blasfeo_pack_dvec(nx, v, 1, qp->rqz + k, nu);
        else if (!strcmp(field, "zl")) blasfeo_pack_dvec(ns, v, 1, qp->rqz + k, nu + nx);
        else if (!strcmp(field, "zu")) blasfeo_pack_dvec(ns, v, 1, qp->rqz + k, nu + nx + ns);
        else if (!strcmp(field, "Zl")) blasfeo_pack_dvec(ns, v, 1, qp->Z + k, 0);
        else if (!strcmp(field, "Zu")) blasfeo_pack_dvec(ns, v, 1, qp->Z + k, ns);
        else if (!strcmp(field, "C")) blasfeo_pack_tran_dmat(ng, nx, v, ng, qp->DCt + k, nu, 0);
        else if (!strcmp(field, "D")) blasfeo_pack_tran_dmat(ng, nu, v, ng, qp->DCt + k, 0, 0);
        else
        {
            /* bounds and masks: position in d / d_mask, sign flipped for the upper bounds (ocp_qp_common.c:897-906) */
            const char *names[] = {"lbu", "lbx", "lg", "ubu", "ubx", "ug", "lls", "lus"};
            const int off[] = {0, nbu, nb, nb + ng, nb + ng + nbu, 2 * nb + ng, 2 * nb + 2 * ng, 2 * nb + 2 * ng + ns};
            const double sgn[] = {1, 1, 1, -1, -1, -1, 1, 1};
            int hit = 0;
            for (int q = 0; q < 8; q++)
            {
                char mname[32];
                snprintf(mname, sizeof(mname), "%s_mask", names[q]);
                if (!strcmp(field, names[q])) { for (int e = 0; e < n; e++) BLASFEO_DVECEL(qp->d + k, off[q] + e) = sgn[q] * v[e]; hit = 1; }
                else if (!strcmp(field, mname)) { for (int e = 0; e < n; e++) BLASFEO_DVECEL(qp->d_mask + k, off[q] + e) = v[e]; hit = 1; }
            }
            if (!hit) { fprintf(stderr, "driver: unknown field %s\n", field); exit(2); }
        }
        free(v);
    }
    fclose(f);
    mock_alloc_sol(dim, &c->sol);
    mock_alloc_sol(dim, &c->sens);
    c->sol.misc = &c->info;
    c->sens.misc = &c->sens_info;
    /* d_ocp_qp_seed: seed_g like rqz, seed_b like b, seed_d like d (all zero) */
    c->seed.dim = dim;
    c->seed.seed_g = calloc(N + 1, sizeof(struct blasfeo_dvec)); c->seed.seed_b = calloc(N + 1, sizeof(struct blasfeo_dvec));
    c->seed.seed_d = calloc(N + 1, sizeof(struct blasfeo_dvec)); c->seed.seed_m = calloc(N + 1, sizeof(struct blasfeo_dvec));
    for (int s = 0; s <= N; s++)
    {
        const int nct = 2 * (dim->nb[s] + dim->ng[s] + dim->ns[s]);
        blasfeo_allocate_dvec(dim->nu[s] + dim->nx[s] + 2 * dim->ns[s], c->seed.seed_g + s);
        blasfeo_allocate_dvec(s < N ? dim->nx[s + 1] : 0, c->seed.seed_b + s);
        blasfeo_allocate_dvec(nct, c->seed.seed_d + s); blasfeo_allocate_dvec(nct, c->seed.seed_m + s);
    }
    return c;
}

/* instance i of a batch built from one base QP: ONLY the vectors differ (what distinct capsules of one OCP differ in
 * after linearisation at different points is more, but the adapter's plumbing sees every array of every QP anyway);
 * tests/test_mock_acados.py applies the same formulas */
static void mock_perturb(mock_capsule *c, int i)
{
    if (i == 0) return;
    const struct d_ocp_qp_dim *dim = &c->dim;
    for (int s = 0; s <= dim->N; s++)
    {
        for (int e = 0; e < dim->nu[s] + dim->nx[s]; e++) BLASFEO_DVECEL(c->qp.rqz + s, e) += 0.02 * (((e + 2 * s + 3 * i) % 7) - 3) / 3.0;
        if (s < dim->N) for (int e = 0; e < dim->nx[s + 1]; e++) BLASFEO_DVECEL(c->qp.b + s, e) += 0.005 * (((e + s + i) % 5) - 2) / 2.0;
    }
}

static void mock_write_sol(FILE *g, const struct d_ocp_qp_dim *dim, const struct d_ocp_qp_sol *sol)
{
    for (int s = 0; s <= dim->N; s++)
    {
        const int nv = dim->nu[s] + dim->nx[s] + 2 * dim->ns[s], nx1 = s < dim->N ? dim->nx[s + 1] : 0, nct = 2 * (dim->nb[s] + dim->ng[s] + dim->ns[s]);
        fprintf(g, "ux %d", s); for (int e = 0; e < nv; e++) fprintf(g, " %.17g", BLASFEO_DVECEL(sol->ux + s, e)); fprintf(g, "\n");
        fprintf(g, "pi %d", s); for (int e = 0; e < nx1; e++) fprintf(g, " %.17g", BLASFEO_DVECEL(sol->pi + s, e)); fprintf(g, "\n");
        fprintf(g, "lam %d", s); for (int e = 0; e < nct; e++) fprintf(g, " %.17g", BLASFEO_DVECEL(sol->lam + s, e)); fprintf(g, "\n");
        fprintf(g, "t %d", s); for (int e = 0; e < nct; e++) fprintf(g, " %.17g", BLASFEO_DVECEL(sol->t + s, e)); fprintf(g, "\n");
    }
}

/* binary: [ux_0 .. ux_N, pi_0 .. pi_{N-1}, lam_0 .. lam_N, t_0 .. t_N] as doubles */
static void mock_write_sol_bin(FILE *g, const struct d_ocp_qp_dim *dim, const struct d_ocp_qp_sol *sol)
{
    for (int s = 0; s <= dim->N; s++) fwrite(sol->ux[s].pa, sizeof(double), dim->nu[s] + dim->nx[s] + 2 * dim->ns[s], g);
    for (int s = 0; s < dim->N; s++) fwrite(sol->pi[s].pa, sizeof(double), dim->nx[s + 1], g);
    for (int s = 0; s <= dim->N; s++) fwrite(sol->lam[s].pa, sizeof(double), 2 * (dim->nb[s] + dim->ng[s] + dim->ns[s]), g);
    for (int s = 0; s <= dim->N; s++) fwrite(sol->t[s].pa, sizeof(double), 2 * (dim->nb[s] + dim->ng[s] + dim->ns[s]), g);
}

/* the seed of instance i, acados' way: "ex"-style +1 on BOTH sides of one x0 row (ocp_nlp_common.c:4057-4066), a
 * gradient seed in seed_g (slack entries included), a dynamics seed in seed_b, and a general-row / input-bound seed
 * with the upper part NEGATED like d (:4078-4081) */
static void mock_fill_seed(mock_capsule *c, int i)
{
    const struct d_ocp_qp_dim *dim = &c->dim;
    for (int s = 0; s <= dim->N; s++)
    {
        const int nv = dim->nu[s] + dim->nx[s] + 2 * dim->ns[s], nb = dim->nb[s], ng = dim->ng[s], nbu = dim->nbu[s];
        for (int e = 0; e < nv; e++) BLASFEO_DVECEL(c->seed.seed_g + s, e) = 0.1 * (((e + s + i) % 5) - 2);
        if (s < dim->N) for (int e = 0; e < dim->nx[s + 1]; e++) BLASFEO_DVECEL(c->seed.seed_b + s, e) = 0.05 * (((e + 2 * s + i) % 3) - 1);
        blasfeo_dvecse(2 * (nb + ng + dim->ns[s]), 0.0, c->seed.seed_d + s, 0);
        for (int e = 0; e < nbu; e++)
        {
            /* d lbu = -v, d ubu = +v in natural sign -> stored [-v ... ; -(+v)] */
            const double v = 0.01 * ((e + s + i) % 2 + 1);
            BLASFEO_DVECEL(c->seed.seed_d + s, e) = -v;
            BLASFEO_DVECEL(c->seed.seed_d + s, nb + ng + e) = -v;
        }
        for (int e = 0; e < ng; e++)
        {
            /* a nonlinear row h(x, p): d lg = d ug = -jac in natural sign -> stored [-jac; +jac] */
            const double jac = 0.02 * ((e + i) % 3 - 1);
            BLASFEO_DVECEL(c->seed.seed_d + s, nb + e) = -jac;
            BLASFEO_DVECEL(c->seed.seed_d + s, 2 * nb + ng + e) = jac;
        }
    }
    if (dim->nbxe[0] > 0)
    {
        const int row = c->qp.idxe[0][i % dim->nbxe[0]]; /* position in the bound list; nbu + index for acados' x0 rows */
        BLASFEO_DVECEL(c->seed.seed_d + 0, row) = 1.0;
        BLASFEO_DVECEL(c->seed.seed_d + 0, row + dim->nb[0] + dim->ng[0]) = 1.0;
    }
}

#endif
