/*
 * TEST INFRASTRUCTURE ONLY -- drives integration/ocp_qp_gpu_ipm.c the way acados drives an inner QP plugin, on QP
 * data held in (mock) HPIPM / BLASFEO storage.
 *
 *   driver <qp.txt> <out.txt> [repeat]
 *
 * qp.txt (written by tests/test_mock_acados.py from an AcadosOcpQp): "N", then per stage "dims k nx nu nbx nbu ng ns nbxe",
 * then lines "<field> <k> <n> v0 v1 ..." with column-major matrices and natural-sign bounds.  The driver packs them the
 * way acados' setters do -- BAbt = [B'; A'; b'], RSQrq lower triangle + [r' q'] row, DCt = [D'; C'],
 * d = [lb; lg; -ub; -ug; ls; us], rqz = [r; q; zl; zu] (print.c:220-429, ocp_qp_common.c:897-906) -- in panel-major
 * storage, POISONS what a plugin must not read (last rows of BAbt / RSQrq, strict upper triangle of RSQrq), calls the 17
 * slots in acados' order (sizes -> assign -> defaults -> opts_set -> evaluate; ocp_qp_interface.c:513-563) and writes the
 * solution.  With `repeat` it changes ONLY the vectors b / rqz / d between two evaluates (what ocp_nlp does every SQP
 * iteration) and writes the second solution.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "acados/ocp_qp/ocp_qp_common.h"

void ocp_qp_gpu_ipm_acados_config_initialize_default(void *config_);

#define MAXN 128
static int N;
static struct d_ocp_qp_dim dim;
static struct d_ocp_qp qp;
static struct d_ocp_qp_sol sol;
static qp_info info;

static double *readvec(FILE *f, int n)
{
    double *v = (double *) calloc(n > 0 ? n : 1, sizeof(double));
    for (int i = 0; i < n; i++) if (fscanf(f, "%lf", v + i) != 1) { fprintf(stderr, "driver: short vector\n"); exit(2); }
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "r");
    if (!f || fscanf(f, "%d", &N) != 1) return 2;
    int *arr[10];
    for (int q = 0; q < 10; q++) arr[q] = (int *) calloc(N + 1, sizeof(int));
    dim.nx = arr[0]; dim.nu = arr[1]; dim.nb = arr[2]; dim.nbx = arr[3]; dim.nbu = arr[4]; dim.ng = arr[5]; dim.ns = arr[6];
    dim.nbxe = arr[7]; dim.nbue = arr[8]; dim.nge = arr[9]; dim.N = N;
    qp.dim = &dim; sol.dim = &dim;
    qp.BAbt = calloc(N + 1, sizeof(struct blasfeo_dmat)); qp.RSQrq = calloc(N + 1, sizeof(struct blasfeo_dmat));
    qp.DCt = calloc(N + 1, sizeof(struct blasfeo_dmat));
    qp.b = calloc(N + 1, sizeof(struct blasfeo_dvec)); qp.rqz = calloc(N + 1, sizeof(struct blasfeo_dvec));
    qp.d = calloc(N + 1, sizeof(struct blasfeo_dvec)); qp.d_mask = calloc(N + 1, sizeof(struct blasfeo_dvec));
    qp.m = calloc(N + 1, sizeof(struct blasfeo_dvec)); qp.Z = calloc(N + 1, sizeof(struct blasfeo_dvec));
    qp.idxb = calloc(N + 1, sizeof(int *)); qp.idxs_rev = calloc(N + 1, sizeof(int *)); qp.idxe = calloc(N + 1, sizeof(int *));
    qp.diag_H_flag = calloc(N + 1, sizeof(int));
    sol.ux = calloc(N + 1, sizeof(struct blasfeo_dvec)); sol.pi = calloc(N + 1, sizeof(struct blasfeo_dvec));
    sol.lam = calloc(N + 1, sizeof(struct blasfeo_dvec)); sol.t = calloc(N + 1, sizeof(struct blasfeo_dvec));
    sol.misc = &info;

    char field[64];
    int k, n;
    int allocated = 0;
    while (fscanf(f, "%63s %d", field, &k) == 2)
    {
        if (!strcmp(field, "dims"))
        {
            if (fscanf(f, "%d %d %d %d %d %d %d", &dim.nx[k], &dim.nu[k], &dim.nbx[k], &dim.nbu[k], &dim.ng[k], &dim.ns[k], &dim.nbxe[k]) != 7) return 2;
            dim.nb[k] = dim.nbx[k] + dim.nbu[k];
            continue;
        }
        if (!allocated)
        {
            for (int s = 0; s <= N; s++)
            {
                const int nu = dim.nu[s], nx = dim.nx[s], nx1 = s < N ? dim.nx[s + 1] : 0, nb = dim.nb[s], ng = dim.ng[s], ns = dim.ns[s];
                const int nct = 2 * (nb + ng + ns);
                blasfeo_allocate_dmat(nu + nx + 1, nx1, qp.BAbt + s);
                blasfeo_allocate_dmat(nu + nx + 1, nu + nx, qp.RSQrq + s);
                blasfeo_allocate_dmat(nu + nx, ng, qp.DCt + s);
                blasfeo_allocate_dvec(nx1, qp.b + s); blasfeo_allocate_dvec(nu + nx + 2 * ns, qp.rqz + s);
                blasfeo_allocate_dvec(nct, qp.d + s); blasfeo_allocate_dvec(nct, qp.d_mask + s); blasfeo_allocate_dvec(nct, qp.m + s);
                blasfeo_allocate_dvec(2 * ns, qp.Z + s);
                blasfeo_dvecse(nct, 1.0, qp.d_mask + s, 0);
                qp.idxb[s] = calloc(nb + 1, sizeof(int)); qp.idxs_rev[s] = calloc(nb + ng + 1, sizeof(int)); qp.idxe[s] = calloc(nb + 1, sizeof(int));
                for (int e = 0; e < dim.nbu[s]; e++) qp.idxb[s][e] = e;
                for (int e = 0; e < dim.nbx[s]; e++) qp.idxb[s][dim.nbu[s] + e] = nu + e;
                for (int e = 0; e < nb + ng; e++) qp.idxs_rev[s][e] = -1;
                blasfeo_allocate_dvec(nu + nx + 2 * ns, sol.ux + s); blasfeo_allocate_dvec(nx1, sol.pi + s);
                blasfeo_allocate_dvec(nct, sol.lam + s); blasfeo_allocate_dvec(nct, sol.t + s);
                /* what a plugin must NOT read: last rows of BAbt / RSQrq and the strict upper triangle of RSQrq */
                for (int c = 0; c < nx1; c++) BLASFEO_DMATEL(qp.BAbt + s, nu + nx, c) = 1e30;
                for (int c = 0; c < nu + nx; c++)
                {
                    BLASFEO_DMATEL(qp.RSQrq + s, nu + nx, c) = -1e30;
                    for (int r = 0; r < c; r++) BLASFEO_DMATEL(qp.RSQrq + s, r, c) = 7e77;
                }
            }
            allocated = 1;
        }
        if (fscanf(f, "%d", &n) != 1) return 2;
        const int nu = dim.nu[k], nx = dim.nx[k], nx1 = k < N ? dim.nx[k + 1] : 0, nbu = dim.nbu[k], nb = dim.nb[k], ng = dim.ng[k], ns = dim.ns[k];
        if (!strcmp(field, "idxb") || !strcmp(field, "idxs_rev") || !strcmp(field, "idxe"))
        {
            int *dst = !strcmp(field, "idxb") ? qp.idxb[k] : !strcmp(field, "idxs_rev") ? qp.idxs_rev[k] : qp.idxe[k];
            for (int e = 0; e < n; e++) if (fscanf(f, "%d", dst + e) != 1) return 2;
            continue;
        }
        double *v = readvec(f, n);
        /* d_ocp_qp_set_* semantics */
        if (!strcmp(field, "A")) blasfeo_pack_tran_dmat(nx1, nx, v, nx1, qp.BAbt + k, nu, 0);
        else if (!strcmp(field, "B")) blasfeo_pack_tran_dmat(nx1, nu, v, nx1, qp.BAbt + k, 0, 0);
        else if (!strcmp(field, "b")) blasfeo_pack_dvec(nx1, v, 1, qp.b + k, 0);
        else if (!strcmp(field, "Q")) { for (int c = 0; c < nx; c++) for (int r = c; r < nx; r++) BLASFEO_DMATEL(qp.RSQrq + k, nu + r, nu + c) = v[r + nx * c]; }
        else if (!strcmp(field, "R")) { for (int c = 0; c < nu; c++) for (int r = c; r < nu; r++) BLASFEO_DMATEL(qp.RSQrq + k, r, c) = v[r + nu * c]; }
        else if (!strcmp(field, "S")) blasfeo_pack_tran_dmat(nu, nx, v, nu, qp.RSQrq + k, nu, 0); /* S (nu x nx) stored as S' in the lower-left block */
        else if (!strcmp(field, "r")) blasfeo_pack_dvec(nu, v, 1, qp.rqz + k, 0);
        else if (!strcmp(field, "q")) blasfeo_pack_dvec(nx, v, 1, qp.rqz + k, nu);
        else if (!strcmp(field, "zl")) blasfeo_pack_dvec(ns, v, 1, qp.rqz + k, nu + nx);
        else if (!strcmp(field, "zu")) blasfeo_pack_dvec(ns, v, 1, qp.rqz + k, nu + nx + ns);
        else if (!strcmp(field, "Zl")) blasfeo_pack_dvec(ns, v, 1, qp.Z + k, 0);
        else if (!strcmp(field, "Zu")) blasfeo_pack_dvec(ns, v, 1, qp.Z + k, ns);
        else if (!strcmp(field, "C")) blasfeo_pack_tran_dmat(ng, nx, v, ng, qp.DCt + k, nu, 0);
        else if (!strcmp(field, "D")) blasfeo_pack_tran_dmat(ng, nu, v, ng, qp.DCt + k, 0, 0);
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
                if (!strcmp(field, names[q])) { for (int e = 0; e < n; e++) BLASFEO_DVECEL(qp.d + k, off[q] + e) = sgn[q] * v[e]; hit = 1; }
                else if (!strcmp(field, mname)) { for (int e = 0; e < n; e++) BLASFEO_DVECEL(qp.d_mask + k, off[q] + e) = v[e]; hit = 1; }
            }
            if (!hit) { fprintf(stderr, "driver: unknown field %s\n", field); return 2; }
        }
        free(v);
    }
    fclose(f);

    /* ---- the plugin, driven through its 17 slots in acados' order ---- */
    qp_solver_config config;
    memset(&config, 0, sizeof(config));
    ocp_qp_gpu_ipm_acados_config_initialize_default(&config);
    void **slots = (void **) &config;
    for (int q = 0; q < 17; q++) if (!slots[q]) { fprintf(stderr, "driver: slot %d empty\n", q); return 3; }
    void *opts = config.opts_assign(&config, &dim, calloc(1, config.opts_calculate_size(&config, &dim)));
    config.opts_initialize_default(&config, &dim, opts);
    double tol = 1e-8;
    int itmax = 50, pl = 0, ws = 0;
    config.opts_set(&config, opts, "tol_stat", &tol); config.opts_set(&config, opts, "tol_eq", &tol);
    config.opts_set(&config, opts, "tol_ineq", &tol); config.opts_set(&config, opts, "tol_comp", &tol);
    config.opts_set(&config, opts, "iter_max", &itmax); config.opts_set(&config, opts, "print_level", &pl);
    config.opts_set(&config, opts, "warm_start", &ws);
    config.opts_update(&config, &dim, opts);
    void *mem = config.memory_assign(&config, &dim, opts, calloc(1, config.memory_calculate_size(&config, &dim, opts)));
    void *work = calloc(1, config.workspace_calculate_size(&config, &dim, opts) + 8);

    int status = config.evaluate(&config, &qp, &sol, opts, mem, work);
    if (argc > 3)
    {
        /* what ocp_nlp changes between two SQP iterations: ONLY the vectors (ocp_nlp_common.c:3119-3138) */
        for (int s = 0; s <= N; s++)
        {
            for (int e = 0; e < dim.nu[s] + dim.nx[s]; e++) BLASFEO_DVECEL(qp.rqz + s, e) += 0.05 * ((e + s) % 3 - 1);
            if (s < N) for (int e = 0; e < dim.nx[s + 1]; e++) BLASFEO_DVECEL(qp.b + s, e) += 0.01 * ((e + 2 * s) % 3 - 1);
        }
        status = config.evaluate(&config, &qp, &sol, opts, mem, work);
    }
    int iter = -1, st2 = -1;
    config.memory_get(&config, mem, "iter", &iter);
    config.memory_get(&config, mem, "status", &st2);

    FILE *g = fopen(argv[2], "w");
    fprintf(g, "status %d %d iter %d %d t_computed %d\n", status, st2, iter, info.num_iter, info.t_computed);
    for (int s = 0; s <= N; s++)
    {
        const int nv = dim.nu[s] + dim.nx[s] + 2 * dim.ns[s], nx1 = s < N ? dim.nx[s + 1] : 0, nct = 2 * (dim.nb[s] + dim.ng[s] + dim.ns[s]);
        fprintf(g, "ux %d", s); for (int e = 0; e < nv; e++) fprintf(g, " %.17g", BLASFEO_DVECEL(sol.ux + s, e)); fprintf(g, "\n");
        fprintf(g, "pi %d", s); for (int e = 0; e < nx1; e++) fprintf(g, " %.17g", BLASFEO_DVECEL(sol.pi + s, e)); fprintf(g, "\n");
        fprintf(g, "lam %d", s); for (int e = 0; e < nct; e++) fprintf(g, " %.17g", BLASFEO_DVECEL(sol.lam + s, e)); fprintf(g, "\n");
        fprintf(g, "t %d", s); for (int e = 0; e < nct; e++) fprintf(g, " %.17g", BLASFEO_DVECEL(sol.t + s, e)); fprintf(g, "\n");
    }
    /* Riccati gains through the solver_get slot (ocp_nlp_ddp.c:373-377) */
    {
        const int nu = dim.nu[1], nx = dim.nx[1];
        double *K = calloc(nu * nx + 1, sizeof(double));
        config.solver_get(&config, &qp, &sol, opts, mem, "K", 1, K, nu, nx);
        fprintf(g, "K 1"); for (int e = 0; e < nu * nx; e++) fprintf(g, " %.17g", K[e]); fprintf(g, "\n");
    }
    fclose(g);
    config.terminate(&config, mem, work);
    return 0;
}
