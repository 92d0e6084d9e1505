/*
 * TEST INFRASTRUCTURE ONLY -- the reference's unit test of its QP solvers (test/ocp_qp/test_qpsolvers.cpp:117-268, "mass spring
 * example", the SPARSE-solver branch with N2 in {15, 5, 3}) restated in C around the solver this repository REGISTERS:
 *
 *     plan.qp_solver = PARTIAL_CONDENSING_GPU_IPM;                                   (or ..._create_from_name("PARTIAL_CONDENSING_GPU_IPM"))
 *     config = ocp_qp_xcond_solver_config_create(plan);      -> the PATCHED switch of interfaces/acados_c/ocp_qp_interface.c:91-182
 *     dims   = ocp_qp_xcond_solver_dims_create(config, N) + ocp_qp_xcond_solver_dims_set(...)
 *     opts   = ocp_qp_xcond_solver_opts_create(config, dims); ocp_qp_xcond_solver_opts_set(config, opts, "cond_N", &N2);
 *     solver = ocp_qp_create(config, dims, opts); status = ocp_qp_solve(solver, qp_in, qp_out);
 *     ocp_qp_inf_norm_residuals(dims->orig_dims, qp_in, qp_out, res);                REQUIRE(status == 0); REQUIRE(max(res) <= tol);
 *
 * Compiled from a PATCHED COPY of the reference files (integration/acados.patch applied to interfaces/acados_c/ocp_qp_interface.{c,h}
 * and acados/ocp_qp/ocp_qp_xcond_solver.c) with -DACADOS_WITH_GPU_IPM, linked with the reference's unmodified ocp_qp_common.c /
 * utils and the two plugin files (integration/ocp_qp_gpu_ipm.c, ocp_qp_gpu_pcond.c).  HPIPM / BLASFEO: tests/mock_hpipm; the
 * initialisers of the reference's OTHER solvers, which the switch references, are abort stubs (acados_c_stubs.c).
 *
 *   acados_c_driver <qp.txt> <out.txt> <N2> [<N2> ...] [--by-name]
 * out.txt: per N2 a line "N2 .. status .. iter .. xcond_N .. res g b d m" followed by the solution lines of that solve.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "acados/ocp_qp/ocp_qp_common.h"
#include "acados/ocp_qp/ocp_qp_xcond_solver.h"
#include "acados/utils/types.h"
#include "acados_c/ocp_qp_interface.h"

#include "qp_loader.h"

static void copy_loaded_qp(mock_capsule *c, ocp_qp_in *in)
{
    struct d_ocp_qp_dim *d = &c->dim;
    for (int k = 0; k <= d->N; k++)
    {
        const int nv = d->nu[k] + d->nx[k], nx1 = k < d->N ? d->nx[k + 1] : 0, nb = d->nb[k], ng = d->ng[k], ns = d->ns[k], nct = 2 * (nb + ng + ns);
        for (int j = 0; j < nx1; j++) for (int i = 0; i <= nv; i++) BLASFEO_DMATEL(in->BAbt + k, i, j) = BLASFEO_DMATEL(c->qp.BAbt + k, i, j);
        for (int j = 0; j < nv; j++) for (int i = 0; i <= nv; i++) BLASFEO_DMATEL(in->RSQrq + k, i, j) = BLASFEO_DMATEL(c->qp.RSQrq + k, i, j);
        for (int j = 0; j < ng; j++) for (int i = 0; i < nv; i++) BLASFEO_DMATEL(in->DCt + k, i, j) = BLASFEO_DMATEL(c->qp.DCt + k, i, j);
        for (int i = 0; i < nx1; i++) BLASFEO_DVECEL(in->b + k, i) = BLASFEO_DVECEL(c->qp.b + k, i);
        for (int i = 0; i < nv + 2 * ns; i++) BLASFEO_DVECEL(in->rqz + k, i) = BLASFEO_DVECEL(c->qp.rqz + k, i);
        for (int i = 0; i < nct; i++)
        {
            BLASFEO_DVECEL(in->d + k, i) = BLASFEO_DVECEL(c->qp.d + k, i);
            BLASFEO_DVECEL(in->d_mask + k, i) = BLASFEO_DVECEL(c->qp.d_mask + k, i);
            BLASFEO_DVECEL(in->m + k, i) = 0.0;
        }
        for (int i = 0; i < 2 * ns; i++) BLASFEO_DVECEL(in->Z + k, i) = BLASFEO_DVECEL(c->qp.Z + k, i);
        memcpy(in->idxb[k], c->qp.idxb[k], sizeof(int) * (size_t) nb);
        memcpy(in->idxs_rev[k], c->qp.idxs_rev[k], sizeof(int) * (size_t) (nb + ng));
        memcpy(in->idxe[k], c->qp.idxe[k], sizeof(int) * (size_t) d->nbxe[k]);
    }
}

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    mock_capsule *cap = mock_load_qp(argv[1]);
    const int N = cap->dim.N;
    int by_name = 0;
    for (int a = 3; a < argc; a++) if (!strcmp(argv[a], "--by-name")) by_name = 1;
    FILE *g = fopen(argv[2], "w");
    for (int a = 3; a < argc; a++)
    {
        if (argv[a][0] == '-') continue;
        int N2 = atoi(argv[a]);
        /* config: the plan -> the patched switch; or the name -> the patched string table */
        ocp_qp_xcond_solver_config *config;
        if (by_name) config = ocp_qp_xcond_solver_config_create_from_name("PARTIAL_CONDENSING_GPU_IPM");
        else
        {
            ocp_qp_solver_plan_t plan;
            plan.qp_solver = PARTIAL_CONDENSING_GPU_IPM;
            config = ocp_qp_xcond_solver_config_create(plan);
        }
        ocp_qp_xcond_solver_dims *dims = ocp_qp_xcond_solver_dims_create(config, N);
        const char *names[] = {"nx", "nu", "nbx", "nbu", "ng", "ns", "nbxe"};
        int *vals[] = {cap->dim.nx, cap->dim.nu, cap->dim.nbx, cap->dim.nbu, cap->dim.ng, cap->dim.ns, cap->dim.nbxe};
        for (int k = 0; k <= N; k++)
            for (int q = 0; q < 7; q++) ocp_qp_xcond_solver_dims_set(config, dims, k, names[q], &vals[q][k]);
        ocp_qp_in *qp_in = ocp_qp_in_create(dims->orig_dims);
        ocp_qp_out *qp_out = ocp_qp_out_create(dims->orig_dims);
        copy_loaded_qp(cap, qp_in);
        void *opts = ocp_qp_xcond_solver_opts_create(config, dims);
        double tol = 1e-8;
        int itmax = 50;
        ocp_qp_xcond_solver_opts_set(config, opts, "tol_stat", &tol); ocp_qp_xcond_solver_opts_set(config, opts, "tol_eq", &tol);
        ocp_qp_xcond_solver_opts_set(config, opts, "tol_ineq", &tol); ocp_qp_xcond_solver_opts_set(config, opts, "tol_comp", &tol);
        ocp_qp_xcond_solver_opts_set(config, opts, "iter_max", &itmax);
        ocp_qp_xcond_solver_opts_set(config, opts, "cond_N", &N2); /* set_N2 of the unit test (:98-112) */
        ocp_qp_solver *solver = ocp_qp_create(config, dims, opts);
        const int status = ocp_qp_solve(solver, qp_in, qp_out);
        double res[4];
        ocp_qp_inf_norm_residuals(dims->orig_dims, qp_in, qp_out, res);
        int iter = -1;
        double t_tot = -1.0;
        ocp_qp_xcond_solver_get_scalar(solver, qp_out, "iter", &iter);
        ocp_qp_xcond_solver_get_scalar(solver, qp_out, "time_tot", &t_tot);
        ocp_qp_dims *xd = NULL;
        config->xcond->dims_get(config->xcond, dims->xcond_dims, "xcond_dims", &xd);
        /* the per-iteration statistics through the reference's ocp_qp_solver_get_stats (ocp_qp_interface.c:612-625: memory_get "iter",
         * "stat", "stat_m"); the last row holds the residuals the solver stopped on */
        double *stats = calloc((size_t) 20 * (size_t) (itmax + 2), sizeof(double));
        int stat_m = 0;
        ocp_qp_solver_get_stats(solver, stats, "PARTIAL_CONDENSING_GPU_IPM");
        solver->config->memory_get(solver->config, solver->mem, "stat_m", &stat_m);
        const double *last = stats + (size_t) stat_m * (size_t) iter;
        fprintf(g, "N2 %d status %d iter %d xcond_N %d res %.17g %.17g %.17g %.17g time_tot %.6g stat_m %d stat_last %.17g %.17g %.17g %.17g %.17g\n", N2, status,
                iter, xd->N, res[0], res[1], res[2], res[3], t_tot, stat_m, last[6], last[7], last[8], last[9], last[10]);
        free(stats);
        mock_write_sol(g, &cap->dim, qp_out);
        /* the PATCHED terminate of the reference's outer solver releases the condensing module's device batch */
        config->terminate(config, solver->mem, solver->work);
        ocp_qp_solver_destroy(solver);
        ocp_qp_xcond_solver_opts_free(opts);
        ocp_qp_out_free(qp_out);
        ocp_qp_in_free(qp_in);
        ocp_qp_xcond_solver_dims_free(dims);
        ocp_qp_xcond_solver_config_free(config);
    }
    fclose(g);
    return 0;
}
