/*
 * TEST INFRASTRUCTURE ONLY -- the reference's OWN orchestration around this repository's plugin.
 *
 * Linked in, compiled UNMODIFIED from /root/reference: acados/ocp_qp/ocp_qp_xcond_solver.c (the 22-slot solver acados' ocp_nlp
 * holds: dims / opts routing / memory carving / ocp_qp_xcond_solve :529-587 / condense_lhs + condense_rhs_and_solve :591-669 /
 * eval_forw_sens :672-700 / memory_get), acados/ocp_qp/ocp_qp_common.c (containers, ocp_qp_compute_t :874-921,
 * ocp_qp_res_compute / _nrm_inf :559-667), acados/utils/mem.c, timing.c.  HPIPM / BLASFEO: tests/mock_hpipm (stand-ins).
 * config->qp_solver = integration/ocp_qp_gpu_ipm.c (this repository), config->xcond = copy_xcond.c (N2 = N).
 *
 *   ref_xcond_driver <qp.txt> <out.txt>
 * out.txt: "status .. iter .. iter_info .. t_computed .. status_mem .. rti_status .. t_max_diff .. res g b d m", then the solution
 * lines (ux, pi, lam, t per stage) of the plain solve; the RTI-split solve and the reference's compute_t are compared inside.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "acados/ocp_qp/ocp_qp_common.h"
#include "acados/ocp_qp/ocp_qp_xcond_solver.h"
#include "acados/utils/types.h"

#include "qp_loader.h"

void ocp_qp_gpu_ipm_acados_config_initialize_default(void *config_);
void copy_xcond_config_initialize_default(void *config_);
/* acados/dense_qp/dense_qp_common.c is not linked (full condensing is not on this path); the symbol is referenced by
 * ocp_qp_xcond_solver_dims_get_ for the "fcond" module only */
void dense_qp_dims_get(void *config_, void *dims, const char *field, int *value) { printf("dense_qp_dims_get: not on this path\n"); exit(1); }

static void copy_loaded_qp(mock_capsule *c, ocp_qp_in *in)
{
    /* the QP held by the loader's containers -> the containers the REFERENCE created (same accessors acados' setters use) */
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
    if (argc < 3) return 2;
    mock_capsule *cap = mock_load_qp(argv[1]);
    const int N = cap->dim.N;

    /* config: the reference's calculate_size / assign / initialize_default, then the two sub-vtables replaced */
    ocp_qp_xcond_solver_config *config = ocp_qp_xcond_solver_config_assign(calloc(1, ocp_qp_xcond_solver_config_calculate_size()));
    ocp_qp_xcond_solver_config_initialize_default(config);
    ocp_qp_gpu_ipm_acados_config_initialize_default(config->qp_solver);
    copy_xcond_config_initialize_default(config->xcond);

    /* dims through the reference's slots (ocp_qp_interface.c:262-300 does the same) */
    ocp_qp_xcond_solver_dims *dims = config->dims_assign(config, N, calloc(1, config->dims_calculate_size(config, N)));
    const char *names[] = {"nx", "nu", "nbx", "nbu", "ng", "ns", "nbxe"};
    int *vals[] = {cap->dim.nx, cap->dim.nu, cap->dim.nbx, cap->dim.nbu, cap->dim.ng, cap->dim.ns, cap->dim.nbxe};
    for (int k = 0; k <= N; k++)
        for (int q = 0; q < 7; q++) config->dims_set(config, dims, k, names[q], &vals[q][k]);

    /* containers: the reference's own */
    ocp_qp_in *qp_in = ocp_qp_in_assign(dims->orig_dims, calloc(1, ocp_qp_in_calculate_size(dims->orig_dims)));
    ocp_qp_out *qp_out = ocp_qp_out_assign(dims->orig_dims, calloc(1, ocp_qp_out_calculate_size(dims->orig_dims)));
    ocp_qp_out *qp_out2 = ocp_qp_out_assign(dims->orig_dims, calloc(1, ocp_qp_out_calculate_size(dims->orig_dims)));
    copy_loaded_qp(cap, qp_in);

    /* opts: "cond_" strings go to the condensing module, the rest to the inner solver (ocp_qp_xcond_solver.c:283-311) */
    void *opts = config->opts_assign(config, dims, calloc(1, config->opts_calculate_size(config, dims)));
    config->opts_initialize_default(config, dims, opts);
    double tol = 1e-8;
    int itmax = 50, condN = N, ws = 0, pl = 0;
    config->opts_set(config, opts, "tol_stat", &tol); config->opts_set(config, opts, "tol_eq", &tol);
    config->opts_set(config, opts, "tol_ineq", &tol); config->opts_set(config, opts, "tol_comp", &tol);
    config->opts_set(config, opts, "iter_max", &itmax); config->opts_set(config, opts, "warm_start", &ws);
    config->opts_set(config, opts, "print_level", &pl); config->opts_set(config, opts, "cond_N", &condN);
    config->opts_update(config, dims, opts);

    void *mem = config->memory_assign(config, dims, opts, calloc(1, config->memory_calculate_size(config, dims, opts)));
    void *work = calloc(1, config->workspace_calculate_size(config, dims, opts) + 8);

    /* the solve: the reference's ocp_qp_xcond_solve */
    const int status = config->evaluate(config, dims, qp_in, qp_out, opts, mem, work);
    qp_info *info = (qp_info *) qp_out->misc;
    int iter = -1, st_mem = -1;
    double t_call = -1.0;
    config->memory_get(config, mem, "iter", &iter);
    config->memory_get(config, mem, "status", &st_mem);
    config->memory_get(config, mem, "time_qp_solver_call", &t_call);

    /* the reference's ocp_qp_compute_t on the plugin's (ux): must reproduce the plugin's t on every unmasked side */
    ocp_qp_out_copy(qp_out, qp_out2);
    ocp_qp_compute_t(qp_in, qp_out2);
    double t_diff = 0.0;
    for (int k = 0; k <= N; k++)
    {
        const int nct = 2 * (cap->dim.nb[k] + cap->dim.ng[k] + cap->dim.ns[k]);
        for (int i = 0; i < nct; i++)
            if (BLASFEO_DVECEL(qp_in->d_mask + k, i) != 0.0)
                t_diff = fmax(t_diff, fabs(BLASFEO_DVECEL(qp_out2->t + k, i) - BLASFEO_DVECEL(qp_out->t + k, i)));
    }
    /* the reference's residual entry (ocp_qp_common.c:559-667; HPIPM's d_ocp_qp_res_compute behind it is the stand-in's) */
    ocp_qp_res *res = ocp_qp_res_assign(dims->orig_dims, calloc(1, ocp_qp_res_calculate_size(dims->orig_dims)));
    ocp_qp_res_ws *res_ws = ocp_qp_res_workspace_assign(dims->orig_dims, calloc(1, ocp_qp_res_workspace_calculate_size(dims->orig_dims)));
    double nrm[4];
    ocp_qp_res_compute(qp_in, qp_out, res, res_ws);
    ocp_qp_res_compute_nrm_inf(res, nrm);

    /* RTI split through the reference's two entries: same solution */
    ocp_qp_out *qp_out3 = ocp_qp_out_assign(dims->orig_dims, calloc(1, ocp_qp_out_calculate_size(dims->orig_dims)));
    int rti = config->condense_lhs(config, dims, qp_in, qp_out3, opts, mem, work);
    if (rti == 0) rti = config->condense_rhs_and_solve(config, dims, qp_in, qp_out3, opts, mem, work);
    double rti_diff = 0.0;
    for (int k = 0; k <= N; k++)
        for (int i = 0; i < cap->dim.nu[k] + cap->dim.nx[k] + 2 * cap->dim.ns[k]; i++)
            rti_diff = fmax(rti_diff, fabs(BLASFEO_DVECEL(qp_out3->ux + k, i) - BLASFEO_DVECEL(qp_out->ux + k, i)));

    FILE *g = fopen(argv[2], "w");
    fprintf(g, "status %d iter %d iter_info %d t_computed %d status_mem %d rti_status %d\n", status, iter, info->num_iter, info->t_computed, st_mem, rti);
    fprintf(g, "checks t_max_diff %.17g rti_max_diff %.17g res %.17g %.17g %.17g %.17g time_call %.6g total %.6g\n", t_diff, rti_diff, nrm[0], nrm[1],
            nrm[2], nrm[3], t_call, info->total_time);
    for (int k = 0; k <= N; k++)
    {
        const int nv = cap->dim.nu[k] + cap->dim.nx[k] + 2 * cap->dim.ns[k], nx1 = k < N ? cap->dim.nx[k + 1] : 0;
        const int nct = 2 * (cap->dim.nb[k] + cap->dim.ng[k] + cap->dim.ns[k]);
        fprintf(g, "ux %d %d", k, nv); for (int i = 0; i < nv; i++) fprintf(g, " %.17g", BLASFEO_DVECEL(qp_out->ux + k, i)); fprintf(g, "\n");
        fprintf(g, "pi %d %d", k, nx1); for (int i = 0; i < nx1; i++) fprintf(g, " %.17g", BLASFEO_DVECEL(qp_out->pi + k, i)); fprintf(g, "\n");
        fprintf(g, "lam %d %d", k, nct); for (int i = 0; i < nct; i++) fprintf(g, " %.17g", BLASFEO_DVECEL(qp_out->lam + k, i)); fprintf(g, "\n");
        fprintf(g, "t %d %d", k, nct); for (int i = 0; i < nct; i++) fprintf(g, " %.17g", BLASFEO_DVECEL(qp_out->t + k, i)); fprintf(g, "\n");
    }
    fclose(g);
    config->terminate(config, mem, work);
    return 0;
}
