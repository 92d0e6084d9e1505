/*
 * TEST INFRASTRUCTURE ONLY -- the reference's OWN orchestration around this repository's plugin.
 *
 * Linked in, compiled UNMODIFIED from /root/reference: acados/ocp_qp/ocp_qp_xcond_solver.c (the 22-slot solver acados' ocp_nlp
 * holds: dims / opts routing / memory carving / ocp_qp_xcond_solve :529-587 / condense_lhs + condense_rhs_and_solve :591-669 /
 * eval_forw_sens :672-700 / memory_get), acados/ocp_qp/ocp_qp_common.c (containers, ocp_qp_compute_t :874-921,
 * ocp_qp_res_compute / _nrm_inf :559-667), acados/utils/mem.c, timing.c.  HPIPM / BLASFEO: tests/mock_hpipm (stand-ins).
 * config->qp_solver = integration/ocp_qp_gpu_ipm.c (this repository); config->xcond = integration/ocp_qp_gpu_pcond.c, the device
 * condensing behind acados' own types (--xcond gpu, the default), or copy_xcond.c (--xcond copy: N2 = N stand-in).
 *
 *   ref_xcond_driver <qp.txt> <out.txt> [--xcond gpu|copy] [--cond-N n] [--block-size a,b,...] [--sens]
 * out.txt: "status .. iter .. iter_info .. t_computed .. status_mem .. rti_status .. xcond_N .. xcond_nu0 ..", a "checks" line
 * (t_max_diff, rti_max_diff, res g b d m, warm_iter), then the solution lines (ux, pi, lam, t per stage) of the plain solve; with
 * --sens a line "sens" and the forward sensitivities of the reference's eval_forw_sens (:672-700) on a seed built acados' way.
 * The RTI-split solve (condense_lhs / condense_rhs_and_solve with ONLY the vectors changed in between), a warm-started solve
 * (condense_qp_out) and the reference's compute_t are compared inside.
 *
 *   ref_xcond_driver batch <n> <qp.txt> <out.bin> [--cond-N n] [--block-size a,b,...] [reps]
 * n capsules, each with the reference's 22-slot solver around { ocp_qp_gpu_pcond.c, ocp_qp_gpu_ipm.c } (one memory each, created by
 * the reference's memory_assign; QP i = the base QP with vectors perturbed by mock_perturb(i)), solved by ONE call of
 * ocp_qp_gpu_xcond_solver_acados_evaluate_batch -- condensing, IPM and expansion fused on the device.  out.bin: per instance the
 * solution as raw doubles; stdout: "batch n .. ms_per_call .. status .. fused_vs_orchestrated ..", then per instance
 * "i status iter iter_info t_computed" read through the reference's memory_get.  Capsules 0 and n-1 are ALSO solved one by one
 * through the reference's ocp_qp_xcond_solve (the host round trip per capsule): fused_vs_orchestrated is the largest difference.
 */
#include <math.h>
#include <time.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "acados/ocp_qp/ocp_qp_common.h"
#include "acados/ocp_qp/ocp_qp_xcond_solver.h"
#include "acados/utils/types.h"

#include "qp_loader.h"

void ocp_qp_gpu_ipm_acados_config_initialize_default(void *config_);
void copy_xcond_config_initialize_default(void *config_);
void ocp_qp_gpu_pcond_acados_config_initialize_default(void *config_);
void ocp_qp_gpu_pcond_acados_memory_release(void *mem_);
int ocp_qp_gpu_xcond_solver_acados_evaluate_batch(void *config_, ocp_qp_xcond_solver_dims *dims, int n, ocp_qp_in **qp_in, ocp_qp_out **qp_out,
                                                  void *opts_, void **mem_, void *work_);
int ocp_qp_gpu_xcond_solver_acados_condense_lhs_batch(void *config, ocp_qp_xcond_solver_dims *dims, int n, ocp_qp_in **qp_in, void *opts, void **mem,
                                                      void *work);
int ocp_qp_gpu_xcond_solver_acados_condense_rhs_and_solve_batch(void *config, ocp_qp_xcond_solver_dims *dims, int n, ocp_qp_in **qp_in,
                                                                ocp_qp_out **qp_out, void *opts, void **mem, void *work);
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

typedef struct
{
    int use_gpu_xcond, cond_N, n_blocks, sens;
    int blocks[256];
} drv_args;

static int parse_flags(int argc, char **argv, int first, drv_args *a, int *reps)
{
    a->use_gpu_xcond = 1; a->cond_N = -1; a->n_blocks = 0; a->sens = 0;
    for (int q = first; q < argc; q++)
    {
        if (!strcmp(argv[q], "--xcond") && q + 1 < argc) a->use_gpu_xcond = strcmp(argv[++q], "copy") != 0;
        else if (!strcmp(argv[q], "--cond-N") && q + 1 < argc) a->cond_N = atoi(argv[++q]);
        else if (!strcmp(argv[q], "--block-size") && q + 1 < argc)
        {
            char *tok = strtok(argv[++q], ",");
            while (tok && a->n_blocks < 256) { a->blocks[a->n_blocks++] = atoi(tok); tok = strtok(NULL, ","); }
        }
        else if (!strcmp(argv[q], "--sens")) a->sens = 1;
        else if (reps) *reps = atoi(argv[q]);
        else return -1;
    }
    return 0;
}

/* config + dims + opts the way ocp_qp_interface.c:262-300 / ocp_nlp build them, through the reference's own slots */
static ocp_qp_xcond_solver_config *make_config(const drv_args *a)
{
    ocp_qp_xcond_solver_config *config = ocp_qp_xcond_solver_config_assign(calloc(1, ocp_qp_xcond_solver_config_calculate_size()));
    ocp_qp_xcond_solver_config_initialize_default(config);
    ocp_qp_gpu_ipm_acados_config_initialize_default(config->qp_solver);
    if (a->use_gpu_xcond) ocp_qp_gpu_pcond_acados_config_initialize_default(config->xcond);
    else copy_xcond_config_initialize_default(config->xcond);
    return config;
}

static ocp_qp_xcond_solver_dims *make_dims(ocp_qp_xcond_solver_config *config, mock_capsule *cap)
{
    const int N = cap->dim.N;
    ocp_qp_xcond_solver_dims *dims = config->dims_assign(config, N, calloc(1, config->dims_calculate_size(config, N)));
    const char *names[] = {"nx", "nu", "nbx", "nbu", "ng", "ns", "nbxe"};
    int *vals[] = {cap->dim.nx, cap->dim.nu, cap->dim.nbx, cap->dim.nbu, cap->dim.ng, cap->dim.ns, cap->dim.nbxe};
    for (int k = 0; k <= N; k++)
        for (int q = 0; q < 7; q++) config->dims_set(config, dims, k, names[q], &vals[q][k]);
    return dims;
}

static void *make_xopts(ocp_qp_xcond_solver_config *config, ocp_qp_xcond_solver_dims *dims, const drv_args *a, int N)
{
    /* "cond_" strings go to the condensing module, the rest to the inner solver (ocp_qp_xcond_solver.c:283-311) */
    void *opts = config->opts_assign(config, dims, calloc(1, config->opts_calculate_size(config, dims)));
    config->opts_initialize_default(config, dims, opts);
    double tol = 1e-8;
    int itmax = 50, condN = a->cond_N > 0 ? a->cond_N : N, ws = 0, pl = 0;
    config->opts_set(config, opts, "tol_stat", &tol); config->opts_set(config, opts, "tol_eq", &tol);
    config->opts_set(config, opts, "tol_ineq", &tol); config->opts_set(config, opts, "tol_comp", &tol);
    config->opts_set(config, opts, "iter_max", &itmax); config->opts_set(config, opts, "warm_start", &ws);
    config->opts_set(config, opts, "print_level", &pl); config->opts_set(config, opts, "cond_N", &condN);
    if (a->n_blocks) config->opts_set(config, opts, "cond_block_size", (void *) a->blocks); /* acados_solver.in.c sends it the same way */
    config->opts_update(config, dims, opts);
    return opts;
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

static double out_diff(const struct d_ocp_qp_dim *d, ocp_qp_out *a, ocp_qp_out *b)
{
    double m = 0.0;
    for (int k = 0; k <= d->N; k++)
    {
        for (int i = 0; i < d->nu[k] + d->nx[k] + 2 * d->ns[k]; i++) m = fmax(m, fabs(BLASFEO_DVECEL(a->ux + k, i) - BLASFEO_DVECEL(b->ux + k, i)));
        if (k < d->N) for (int i = 0; i < d->nx[k + 1]; i++) m = fmax(m, fabs(BLASFEO_DVECEL(a->pi + k, i) - BLASFEO_DVECEL(b->pi + k, i)));
    }
    return m;
}

static int run_batch(int argc, char **argv)
{
    if (argc < 5) return 2;
    const int n = atoi(argv[2]);
    drv_args a;
    int reps = 1;
    if (parse_flags(argc, argv, 5, &a, &reps) != 0) return 2;
    a.use_gpu_xcond = 1;
    mock_capsule **caps = calloc(n, sizeof(*caps));
    for (int i = 0; i < n; i++) { caps[i] = mock_load_qp(argv[3]); mock_perturb(caps[i], i); }
    const int N = caps[0]->dim.N;
    ocp_qp_xcond_solver_config *config = make_config(&a);
    ocp_qp_xcond_solver_dims *dims = make_dims(config, caps[0]);
    void *opts = make_xopts(config, dims, &a, N);
    ocp_qp_in **ins = calloc(n, sizeof(void *));
    ocp_qp_out **outs = calloc(n, sizeof(void *));
    void **mems = calloc(n, sizeof(void *));
    const acados_size_t msz = config->memory_calculate_size(config, dims, opts);
    for (int i = 0; i < n; i++)
    {
        ins[i] = ocp_qp_in_assign(dims->orig_dims, calloc(1, ocp_qp_in_calculate_size(dims->orig_dims)));
        outs[i] = ocp_qp_out_assign(dims->orig_dims, calloc(1, ocp_qp_out_calculate_size(dims->orig_dims)));
        copy_loaded_qp(caps[i], ins[i]);
        mems[i] = config->memory_assign(config, dims, opts, calloc(1, msz));
    }
    void *work = calloc(1, config->workspace_calculate_size(config, dims, opts) + 8);
    int status = 0;
    double best = 1e300;
    for (int r = 0; r < reps; r++)
    {
        const double t0 = now_s();
        status = ocp_qp_gpu_xcond_solver_acados_evaluate_batch(config, dims, n, ins, outs, opts, mems, work);
        const double dt = now_s() - t0;
        if (dt < best) best = dt;
    }
    /* where the time of the last ONE-CALL evaluate went (extension fields of the adapter's memory_get; qp_info.solve_QP_time = the device's own event time) */
    double t_unpack = 0.0, t_pack = 0.0, t_call = 0.0;
    int zero_copy = 0;
    {
        void *sm = ((ocp_qp_xcond_solver_memory *) mems[0])->solver_memory;
        config->qp_solver->memory_get(config->qp_solver, sm, "time_unpack_in", &t_unpack);
        config->qp_solver->memory_get(config->qp_solver, sm, "time_pack_out", &t_pack);
        config->qp_solver->memory_get(config->qp_solver, sm, "time_qp_solver_call", &t_call);
        config->qp_solver->memory_get(config->qp_solver, sm, "zero_copy", &zero_copy);
    }
    const double t_dev = ((qp_info *) outs[0]->misc)->solve_QP_time;
    /* the same QPs as the two halves of an RTI step (the batch counterparts of condense_lhs / condense_rhs_and_solve): preparation sends
     * everything and condenses the matrices, feedback reads and sends only the vector members.  The feedback solution must equal the
     * one-call solution (same data): compared below.  Best of `reps` pairs. */
    double best_prep = 1e300, best_fb = 1e300, rti_diff = 0.0;
    int rti_status = 0, rti_upload = -1;
    double fb_unpack = 0.0, fb_call = 0.0, fb_pack = 0.0;
    {
        ocp_qp_out **outs2 = calloc(n, sizeof(void *));
        for (int i = 0; i < n; i++) outs2[i] = ocp_qp_out_assign(dims->orig_dims, calloc(1, ocp_qp_out_calculate_size(dims->orig_dims)));
        for (int r = 0; r < reps; r++)
        {
            double t0 = now_s();
            int s1 = ocp_qp_gpu_xcond_solver_acados_condense_lhs_batch(config, dims, n, ins, opts, mems, work);
            double dt = now_s() - t0;
            if (dt < best_prep) best_prep = dt;
            t0 = now_s();
            int s2 = ocp_qp_gpu_xcond_solver_acados_condense_rhs_and_solve_batch(config, dims, n, ins, outs2, opts, mems, work);
            dt = now_s() - t0;
            if (dt < best_fb) best_fb = dt;
            if (s1 != 0) rti_status = s1;
            if (s2 != 0) rti_status = s2;
        }
        for (int i = 0; i < n; i++) rti_diff = fmax(rti_diff, out_diff(&caps[i]->dim, outs[i], outs2[i]));
        config->qp_solver->memory_get(config->qp_solver, ((ocp_qp_xcond_solver_memory *) mems[0])->solver_memory, "upload_doubles", &rti_upload);
        config->qp_solver->memory_get(config->qp_solver, ((ocp_qp_xcond_solver_memory *) mems[0])->solver_memory, "time_unpack_in", &fb_unpack);
        config->qp_solver->memory_get(config->qp_solver, ((ocp_qp_xcond_solver_memory *) mems[0])->solver_memory, "time_qp_solver_call", &fb_call);
        config->qp_solver->memory_get(config->qp_solver, ((ocp_qp_xcond_solver_memory *) mems[0])->solver_memory, "time_pack_out", &fb_pack);
        /* (the per-capsule lines below report the state after the LAST call: the feedback half) */
    }
    /* capsules 0 and n-1 once more through the reference's per-capsule ocp_qp_xcond_solve (device condensing module, host round trip) */
    double fvo = 0.0;
    ocp_qp_out *o2 = ocp_qp_out_assign(dims->orig_dims, calloc(1, ocp_qp_out_calculate_size(dims->orig_dims)));
    int st_one = 0;
    for (int pick = 0; pick < 2; pick++)
    {
        const int i = pick ? n - 1 : 0;
        void *m1 = config->memory_assign(config, dims, opts, calloc(1, msz));
        const int s1 = config->evaluate(config, dims, ins[i], o2, opts, m1, work);
        if (s1 != 0) st_one = s1;
        fvo = fmax(fvo, out_diff(&caps[i]->dim, outs[i], o2));
        ocp_qp_gpu_pcond_acados_memory_release(((ocp_qp_xcond_solver_memory *) m1)->xcond_memory);
        config->terminate(config, m1, work);
    }
    ocp_qp_dims *xd = NULL;
    config->xcond->dims_get(config->xcond, dims->xcond_dims, "xcond_dims", &xd);
    /* the reference's residual entry (ocp_qp_common.c:559-667) on EVERY capsule's (qp_in, qp_out) of the batch call */
    double res_max = 0.0;
    {
        ocp_qp_res *res = ocp_qp_res_assign(dims->orig_dims, calloc(1, ocp_qp_res_calculate_size(dims->orig_dims)));
        ocp_qp_res_ws *res_ws = ocp_qp_res_workspace_assign(dims->orig_dims, calloc(1, ocp_qp_res_workspace_calculate_size(dims->orig_dims)));
        for (int i = 0; i < n; i++)
        {
            double nrm[4];
            ocp_qp_res_compute(ins[i], outs[i], res, res_ws);
            ocp_qp_res_compute_nrm_inf(res, nrm);
            for (int q = 0; q < 4; q++) res_max = fmax(res_max, nrm[q]);
        }
    }
    int cond_active = -1; /* stages of the QP the device IPM ran on in the batch call: the condensed one */
    config->qp_solver->memory_get(config->qp_solver, ((ocp_qp_xcond_solver_memory *) mems[0])->solver_memory, "cond_N_active", &cond_active);
    printf("batch n %d ms_per_call %.6f status %d fused_vs_orchestrated %.17g orchestrated_status %d xcond_N %d xcond_nu0 %d cond_N_active %d res_max %.17g "
           "unpack_in_ms %.4f copy_and_device_ms %.4f device_solve_ms %.4f pack_out_ms %.4f threads %d rti_preparation_ms %.6f rti_feedback_ms %.6f "
           "rti_status %d rti_vs_one_call %.17g rti_feedback_upload_doubles %d fb_unpack_in_ms %.4f fb_copy_and_device_ms %.4f fb_pack_out_ms %.4f zero_copy %d end\n", n,
           best * 1e3, status, fvo, st_one, xd->N, xd->nu[0], cond_active, res_max, t_unpack * 1e3, t_call * 1e3, t_dev * 1e3, t_pack * 1e3, omp_get_max_threads(),
           best_prep * 1e3, best_fb * 1e3, rti_status, rti_diff, rti_upload, fb_unpack * 1e3, fb_call * 1e3, fb_pack * 1e3, zero_copy);
    FILE *g = fopen(argv[4], "wb");
    for (int i = 0; i < n; i++)
    {
        int iter = -1, st = -1;
        config->memory_get(config, mems[i], "iter", &iter);
        config->memory_get(config, mems[i], "status", &st);
        qp_info *info = (qp_info *) outs[i]->misc;
        printf("%d %d %d %d %d\n", i, st, iter, info->num_iter, info->t_computed);
        mock_write_sol_bin(g, &caps[i]->dim, outs[i]);
    }
    fclose(g);
    for (int i = 0; i < n; i++) config->terminate(config, mems[i], work);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 2 && !strcmp(argv[1], "batch")) return run_batch(argc, argv);
    if (argc < 3) return 2;
    drv_args a;
    if (parse_flags(argc, argv, 3, &a, NULL) != 0) return 2;
    mock_capsule *cap = mock_load_qp(argv[1]);
    const int N = cap->dim.N;

    /* config: the reference's calculate_size / assign / initialize_default, then the two sub-vtables replaced */
    ocp_qp_xcond_solver_config *config = make_config(&a);
    ocp_qp_xcond_solver_dims *dims = make_dims(config, cap);

    /* containers: the reference's own */
    ocp_qp_in *qp_in = ocp_qp_in_assign(dims->orig_dims, calloc(1, ocp_qp_in_calculate_size(dims->orig_dims)));
    ocp_qp_out *qp_out = ocp_qp_out_assign(dims->orig_dims, calloc(1, ocp_qp_out_calculate_size(dims->orig_dims)));
    ocp_qp_out *qp_out2 = ocp_qp_out_assign(dims->orig_dims, calloc(1, ocp_qp_out_calculate_size(dims->orig_dims)));
    copy_loaded_qp(cap, qp_in);

    void *opts = make_xopts(config, dims, &a, N);
    void *mem = config->memory_assign(config, dims, opts, calloc(1, config->memory_calculate_size(config, dims, opts)));
    void *work = calloc(1, config->workspace_calculate_size(config, dims, opts) + 8);
    ocp_qp_dims *xd = NULL;
    config->xcond->dims_get(config->xcond, dims->xcond_dims, "xcond_dims", &xd);

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

    /* RTI split through the reference's two entries.  What ocp_nlp does between them: ONLY the vectors of qp_in change (here: they are
     * scrambled before condense_lhs and restored after it -- a module that condensed vectors in the lhs phase, or matrices only in
     * the rhs phase from stale data, returns another solution) */
    ocp_qp_out *qp_out3 = ocp_qp_out_assign(dims->orig_dims, calloc(1, ocp_qp_out_calculate_size(dims->orig_dims)));
    for (int k = 0; k <= N; k++)
    {
        for (int i = 0; i < cap->dim.nu[k] + cap->dim.nx[k]; i++) BLASFEO_DVECEL(qp_in->rqz + k, i) += 0.37;
        if (k < N) for (int i = 0; i < cap->dim.nx[k + 1]; i++) BLASFEO_DVECEL(qp_in->b + k, i) -= 0.11;
    }
    int rti = config->condense_lhs(config, dims, qp_in, qp_out3, opts, mem, work);
    for (int k = 0; k <= N; k++)
    {
        for (int i = 0; i < cap->dim.nu[k] + cap->dim.nx[k]; i++) BLASFEO_DVECEL(qp_in->rqz + k, i) -= 0.37;
        if (k < N) for (int i = 0; i < cap->dim.nx[k + 1]; i++) BLASFEO_DVECEL(qp_in->b + k, i) += 0.11;
    }
    if (rti == 0) rti = config->condense_rhs_and_solve(config, dims, qp_in, qp_out3, opts, mem, work);
    const double rti_diff = out_diff(&cap->dim, qp_out3, qp_out);

    /* warm start from the solution (warm_start 2 + initialize_next_xcond_qp_from_qp_out: the reference's xcond solver restates
     * qp_out in the condensed variables through the module's condense_qp_out, ocp_qp_xcond_solver.c:554-565) */
    int warm_iter = -1, ws2 = 2, yes = 1;
    {
        config->opts_set(config, opts, "warm_start", &ws2);
        config->opts_set(config, opts, "initialize_next_xcond_qp_from_qp_out", &yes);
        ocp_qp_out_copy(qp_out, qp_out3);
        const int s3 = config->evaluate(config, dims, qp_in, qp_out3, opts, mem, work);
        config->memory_get(config, mem, "iter", &warm_iter);
        if (s3 != 0 || out_diff(&cap->dim, qp_out3, qp_out) > 1e-6) warm_iter = 1000 + warm_iter;
        int ws0 = 0, no = 0;
        config->opts_set(config, opts, "warm_start", &ws0);
        config->opts_set(config, opts, "initialize_next_xcond_qp_from_qp_out", &no);
    }

    FILE *g = fopen(argv[2], "w");
    fprintf(g, "status %d iter %d iter_info %d t_computed %d status_mem %d rti_status %d xcond_N %d xcond_nu0 %d warm_iter %d xcond_nx0 %d xcond_nbx0 %d\n", status, iter,
            info->num_iter, info->t_computed, st_mem, rti, xd->N, xd->nu[0], warm_iter, xd->nx[0], xd->nbx[0]);
    fprintf(g, "checks t_max_diff %.17g rti_max_diff %.17g res %.17g %.17g %.17g %.17g time_call %.6g total %.6g\n", t_diff, rti_diff, nrm[0], nrm[1],
            nrm[2], nrm[3], t_call, info->total_time);
    mock_write_sol(g, &cap->dim, qp_out);
    if (a.sens)
    {
        /* forward sensitivities through the reference's eval_forw_sens (ocp_qp_xcond_solver.c:672-700): condense_rhs_seed ->
         * qp_solver->eval_forw_sens on the condensed seed -> expand_sol_seed.  A cold solve first: the factorisation the
         * sensitivity reuses is the last solve's */
        config->evaluate(config, dims, qp_in, qp_out, opts, mem, work);
        mock_fill_seed(cap, 0);
        ocp_qp_seed *seed = ocp_qp_seed_assign(dims->orig_dims, calloc(1, ocp_qp_seed_calculate_size(dims->orig_dims)));
        for (int k = 0; k <= N; k++)
        {
            const int nct = 2 * (cap->dim.nb[k] + cap->dim.ng[k] + cap->dim.ns[k]);
            for (int i = 0; i < cap->dim.nu[k] + cap->dim.nx[k] + 2 * cap->dim.ns[k]; i++) BLASFEO_DVECEL(seed->seed_g + k, i) = BLASFEO_DVECEL(cap->seed.seed_g + k, i);
            if (k < N) for (int i = 0; i < cap->dim.nx[k + 1]; i++) BLASFEO_DVECEL(seed->seed_b + k, i) = BLASFEO_DVECEL(cap->seed.seed_b + k, i);
            for (int i = 0; i < nct; i++) { BLASFEO_DVECEL(seed->seed_d + k, i) = BLASFEO_DVECEL(cap->seed.seed_d + k, i); BLASFEO_DVECEL(seed->seed_m + k, i) = 0.0; }
        }
        ocp_qp_out *sens = ocp_qp_out_assign(dims->orig_dims, calloc(1, ocp_qp_out_calculate_size(dims->orig_dims)));
        config->eval_forw_sens(config, dims, qp_in, seed, sens, opts, mem, work);
        fprintf(g, "sens\n");
        mock_write_sol(g, &cap->dim, sens);
    }
    fclose(g);
    if (a.use_gpu_xcond) ocp_qp_gpu_pcond_acados_memory_release(((ocp_qp_xcond_solver_memory *) mem)->xcond_memory);
    config->terminate(config, mem, work);
    return 0;
}
