/*
 * TEST INFRASTRUCTURE ONLY -- drives integration/ocp_qp_gpu_ipm.c the way acados drives an inner QP plugin, on QP
 * data held in (mock) HPIPM / BLASFEO storage (tests/mock_acados/qp_loader.h packs it the way acados' setters do and
 * poisons everything a plugin must not read).
 *
 *   driver <qp.txt> <out.txt> [repeat | sens]
 *       one capsule through the 17 slots in acados' order (sizes -> assign -> defaults -> opts_set -> evaluate;
 *       ocp_qp_interface.c:513-563).  `repeat`: ONLY the vectors b / rqz change between two evaluates (what ocp_nlp does
 *       every SQP iteration), the second solution is written.  `sens`: after the solve a seed built acados' way
 *       (ocp_nlp_common.c:4057-4081) goes through eval_forw_sens; solution, then "sens" lines, are written.
 *
 *   driver batch <n> <qpA.txt> <qpB.txt|-> <out.bin> [sens] [reps]
 *       n capsules (instance i = base QP A, or B for odd i when given, with the vectors perturbed by mock_perturb(i)),
 *       one memory each, solved by ONE call of ocp_qp_gpu_ipm_acados_evaluate_batch -- the replacement of the per-capsule
 *       loop of acados_solver.in.c:3222-3243; with `sens` followed by ocp_qp_gpu_ipm_acados_eval_sens_batch on per-instance
 *       seeds.  out.bin: per instance the solution (and then the sensitivities) as raw doubles
 *       [ux_0..ux_N, pi_0..pi_{N-1}, lam_0..lam_N, t_0..t_N]; stdout: one line "batch n .. ms_per_call .. status ..",
 *       then per instance "i status iter".
 *
 *   driver rendezvous <n> <qp.txt> <out.bin>
 *       what an UNMODIFIED `_acados_batch_solve` does (acados_solver.in.c:3232-3236): n threads, each running its own
 *       capsule's loop -- capsule i makes 1 + i % 3 "SQP iterations" (vectors perturbed, then the plugin's `evaluate` SLOT
 *       through the vtable) and returns.  With a rendezvous in the solver options the n evaluates of an iteration reach the
 *       GPU as one batch although nobody calls a batch entry.  out.bin: every capsule's last solution.
 */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "qp_loader.h"

void ocp_qp_gpu_ipm_acados_config_initialize_default(void *config_);
int ocp_qp_gpu_ipm_acados_evaluate_batch(void *config, int n, void **qp_in, void **qp_out, void *opts, void **mem, void *work);
void ocp_qp_gpu_ipm_acados_eval_sens_batch(void *config, int n, void **qp_in, void **seed, void **sens_qp_out, void *opts, void **mem, void *work);
void *ocp_qp_gpu_ipm_acados_rendezvous_create(int n_capsules);
void ocp_qp_gpu_ipm_acados_rendezvous_destroy(void *r);
void ocp_qp_gpu_ipm_acados_rendezvous_leave(void *r);

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

static void *make_opts(qp_solver_config *config, struct d_ocp_qp_dim *dim)
{
    void *opts = config->opts_assign(config, dim, calloc(1, config->opts_calculate_size(config, dim)));
    config->opts_initialize_default(config, dim, opts);
    double tol = 1e-8;
    int itmax = 50, pl = 0, ws = 0;
    config->opts_set(config, opts, "tol_stat", &tol); config->opts_set(config, opts, "tol_eq", &tol);
    config->opts_set(config, opts, "tol_ineq", &tol); config->opts_set(config, opts, "tol_comp", &tol);
    config->opts_set(config, opts, "iter_max", &itmax); config->opts_set(config, opts, "print_level", &pl);
    config->opts_set(config, opts, "warm_start", &ws);
    config->opts_update(config, dim, opts);
    return opts;
}

static int run_batch(int argc, char **argv)
{
    if (argc < 6) return 2;
    const int n = atoi(argv[2]);
    const int two = strcmp(argv[4], "-") != 0;
    int do_sens = 0, reps = 1;
    for (int a = 6; a < argc; a++) { if (!strcmp(argv[a], "sens")) do_sens = 1; else reps = atoi(argv[a]); }
    qp_solver_config config;
    memset(&config, 0, sizeof(config));
    ocp_qp_gpu_ipm_acados_config_initialize_default(&config);

    mock_capsule **caps = calloc(n, sizeof(*caps));
    void **ins = calloc(n, sizeof(void *)), **outs = calloc(n, sizeof(void *)), **mems = calloc(n, sizeof(void *));
    void **seeds = calloc(n, sizeof(void *)), **sens = calloc(n, sizeof(void *));
    for (int i = 0; i < n; i++)
    {
        caps[i] = mock_load_qp(argv[two && (i & 1) ? 4 : 3]);
        mock_perturb(caps[i], i);
        ins[i] = &caps[i]->qp; outs[i] = &caps[i]->sol; seeds[i] = &caps[i]->seed; sens[i] = &caps[i]->sens;
    }
    void *opts = make_opts(&config, &caps[0]->dim);
    for (int i = 0; i < n; i++)
        mems[i] = config.memory_assign(&config, &caps[i]->dim, opts, calloc(1, config.memory_calculate_size(&config, &caps[i]->dim, opts)));

    int status = ocp_qp_gpu_ipm_acados_evaluate_batch(&config, n, ins, outs, opts, mems, NULL); /* builds the device batches */
    double best = 1e30;
    for (int r = 0; r < reps; r++)
    {
        const double t0 = now_s();
        status = ocp_qp_gpu_ipm_acados_evaluate_batch(&config, n, ins, outs, opts, mems, NULL);
        const double dt = now_s() - t0;
        if (dt < best) best = dt;
    }
    double sens_s = 0.0;
    if (do_sens)
    {
        for (int i = 0; i < n; i++) mock_fill_seed(caps[i], i);
        const double t0 = now_s();
        ocp_qp_gpu_ipm_acados_eval_sens_batch(&config, n, ins, seeds, sens, opts, mems, NULL);
        sens_s = now_s() - t0;
    }
    const char *kname = "";
    config.memory_get(&config, mems[0], "kernel_name", &kname);
    int it_max = 0;
    for (int i = 0; i < n; i++) if (caps[i]->info.num_iter > it_max) it_max = caps[i]->info.num_iter;
    int zero_copy = 0;
    config.memory_get(&config, mems[0], "zero_copy", &zero_copy);
    printf("batch n %d ms_per_call %.3f status %d interface_ms %.3f solve_ms %.3f sens_ms %.3f iter_max %d zero_copy %d\n", n, best * 1e3, status,
           caps[0]->info.interface_time * 1e3, caps[0]->info.solve_QP_time * 1e3, sens_s * 1e3, it_max, zero_copy);
    fprintf(stderr, "kernel of capsule 0: %s\n", kname);
    FILE *g = fopen(argv[5], "wb");
    for (int i = 0; i < n; i++)
    {
        int iter = -1, st = -1;
        config.memory_get(&config, mems[i], "iter", &iter);
        config.memory_get(&config, mems[i], "status", &st);
        printf("%d %d %d %d %d\n", i, st, iter, caps[i]->info.num_iter, caps[i]->info.t_computed);
        mock_write_sol_bin(g, &caps[i]->dim, &caps[i]->sol);
        if (do_sens) mock_write_sol_bin(g, &caps[i]->dim, &caps[i]->sens);
    }
    fclose(g);
    /* the generated batch loops call the PER-CAPSULE slots inside `#pragma omp parallel for` (acados_solver.in.c:3292-3337:
     * eval_param_sens / eval_solution_sens_adj_p per capsule; ocp_nlp_ddp.c:373-377 reads K, k per capsule): after a batched
     * solve the capsules of a class share one device batch and its staging -- every capsule's slot result, computed with
     * all capsules in flight at once, must be its share of the batched call (sensitivities) resp. what the serial call
     * returns (gains) */
    if (do_sens)
    {
        double worst = 0.0, worst_k = 0.0;
        struct d_ocp_qp_sol *chk = calloc(n, sizeof(*chk));
        double **Kser = calloc(n, sizeof(double *)), **Kpar = calloc(n, sizeof(double *));
        for (int i = 0; i < n; i++)
        {
            mock_alloc_sol(&caps[i]->dim, chk + i);
            const int nu = caps[i]->dim.nu[1], nx = caps[i]->dim.nx[1];
            Kser[i] = calloc((size_t) (nu * nx + 1), sizeof(double)); Kpar[i] = calloc((size_t) (nu * nx + 1), sizeof(double));
            if (nu > 0) config.solver_get(&config, &caps[i]->qp, &caps[i]->sol, opts, mems[i], "K", 1, Kser[i], nu, nx);
        }
#pragma omp parallel for schedule(dynamic, 1) num_threads(8)
        for (int i = 0; i < n; i++)
        {
            config.eval_forw_sens(&config, &caps[i]->qp, &caps[i]->seed, chk + i, opts, mems[i], NULL);
            const int nu = caps[i]->dim.nu[1], nx = caps[i]->dim.nx[1];
            if (nu > 0) config.solver_get(&config, &caps[i]->qp, &caps[i]->sol, opts, mems[i], "K", 1, Kpar[i], nu, nx);
        }
        for (int i = 0; i < n; i++)
        {
            mock_capsule *c = caps[i];
            for (int s = 0; s <= c->dim.N; s++)
                for (int e = 0; e < c->dim.nu[s] + c->dim.nx[s]; e++)
                    worst = fmax(worst, fabs(BLASFEO_DVECEL(chk[i].ux + s, e) - BLASFEO_DVECEL(c->sens.ux + s, e)));
            for (int e = 0; e < c->dim.nu[1] * c->dim.nx[1]; e++) worst_k = fmax(worst_k, fabs(Kser[i][e] - Kpar[i][e]));
        }
        printf("threaded_slots_vs_batch_sens %.3e\n", worst);
        printf("threaded_slots_gain_K %.3e\n", worst_k);
    }
    /* a single-capsule slot after a batch call: the sensitivity of capsule n-1 alone through its own memory must equal
     * its share of the batched call */
    if (do_sens)
    {
        mock_capsule *c = caps[n - 1];
        struct d_ocp_qp_sol chk;
        mock_alloc_sol(&c->dim, &chk);
        config.eval_forw_sens(&config, &c->qp, &c->seed, &chk, opts, mems[n - 1], NULL);
        double worst = 0.0;
        for (int s = 0; s <= c->dim.N; s++)
            for (int e = 0; e < c->dim.nu[s] + c->dim.nx[s]; e++)
                worst = fmax(worst, fabs(BLASFEO_DVECEL(chk.ux + s, e) - BLASFEO_DVECEL(c->sens.ux + s, e)));
        printf("single_vs_batch_sens %.3e\n", worst);
    }
    /* regrouping: the first half of the capsules is solved again as a smaller batch (the owner rebuilds the group), then the
     * LAST capsule -- whose memory still remembers the released group -- goes through the single-QP slot and must be
     * answered from its own batch, with the same solution as in the big batch */
    if (n >= 4)
    {
        mock_capsule *c = caps[n - 1];
        struct d_ocp_qp_sol before;
        mock_alloc_sol(&c->dim, &before);
        for (int s = 0; s <= c->dim.N; s++)
            for (int e = 0; e < c->dim.nu[s] + c->dim.nx[s]; e++) BLASFEO_DVECEL(before.ux + s, e) = BLASFEO_DVECEL(c->sol.ux + s, e);
        const int st_half = ocp_qp_gpu_ipm_acados_evaluate_batch(&config, n / 2, ins, outs, opts, mems, NULL);
        int st_last = -1, it_last = -1;
        config.memory_get(&config, mems[n - 1], "status", &st_last);   /* answered from the memory itself */
        config.memory_get(&config, mems[n - 1], "iter", &it_last);
        const int st_single = config.evaluate(&config, &c->qp, &c->sol, opts, mems[n - 1], NULL);
        double worst = 0.0;
        for (int s = 0; s <= c->dim.N; s++)
            for (int e = 0; e < c->dim.nu[s] + c->dim.nx[s]; e++)
                worst = fmax(worst, fabs(BLASFEO_DVECEL(before.ux + s, e) - BLASFEO_DVECEL(c->sol.ux + s, e)));
        printf("regroup half_status %d last_status %d last_iter %d single_status %d single_vs_batch %.3e\n", st_half, st_last, it_last, st_single, worst);
        config.terminate(&config, mems[n - 1], NULL);
    }
    config.terminate(&config, mems[0], NULL);
    return 0;
}

typedef struct
{
    qp_solver_config *config;
    mock_capsule *cap;
    void *opts, *mem, *rv;
    int index, iters, status, qp_iter_last;
} rv_thread;

static void *rv_capsule_loop(void *arg)
{
    rv_thread *t = (rv_thread *) arg;
    for (int j = 0; j < t->iters; j++)
    {
        mock_perturb(t->cap, t->index + 100 * j);   /* "linearisation": new vectors */
        t->status = t->config->evaluate(t->config, &t->cap->qp, &t->cap->sol, t->opts, t->mem, NULL);
        t->qp_iter_last = t->cap->info.num_iter;
    }
    ocp_qp_gpu_ipm_acados_rendezvous_leave(t->rv);  /* ocp_nlp_solve returned */
    return NULL;
}

static int run_rendezvous(int argc, char **argv)
{
    if (argc < 5) return 2;
    const int n = atoi(argv[2]);
    qp_solver_config config;
    memset(&config, 0, sizeof(config));
    ocp_qp_gpu_ipm_acados_config_initialize_default(&config);
    void *rv = ocp_qp_gpu_ipm_acados_rendezvous_create(n);
    rv_thread *th = calloc(n, sizeof(*th));
    pthread_t *tid = calloc(n, sizeof(*tid));
    for (int i = 0; i < n; i++)
    {
        th[i].config = &config; th[i].cap = mock_load_qp(argv[3]); th[i].index = i; th[i].iters = 1 + i % 3; th[i].rv = rv;
        th[i].opts = make_opts(&config, &th[i].cap->dim);          /* every capsule has its own opts object, as in acados */
        config.opts_set(&config, th[i].opts, "rendezvous", rv);
        th[i].mem = config.memory_assign(&config, &th[i].cap->dim, th[i].opts, calloc(1, config.memory_calculate_size(&config, &th[i].cap->dim, th[i].opts)));
    }
    for (int i = 0; i < n; i++) pthread_create(tid + i, NULL, rv_capsule_loop, th + i);
    for (int i = 0; i < n; i++) pthread_join(tid[i], NULL);
    FILE *g = fopen(argv[4], "wb");
    for (int i = 0; i < n; i++)
    {
        int st = -1;
        config.memory_get(&config, th[i].mem, "status", &st);
        printf("%d %d %d %d\n", i, th[i].status, st, th[i].qp_iter_last);
        mock_write_sol_bin(g, &th[i].cap->dim, &th[i].cap->sol);
    }
    fclose(g);
    config.terminate(&config, th[0].mem, NULL);
    ocp_qp_gpu_ipm_acados_rendezvous_destroy(rv);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 2 && !strcmp(argv[1], "batch")) return run_batch(argc, argv);
    if (argc >= 2 && !strcmp(argv[1], "rendezvous")) return run_rendezvous(argc, argv);
    if (argc < 3) return 2;
    mock_capsule *c = mock_load_qp(argv[1]);
    struct d_ocp_qp_dim *dim = &c->dim;
    const int N = dim->N;

    /* ---- the plugin, driven through its 17 slots in acados' order ---- */
    qp_solver_config config;
    memset(&config, 0, sizeof(config));
    ocp_qp_gpu_ipm_acados_config_initialize_default(&config);
    void **slots = (void **) &config;
    for (int q = 0; q < 17; q++) if (!slots[q]) { fprintf(stderr, "driver: slot %d empty\n", q); return 3; }
    void *opts = make_opts(&config, dim);
    void *mem = config.memory_assign(&config, dim, opts, calloc(1, config.memory_calculate_size(&config, dim, opts)));
    void *work = calloc(1, config.workspace_calculate_size(&config, dim, opts) + 8);

    int status = config.evaluate(&config, &c->qp, &c->sol, opts, mem, work);
    if (argc > 3 && !strcmp(argv[3], "repeat"))
    {
        /* what ocp_nlp changes between two SQP iterations: ONLY the vectors (ocp_nlp_common.c:3119-3138) */
        for (int s = 0; s <= N; s++)
        {
            for (int e = 0; e < dim->nu[s] + dim->nx[s]; e++) BLASFEO_DVECEL(c->qp.rqz + s, e) += 0.05 * ((e + s) % 3 - 1);
            if (s < N) for (int e = 0; e < dim->nx[s + 1]; e++) BLASFEO_DVECEL(c->qp.b + s, e) += 0.01 * ((e + 2 * s) % 3 - 1);
        }
        status = config.evaluate(&config, &c->qp, &c->sol, opts, mem, work);
    }
    int iter = -1, st2 = -1;
    config.memory_get(&config, mem, "iter", &iter);
    config.memory_get(&config, mem, "status", &st2);

    FILE *g = fopen(argv[2], "w");
    fprintf(g, "status %d %d iter %d %d t_computed %d\n", status, st2, iter, c->info.num_iter, c->info.t_computed);
    mock_write_sol(g, dim, &c->sol);
    /* Riccati gains through the solver_get slot (ocp_nlp_ddp.c:373-377) */
    {
        const int nu = dim->nu[1], nx = dim->nx[1];
        double *K = calloc(nu * nx + 1, sizeof(double));
        config.solver_get(&config, &c->qp, &c->sol, opts, mem, "K", 1, K, nu, nx);
        fprintf(g, "K 1"); for (int e = 0; e < nu * nx; e++) fprintf(g, " %.17g", K[e]); fprintf(g, "\n");
    }
    if (argc > 3 && !strcmp(argv[3], "sens"))
    {
        /* ocp_nlp_common_eval_param_sens (ocp_nlp_common.c:4039-4105): zero seed, fill, eval_forw_sens, read ux pi lam */
        mock_fill_seed(c, 0);
        config.eval_forw_sens(&config, &c->qp, &c->seed, &c->sens, opts, mem, work);
        fprintf(g, "sens 0\n");
        mock_write_sol(g, dim, &c->sens);
    }
    fclose(g);
    config.terminate(&config, mem, work);
    return 0;
}
