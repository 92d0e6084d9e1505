/*
 * TEST INFRASTRUCTURE ONLY -- the lock-step batched RTI loop RUN end to end (second half): n capsules of a linear MPC
 * problem, each an `ocp_nlp` of the reference's own SQP_RTI (integration/acados.patch applied) + LINEAR_LS cost + DISCRETE_MODEL
 * dynamics + BGH constraints + NO_REGULARIZE + FIXED_STEP, QP solver PARTIAL_CONDENSING_GPU_IPM from the plan -- built by
 * tests/lockstep_build.py from the reference's sources (HPIPM / BLASFEO: the stand-ins of tests/mock_hpipm) and linked against this
 * repository's library.  What a generated `acados_solver_<name>.c` would do, with the two batch functions it would contain taken
 * verbatim from the PATCHED template (lockstep_batch_fns.inc is cut out of c_templates_tera/acados_solver.in.c by the build script,
 * `{{ name }}` -> mpc):
 *
 *   mpc_acados_batch_solve_gpu_qp   every capsule's RTI step with ONE device batch for the n QPs
 *                                   (batch_qp_phase 1 -> ocp_qp_gpu_xcond_solver_acados_evaluate_batch -> batch_qp_phase 2)
 *   mpc_acados_batch_solve          the reference's per-capsule OpenMP loop (acados_solver.in.c:3222-3243), on TWINS of a subset
 *
 * Closed loop: after every RTI step x0 := x1 of the step's solution (the plant is the model), new initial-state bounds at stage 0.
 * The "linearisation" of every step goes through the reference's module code: gradient / constraint residuals / dynamics offsets are
 * written into rqz / d / b of the capsule's qp_in through the aliased pointers (ocp_nlp_common.c:2797-2894, 3119-3138).
 *
 *   lockstep_driver model.txt out.txt n_twins iterations cond_N threads [split]
 *
 * split = 1: every step as the two halves of real-time iteration -- rti_phase PREPARATION for all capsules (matrices to the device once,
 * condensed there), the new x0, rti_phase FEEDBACK (only the QPs' vector members travel) -- through the same batch function.
 */
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "acados/ocp_nlp/ocp_nlp_common.h"
#include "acados/ocp_nlp/ocp_nlp_sqp_rti.h"
#include "acados/ocp_qp/ocp_qp_gpu_ipm.h"
#include "acados/utils/external_function_generic.h"
#include "acados_c/ocp_nlp_interface.h"

typedef struct mpc_solver_capsule
{
    ocp_nlp_in *nlp_in;
    ocp_nlp_out *nlp_out;
    ocp_nlp_out *sens_out;
    ocp_nlp_solver *nlp_solver;
    void *nlp_opts;
    ocp_nlp_plan_t *nlp_solver_plan;
    ocp_nlp_config *nlp_config;
    ocp_nlp_dims *nlp_dims;
} mpc_solver_capsule;

#include "lockstep_batch_fns.inc"

/* x+ = A x + B u as an external function of the discrete-dynamics module (ocp_nlp_dynamics_disc.c:711-742: in = {x, u} as
 * BLASFEO_DVEC_ARGS, out = {fun as BLASFEO_DVEC_ARGS, [B A]' as BLASFEO_DMAT_ARGS}) */
typedef struct
{
    external_function_generic base;
    int nx, nu, with_jac;
    const double *A, *B;
} lti_fun;

static size_t lti_ws(void *f) { (void) f; return 0; }
static void lti_set_ws(void *f, void *w) { (void) f; (void) w; }
static void lti_eval(void *self, ext_fun_arg_t *tin, void **in, ext_fun_arg_t *tout, void **out)
{
    const lti_fun *f = (const lti_fun *) self;
    const struct blasfeo_dvec_args *x = (const struct blasfeo_dvec_args *) in[0], *u = (const struct blasfeo_dvec_args *) in[1];
    struct blasfeo_dvec_args *fo = (struct blasfeo_dvec_args *) out[0];
    (void) tin; (void) tout;
    for (int r = 0; r < f->nx; r++)
    {
        double a = 0.0;
        for (int c = 0; c < f->nx; c++) a += f->A[r + f->nx * c] * BLASFEO_DVECEL(x->x, x->xi + c);
        for (int c = 0; c < f->nu; c++) a += f->B[r + f->nx * c] * BLASFEO_DVECEL(u->x, u->xi + c);
        BLASFEO_DVECEL(fo->x, fo->xi + r) = a;
    }
    if (f->with_jac)
    {
        struct blasfeo_dmat_args *j = (struct blasfeo_dmat_args *) out[1];
        for (int r = 0; r < f->nx; r++)
        {
            for (int c = 0; c < f->nu; c++) BLASFEO_DMATEL(j->A, j->ai + c, j->aj + r) = f->B[r + f->nx * c];
            for (int c = 0; c < f->nx; c++) BLASFEO_DMATEL(j->A, j->ai + f->nu + c, j->aj + r) = f->A[r + f->nx * c];
        }
    }
}

typedef struct
{
    int N, nx, nu, n;
    double *A, *B, *Q, *R, *QN, umax, *x0;
} model_t;

static double *read_doubles(FILE *f, int cnt)
{
    double *p = (double *) malloc(sizeof(double) * (size_t) (cnt > 0 ? cnt : 1));
    for (int i = 0; i < cnt; i++)
        if (fscanf(f, "%lf", p + i) != 1) { fprintf(stderr, "lockstep_driver: short model file\n"); exit(2); }
    return p;
}

static lti_fun g_fun, g_fun_jac;

static mpc_solver_capsule *capsule_create(const model_t *m, const double *x0, int cond_N)
{
    const int N = m->N, nx = m->nx, nu = m->nu;
    mpc_solver_capsule *c = (mpc_solver_capsule *) calloc(1, sizeof(*c));
    /* plan (acados_solver.in.c:217-259) */
    ocp_nlp_plan_t *plan = ocp_nlp_plan_create(N);
    c->nlp_solver_plan = plan;
    plan->nlp_solver = SQP_RTI;
    plan->ocp_qp_solver_plan.qp_solver = PARTIAL_CONDENSING_GPU_IPM;
    plan->relaxed_ocp_qp_solver_plan.qp_solver = PARTIAL_CONDENSING_GPU_IPM;
    for (int i = 0; i <= N; i++) { plan->nlp_cost[i] = LINEAR_LS; plan->nlp_constraints[i] = BGH; }
    for (int i = 0; i < N; i++) { plan->nlp_dynamics[i] = DISCRETE_MODEL; plan->sim_solver_plan[i].sim_solver = INVALID_SIM_SOLVER; }
    plan->regularization = NO_REGULARIZE;
    plan->globalization = FIXED_STEP;
    c->nlp_config = ocp_nlp_config_create(*plan);
    ocp_nlp_config *cfg = c->nlp_config;
    /* dims (:262-447) */
    c->nlp_dims = ocp_nlp_dims_create(cfg);
    ocp_nlp_dims *dims = c->nlp_dims;
    int *vnx = (int *) calloc((size_t) N + 1, sizeof(int)), *vnu = (int *) calloc((size_t) N + 1, sizeof(int)), *zero = (int *) calloc((size_t) N + 1, sizeof(int));
    for (int i = 0; i <= N; i++) { vnx[i] = nx; vnu[i] = i < N ? nu : 0; }
    ocp_nlp_dims_set_opt_vars(cfg, dims, "nx", vnx);
    ocp_nlp_dims_set_opt_vars(cfg, dims, "nu", vnu);
    ocp_nlp_dims_set_opt_vars(cfg, dims, "nz", zero);
    ocp_nlp_dims_set_opt_vars(cfg, dims, "ns", zero);
    ocp_nlp_dims_set_opt_vars(cfg, dims, "np", zero);
    for (int i = 0; i <= N; i++)
    {
        int nbx = i == 0 ? nx : 0, nbu = i < N ? nu : 0, nbxe = i == 0 ? nx : 0, z = 0, ny = i < N ? nx + nu : nx;
        ocp_nlp_dims_set_constraints(cfg, dims, i, "nbx", &nbx);
        ocp_nlp_dims_set_constraints(cfg, dims, i, "nbu", &nbu);
        ocp_nlp_dims_set_constraints(cfg, dims, i, "nsbx", &z);
        ocp_nlp_dims_set_constraints(cfg, dims, i, "nsbu", &z);
        ocp_nlp_dims_set_constraints(cfg, dims, i, "ng", &z);
        ocp_nlp_dims_set_constraints(cfg, dims, i, "nsg", &z);
        ocp_nlp_dims_set_constraints(cfg, dims, i, "nbxe", &nbxe);
        ocp_nlp_dims_set_constraints(cfg, dims, i, "nh", &z);
        ocp_nlp_dims_set_constraints(cfg, dims, i, "nsh", &z);
        ocp_nlp_dims_set_cost(cfg, dims, i, "ny", &ny);
    }
    free(vnx); free(vnu); free(zero);
    c->nlp_opts = ocp_nlp_solver_opts_create(cfg, dims);
    c->nlp_out = ocp_nlp_out_create(cfg, dims);
    c->sens_out = ocp_nlp_out_create(cfg, dims);
    c->nlp_in = ocp_nlp_in_create(cfg, dims);
    ocp_nlp_in *in = c->nlp_in;
    /* model (:1012-1800): cost 1/2 |[x; u]|^2_W, y = Vx x + Vu u */
    const int ny = nx + nu;
    double *Vx = (double *) calloc((size_t) ny * nx, sizeof(double)), *Vu = (double *) calloc((size_t) ny * nu, sizeof(double));
    double *W = (double *) calloc((size_t) ny * ny, sizeof(double)), *yref = (double *) calloc((size_t) ny, sizeof(double));
    double *VxN = (double *) calloc((size_t) nx * nx, sizeof(double)), *WN = (double *) calloc((size_t) nx * nx, sizeof(double));
    for (int r = 0; r < nx; r++) { Vx[r + ny * r] = 1.0; W[r + ny * r] = m->Q[r]; VxN[r + nx * r] = 1.0; WN[r + nx * r] = m->QN[r]; }
    for (int r = 0; r < nu; r++) { Vu[nx + r + ny * r] = 1.0; W[nx + r + ny * (nx + r)] = m->R[r]; }
    for (int i = 0; i < N; i++)
    {
        ocp_nlp_cost_model_set(cfg, dims, in, i, "Vx", Vx);
        ocp_nlp_cost_model_set(cfg, dims, in, i, "Vu", Vu);
        ocp_nlp_cost_model_set(cfg, dims, in, i, "W", W);
        ocp_nlp_cost_model_set(cfg, dims, in, i, "yref", yref);
        ocp_nlp_dynamics_model_set(cfg, dims, in, i, "disc_dyn_fun", &g_fun);
        ocp_nlp_dynamics_model_set(cfg, dims, in, i, "disc_dyn_fun_jac", &g_fun_jac);
    }
    ocp_nlp_cost_model_set(cfg, dims, in, N, "Vx", VxN);
    ocp_nlp_cost_model_set(cfg, dims, in, N, "W", WN);
    ocp_nlp_cost_model_set(cfg, dims, in, N, "yref", yref);
    free(Vx); free(Vu); free(W); free(yref); free(VxN); free(WN);
    int *idx = (int *) calloc((size_t) (nx > nu ? nx : nu), sizeof(int));
    double *lo = (double *) calloc((size_t) nu, sizeof(double)), *hi = (double *) calloc((size_t) nu, sizeof(double));
    for (int r = 0; r < (nx > nu ? nx : nu); r++) idx[r] = r;
    for (int r = 0; r < nu; r++) { lo[r] = -m->umax; hi[r] = m->umax; }
    ocp_nlp_constraints_model_set(cfg, dims, in, c->nlp_out, 0, "idxbx", idx);
    ocp_nlp_constraints_model_set(cfg, dims, in, c->nlp_out, 0, "lbx", (void *) x0);
    ocp_nlp_constraints_model_set(cfg, dims, in, c->nlp_out, 0, "ubx", (void *) x0);
    ocp_nlp_constraints_model_set(cfg, dims, in, c->nlp_out, 0, "idxbxe", idx);
    for (int i = 0; i < N; i++)
    {
        ocp_nlp_constraints_model_set(cfg, dims, in, c->nlp_out, i, "idxbu", idx);
        ocp_nlp_constraints_model_set(cfg, dims, in, c->nlp_out, i, "lbu", lo);
        ocp_nlp_constraints_model_set(cfg, dims, in, c->nlp_out, i, "ubu", hi);
    }
    free(idx); free(lo); free(hi);
    /* options (:2300-2990) */
    int iter_max = 50, ws = 0;
    ocp_nlp_solver_opts_set(cfg, c->nlp_opts, "qp_cond_N", &cond_N);
    ocp_nlp_solver_opts_set(cfg, c->nlp_opts, "qp_iter_max", &iter_max);
    ocp_nlp_solver_opts_set(cfg, c->nlp_opts, "qp_warm_start", &ws);
    c->nlp_solver = ocp_nlp_solver_create(cfg, dims, c->nlp_opts, in);
    if (ocp_nlp_precompute(c->nlp_solver, in, c->nlp_out) != ACADOS_SUCCESS) { fprintf(stderr, "lockstep_driver: ocp_nlp_precompute failed\n"); exit(2); }
    return c;
}

static void capsule_set_x0(mpc_solver_capsule *c, double *x0)
{
    ocp_nlp_constraints_model_set(c->nlp_config, c->nlp_dims, c->nlp_in, c->nlp_out, 0, "lbx", x0);
    ocp_nlp_constraints_model_set(c->nlp_config, c->nlp_dims, c->nlp_in, c->nlp_out, 0, "ubx", x0);
}

int main(int argc, char **argv)
{
    if (argc < 7) { fprintf(stderr, "usage: lockstep_driver model.txt out.txt n_twins iterations cond_N threads\n"); return 2; }
    FILE *f = fopen(argv[1], "r");
    if (!f) { perror(argv[1]); return 2; }
    model_t m;
    if (fscanf(f, "%d %d %d %d", &m.N, &m.nx, &m.nu, &m.n) != 4) return 2;
    m.A = read_doubles(f, m.nx * m.nx); m.B = read_doubles(f, m.nx * m.nu);
    m.Q = read_doubles(f, m.nx); m.R = read_doubles(f, m.nu); m.QN = read_doubles(f, m.nx);
    double *um = read_doubles(f, 1); m.umax = um[0];
    m.x0 = read_doubles(f, m.n * m.nx);
    fclose(f);
    const int n_twins = atoi(argv[3]), iters = atoi(argv[4]), cond_N = atoi(argv[5]), threads = atoi(argv[6]);
    const int split = argc > 7 ? atoi(argv[7]) : 0;
    const int n = m.n, nx = m.nx, nu = m.nu;
    g_fun.base.evaluate = &lti_eval; g_fun.base.get_external_workspace_requirement = &lti_ws; g_fun.base.set_external_workspace = &lti_set_ws;
    g_fun.nx = nx; g_fun.nu = nu; g_fun.A = m.A; g_fun.B = m.B; g_fun.with_jac = 0;
    g_fun_jac = g_fun; g_fun_jac.with_jac = 1;

    mpc_solver_capsule **lock = (mpc_solver_capsule **) calloc((size_t) n, sizeof(*lock));
    mpc_solver_capsule **twin = (mpc_solver_capsule **) calloc((size_t) (n_twins > 0 ? n_twins : 1), sizeof(*twin));
    int *twin_of = (int *) calloc((size_t) (n_twins > 0 ? n_twins : 1), sizeof(int));
    for (int i = 0; i < n; i++) lock[i] = capsule_create(&m, m.x0 + (size_t) i * nx, cond_N);
    for (int j = 0; j < n_twins; j++)
    {
        twin_of[j] = n_twins > 1 ? (int) ((long) j * (n - 1) / (n_twins - 1)) : 0;
        twin[j] = capsule_create(&m, m.x0 + (size_t) twin_of[j] * nx, cond_N);
    }
    int *st_lock = (int *) calloc((size_t) n, sizeof(int)), *st_twin = (int *) calloc((size_t) (n_twins > 0 ? n_twins : 1), sizeof(int));
    double *x1 = (double *) calloc((size_t) nx, sizeof(double)), *u0 = (double *) calloc((size_t) nu, sizeof(double));
    FILE *o = fopen(argv[2], "w");
    if (!o) { perror(argv[2]); return 2; }
    double t_lock = 0.0, t_twin = 0.0;
    double *x0_next = (double *) calloc((size_t) (n + n_twins) * nx, sizeof(double));
    for (int it = 0; it < iters; it++)
    {
        for (int half = split ? 1 : 0; half <= (split ? 2 : 0); half++)
        {
            /* half 0: preparation + feedback in one call; 1: preparation; 2: feedback (the measured state arrives in between) */
            for (int i = 0; i < n + n_twins; i++)
            {
                mpc_solver_capsule *c = i < n ? lock[i] : twin[i - n];
                int ph = half;
                ocp_nlp_solver_opts_set(c->nlp_config, c->nlp_opts, "rti_phase", &ph);
                if (half != 1 && it > 0) capsule_set_x0(c, x0_next + (size_t) i * nx); /* closed loop: the plant is the model */
            }
            double t0 = omp_get_wtime();
            mpc_acados_batch_solve_gpu_qp(lock, st_lock, n, threads);
            t_lock += omp_get_wtime() - t0;
            t0 = omp_get_wtime();
            if (n_twins > 0) mpc_acados_batch_solve(twin, st_twin, n_twins, threads);
            t_twin += omp_get_wtime() - t0;
        }
        for (int pass = 0; pass < 2; pass++)
        {
            mpc_solver_capsule **cs = pass ? twin : lock;
            const int cnt = pass ? n_twins : n;
            for (int i = 0; i < cnt; i++)
            {
                mpc_solver_capsule *c = cs[i];
                int qp_iter = -1, qp_status = -1;
                ocp_nlp_get(c->nlp_solver, "qp_iter", &qp_iter);
                ocp_nlp_get(c->nlp_solver, "qp_status", &qp_status);
                ocp_nlp_out_get(c->nlp_config, c->nlp_dims, c->nlp_out, 0, "u", u0);
                ocp_nlp_out_get(c->nlp_config, c->nlp_dims, c->nlp_out, 1, "x", x1);
                fprintf(o, "%s %d %d status %d qp_status %d qp_iter %d u0", pass ? "twin" : "lock", it, pass ? twin_of[i] : i, pass ? st_twin[i] : st_lock[i],
                        qp_status, qp_iter);
                for (int r = 0; r < nu; r++) fprintf(o, " %.17g", u0[r]);
                fprintf(o, " x1");
                for (int r = 0; r < nx; r++) fprintf(o, " %.17g", x1[r]);
                fprintf(o, "\n");
                memcpy(x0_next + (size_t) (pass ? n + i : i) * nx, x1, sizeof(double) * (size_t) nx);
            }
        }
    }
    {
        /* what the LAST batch call sent per QP (extension field of the adapter's memory_get): the whole input blob, or its vector part */
        ocp_nlp_memory *nm;
        int up = -1;
        ocp_nlp_get(lock[0]->nlp_solver, "nlp_mem", &nm);
        ocp_qp_xcond_solver_config *qc = lock[0]->nlp_config->qp_solver;
        qc->qp_solver->memory_get(qc->qp_solver, ((ocp_qp_xcond_solver_memory *) nm->qp_solver_mem)->solver_memory, "upload_doubles", &up);
        fprintf(o, "time lock %.6f twin %.6f upload_doubles %d\n", t_lock, t_twin, up);
    }
    fclose(o);
    for (int i = 0; i < n + n_twins; i++)
    {
        mpc_solver_capsule *c = i < n ? lock[i] : twin[i - n];
        ocp_nlp_solver_destroy(c->nlp_solver);
        ocp_nlp_in_destroy(c->nlp_in); ocp_nlp_out_destroy(c->nlp_out); ocp_nlp_out_destroy(c->sens_out);
        ocp_nlp_solver_opts_destroy(c->nlp_opts); ocp_nlp_dims_destroy(c->nlp_dims); ocp_nlp_config_destroy(c->nlp_config);
        ocp_nlp_plan_destroy(c->nlp_solver_plan);
        free(c);
    }
    return 0;
}
