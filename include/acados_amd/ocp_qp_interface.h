/*
 * ocp_qp_interface.h -- acados-shaped C API of the MI355X OCP-QP backend.
 *
 * Every entry point below has the name, argument meaning and error behaviour of the
 * reference function it stands in for (paths relative to /root/reference):
 *
 *   types            acados/ocp_qp/ocp_qp_common.h:49-54 (ocp_qp_dims/in/out are typedefs of
 *                    HPIPM structs there; here they are plain column-major C containers,
 *                    because HPIPM/BLASFEO headers are absent -- see INTEGRATION.md for the
 *                    blasfeo_unpack adapter a maintainer adds in an acados build)
 *   qp_info          acados/ocp_qp/ocp_qp_common.h:114-122
 *   qp_solver_config acados/ocp_qp/ocp_qp_common.h:60-79   (17 function pointers, same order)
 *   ocp_qp_gpu_ipm_* acados/ocp_qp/ocp_qp_hpipm.c:60-540   (the slot this backend takes)
 *   xcond solver     acados/ocp_qp/ocp_qp_xcond_solver.h:81-107, .c:529-587
 *   create/solve/get interfaces/acados_c/ocp_qp_interface.c:185-259, 300-480, 513-571
 *   batch entry      NEW (SURVEY 8b "Threading"): replaces the OpenMP loop of
 *                    c_templates_tera/acados_solver.in.c:3222-3243
 *
 * Status codes are acados' return_values_t (acados/utils/types.h:74-87).
 */
#ifndef ACADOS_AMD_OCP_QP_INTERFACE_H_
#define ACADOS_AMD_OCP_QP_INTERFACE_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef size_t acados_size_t;

typedef enum
{
    ACADOS_UNKNOWN = -1,
    ACADOS_SUCCESS = 0,
    ACADOS_NAN_DETECTED = 1,
    ACADOS_MAXITER = 2,
    ACADOS_MINSTEP = 3,
    ACADOS_QP_FAILURE = 4,
    ACADOS_READY = 5,
    ACADOS_UNBOUNDED = 6,
    ACADOS_TIMEOUT = 7,
    ACADOS_QPSCALING_BOUNDS_NOT_SATISFIED = 8,
    ACADOS_INFEASIBLE = 9,
} return_values_t;

/* fields acados reads from struct d_ocp_qp_dim (print.c:197-209, ocp_qp_common.c:166-169) */
typedef struct
{
    int N;
    int *nx, *nu, *nb, *nbx, *nbu, *ng, *ns, *nbxe, *nbue, *nge;
} ocp_qp_dims;

typedef struct
{
    double solve_QP_time;
    double condensing_time;
    double interface_time;
    double total_time;
    int num_iter;
    int t_computed;
} qp_info;

/* plain column-major containers (per-stage arrays, natural-sign bounds) */
typedef struct
{
    ocp_qp_dims *dim;
    double **A, **B, **b, **Q, **S, **R, **q, **r;
    int **idxb;
    double **lb, **ub, **lb_mask, **ub_mask;            /* nb entries: [bu; bx] */
    double **C, **D, **lg, **ug, **lg_mask, **ug_mask;
    double **Zl, **Zu, **zl, **zu, **lls, **lus, **lls_mask, **lus_mask;
    int **idxs_rev, **idxe;
} ocp_qp_in;

typedef struct
{
    ocp_qp_dims *dim;
    double **ux;  /* [u; x; sl; su] */
    double **pi;
    double **lam; /* [lb lg ub ug ls us] */
    double **t;
    void *misc;   /* qp_info */
} ocp_qp_out;

/* seed of a solution-sensitivity solve (struct d_ocp_qp_seed behind acados' ocp_qp_seed, ocp_qp_common.h):
 * derivative of the problem data w.r.t. a parameter.  seed_g[k] = d[r; q; zl; zu], seed_b[k] = d b,
 * seed_d[k] = d[lb lg ub ug ls us] in natural sign (ocp_nlp_common.c:4057-4064 seeds +1 on both sides of an x0 row),
 * seed_m[k] (complementarity; not used by acados, ignored) */
typedef struct
{
    ocp_qp_dims *dim;
    double **seed_g, **seed_b, **seed_d, **seed_m;
} ocp_qp_seed;

typedef struct
{
    void (*dims_set)(void *config_, void *dims_, int stage, const char *field, int *value);
    acados_size_t (*opts_calculate_size)(void *config, void *dims);
    void *(*opts_assign)(void *config, void *dims, void *raw_memory);
    void (*opts_initialize_default)(void *config, void *dims, void *opts);
    void (*opts_update)(void *config, void *dims, void *opts);
    void (*opts_set)(void *config_, void *opts_, const char *field, void *value);
    void (*opts_get)(void *config_, void *opts_, const char *field, void *value);
    acados_size_t (*memory_calculate_size)(void *config, void *dims, void *opts);
    void *(*memory_assign)(void *config, void *dims, void *opts, void *raw_memory);
    void (*memory_get)(void *config_, void *mem_, const char *field, void *value);
    acados_size_t (*workspace_calculate_size)(void *config, void *dims, void *opts);
    int (*evaluate)(void *config, void *qp_in, void *qp_out, void *opts, void *mem, void *work);
    void (*solver_get)(void *config_, void *qp_in_, void *qp_out_, void *opts_, void *mem_, const char *field,
                       int stage, void *value, int size1, int size2);
    void (*memory_reset)(void *config, void *qp_in, void *qp_out, void *opts, void *mem, void *work);
    void (*eval_forw_sens)(void *config, void *qp_in, void *seed, void *qp_out, void *opts, void *mem, void *work);
    void (*eval_adj_sens)(void *config, void *qp_in, void *seed, void *qp_out, void *opts, void *mem, void *work);
    void (*terminate)(void *config, void *mem, void *work);
} qp_solver_config;

/* ---- dims / in / out (ocp_qp_common.c:100-260, ocp_qp_interface.c:300-480) ---- */
acados_size_t ocp_qp_dims_calculate_size(int N);
ocp_qp_dims *ocp_qp_dims_assign(int N, void *raw_memory);
ocp_qp_dims *ocp_qp_dims_create(int N);
void ocp_qp_dims_set(void *config_, void *dims, int stage, const char *field, int *value);
void ocp_qp_dims_get(void *config_, void *dims, int stage, const char *field, int *value);
void ocp_qp_dims_free(void *dims);

acados_size_t ocp_qp_in_calculate_size(ocp_qp_dims *dims);
ocp_qp_in *ocp_qp_in_assign(ocp_qp_dims *dims, void *raw_memory);
ocp_qp_in *ocp_qp_in_create(ocp_qp_dims *dims);
void ocp_qp_in_set(void *config, ocp_qp_in *in, int stage, char *field, void *value);
void ocp_qp_in_free(void *in);

acados_size_t ocp_qp_out_calculate_size(ocp_qp_dims *dims);
ocp_qp_out *ocp_qp_out_assign(ocp_qp_dims *dims, void *raw_memory);
ocp_qp_out *ocp_qp_out_create(ocp_qp_dims *dims);
void ocp_qp_out_get(ocp_qp_out *out, int stage, const char *field, void *value);
void ocp_qp_out_free(void *out);

ocp_qp_seed *ocp_qp_seed_create(ocp_qp_dims *dims); /* zero-initialised */
void ocp_qp_seed_free(void *seed);

/* t = slack of every inequality at the current ux (ocp_qp_common.c:874-921) */
void ocp_qp_compute_t(ocp_qp_in *qp_in, ocp_qp_out *qp_out);

/* ---- inner plugin: the qp_solver_config slot (ocp_qp_hpipm.c:517-540) ---- */
void ocp_qp_gpu_ipm_config_initialize_default(void *config);
acados_size_t ocp_qp_gpu_ipm_opts_calculate_size(void *config, void *dims);
void *ocp_qp_gpu_ipm_opts_assign(void *config, void *dims, void *raw_memory);
void ocp_qp_gpu_ipm_opts_initialize_default(void *config, void *dims, void *opts);
void ocp_qp_gpu_ipm_opts_update(void *config, void *dims, void *opts);
void ocp_qp_gpu_ipm_opts_set(void *config, void *opts, const char *field, void *value);
void ocp_qp_gpu_ipm_opts_get(void *config, void *opts, const char *field, void *value);
acados_size_t ocp_qp_gpu_ipm_memory_calculate_size(void *config, void *dims, void *opts);
void *ocp_qp_gpu_ipm_memory_assign(void *config, void *dims, void *opts, void *raw_memory);
void ocp_qp_gpu_ipm_memory_get(void *config, void *mem, const char *field, void *value);
acados_size_t ocp_qp_gpu_ipm_workspace_calculate_size(void *config, void *dims, void *opts);
int ocp_qp_gpu_ipm(void *config, void *qp_in, void *qp_out, void *opts, void *mem, void *work);
void ocp_qp_gpu_ipm_solver_get(void *config, void *qp_in, void *qp_out, void *opts, void *mem, const char *field,
                               int stage, void *value, int size1, int size2);
void ocp_qp_gpu_ipm_memory_reset(void *config, void *qp_in, void *qp_out, void *opts, void *mem, void *work);
void ocp_qp_gpu_ipm_eval_forw_sens(void *config, void *qp_in, void *seed, void *qp_out, void *opts, void *mem, void *work);
void ocp_qp_gpu_ipm_eval_adj_sens(void *config, void *qp_in, void *seed, void *qp_out, void *opts, void *mem, void *work);
void ocp_qp_gpu_ipm_terminate(void *config, void *mem, void *work);
/* batch extension of `evaluate`: n QPs of identical structure in one device batch.
 * mem[0] owns the device batch; status[i] receives the per-instance acados status.
 * Returns the worst status. */
int ocp_qp_gpu_ipm_evaluate_batch(void *config, int n, void **qp_in, void **qp_out, void *opts, void **mem,
                                  void *work, int *status);

/* ---- outer level as the Python/C drivers use it (ocp_qp_interface.c:185-259, 513-650;
 *      ctypes call list in acados_ocp_qp_solver.py:108-168) ---- */
typedef struct ocp_qp_xcond_solver_config_ ocp_qp_xcond_solver_config;
typedef struct ocp_qp_xcond_solver_dims_ ocp_qp_xcond_solver_dims;
typedef struct ocp_qp_solver_ ocp_qp_solver;

/* accepted names: "PARTIAL_CONDENSING_GPU_IPM" and, as the drop-in alias that keeps
 * existing scripts unchanged, "PARTIAL_CONDENSING_HPIPM"; anything else returns NULL
 * after printing the reference's message */
ocp_qp_xcond_solver_config *ocp_qp_xcond_solver_config_create_from_name(const char *qp_solver_name);
void ocp_qp_xcond_solver_config_free(ocp_qp_xcond_solver_config *config);
ocp_qp_xcond_solver_dims *ocp_qp_xcond_solver_dims_create(ocp_qp_xcond_solver_config *config, int N);
void ocp_qp_xcond_solver_dims_set(void *config, ocp_qp_xcond_solver_dims *dims, int stage, const char *field, int *value);
void ocp_qp_xcond_solver_dims_free(ocp_qp_xcond_solver_dims *dims);
void *ocp_qp_xcond_solver_opts_create(ocp_qp_xcond_solver_config *config, ocp_qp_xcond_solver_dims *dims);
void ocp_qp_xcond_solver_opts_set(ocp_qp_xcond_solver_config *config, void *opts, const char *field, void *value);
void ocp_qp_xcond_solver_opts_free(void *opts);
ocp_qp_in *ocp_qp_in_create_from_xcond_dims(ocp_qp_xcond_solver_dims *dims);
ocp_qp_out *ocp_qp_out_create_from_xcond_dims(ocp_qp_xcond_solver_dims *dims);
ocp_qp_solver *ocp_qp_create(ocp_qp_xcond_solver_config *config, ocp_qp_xcond_solver_dims *dims, void *opts);
void ocp_qp_solver_destroy(ocp_qp_solver *solver);
int ocp_qp_solve(ocp_qp_solver *solver, ocp_qp_in *qp_in, ocp_qp_out *qp_out);
/* batch extension: n (qp_in, qp_out) pairs of identical structure, one device batch */
int ocp_qp_solve_batch(ocp_qp_solver *solver, int n, ocp_qp_in **qp_in, ocp_qp_out **qp_out, int *status);
void ocp_qp_xcond_solver_get_scalar(ocp_qp_solver *solver, ocp_qp_out *qp_out, const char *field, void *value);
void ocp_qp_solver_get_stats(ocp_qp_solver *solver, double *stat, const char *qp_solver_name);
/* solution sensitivities through the eval_forw_sens / eval_adj_sens slots (ocp_nlp reaches them through the
 * vtable, ocp_nlp_common.c:4091, 4141): d(solution)/d(parameter) of the QP solved last, into sens_out (ux, pi, lam, t) */
void ocp_qp_solver_eval_forw_sens(ocp_qp_solver *solver, ocp_qp_in *qp_in, ocp_qp_seed *seed, ocp_qp_out *sens_out);
void ocp_qp_solver_eval_adj_sens(ocp_qp_solver *solver, ocp_qp_in *qp_in, ocp_qp_seed *seed, ocp_qp_out *sens_out);
/* ---- condensing-only boundary: interfaces/acados_c/condensing_interface.h:61-75 (ocp_qp_condensing_create,
 *      ocp_qp_condense, ocp_qp_expand) over the `condensing` / `expansion` slots of ocp_qp_xcond_config
 *      (acados/ocp_qp/ocp_qp_common.h:84-107; partial condensing: ocp_qp_partial_condensing.c:523-556, :664-689).
 * cond_N = N2; block_size: N2 + 1 entries summing to N with a trailing 0, or NULL for the default split
 * (d_part_cond_qp_compute_block_size: N / N2 each, remainder to the first blocks).  _get_xcond_dims is what
 * dims_get("xcond_dims") answers in the reference: create the condensed ocp_qp_in / ocp_qp_out from it.
 * ocp_qp_condense writes the condensed QP (data, index sets, masks) into xcond_qp_in; ocp_qp_expand takes a solution
 * of the condensed QP (ux, pi, lam, t) and writes the solution of the original one.  Both return ACADOS_SUCCESS (0) or
 * ACADOS_QP_FAILURE. */
typedef struct ocp_qp_condensing_module_ ocp_qp_condensing_module;
ocp_qp_condensing_module *ocp_qp_condensing_create(ocp_qp_dims *dims, int cond_N, const int *block_size);
void ocp_qp_condensing_free(ocp_qp_condensing_module *module);
ocp_qp_dims *ocp_qp_condensing_get_xcond_dims(ocp_qp_condensing_module *module);
int ocp_qp_condense(ocp_qp_condensing_module *module, void *qp_in, void *xcond_qp_in);
int ocp_qp_expand(ocp_qp_condensing_module *module, void *xcond_qp_out, void *qp_out);
/* Riccati quantities of the last factorisation through the solver_get slot: field in P p K k Lr
 * (ocp_qp_hpipm.c:417-478); column-major; u = K x + k */
void ocp_qp_solver_get_ric(ocp_qp_solver *solver, ocp_qp_in *qp_in, ocp_qp_out *qp_out, const char *field, int stage,
                           void *value, int size1, int size2);

#ifdef __cplusplus
}
#endif
#endif
