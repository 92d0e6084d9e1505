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

#include <stdbool.h>
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

/* KKT residual vectors (struct d_ocp_qp_res behind acados' ocp_qp_res, ocp_qp_common.h:52): res_g[k] nu+nx+2ns
 * (stationarity w.r.t. [u; x; sl; su]), res_b[k] nx_{k+1}, res_d[k] / res_m[k] 2(nb+ng+ns) ordered [lb lg ub ug ls us] */
typedef struct
{
    ocp_qp_dims *dim;
    double **res_g, **res_b, **res_d, **res_m;
} ocp_qp_res;
typedef struct ocp_qp_res_ws_ ocp_qp_res_ws; /* owns a one-instance device batch */

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

/* the condensing vtable: acados/ocp_qp/ocp_qp_common.h:84-107, 20 function pointers, same order and signatures */
typedef struct
{
    acados_size_t (*dims_calculate_size)(void *config, int N);
    void *(*dims_assign)(void *config, int N, void *raw_memory);
    void (*dims_set)(void *config, void *dims_, int stage, const char *field, int *value);
    void (*dims_get)(void *config, void *dims, const char *field, void *value);
    acados_size_t (*opts_calculate_size)(void *dims);
    void *(*opts_assign)(void *dims, void *raw_memory);
    void (*opts_initialize_default)(void *dims, void *opts);
    void (*opts_update)(void *dims, void *opts);
    void (*opts_set)(void *opts_, const char *field, void *value);
    acados_size_t (*memory_calculate_size)(void *dims, void *opts);
    void *(*memory_assign)(void *dims, void *opts, void *raw_memory);
    void (*memory_get)(void *config, void *mem, const char *field, void *value);
    acados_size_t (*workspace_calculate_size)(void *dims, void *opts);
    int (*condensing)(void *qp_in, void *x_cond_qp_in, void *opts, void *mem, void *work);
    int (*condense_rhs)(void *qp_in, void *x_cond_qp_in, void *opts, void *mem, void *work);
    int (*condense_rhs_seed)(void *qp_in, void *seed, void *xcond_seed, void *opts, void *mem, void *work);
    int (*condense_lhs)(void *qp_in, void *x_cond_qp_in, void *opts, void *mem, void *work);
    int (*condense_qp_out)(void *qp_in, void *x_cond_qp_in, void *qp_out, void *p_cond_qp_out, void *opts, void *mem, void *work);
    int (*expansion)(void *qp_in, void *qp_out, void *opts, void *mem, void *work);
    int (*expand_sol_seed)(void *qp_in, void *qp_out, void *opts, void *mem, void *work);
} ocp_qp_xcond_config;

/* the xcond-solver level: acados/ocp_qp/ocp_qp_xcond_solver.h:47-107 (dims / opts / memory / workspace structs and the
 * 22 function pointers + the two sub-vtables, same order and signatures) */
typedef struct
{
    ocp_qp_dims *orig_dims;
    void *xcond_dims;
} ocp_qp_xcond_solver_dims;

typedef struct ocp_qp_xcond_solver_opts_
{
    void *xcond_opts;
    void *qp_solver_opts;
    bool initialize_next_xcond_qp_from_qp_out;
} ocp_qp_xcond_solver_opts;

typedef struct ocp_qp_xcond_solver_memory_
{
    void *xcond_memory;
    void *solver_memory;
    void *xcond_qp_in;
    void *xcond_qp_out;
    void *xcond_seed;
} ocp_qp_xcond_solver_memory;

typedef struct ocp_qp_xcond_solver_workspace_
{
    void *xcond_work;
    void *qp_solver_work;
} ocp_qp_xcond_solver_workspace;

typedef struct
{
    acados_size_t (*dims_calculate_size)(void *config, int N);
    ocp_qp_xcond_solver_dims *(*dims_assign)(void *config, int N, void *raw_memory);
    void (*dims_set)(void *config_, ocp_qp_xcond_solver_dims *dims, int stage, const char *field, int *value);
    void (*dims_get)(void *config_, ocp_qp_xcond_solver_dims *dims, int stage, const char *field, int *value);
    acados_size_t (*opts_calculate_size)(void *config, ocp_qp_xcond_solver_dims *dims);
    void *(*opts_assign)(void *config, ocp_qp_xcond_solver_dims *dims, void *raw_memory);
    void (*opts_initialize_default)(void *config, ocp_qp_xcond_solver_dims *dims, void *opts);
    void (*opts_update)(void *config, ocp_qp_xcond_solver_dims *dims, void *opts);
    void (*opts_set)(void *config_, void *opts_, const char *field, void *value);
    void (*opts_get)(void *config_, void *opts_, const char *field, void *value);
    acados_size_t (*memory_calculate_size)(void *config, ocp_qp_xcond_solver_dims *dims, void *opts);
    void *(*memory_assign)(void *config, ocp_qp_xcond_solver_dims *dims, void *opts, void *raw_memory);
    void (*memory_get)(void *config_, void *mem_, const char *field, void *value);
    void (*solver_get)(void *config_, ocp_qp_in *qp_in, ocp_qp_out *qp_out, void *opts_, void *mem_, const char *field, int stage,
                       void *value, int size1, int size2);
    void (*memory_reset)(void *config, ocp_qp_xcond_solver_dims *dims, ocp_qp_in *qp_in, ocp_qp_out *qp_out, void *opts, void *mem,
                         void *work);
    acados_size_t (*workspace_calculate_size)(void *config, ocp_qp_xcond_solver_dims *dims, void *opts);
    int (*evaluate)(void *config, ocp_qp_xcond_solver_dims *dims, ocp_qp_in *qp_in, ocp_qp_out *qp_out, void *opts, void *mem, void *work);
    int (*condense_lhs)(void *config, ocp_qp_xcond_solver_dims *dims, ocp_qp_in *qp_in, ocp_qp_out *qp_out, void *opts, void *mem,
                        void *work);
    int (*condense_rhs_and_solve)(void *config, ocp_qp_xcond_solver_dims *dims, ocp_qp_in *qp_in, ocp_qp_out *qp_out, void *opts,
                                  void *mem, void *work);
    void (*eval_forw_sens)(void *config, ocp_qp_xcond_solver_dims *dims, ocp_qp_in *qp_in, ocp_qp_seed *seed, ocp_qp_out *sens_qp_out,
                           void *opts, void *mem, void *work);
    void (*eval_adj_sens)(void *config, ocp_qp_xcond_solver_dims *dims, ocp_qp_in *qp_in, ocp_qp_seed *seed, ocp_qp_out *sens_qp_out,
                          void *opts, void *mem, void *work);
    void (*terminate)(void *config, void *mem, void *work);
    qp_solver_config *qp_solver;
    ocp_qp_xcond_config *xcond;
} ocp_qp_xcond_solver_config;

typedef struct ocp_qp_xcond_solver
{
    ocp_qp_xcond_solver_config *config;
    ocp_qp_xcond_solver_dims *dims;
    ocp_qp_xcond_solver_opts *opts;
    ocp_qp_xcond_solver_memory *mem;
    ocp_qp_xcond_solver_workspace *work;
} ocp_qp_xcond_solver;
typedef ocp_qp_xcond_solver ocp_qp_solver; /* interfaces/acados_c/ocp_qp_interface.h */

/* dims / opts / memory of the DEVICE partial-condensing module behind ocp_qp_xcond_config (the reference's are
 * ocp_qp_partial_condensing.h:49-95 around HPIPM structs).  `condensed` = 0: this QP class is solved in the full space
 * (N2 = N requested, or beyond the kernel limits): xcond dims = original dims, condensing / expansion are copies. */
typedef struct
{
    ocp_qp_dims *orig_dims;
    ocp_qp_dims *pcond_dims;
    int *block_size; /* N2 + 1 entries in use */
    int condensed;
    /* the condensed dims come from the device library (a probe batch); the result is kept for the (orig dims, N2, block
     * sizes) it was computed for, so that the size queries acados makes (calculate_size, assign, ...) stay cheap */
    unsigned long long probe_key;
    int probe_valid;
} ocp_qp_partial_condensing_dims;

typedef struct
{
    int N2;
    int N2_bkp;
    int ric_alg;
    int mem_qp_in;
    int full_condensing;     /* the FULL_CONDENSING flavour: N2 = 1; past what one condensed stage carries: the dense path inside the solve (dense_kernels.hpp) */
    int *block_size;
    bool block_size_was_set;
} ocp_qp_partial_condensing_opts;

typedef struct
{
    ocp_qp_in *pcond_qp_in;
    ocp_qp_out *pcond_qp_out;
    ocp_qp_seed *pcond_qp_seed;
    qp_info *qp_out_info;    /* = pcond_qp_out->misc */
    double time_qp_xcond;
    ocp_qp_in *ptr_qp_in;
    ocp_qp_in *ptr_pcond_qp_in;
    ocp_qp_seed *ptr_qp_seed; /* seed of the last condense_rhs_seed (needed by expand_sol_seed, :648) */
    ocp_qp_partial_condensing_dims *dims;
    void *device;            /* device-side state (the original QP as a one-instance HBM batch) */
} ocp_qp_partial_condensing_memory;

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

acados_size_t ocp_qp_seed_calculate_size(ocp_qp_dims *dims);
ocp_qp_seed *ocp_qp_seed_assign(ocp_qp_dims *dims, void *raw_memory); /* zero-initialised */
ocp_qp_seed *ocp_qp_seed_create(ocp_qp_dims *dims);
void ocp_qp_seed_free(void *seed);

/* ---- KKT residuals of an arbitrary (qp_in, qp_out): ocp_qp_res_* of acados/ocp_qp/ocp_qp_common.c:497-667 and the wrapper
 *      ocp_qp_inf_norm_residuals (interfaces/acados_c/ocp_qp_interface.c:642-650) the reference's unit test asserts on
 *      (test/ocp_qp/test_qpsolvers.cpp:240-251).  Evaluated on the device by a kernel independent of the IPM sweeps.
 *      res = [stationarity, dynamics, inequalities, complementarity] inf-norms. */
acados_size_t ocp_qp_res_calculate_size(ocp_qp_dims *dims);
ocp_qp_res *ocp_qp_res_assign(ocp_qp_dims *dims, void *raw_memory);
ocp_qp_res *ocp_qp_res_create(ocp_qp_dims *dims);
void ocp_qp_res_free(void *res);
acados_size_t ocp_qp_res_workspace_calculate_size(ocp_qp_dims *dims);
ocp_qp_res_ws *ocp_qp_res_workspace_assign(ocp_qp_dims *dims, void *raw_memory);
ocp_qp_res_ws *ocp_qp_res_workspace_create(ocp_qp_dims *dims);
void ocp_qp_res_workspace_free(ocp_qp_res_ws *ws); /* releases the device batch + the block of _create */
void ocp_qp_res_compute(ocp_qp_in *qp_in, ocp_qp_out *qp_out, ocp_qp_res *qp_res, ocp_qp_res_ws *res_ws);
void ocp_qp_res_compute_nrm_inf(ocp_qp_res *qp_res, double res[4]);
void ocp_qp_inf_norm_residuals(ocp_qp_dims *dims, ocp_qp_in *qp_in, ocp_qp_out *qp_out, double *res);

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

/* ---- the condensing module: the 20 slots of ocp_qp_xcond_config on the DEVICE condensing kernels
 *      (fill: ocp_qp_partial_condensing.c:720-750).  opts strings: "N" "N_bkp" "ric_alg" "block_size" (:269-312);
 *      dims_get "xcond_dims" (:138-155); memory_get "xcond_qp_in" "xcond_qp_out" "xcond_seed" "qp_out_info"
 *      "time_qp_xcond" (:467-504).  _fcond_ = the same module condensing to ONE block (FULL_CONDENSING_GPU_IPM). ---- */
void ocp_qp_gpu_pcond_config_initialize_default(void *config);
void ocp_qp_gpu_fcond_config_initialize_default(void *config);
acados_size_t ocp_qp_gpu_pcond_dims_calculate_size(void *config, int N);
void *ocp_qp_gpu_pcond_dims_assign(void *config, int N, void *raw_memory);
void ocp_qp_gpu_pcond_dims_set(void *config, void *dims, int stage, const char *field, int *value);
void ocp_qp_gpu_pcond_dims_get(void *config, void *dims, const char *field, void *value);
acados_size_t ocp_qp_gpu_pcond_opts_calculate_size(void *dims);
void *ocp_qp_gpu_pcond_opts_assign(void *dims, void *raw_memory);
void ocp_qp_gpu_pcond_opts_initialize_default(void *dims, void *opts);
void ocp_qp_gpu_fcond_opts_initialize_default(void *dims, void *opts);
void ocp_qp_gpu_pcond_opts_update(void *dims, void *opts);
void ocp_qp_gpu_pcond_opts_set(void *opts, const char *field, void *value);
acados_size_t ocp_qp_gpu_pcond_memory_calculate_size(void *dims, void *opts);
void *ocp_qp_gpu_pcond_memory_assign(void *dims, void *opts, void *raw_memory);
void ocp_qp_gpu_pcond_memory_get(void *config, void *mem, const char *field, void *value);
acados_size_t ocp_qp_gpu_pcond_workspace_calculate_size(void *dims, void *opts);
int ocp_qp_gpu_pcond_condensing(void *qp_in, void *pcond_qp_in, void *opts, void *mem, void *work);
int ocp_qp_gpu_pcond_condense_lhs(void *qp_in, void *pcond_qp_in, void *opts, void *mem, void *work);
int ocp_qp_gpu_pcond_condense_rhs(void *qp_in, void *pcond_qp_in, void *opts, void *mem, void *work);
int ocp_qp_gpu_pcond_condense_qp_out(void *qp_in, void *pcond_qp_in, void *qp_out, void *pcond_qp_out, void *opts, void *mem, void *work);
int ocp_qp_gpu_pcond_condense_rhs_seed(void *qp_in, void *seed, void *pcond_seed, void *opts, void *mem, void *work);
int ocp_qp_gpu_pcond_expansion(void *pcond_qp_out, void *qp_out, void *opts, void *mem, void *work);
int ocp_qp_gpu_pcond_expand_sol_seed(void *pcond_qp_out, void *qp_out, void *opts, void *mem, void *work);
/* device-side resources of a module memory (the xcond solver's terminate and ocp_qp_condensing_free call it) */
void ocp_qp_gpu_pcond_memory_release(void *mem);

/* ---- condensing-only boundary, names and call sequence of interfaces/acados_c/condensing_interface.h:42-75 ---- */
typedef enum { PARTIAL_CONDENSING, FULL_CONDENSING } condensing_t;
typedef struct { condensing_t condensing_type; } condensing_plan;
typedef struct
{
    ocp_qp_xcond_config *config;
    void *dims;
    void *opts;
    void *mem;
    void *work;
} condensing_module;
ocp_qp_xcond_config *ocp_qp_condensing_config_create(condensing_plan *plan);
void *ocp_qp_condensing_dims_create(ocp_qp_xcond_config *config, int N); /* calloc + config->dims_assign; set with config->dims_set */
void *ocp_qp_condensing_opts_create(ocp_qp_xcond_config *config, void *dims_);
acados_size_t ocp_qp_condensing_calculate_size(ocp_qp_xcond_config *config, void *dims_, void *opts_);
condensing_module *ocp_qp_condensing_assign(ocp_qp_xcond_config *config, void *dims_, void *opts_, void *raw_memory);
condensing_module *ocp_qp_condensing_create(ocp_qp_xcond_config *config, void *dims_, void *opts_);
void ocp_qp_condensing_free(condensing_module *module); /* device batch + the block of _create */
/* (qp_in, xcond_qp_in) resp. (xcond_qp_out, qp_out), as condensing_interface.c:116-124 forwards them */
int ocp_qp_condense(condensing_module *module, void *qp_in, void *qp_out);
int ocp_qp_expand(condensing_module *module, void *qp_in, void *qp_out);

/* ---- the xcond-solver level: the 22 slots of ocp_qp_xcond_solver_config (fill: ocp_qp_xcond_solver.c:744-770) as ONE fused
 *      device path: evaluate = pack the original QP, condense (cond_N / cond_block_size of the opts of this call), solve,
 *      expand, unpack.  condense_lhs / condense_rhs_and_solve keep the matrix part resident in HBM between the two calls. ---- */
void ocp_qp_gpu_xcond_solver_config_initialize_default(void *config);
acados_size_t ocp_qp_gpu_xcond_solver_dims_calculate_size(void *config, int N);
ocp_qp_xcond_solver_dims *ocp_qp_gpu_xcond_solver_dims_assign(void *config, int N, void *raw_memory);
void ocp_qp_gpu_xcond_solver_dims_set_(void *config, ocp_qp_xcond_solver_dims *dims, int stage, const char *field, int *value);
void ocp_qp_gpu_xcond_solver_dims_get_(void *config, ocp_qp_xcond_solver_dims *dims, int stage, const char *field, int *value);
acados_size_t ocp_qp_gpu_xcond_solver_opts_calculate_size(void *config, ocp_qp_xcond_solver_dims *dims);
void *ocp_qp_gpu_xcond_solver_opts_assign(void *config, ocp_qp_xcond_solver_dims *dims, void *raw_memory);
void ocp_qp_gpu_xcond_solver_opts_initialize_default(void *config, ocp_qp_xcond_solver_dims *dims, void *opts);
void ocp_qp_gpu_xcond_solver_opts_update(void *config, ocp_qp_xcond_solver_dims *dims, void *opts);
void ocp_qp_gpu_xcond_solver_opts_set_(void *config, void *opts, const char *field, void *value);
void ocp_qp_gpu_xcond_solver_opts_get_(void *config, void *opts, const char *field, void *value);
acados_size_t ocp_qp_gpu_xcond_solver_memory_calculate_size(void *config, ocp_qp_xcond_solver_dims *dims, void *opts);
void *ocp_qp_gpu_xcond_solver_memory_assign(void *config, ocp_qp_xcond_solver_dims *dims, void *opts, void *raw_memory);
void ocp_qp_gpu_xcond_solver_memory_get(void *config, void *mem, const char *field, void *value);
void ocp_qp_gpu_xcond_solver_get(void *config, ocp_qp_in *qp_in, ocp_qp_out *qp_out, void *opts, void *mem, const char *field, int stage,
                                 void *value, int size1, int size2);
void ocp_qp_gpu_xcond_solver_memory_reset(void *config, ocp_qp_xcond_solver_dims *dims, ocp_qp_in *qp_in, ocp_qp_out *qp_out, void *opts,
                                          void *mem, void *work);
acados_size_t ocp_qp_gpu_xcond_solver_workspace_calculate_size(void *config, ocp_qp_xcond_solver_dims *dims, void *opts);
int ocp_qp_gpu_xcond_solve(void *config, ocp_qp_xcond_solver_dims *dims, ocp_qp_in *qp_in, ocp_qp_out *qp_out, void *opts, void *mem,
                           void *work);
int ocp_qp_gpu_xcond_condense_lhs(void *config, ocp_qp_xcond_solver_dims *dims, ocp_qp_in *qp_in, ocp_qp_out *qp_out, void *opts,
                                  void *mem, void *work);
int ocp_qp_gpu_xcond_condense_rhs_and_solve(void *config, ocp_qp_xcond_solver_dims *dims, ocp_qp_in *qp_in, ocp_qp_out *qp_out,
                                            void *opts, void *mem, void *work);
void ocp_qp_gpu_xcond_solver_eval_forw_sens(void *config, ocp_qp_xcond_solver_dims *dims, ocp_qp_in *qp_in, ocp_qp_seed *seed,
                                            ocp_qp_out *sens_qp_out, void *opts, void *mem, void *work);
void ocp_qp_gpu_xcond_solver_eval_adj_sens(void *config, ocp_qp_xcond_solver_dims *dims, ocp_qp_in *qp_in, ocp_qp_seed *seed,
                                           ocp_qp_out *sens_qp_out, void *opts, void *mem, void *work);
void ocp_qp_gpu_xcond_solver_terminate(void *config, void *mem, void *work);

/* ---- outer level as the Python/C drivers use it (interfaces/acados_c/ocp_qp_interface.c:185-259, 262-331, 483-650;
 *      ctypes call list in acados_ocp_qp_solver.py:108-168) ---- */
acados_size_t ocp_qp_xcond_solver_config_calculate_size(void);
ocp_qp_xcond_solver_config *ocp_qp_xcond_solver_config_assign(void *raw_memory);
/* accepted names: "PARTIAL_CONDENSING_GPU_IPM", as the drop-in alias that keeps existing scripts unchanged
 * "PARTIAL_CONDENSING_HPIPM", and "FULL_CONDENSING_GPU_IPM" (one block where nx + N nu <= 64; the dense path of dense_kernels.hpp beyond);
 * anything else returns NULL after printing the reference's message */
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
acados_size_t ocp_qp_calculate_size(ocp_qp_xcond_solver_config *config, ocp_qp_xcond_solver_dims *dims, void *opts_);
ocp_qp_solver *ocp_qp_assign(ocp_qp_xcond_solver_config *config, ocp_qp_xcond_solver_dims *dims, void *opts_, void *raw_memory);
ocp_qp_solver *ocp_qp_create(ocp_qp_xcond_solver_config *config, ocp_qp_xcond_solver_dims *dims, void *opts);
void ocp_qp_solver_destroy(ocp_qp_solver *solver); /* terminate slot (device-side resources) + the block of ocp_qp_create */
int ocp_qp_solve(ocp_qp_solver *solver, ocp_qp_in *qp_in, ocp_qp_out *qp_out);
/* RTI split through the condense_lhs / condense_rhs_and_solve slots (ocp_nlp_sqp_rti.c:509, 1115) */
int ocp_qp_condense_lhs(ocp_qp_solver *solver, ocp_qp_in *qp_in, ocp_qp_out *qp_out);
int ocp_qp_condense_rhs_and_solve(ocp_qp_solver *solver, ocp_qp_in *qp_in, ocp_qp_out *qp_out);
/* batch extension: n (qp_in, qp_out) pairs of identical structure, one device batch */
int ocp_qp_solve_batch(ocp_qp_solver *solver, int n, ocp_qp_in **qp_in, ocp_qp_out **qp_out, int *status);
void ocp_qp_xcond_solver_get_scalar(ocp_qp_solver *solver, ocp_qp_out *qp_out, const char *field, void *value);
void ocp_qp_solver_get_stats(ocp_qp_solver *solver, double *stat, const char *qp_solver_name);
/* solution sensitivities through the eval_forw_sens / eval_adj_sens slots (ocp_nlp reaches them through the
 * vtable, ocp_nlp_common.c:4091, 4141): d(solution)/d(parameter) of the QP solved last, into sens_out (ux, pi, lam, t) */
void ocp_qp_solver_eval_forw_sens(ocp_qp_solver *solver, ocp_qp_in *qp_in, ocp_qp_seed *seed, ocp_qp_out *sens_out);
void ocp_qp_solver_eval_adj_sens(ocp_qp_solver *solver, ocp_qp_in *qp_in, ocp_qp_seed *seed, ocp_qp_out *sens_out);
/* Riccati quantities of the last factorisation through the solver_get slot: field in P p K k Lr
 * (ocp_qp_hpipm.c:417-478); column-major; u = K x + k */
void ocp_qp_solver_get_ric(ocp_qp_solver *solver, ocp_qp_in *qp_in, ocp_qp_out *qp_out, const char *field, int stage,
                           void *value, int size1, int size2);

#ifdef __cplusplus
}
#endif
#endif
