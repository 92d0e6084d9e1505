/*
 * ocp_qp_oracle.h -- CPU oracle for the batched OCP-QP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (acados_amd/, include/)
 * may include, link or call this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / CPU baseline.
 *
 * What it restates (citations relative to /root/reference):
 *   - problem formulation and field semantics:
 *       interfaces/acados_template/acados_template/acados_ocp_qp.py:24-45
 *       interfaces/acados_template/acados_template/acados_casadi_ocp_qp.py:52-260
 *   - constraint / multiplier ordering  lam,t = [lb; lg; ub; ug; ls; us], upper
 *     bounds entering with flipped sign, slack coupling through idxs_rev:
 *       acados/ocp_qp/ocp_qp_common.c:874-921   (ocp_qp_compute_t, restated verbatim
 *                                               in oqp_compute_t)
 *   - solver contract (cold start zeroes ux, option names, status codes):
 *       acados/ocp_qp/ocp_qp_hpipm.c:101-183, 314-405 ; acados/utils/types.h:74-87
 *   - the arithmetic itself (d_ocp_qp_ipm_solve) lives in giaf/hpipm (branch
 *     `stable`, SHA unrecoverable) which is ABSENT from /root/reference
 *     (external/hpipm is an empty submodule directory).  The algorithm below is a
 *     from-scratch restatement of its published method (Frison & Diehl 2020):
 *     infeasible-start primal-dual Mehrotra predictor-corrector IPM whose Newton
 *     systems are solved with a square-root Riccati recursion.
 *
 * PARITY PINNING: pinned at solution level against the reference's own golden
 * fixtures (tests/golden/: last_qp_*.json -> sqp_sol_*.json, lam/pi @1e-5,
 * examples/acados_python/tests/qp_test/test_ocpqp_solver.py:43-53) and against an
 * independent dense KKT solve (SciPy) on the input-only casadi_tests fixtures.
 * Iteration counts and per-iteration statistics are UNPINNED (no HPIPM here): the
 * safeguards of the iteration -- the conditional corrector (a step that would more
 * than double the duality measure is taken again from the centering term alone),
 * the scaling of the step to the boundary ((1 - a) 0.99 + a 0.9999999), sigma = (mu_aff/mu)^3 -- are restated from upstream
 * knowledge of HPIPM's solve loop; the golden vectors pin where the iteration ends,
 * not its path (ocp_qp_oracle.c, "conditional corrector").
 */
#ifndef OCP_QP_ORACLE_H_
#define OCP_QP_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oqp oqp;

/* options: names follow ocp_qp_hpipm_opts_set / d_ocp_qp_ipm_arg_set strings */
typedef struct
{
    double mu0;        /* initial barrier parameter (acados override: 1e0, ocp_qp_hpipm.c:111) */
    double tol_stat;   /* res_g_max */
    double tol_eq;     /* res_b_max */
    double tol_ineq;   /* res_d_max */
    double tol_comp;   /* res_m_max */
    double alpha_min;  /* 1e-8 (ocp_qp_hpipm.c:110) */
    double tau_min;    /* complementarity target floor */
    double lam_min;    /* 1e-16 */
    double t_min;      /* 1e-16 */
    double reg_prim;   /* 1e-15 */
    int iter_max;      /* 50 */
    int pred_corr;     /* 1: Mehrotra predictor-corrector */
    int cond_pred_corr;/* 1: the step is taken again with the centering term alone when it would more than double the duality measure */
    int warm_start;    /* 0: cold */
    int print_level;
    int t0_init;       /* 2; cold start of (t, lam): acados_ocp_options.py:1128-1143 */
} oqp_opts;

void oqp_opts_default(oqp_opts *opts);

/* acados status codes (acados/utils/types.h:74-87) */
enum { OQP_SUCCESS = 0, OQP_NAN_DETECTED = 1, OQP_MAXITER = 2, OQP_MINSTEP = 3, OQP_INFEASIBLE = 9 };

oqp *oqp_create(int N, const int *nx, const int *nu, const int *nbx, const int *nbu,
                const int *ng, const int *ns);
void oqp_free(oqp *qp);
/* a second, independent handle holding the same problem data */
oqp *oqp_clone(const oqp *src);

/* field keys are those of ocp_qp_in_set / d_ocp_qp_set as used by
 * acados_ocp_qp_solver.py:277-292: A B b Q S R q r idxb idxbx idxbu lbx ubx lbu ubu
 * C D lg ug Zl Zu zl zu lls lus idxs_rev idxe lbx_mask ubx_mask lbu_mask ubu_mask
 * lg_mask ug_mask lls_mask lus_mask.  Matrices are column-major.  Integer fields
 * take int*.  For idxe pass `n` entries (set nbxe first through oqp_set_nbxe). */
int oqp_set(oqp *qp, const char *field, int stage, const void *value);
void oqp_set_nbxe(oqp *qp, int stage, int nbxe);

int oqp_solve(oqp *qp, const oqp_opts *opts);

/* x u sl su pi lam t  (lam, t: 2*(nb+ng+ns) entries, ordering [lb lg ub ug ls us]) */
int oqp_get(const oqp *qp, const char *field, int stage, double *value);
/* factorise the Newton system at the current iterate; then oqp_get "ric_L" (n x n col-major
 * lower Cholesky factor of the stage matrix, variables [u;x]) and "ric_l" (n) are valid */
void oqp_refactor(oqp *qp, const oqp_opts *opts);
int oqp_get_iter(const oqp *qp);
/* stat: (iter+1) x 20 row-major, HPIPM column legend (acados_ocp_qp_solver.py:431-451) */
const double *oqp_get_stat(const oqp *qp);

/* restates ocp_qp_compute_t (ocp_qp_common.c:874-921) on the current ux */
void oqp_compute_t(oqp *qp);
/* the four KKT residual inf-norms of the current (ux,pi,lam,t):
 * res[0]=stat res[1]=eq res[2]=ineq res[3]=comp  (ocp_qp_res_compute + _nrm_inf,
 * ocp_qp_common.c:559-667; formulas from the KKT system in ocp_qp_clarabel.c:493-683) */
void oqp_res_nrm_inf(oqp *qp, double res[4]);

/* OpenMP batch idiom of acados_solver.in.c:3222-3243: independent QPs, one per
 * loop iteration. returns number of non-zero statuses; status[i] filled. */
int oqp_solve_batch(oqp **qps, int n, const oqp_opts *opts, int *status, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
