/*
 * ocp_qp_gpu_ipm.c -- the acados-side adapter of the MI355X OCP-QP backend: the file a maintainer drops into
 * acados/ocp_qp/ next to ocp_qp_hpipm.c.  It is written against acados' OWN types -- ocp_qp_in / ocp_qp_out are HPIPM's
 * d_ocp_qp / d_ocp_qp_sol holding BLASFEO matrices (acados/ocp_qp/ocp_qp_common.h:49-54), panel-major in the default
 * build (external/CMakeLists.txt:46) -- and talks to libacados_amd_qp.so through the device-batch C-ABI
 * (include/acados_amd/ocp_qp_gpu_batch.h) only.  Two entry levels:
 *
 *   1. the 17 slots of qp_solver_config (ocp_qp_common.h:60-79), one QP per call:
 *          ocp_qp_gpu_ipm_acados_config_initialize_default(config->qp_solver);
 *      in the `case PARTIAL_CONDENSING_GPU_IPM:` of ocp_qp_xcond_solver_config_initialize_from_plan
 *      (interfaces/acados_c/ocp_qp_interface.c:91-182), with HPIPM's partial condensing in the xcond slot
 *      (INTEGRATION.md section 3);
 *   2. the BATCH entries the reference does not have (SURVEY 8b "Threading"): n capsules' QPs -- acados structs, one
 *      `mem` each -- go to the GPU as ONE device batch per structure class:
 *          ocp_qp_gpu_ipm_acados_evaluate_batch(config, n, qp_in[], qp_out[], opts, mem[], work)
 *          ocp_qp_gpu_ipm_acados_eval_sens_batch(config, n, qp_in[], seed[], sens_qp_out[], opts, mem[], work)
 *      They replace the per-capsule loops `#pragma omp parallel for ... ocp_nlp_solve(capsule[i])` ->
 *      config->evaluate(...) of acados_solver.in.c:3222-3243 (call site ocp_nlp_common.c:4517) and
 *      acados_solver.in.c:3292-3337 (call sites ocp_nlp_common.c:4091, 4141): host threads unpack the n panel-major QPs
 *      into one pinned blob, ONE host->device copy + ONE scatter launch, one device batch solve, one gather launch + one
 *      copy back, host threads pack the n qp_out.  QPs that differ in structure (dims, idxb, idxs_rev, idxe) are bucketed
 *      by structure signature, one device batch per bucket, buckets solved concurrently.
 *
 * Data access rule followed (pattern of acados/ocp_qp/ocp_qp_clarabel.c:205-683, 1018-1072): matrices only
 * through blasfeo_unpack_dmat / blasfeo_unpack_tran_dmat, vectors through blasfeo_unpack_dvec / blasfeo_pack_dvec;
 * r, q, b are taken from the VECTORS rqz / b, never from the last rows of RSQrq / BAbt (ocp_nlp writes only the vectors,
 * ocp_nlp_common.c:3119-3138); d = [lb; lg; -ub; -ug; ls; us] (ocp_qp_common.c:897-906) -> natural-sign bounds; every
 * member array is re-read on every evaluate (they alias ocp_nlp memory, ocp_nlp_common.c:2797-2894).
 *
 * Memory rule: opts and memory are carved from the caller's block (sizes from dims) -- staging, structure signature and
 * the segment tables of the single-QP path included: no malloc in `evaluate`.  Kept outside the block: the device batch
 * and its stream (released by `terminate`, ocp_qp_common.h:78) and, for the batch entries, the per-bucket tables and
 * pinned staging, sized once per (n, structure) and owned by mem[0] of the call (released by its terminate / reset).
 *
 * In this repository the file is compiled and RUN in the test tiers against tests/mock_acados/include (stand-ins for
 * the HPIPM / BLASFEO / acados declarations restated from the fields acados touches): tests/test_mock_acados.py.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#if defined(_OPENMP)
#include <omp.h>
#endif

#include "acados/ocp_qp/ocp_qp_common.h"
#include "acados/utils/types.h"
#include "blasfeo/include/blasfeo_d_aux.h"

#include "acados_amd/ocp_qp_gpu_batch.h"
#include "ocp_qp_gpu_segments.h" /* segment tables: device blobs <-> BLASFEO containers (shared with ocp_qp_gpu_pcond.c) */

typedef struct
{
    double mu0, tol_stat, tol_eq, tol_ineq, tol_comp, alpha_min, tau_min, reg_prim, t0_min, lam0_min;
    double tol_comp_soft_scale; /* backend-specific: exit tolerance on complementarity of soft-constrained classes = tol_comp * this */
    int iter_max, warm_start, print_level, ric_alg, t0_init, update_fact_exit, cond_pred_corr;
    /* extension (not an HPIPM name; the outer solver never forwards "cond_" strings, ocp_qp_xcond_solver.c:294-297): partial
     * condensing INSIDE the device solve -- the QPs handed to evaluate / evaluate_batch are the original ones, condensing, IPM and
     * expansion run back to back on the device.  Set by ocp_qp_gpu_xcond_solver_acados_evaluate_batch (ocp_qp_gpu_pcond.c) from
     * the condensing module's options, or directly by a harness; 0 / N: off */
    int cond_N, cond_block_size_set, cond_block_cap;
    int *cond_block_size;        /* carved behind the struct: N + 2 entries */
    struct ocp_qp_gpu_ipm_rendezvous_ *rendezvous; /* set: `evaluate` waits for the other capsules and the QPs go as one batch */
} ocp_qp_gpu_ipm_opts;

/* one structure class = one device batch */
typedef struct
{
    union { gpu_layout lay; struct { GPU_LAYOUT_MEMBERS }; }; /* bk->batch, bk->seg_in, ... ARE bk->lay.batch, ... */
    int n;                       /* instances */
    int *sig, sig_len, sig_cap;  /* structure the batch was built for */
    double *blob_in, *blob_out;  /* staging: carved (single QP) or pinned (batch entries) */
    int blob_clean;              /* batch entries: every position of blob_in that no segment writes is zero (nothing but QP data has been staged
                                    there since the last full clear): the per-call memset -- a second pass over 85 KB per QP -- is skipped */
    size_t cap_in, cap_out;      /* doubles */
    int lhs_resident;            /* batch entries: the matrices of every instance are on the device and (cond_N < N) condensed -- set by the
                                    condense_lhs batch entry, consumed by the condense_rhs_and_solve one (an RTI step's two halves) */
    int *members;                /* batch entries: index of each instance in the caller's arrays */
    /* batch entries, zero-copy (zc_prepare): sources per instance, the class's word tables on the device (1) / refused (-1), the cn the
     * tables were built with and the doubles they read of every source, this call's source addresses (pinned, n x zc_P) */
    int zc_P, zc_tab;
    int *zc_cn, *zc_ext;
    const void **zc_ptrs;
    int *st, *it;                /* per-instance status / iterations of the last solve */
    int status;                  /* worst status of the last solve */
    /* After a batched solve (evaluate_batch / a rendezvous round) every capsule of a structure class shares this bucket:
     * its device batch, its pinned staging, its seed state.  The generated batch loops call the per-capsule slots
     * solver_get / eval_forw_sens / eval_adj_sens inside `#pragma omp parallel for` (acados_solver.in.c:3292-3337): staging,
     * device pass and read-back of one capsule are one critical section (round-3 review: they raced). */
    pthread_mutex_t mu;
    int mu_live;
} gpu_bucket;

static void bucket_lock(gpu_bucket *bk)
{
    if (!bk->mu_live) { pthread_mutex_init(&bk->mu, NULL); bk->mu_live = 1; } /* own bucket: first use is single-threaded (evaluate) */
    pthread_mutex_lock(&bk->mu);
}
static void bucket_unlock(gpu_bucket *bk) { pthread_mutex_unlock(&bk->mu); }

struct ocp_qp_gpu_ipm_memory_;
typedef struct gpu_group_
{
    struct ocp_qp_gpu_ipm_memory_ *owner; /* mem[0] of the call that built the group: its terminate releases it */
    int n, nbk;
    gpu_bucket *bk;
    int *bucket_of, *pos_of;     /* per caller index */
    int *scratch, scratch_cap, scratch_nt; /* signature scratch: scratch_cap ints for each of scratch_nt host threads */
    /* The header of a group is never handed back to the allocator: when a group is released its resources go, the header
     * goes to a pool with its generation advanced.  A memory that still points at it (a capsule of an earlier batch call
     * whose owner has rebuilt or released the group) sees the generation mismatch and falls back to its own bucket
     * instead of following a dangling pointer. */
    unsigned gen;
    struct gpu_group_ *next_free;
    /* zero-copy (zc_prepare): 0 not tried yet, 1 the capsules' QP memory is registered with the device, -1 off for this group */
    int zc_state, zc_nr, zc_rereg;
    struct zc_range_ *zc_r;      /* the registered blocks, sorted by address */
    int *zc_hint;                /* per caller index: the block its last source was found in */
} gpu_group;

typedef struct ocp_qp_gpu_ipm_memory_
{
    gpu_bucket own;              /* the single-QP path: a bucket of one, everything carved */
    gpu_group *group;            /* batch entries: the group this memory's QP was solved in last (NULL: own) ... */
    unsigned group_gen;          /* ... its generation at that time (mem_group() checks it) ... */
    int g_bucket, g_pos;         /* ... and where */
    int rv_index;                /* slot of this memory's capsule in its rendezvous (-1: none yet) */
    int *sig_scratch;
    double time_qp_solver_call;
    int zero_copy;               /* extension: the last batch call's QP data was gathered by the device from the capsules' memory (zc_prepare) */
    int upload_doubles;          /* extension: doubles per QP the last batch call sent to the device (the whole input blob, or its vector part) */
    double time_unpack_in, time_pack_out; /* extension: host time spent reading qp_in into / writing qp_out from the staging blobs */
    int iter, status;
    /* per-iteration statistics of this memory's QP, HPIPM-shaped (ocp_qp_hpipm.c:255-297 "stat" / "stat_m": what
     * ocp_qp_solver_get_stats and the Python getter read): stat_m columns x (iter + 1) rows, carved, filled on demand from the device
     * table (the library keeps it for the first 64 instances of a batch) */
    double *stat;
    int stat_m, stat_rows;
} ocp_qp_gpu_ipm_memory;

/* the group a memory belongs to, if that group is still the one it was solved in */
static gpu_group *mem_group(const ocp_qp_gpu_ipm_memory *m)
{
    return m->group && m->group->gen == m->group_gen && m->group->owner ? m->group : NULL;
}

typedef struct zc_range_ { char *lo, *hi; } zc_range;

static int rendezvous_evaluate(struct ocp_qp_gpu_ipm_rendezvous_ *r, void *config, void *qp_in, void *qp_out, void *opts, ocp_qp_gpu_ipm_memory *m);
static void group_release(gpu_group *g);

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}


/* ------------------------------------------------------------------ dims / opts (ocp_qp_hpipm.c:60-183) */

static void gpu_dims_set(void *config_, void *dims_, int stage, const char *field, int *value)
{
    /* ocp_qp_common.c:147-153 -> d_ocp_qp_dim_set: the slot acados fills with ocp_qp_dims_set */
    ocp_qp_dims *d = (ocp_qp_dims *) dims_;
    int *dst = NULL;
    if (!strcmp(field, "nx")) dst = d->nx; else if (!strcmp(field, "nu")) dst = d->nu; else if (!strcmp(field, "nbx")) dst = d->nbx;
    else if (!strcmp(field, "nbu")) dst = d->nbu; else if (!strcmp(field, "ng")) dst = d->ng; else if (!strcmp(field, "ns")) dst = d->ns;
    else if (!strcmp(field, "nbxe")) dst = d->nbxe; else if (!strcmp(field, "nbue")) dst = d->nbue; else if (!strcmp(field, "nge")) dst = d->nge;
    else { printf("\nerror: ocp_qp_dims_set: field %s not available\n", field); exit(1); }
    dst[stage] = *value;
    d->nb[stage] = d->nbx[stage] + d->nbu[stage];
}

static acados_size_t gpu_opts_calculate_size(void *config, void *dims_)
{
    return size8(sizeof(ocp_qp_gpu_ipm_opts) + sizeof(int) * (size_t) (((const ocp_qp_dims *) dims_)->N + 2) + 2 * 8);
}
static void *gpu_opts_assign(void *config, void *dims_, void *raw_memory)
{
    char *c = align8((char *) raw_memory);
    ocp_qp_gpu_ipm_opts *o = (ocp_qp_gpu_ipm_opts *) c;
    o->cond_block_cap = ((const ocp_qp_dims *) dims_)->N + 2;
    o->cond_block_size = (int *) align8(c + sizeof(ocp_qp_gpu_ipm_opts));
    return o;
}

static void gpu_opts_initialize_default(void *config, void *dims, void *opts_)
{
    ocp_qp_gpu_ipm_opts *o = (ocp_qp_gpu_ipm_opts *) opts_;
    /* mode BALANCE + the acados overrides, ocp_qp_hpipm.c:101-113 */
    o->mu0 = 1e0; o->tol_stat = 1e-6; o->tol_eq = 1e-8; o->tol_ineq = 1e-8; o->tol_comp = 1e-8; o->alpha_min = 1e-8;
    o->tau_min = 0.0; o->reg_prim = 1e-15; o->t0_min = 1e-16; o->lam0_min = 1e-16;
    o->iter_max = 50; o->warm_start = 0; o->print_level = 0; o->ric_alg = 1; o->t0_init = 2; o->update_fact_exit = 0;
    o->cond_pred_corr = 1; o->tol_comp_soft_scale = 1.0;
    o->cond_N = 0; o->cond_block_size_set = 0;
    o->rendezvous = NULL;
}

static void gpu_opts_update(void *config, void *dims, void *opts) {}

static void gpu_opts_set(void *config, void *opts_, const char *field, void *value)
{
    ocp_qp_gpu_ipm_opts *o = (ocp_qp_gpu_ipm_opts *) opts_;
    const double *d = (const double *) value;
    const int *i = (const int *) value;
    if (!strcmp(field, "iter_max")) o->iter_max = *i;
    else if (!strcmp(field, "print_level")) o->print_level = *i;
    else if (!strcmp(field, "tol_stat")) o->tol_stat = *d;
    else if (!strcmp(field, "tol_eq")) o->tol_eq = *d;
    else if (!strcmp(field, "tol_ineq")) o->tol_ineq = *d;
    else if (!strcmp(field, "tol_comp")) o->tol_comp = *d;
    else if (!strcmp(field, "warm_start")) o->warm_start = *i;
    else if (!strcmp(field, "tau_min")) o->tau_min = *d;
    else if (!strcmp(field, "mu0")) { if (*d > 0.0) o->mu0 = *d; }
    else if (!strcmp(field, "t0_init")) o->t0_init = *i;
    else if (!strcmp(field, "ric_alg"))
    {
        /* the device kernels carry the Cholesky factor of P (square-root Riccati); the classical recursion for an indefinite
         * full-space Hessian is not available: refused, not silently replaced */
        if (*i != 1) { printf("\nerror: ocp_qp_gpu_ipm_opts_set: ric_alg = %d not available (only ric_alg = 1)\n", *i); exit(1); }
        o->ric_alg = *i;
    }
    else if (!strcmp(field, "tol_comp_soft_scale")) o->tol_comp_soft_scale = *d;
    else if (!strcmp(field, "cond_N")) { if (*i != o->cond_N) { o->cond_N = *i; o->cond_block_size_set = 0; } }
    else if (!strcmp(field, "cond_block_size"))
    {
        /* cond_N + 1 entries, cond_N first (ocp_qp_partial_condensing.c:305-313) */
        if (o->cond_N <= 0 || o->cond_N + 1 > o->cond_block_cap) { printf("\nerror: ocp_qp_gpu_ipm_opts_set: cond_block_size needs cond_N (1..N-1) first\n"); exit(1); }
        memcpy(o->cond_block_size, i, sizeof(int) * (size_t) (o->cond_N + 1));
        o->cond_block_size_set = 1;
    }
    else if (!strcmp(field, "t0_min")) o->t0_min = *d;
    else if (!strcmp(field, "lam0_min")) o->lam0_min = *d;
    else if (!strcmp(field, "update_fact_exit")) o->update_fact_exit = *i;
    else if (!strcmp(field, "hpipm_mode"))
    {
        /* the acados overrides hold for every mode (ocp_qp_hpipm.c:146-165); of what the modes change inside HPIPM this
         * backend has the conditional corrector: SPEED_ABS switches it off, the other modes are one arithmetic */
        o->cond_pred_corr = strcmp((const char *) value, "SPEED_ABS") ? 1 : 0;
    }
    /* `value` IS the rendezvous (or NULL): reachable from a capsule as ocp_nlp_solver_opts_set(.., "qp_rendezvous", r) */
    else if (!strcmp(field, "rendezvous")) o->rendezvous = (struct ocp_qp_gpu_ipm_rendezvous_ *) value;
    else { printf("\nerror: ocp_qp_gpu_ipm_opts_set: wrong field: %s\n", field); exit(1); }
}

static void gpu_opts_get(void *config, void *opts_, const char *field, void *value)
{
    ocp_qp_gpu_ipm_opts *o = (ocp_qp_gpu_ipm_opts *) opts_;
    if (!strcmp(field, "t0_min")) *(double *) value = o->t0_min;
    else if (!strcmp(field, "lam0_min")) *(double *) value = o->lam0_min;
    else { printf("\nerror: ocp_qp_gpu_ipm_opts_get: field %s not available\n", field); exit(1); }
}

/* ------------------------------------------------------------------ memory */

#define GPU_IPM_STAT_M 20 /* columns of the statistics table, as HPIPM's */
static int stat_rows_for(const void *opts_)
{
    const ocp_qp_gpu_ipm_opts *o = (const ocp_qp_gpu_ipm_opts *) opts_;
    return (o && o->iter_max > 50 ? o->iter_max : 50) + 2;
}

static acados_size_t gpu_memory_calculate_size(void *config, void *dims_, void *opts)
{
    const ocp_qp_dims *d = (const ocp_qp_dims *) dims_;
    const size_t nst = (size_t) d->N + 1;
    return size8(sizeof(ocp_qp_gpu_ipm_memory) + 2 * sizeof(int) * (size_t) sig_len(d)
                 + sizeof(double) * (size_t) (blob_in_cap(d) + blob_out_cap(d))
                 + sizeof(gpu_seg) * nst * (SEGS_IN_PER_STAGE + SEGS_OUT_PER_STAGE + SEGS_SEED_PER_STAGE) + 2 * sizeof(int)
                 + sizeof(double) * GPU_IPM_STAT_M * (size_t) stat_rows_for(opts) + 9 * 8);
}

static void *gpu_memory_assign(void *config, void *dims_, void *opts, void *raw_memory)
{
    const ocp_qp_dims *d = (const ocp_qp_dims *) dims_;
    const int nst = d->N + 1;
    char *c = align8((char *) raw_memory);
    ocp_qp_gpu_ipm_memory *m = (ocp_qp_gpu_ipm_memory *) c;
    memset(m, 0, sizeof(*m));
    c = align8(c + sizeof(*m));
    gpu_bucket *bk = &m->own;
    bk->n = 1;
    bk->cap_in = (size_t) blob_in_cap(d); bk->cap_out = (size_t) blob_out_cap(d);
    bk->blob_in = (double *) c; c += sizeof(double) * bk->cap_in;
    bk->blob_out = (double *) c; c += sizeof(double) * bk->cap_out;
    bk->seg_cap_in = nst * SEGS_IN_PER_STAGE; bk->seg_cap_out = nst * SEGS_OUT_PER_STAGE; bk->seg_cap_seed = nst * SEGS_SEED_PER_STAGE;
    bk->seg_in = (gpu_seg *) c; c += sizeof(gpu_seg) * (size_t) bk->seg_cap_in;
    bk->seg_out = (gpu_seg *) c; c += sizeof(gpu_seg) * (size_t) bk->seg_cap_out;
    bk->seg_seed = (gpu_seg *) c; c += sizeof(gpu_seg) * (size_t) bk->seg_cap_seed;
    bk->sig_cap = sig_len(d);
    bk->sig = (int *) c; c += sizeof(int) * (size_t) bk->sig_cap;
    m->sig_scratch = (int *) c; c += sizeof(int) * (size_t) bk->sig_cap; /* its own scratch: a signature can be longer than a staging blob */
    bk->st = (int *) c; c += sizeof(int);
    bk->it = (int *) c; c += sizeof(int);
    c = align8(c);
    m->stat_m = GPU_IPM_STAT_M;
    m->stat_rows = stat_rows_for(opts);
    m->stat = (double *) c; c += sizeof(double) * GPU_IPM_STAT_M * (size_t) m->stat_rows;
    memset(m->stat, 0, sizeof(double) * GPU_IPM_STAT_M * (size_t) m->stat_rows);
    m->rv_index = -1;
    return m;
}

static void gpu_memory_get(void *config, void *mem_, const char *field, void *value)
{
    ocp_qp_gpu_ipm_memory *m = (ocp_qp_gpu_ipm_memory *) mem_;
    if (!strcmp(field, "time_qp_solver_call")) *(double *) value = m->time_qp_solver_call;
    else if (!strcmp(field, "upload_doubles")) *(int *) value = m->upload_doubles;
    else if (!strcmp(field, "zero_copy")) *(int *) value = m->zero_copy;
    else if (!strcmp(field, "time_unpack_in")) *(double *) value = m->time_unpack_in;
    else if (!strcmp(field, "time_pack_out")) *(double *) value = m->time_pack_out;
    else if (!strcmp(field, "iter")) *(int *) value = m->iter;
    else if (!strcmp(field, "status")) *(int *) value = m->status;
    else if (!strcmp(field, "kernel_name")) /* extension: which kernel family serves this memory's QP (const char *) */
    {
        const gpu_group *gg = mem_group(m);
        const gpu_bucket *bk = gg ? gg->bk + m->g_bucket : &m->own;
        *(const char **) value = bk->batch ? ocp_qp_gpu_batch_kernel_name(bk->batch) : "";
    }
    else if (!strcmp(field, "stat"))
    {
        /* ocp_qp_hpipm.c:262-266: pointer to the table of the last solve (rows 0..iter) */
        const gpu_group *gg = mem_group(m);
        const gpu_bucket *bk = gg ? gg->bk + m->g_bucket : &m->own;
        memset(m->stat, 0, sizeof(double) * (size_t) m->stat_m * (size_t) m->stat_rows);
        if (bk->batch) ocp_qp_gpu_batch_get_stat(bk->batch, gg ? m->g_pos : 0, m->stat, m->stat_rows);
        *(double **) value = m->stat;
    }
    else if (!strcmp(field, "stat_m")) *(int *) value = m->stat_m;
    else if (!strcmp(field, "tau_iter")) *(double *) value = 0.0; /* (the barrier parameter the solver ended on is not tracked per capsule) */
    else if (!strcmp(field, "cond_N_active")) /* extension: stages of the QP the device IPM ran on in the last solve (int; N: not condensed) */
    {
        const gpu_group *gg = mem_group(m);
        const gpu_bucket *bk = gg ? gg->bk + m->g_bucket : &m->own;
        *(int *) value = bk->batch ? (int) ocp_qp_gpu_batch_get_scalar(bk->batch, "cond_N_active") : -1;
    }
    else { printf("\nerror: ocp_qp_gpu_ipm_memory_get: field %s not available\n", field); exit(1); }
}

static acados_size_t gpu_workspace_calculate_size(void *config, void *dims, void *opts) { return 0; }

/* (re)create the device batch of a bucket for the structure of `in` (n instances); -1: the device failed while the batch's
 * structure was built (the batch is gone, the next call tries again) */
static int bucket_build(gpu_bucket *bk, const ocp_qp_in *in, const int *sig, int len)
{
    const ocp_qp_dims *d = in->dim;
    const int N = d->N;
    if (bk->batch) ocp_qp_gpu_batch_destroy(bk->batch);
    bk->batch = ocp_qp_gpu_batch_create(N, d->nx, d->nu, d->nbx, d->nbu, d->ng, d->ns, bk->n, -1);
    if (!bk->batch) { printf("\nerror: ocp_qp_gpu_ipm: no GPU batch could be created (no device or unsupported shape)\n"); exit(1); }
    memcpy(bk->sig, sig, sizeof(int) * len);
    bk->sig_len = len;
    for (int k = 0; k <= N; k++)
    {
        ocp_qp_gpu_batch_set_int(bk->batch, "idxb", k, in->idxb[k], d->nb[k]);
        ocp_qp_gpu_batch_set_int(bk->batch, "idxs_rev", k, in->idxs_rev[k], d->nb[k] + d->ng[k]);
        ocp_qp_gpu_batch_set_int(bk->batch, "idxe", k, in->idxe[k] + d->nbue[k], d->nbxe[k]); /* [bue | bxe | ge]: the bxe part */
    }
    if (gpu_layout_build(&bk->lay, d) != 0)
    {
        ocp_qp_gpu_batch_destroy(bk->batch);
        bk->batch = NULL;
        bk->sig_len = 0;
        return -1;
    }
    return 0;
}

/* the next capsule's QP towards the cache while this one is unpacked: a batch call reads n x 85 KB that n linearisations wrote
 * long ago (cold), block by block -- one touch per 64-byte line of the big members, enough lines in flight to hide DRAM latency */
static void prefetch_qp_in(const ocp_qp_in *in)
{
    const ocp_qp_dims *d = in->dim;
    for (int k = 0; k <= d->N; k++)
    {
        const struct blasfeo_dmat *ms[3] = {in->BAbt + k, in->RSQrq + k, in->DCt + k};
        for (int q = 0; q < 3; q++)
        {
            const char *p = (const char *) ms[q]->pA;
            const int bytes = ms[q]->memsize;
            for (int o = 0; o < bytes; o += 64) __builtin_prefetch(p + o, 0, 1);
        }
        __builtin_prefetch(in->rqz[k].pa, 0, 1); __builtin_prefetch(in->d[k].pa, 0, 1); __builtin_prefetch(in->b[k].pa, 0, 1);
        __builtin_prefetch(in->d_mask[k].pa, 0, 1);
    }
}

static void prefetch_qp_vec(const ocp_qp_in *in)
{
    const ocp_qp_dims *d = in->dim;
    for (int k = 0; k <= d->N; k++)
    {
        const struct blasfeo_dvec *vs[4] = {in->rqz + k, in->d + k, in->d_mask + k, in->b + k};
        for (int q = 0; q < 4; q++)
            for (int o = 0; o < vs[q]->m * (int) sizeof(double); o += 64) __builtin_prefetch((const char *) vs[q]->pa + o, 0, 1);
    }
}

static void apply_opts(ocp_qp_gpu_batch *b, const ocp_qp_gpu_ipm_opts *o, int ws)
{
    ocp_qp_gpu_batch_opts_set(b, "iter_max", &o->iter_max);
    ocp_qp_gpu_batch_opts_set(b, "tol_stat", &o->tol_stat);
    ocp_qp_gpu_batch_opts_set(b, "tol_eq", &o->tol_eq);
    ocp_qp_gpu_batch_opts_set(b, "tol_ineq", &o->tol_ineq);
    ocp_qp_gpu_batch_opts_set(b, "tol_comp", &o->tol_comp);
    ocp_qp_gpu_batch_opts_set(b, "mu0", &o->mu0);
    ocp_qp_gpu_batch_opts_set(b, "t0_init", &o->t0_init);
    ocp_qp_gpu_batch_opts_set(b, "tau_min", &o->tau_min);
    ocp_qp_gpu_batch_opts_set(b, "tol_comp_soft_scale", &o->tol_comp_soft_scale);
    ocp_qp_gpu_batch_opts_set(b, "cond_pred_corr", &o->cond_pred_corr);
    /* partial condensing inside the device solve (0 = off; the library keeps its condensed batch while cond_N is unchanged) */
    ocp_qp_gpu_batch_opts_set(b, "cond_N", &o->cond_N);
    if (o->cond_N > 0 && o->cond_block_size_set) ocp_qp_gpu_batch_opts_set(b, "cond_block_size", o->cond_block_size);
    ocp_qp_gpu_batch_opts_set(b, "t0_min", &o->t0_min);
    ocp_qp_gpu_batch_opts_set(b, "lam0_min", &o->lam0_min);
    ocp_qp_gpu_batch_opts_set(b, "print_level", &o->print_level);
    ocp_qp_gpu_batch_opts_set(b, "warm_start", &ws);
}

/* the device part of a solve: staged blobs -> device batch -> staged solution, statuses.  `staged` 1: the input blob is on the device
 * already (handed over in chunks while it was being filled, evaluate_batch_masked): only its scatter launch is left; 2: there is no host
 * blob -- the device gathers the QP data from the capsules' own memory (zc_prepare filled bk->zc_ptrs) */
enum { BATCH_SOLVE = 0, BATCH_LHS = 1, BATCH_RHS_SOLVE = 2 }; /* evaluate | RTI preparation (condense_lhs) | RTI feedback (condense_rhs_and_solve) */

static void bucket_solve(gpu_bucket *bk, const ocp_qp_gpu_ipm_opts *o, int ws, int staged, int mode)
{
    ocp_qp_gpu_batch *b = bk->batch;
    if (!b) /* the device failed while this bucket's batch was built (bucket_build) */
    {
        for (int i = 0; i < bk->n; i++) { bk->st[i] = ACADOS_QP_FAILURE; bk->it[i] = 0; }
        bk->status = ACADOS_QP_FAILURE;
        return;
    }
    apply_opts(b, o, ws);
    /* the starting point goes in before the pack, which then restores the equality-flagged values (x0) */
    /* a negative return = the device failed (HIP error, reported by the library): every QP of the bucket comes back as
     * ACADOS_QP_FAILURE -- ocp_nlp ends that capsule's solve cleanly (ocp_nlp_sqp.c:720-751) -- and the process, with the
     * other host threads of an MPC fleet in it, lives on */
    if (mode == BATCH_LHS)
    {
        /* RTI preparation of the whole class: matrices (and everything else) to the device, the matrix part of the condensing there */
        const int bad = (staged == 2 ? ocp_qp_gpu_batch_gather_run(b, 0, bk->zc_ptrs)
                                     : (staged ? ocp_qp_gpu_batch_set_bulk_staged(b) : ocp_qp_gpu_batch_set_bulk(b, bk->blob_in, 0))) != 0
                        || ocp_qp_gpu_batch_condense_lhs(b) != 0;
        for (int i = 0; i < bk->n; i++) { bk->st[i] = bad ? ACADOS_QP_FAILURE : ACADOS_SUCCESS; bk->it[i] = 0; }
        bk->status = bad ? ACADOS_QP_FAILURE : ACADOS_SUCCESS;
        bk->lhs_resident = !bad;
        return;
    }
    const int vec_only = mode == BATCH_RHS_SOLVE && bk->lhs_resident;
    bk->lhs_resident = 0;
    if ((ws >= 2 && ocp_qp_gpu_batch_set_bulk_out(b, bk->blob_out, 0) != 0)
        || (staged == 2 ? ocp_qp_gpu_batch_gather_run(b, vec_only ? 2 : 0, bk->zc_ptrs)
                        : (vec_only ? ocp_qp_gpu_batch_set_bulk_vec(b, bk->blob_in, 0)
                                    : (staged ? ocp_qp_gpu_batch_set_bulk_staged(b) : ocp_qp_gpu_batch_set_bulk(b, bk->blob_in, 0)))) != 0
        || (vec_only ? ocp_qp_gpu_batch_condense_rhs_and_solve(b) : ocp_qp_gpu_batch_solve(b)) < 0 || ocp_qp_gpu_batch_get_bulk(b, bk->blob_out, 0) != 0
        || ocp_qp_gpu_batch_get_info(b, "status", bk->st) != 0 || ocp_qp_gpu_batch_get_info(b, "iter", bk->it) != 0)
    {
        for (int i = 0; i < bk->n; i++) { bk->st[i] = ACADOS_QP_FAILURE; bk->it[i] = 0; }
        bk->status = ACADOS_QP_FAILURE;
        return;
    }
    int worst = 0;
    for (int i = 0; i < bk->n; i++)
        if (bk->st[i] != ACADOS_SUCCESS && (worst == 0 || worst == ACADOS_MAXITER)) worst = bk->st[i];
    bk->status = worst;
}

/* ------------------------------------------------------------------ evaluate (ocp_qp_hpipm.c:314-405) */

static int ocp_qp_gpu_ipm_acados(void *config, void *qp_in_, void *qp_out_, void *opts_, void *mem_, void *work)
{
    const double t_start = now_s();
    ocp_qp_in *in = (ocp_qp_in *) qp_in_;
    ocp_qp_out *out = (ocp_qp_out *) qp_out_;
    ocp_qp_gpu_ipm_opts *o = (ocp_qp_gpu_ipm_opts *) opts_;
    ocp_qp_gpu_ipm_memory *m = (ocp_qp_gpu_ipm_memory *) mem_;
    if (o->rendezvous) return rendezvous_evaluate(o->rendezvous, config, qp_in_, qp_out_, opts_, m);
    gpu_bucket *bk = &m->own;
    qp_info *info = (qp_info *) out->misc;

    /* device batch: (re)created when the structure changes; the signature is compared in carved memory */
    {
        if (sig_len(in->dim) > bk->sig_cap) { printf("\nerror: ocp_qp_gpu_ipm: dims of qp_in grew after memory_assign\n"); exit(1); }
        const int len = fill_sig(in, m->sig_scratch);
        if (!bk->batch || bk->sig_len != len || memcmp(bk->sig, m->sig_scratch, sizeof(int) * len) != 0)
        {
            if (bucket_build(bk, in, m->sig_scratch, len) != 0)
            {
                info->num_iter = 0; info->t_computed = 0;
                m->status = ACADOS_QP_FAILURE; m->iter = 0;
                return ACADOS_QP_FAILURE;
            }
            if ((size_t) bk->L_in > bk->cap_in || (size_t) bk->L_out > bk->cap_out || (size_t) bk->L_seed > bk->cap_in)
            {
                printf("\nerror: ocp_qp_gpu_ipm: bulk blob larger than the carved staging\n");
                exit(1);
            }
        }
    }
    /* this memory's QP lives in its own batch from now on; a group this memory OWNS (it was mem[0] of a batch call) goes with
     * it -- nobody else could release it later (round-3 review); the other members see the generation change (mem_group) */
    if (mem_group(m) && m->group->owner == m) group_release(m->group);
    m->group = NULL;

    const int ws = o->warm_start >= 2 ? o->warm_start : 0; /* 1 = 0, acados_ocp_options.py:1029-1031 */
    memset(bk->blob_in, 0, sizeof(double) * (size_t) bk->L_in);
    unpack_qp_in(&bk->lay, in, bk->blob_in);
    if (ws >= 2) unpack_qp_out_duals(&bk->lay, out, bk->blob_out);
    const double t_packed = now_s();

    bucket_solve(bk, o, ws, 0, BATCH_SOLVE);
    const double t_solved = now_s();

    pack_qp_out(&bk->lay, bk->blob_out, out);
    const double t_end = now_s();
    if (info)
    {
        info->solve_QP_time = ocp_qp_gpu_batch_get_scalar(bk->batch, "time_tot");
        info->condensing_time = 0.0;
        info->interface_time = (t_packed - t_start) + (t_end - t_solved);
        info->total_time = t_end - t_start;
        info->num_iter = bk->it[0];
        info->t_computed = 1; /* t comes from the device (ocp_qp_compute_t restated there for the rows the IPM skipped) */
    }
    m->iter = bk->it[0]; m->status = bk->st[0]; m->time_qp_solver_call = t_solved - t_packed;
    return bk->st[0]; /* already return_values_t (acados/utils/types.h:74-87) */
}

/* ------------------------------------------------------------------ the batch entries */

static void bucket_release(gpu_bucket *bk)
{
    if (bk->batch) ocp_qp_gpu_batch_destroy(bk->batch);
    ocp_qp_gpu_host_free(bk->blob_in); ocp_qp_gpu_host_free(bk->blob_out); ocp_qp_gpu_host_free((void *) bk->zc_ptrs);
    free(bk->zc_cn); free(bk->zc_ext);
    free(bk->sig); free(bk->seg_in); free(bk->seg_out); free(bk->seg_seed); free(bk->seg_vec); free(bk->members); free(bk->st); free(bk->it);
    if (bk->mu_live) pthread_mutex_destroy(&bk->mu);
    memset(bk, 0, sizeof(*bk));
}

static pthread_mutex_t g_group_pool_mu = PTHREAD_MUTEX_INITIALIZER;
static gpu_group *g_group_pool = NULL; /* released headers (see gpu_group::gen); the only state of this file outside caller-owned objects */

static void zc_unregister(gpu_group *g)
{
    for (int r = 0; r < g->zc_nr; r++) (void) ocp_qp_gpu_host_unregister(g->zc_r[r].lo);
    free(g->zc_r); g->zc_r = NULL; g->zc_nr = 0;
    if (g->zc_state > 0) g->zc_state = 0;
}

static void group_release(gpu_group *g)
{
    if (!g) return;
    zc_unregister(g);
    free(g->zc_hint);
    for (int q = 0; q < g->nbk; q++) bucket_release(g->bk + q);
    free(g->bk); free(g->bucket_of); free(g->pos_of); free(g->scratch);
    const unsigned gen = g->gen + 1;
    memset(g, 0, sizeof(*g)); /* owner = NULL: not a live group */
    g->gen = gen;
    pthread_mutex_lock(&g_group_pool_mu);
    g->next_free = g_group_pool; g_group_pool = g;
    pthread_mutex_unlock(&g_group_pool_mu);
}

static void *xcalloc(size_t cnt, size_t sz)
{
    void *p = calloc(cnt ? cnt : 1, sz);
    if (!p) { printf("\nerror: ocp_qp_gpu_ipm: out of host memory\n"); exit(1); }
    return p;
}


/* ------------------------------------------------------------------ zero-copy: the device reads the capsules' own QP memory
 *
 * Reading n capsules' qp_in into a pinned blob is a pass over n x 85 KB (C2-shaped QPs) of cold host memory: bound by the host's memory
 * system at 26-29 GB/s on 16 threads however the loops are written (profiles/r06_orch_threads.txt), and the host->device copy comes on
 * top.  A kernel reads host memory that is REGISTERED with the device at the PCIe rate (56 GB/s, profiles/r06_zero_copy_probe.txt).  So
 * the batch entries (all capsules present, probed panel-major storage) register the blocks the member arrays of the n qp_in live in --
 * once per group: ~20 us per block -- and hand the device, per call, only the arrays' addresses; the device library gathers the words into
 * its blob (class-wide word tables: gpu_words_build) and goes on exactly as after _set_bulk.  The memory stays the capsule's: nothing is
 * copied or moved on the host, every call re-reads the addresses from the structs and checks them against the registered blocks (a
 * capsule whose qp_in moved: registered again; more than three times: the group goes back to the blob path).  Any refusal -- registration
 * fails (pages already registered by somebody else, address space holes), a matrix with another cn than the class's -- is the blob path,
 * whose results are the same byte for byte (tests/test_mock_acados.py).  ACADOS_AMD_ZERO_COPY=0 switches it off, =1 uses it at every size
 * (by default a call whose whole QP data exceeds 160 MB keeps the chunked blob where more than 12 host threads fill it: zc_prepare).
 * The blocks are unregistered when the group goes (terminate / memory_reset of the capsule that led the call, a new set of capsules):
 * destroy capsule 0 of a batch first, as the generated `_free` loop does, or the others' memory is freed while still registered.
 */
#if defined(GPU_ZERO_COPY)
#define ZC_PAGE ((size_t) 4096)

/* ACADOS_AMD_ZERO_COPY: 0 never, 1 always, unset: by size (zc_prepare) */
static int zc_enabled(void)
{
    const char *e = getenv("ACADOS_AMD_ZERO_COPY");
    return e && e[0] == '0' ? 0 : (e && e[0] == '1' ? 2 : 1);
}

static int zc_range_cmp(const void *a_, const void *b_)
{
    const zc_range *a = (const zc_range *) a_, *b = (const zc_range *) b_;
    return a->lo < b->lo ? -1 : (a->lo > b->lo);
}

/* sorted ranges -> merged in place (ranges closer than `gap` bytes become one); returns the new count */
static int zc_merge(zc_range *r, int cnt, size_t gap)
{
    if (cnt == 0) return 0;
    qsort(r, cnt, sizeof(zc_range), zc_range_cmp);
    int o = 0;
    for (int i = 1; i < cnt; i++)
    {
        if (r[i].lo <= r[o].hi + gap) { if (r[i].hi > r[o].hi) r[o].hi = r[i].hi; }
        else r[++o] = r[i];
    }
    return o + 1;
}

/* the class's word tables -> device, once per bucket: 1 done, -1 refused */
static int zc_tables(gpu_bucket *bk, const ocp_qp_in *in0)
{
    const int N = in0->dim->N, P = GPU_WORD_MEMBERS * (N + 1);
    if (!bk->batch || bk->ps <= 0 || bk->L_in <= 0) return -1;
    bk->zc_P = P;
    bk->zc_cn = (int *) xcalloc(3 * (N + 1), sizeof(int));
    bk->zc_ext = (int *) xcalloc(2 * (size_t) P, sizeof(int)); /* [the whole blob's | its vector part's] */
    gpu_word_cn(in0, N, bk->zc_cn);
    for (int which = 0; which < 2; which++) /* the whole input blob; its vector part (RTI feedback) */
    {
        const gpu_seg *tab = which ? bk->seg_vec : bk->seg_in;
        const int cnt = which ? bk->n_vec : bk->n_in;
        if (which && bk->L_vec <= 0) break;
        gpu_word *w = NULL;
        const int nw = gpu_words_build(tab, cnt, N, bk->ps, bk->zc_cn, &w);
        if (!w) return -1;
        int *col = (int *) xcalloc(3 * (size_t) nw, sizeof(int));
        unsigned char *neg = (unsigned char *) xcalloc(nw, 1);
        for (int q = 0; q < nw; q++)
        {
            col[q] = w[q].slot; col[nw + q] = w[q].off; col[2 * nw + q] = w[q].pos; neg[q] = w[q].neg;
            int *ext = bk->zc_ext + (which ? P : 0);
            if (w[q].off + 1 > ext[w[q].slot]) ext[w[q].slot] = w[q].off + 1;
        }
        const int rc = ocp_qp_gpu_batch_gather_tables(bk->batch, which ? 2 : 0, P, nw, col, col + nw, col + 2 * nw, neg);
        free(w); free(col); free(neg);
        if (rc != 0) return -1;
    }
    bk->zc_ptrs = (const void **) ocp_qp_gpu_host_alloc(sizeof(void *) * (size_t) bk->n * (size_t) P);
    return bk->zc_ptrs ? 1 : -1;
}

/* register the pages under every source array of every capsule: 0 / -1 */
static int zc_register(gpu_group *g, int n, ocp_qp_in **ins, size_t gap)
{
    int cap = 0;
    for (int q = 0; q < g->nbk; q++) cap += g->bk[q].n * 8; /* (a capsule's arrays are carved from one block or two: grows if not) */
    zc_range *all = (zc_range *) xcalloc(cap, sizeof(zc_range));
    int cnt = 0, Pmax = 0;
    for (int q = 0; q < g->nbk; q++) if (g->bk[q].zc_P > Pmax) Pmax = g->bk[q].zc_P;
    zc_range *one = (zc_range *) xcalloc(Pmax, sizeof(zc_range));
    const void **src = (const void **) xcalloc(Pmax, sizeof(void *));
    for (int i = 0; i < n; i++)
    {
        const gpu_bucket *bk = g->bk + g->bucket_of[i];
        int m = 0;
        gpu_word_sources(ins[i], ins[i]->dim->N, src, 0);
        for (int s = 0; s < bk->zc_P; s++)
        {
            if (bk->zc_ext[s] == 0) continue;
            if (!src[s]) { free(all); free(one); free(src); return -1; }
            one[m].lo = (char *) ((size_t) src[s] & ~(ZC_PAGE - 1));
            one[m].hi = (char *) (((size_t) src[s] + sizeof(double) * (size_t) bk->zc_ext[s] + ZC_PAGE - 1) & ~(ZC_PAGE - 1));
            m++;
        }
        m = zc_merge(one, m, gap);
        if (cnt + m > cap)
        {
            cap = 2 * (cnt + m);
            all = (zc_range *) realloc(all, sizeof(zc_range) * cap);
            if (!all) { printf("\nerror: ocp_qp_gpu_ipm: out of host memory\n"); exit(1); }
        }
        memcpy(all + cnt, one, sizeof(zc_range) * m);
        cnt += m;
    }
    free(one); free(src);
    cnt = zc_merge(all, cnt, 0); /* capsules sharing a page share a block */
    for (int r = 0; r < cnt; r++)
        if (ocp_qp_gpu_host_register(all[r].lo, (size_t) (all[r].hi - all[r].lo)) != 0)
        {
            for (int u = 0; u < r; u++) (void) ocp_qp_gpu_host_unregister(all[u].lo);
            free(all);
            return -1;
        }
    g->zc_r = all; g->zc_nr = cnt;
    return 0;
}

/* block holding [p, p + bytes): index, -1 none */
static int zc_find(const gpu_group *g, const char *p, size_t bytes, int hint)
{
    if (hint >= 0 && hint < g->zc_nr && p >= g->zc_r[hint].lo && p + bytes <= g->zc_r[hint].hi) return hint;
    int lo = 0, hi = g->zc_nr - 1;
    while (lo <= hi)
    {
        const int mid = (lo + hi) / 2;
        if (p < g->zc_r[mid].lo) hi = mid - 1;
        else if (p >= g->zc_r[mid].hi) lo = mid + 1;
        else return p + bytes <= g->zc_r[mid].hi ? mid : -1;
    }
    return -1;
}

/* the structs the next capsule's addresses are read from, towards the cache: 168 BLASFEO structs per C2-shaped capsule, cold, one
 * dependent miss each otherwise */
static void zc_prefetch(const ocp_qp_in *in, int N, int vec)
{
    const char *a[8] = {(const char *) in->b, (const char *) in->rqz, (const char *) in->d, (const char *) in->d_mask, (const char *) in->Z,
                        (const char *) in->BAbt, (const char *) in->RSQrq, (const char *) in->DCt};
    const int bytes_v = (N + 1) * (int) sizeof(struct blasfeo_dvec), bytes_m = (N + 1) * (int) sizeof(struct blasfeo_dmat);
    for (int q = 0; q < (vec ? 5 : 8); q++)
        for (int o = 0; o < (q < 5 ? bytes_v : bytes_m); o += 64) __builtin_prefetch(a[q] + o, 0, 1);
}

/* this call's source addresses into the buckets' pinned tables (`vec`: of the vector members only); 1: all of them inside registered
 * blocks (`check`) and every matrix with the cn the word tables were built for, 0: not */
static int zc_fill(gpu_group *g, int n, ocp_qp_in **ins, int check, int vec)
{
    int ok = 1;
#pragma omp parallel for schedule(static) reduction(&& : ok)
    for (int i = 0; i < n; i++)
    {
        const gpu_bucket *bk = g->bk + g->bucket_of[i];
        const int N = ins[i]->dim->N, P = bk->zc_P;
        const int *ext = vec ? bk->zc_ext + P : bk->zc_ext;
        const void **row = bk->zc_ptrs + (size_t) g->pos_of[i] * (size_t) P;
        if (i + 2 < n) { __builtin_prefetch(ins[i + 2], 0, 1); __builtin_prefetch((const char *) ins[i + 2] + 64, 0, 1); }
        if (i + 1 < n) zc_prefetch(ins[i + 1], N, vec);
        gpu_word_sources(ins[i], N, row, vec);
        for (int k = 0; k <= N && !vec; k++)
            ok = ok && (k == N || ins[i]->BAbt[k].cn == bk->zc_cn[SRC_BAbt * (N + 1) + k]) && ins[i]->RSQrq[k].cn == bk->zc_cn[SRC_RSQrq * (N + 1) + k]
                 && ins[i]->DCt[k].cn == bk->zc_cn[SRC_DCt * (N + 1) + k];
        int hint = g->zc_hint[i];
        for (int s = 0; s < P; s++)
        {
            if (ext[s] == 0) { row[s] = NULL; continue; }
            if (!check) continue;
            hint = row[s] ? zc_find(g, (const char *) row[s], sizeof(double) * (size_t) ext[s], hint) : -1;
            if (hint < 0) { ok = 0; break; }
        }
        g->zc_hint[i] = hint;
    }
    return ok;
}

/* 1: every bucket's zc_ptrs is filled and the device can read what they point at */
static int zc_prepare(gpu_group *g, int n, ocp_qp_in **ins, int vec)
{
    const int mode = zc_enabled();
    if (g->zc_state < 0 || !mode) return 0;
    if (!vec && mode == 1)
    {
        /* The whole QP data of a LARGE call on a host with MANY threads stays on the blob path: both paths are bound by PCIe there, and the
         * gather moves whole 64-byte lines of 168 small arrays per capsule (1.17x the bytes of the packed blob) without the overlap the
         * chunked blob has.  Measured on C3-shaped capsules: 1,024 of them (87 MB) 6.0 ms either way; 4,096 (348 MB) on 16 host threads
         * 15.7-17.4 ms gathered against 14.6-16.6 ms through the blob (profiles/r06_zero_copy_latency.txt) -- but the gather does not
         * care how many host threads there are (16.0-18.7 ms from 4 to 32 threads) while the blob path needs them all (31.9 ms on 4
         * threads, 20.1 on 8, 15.4 on 32: profiles/r06_orch_threads_final.txt).  The vector part (an RTI feedback step) is always gathered. */
        size_t bytes = 0;
        for (int q = 0; q < g->nbk; q++) bytes += sizeof(double) * (size_t) g->bk[q].n * (size_t) g->bk[q].L_in;
#if defined(_OPENMP)
        const int host_threads = omp_get_max_threads();
#else
        const int host_threads = 1;
#endif
        if (bytes > ((size_t) 160 << 20) && host_threads > 12) return 0;
    }
    for (int q = 0; q < g->nbk; q++)
    {
        gpu_bucket *bk = g->bk + q;
        if (bk->zc_tab == 0) bk->zc_tab = zc_tables(bk, ins[bk->members[0]]);
        if (bk->zc_tab < 0) { g->zc_state = -1; return 0; }
    }
    if (!g->zc_hint) g->zc_hint = (int *) xcalloc(n, sizeof(int));
    for (int attempt = 0; attempt < 2; attempt++)
    {
        if (g->zc_state == 0)
        {
            /* one block per capsule where its arrays sit within 64 KB of each other (they are carved from one allocation); a hole in
             * between that is not mapped: block by array */
            if (!zc_fill(g, n, ins, 0, 0) || (zc_register(g, n, ins, 16 * ZC_PAGE) != 0 && zc_register(g, n, ins, 0) != 0)) { g->zc_state = -1; return 0; }
            g->zc_state = 1;
        }
        if (zc_fill(g, n, ins, 1, vec)) return 1;
        /* a qp_in is somewhere else than last time */
        zc_unregister(g);
        if (++g->zc_rereg > 3) break;
    }
    zc_unregister(g);
    g->zc_state = -1;
    return 0;
}
#else
static int zc_prepare(gpu_group *g, int n, ocp_qp_in **ins, int vec) { return 0; }
#endif

/* group of the n QPs of this call: reused as long as n and every QP's structure are what they were, else rebuilt --
 * QPs are bucketed by structure signature, one device batch per bucket */
static gpu_group *group_for(int n, ocp_qp_in **ins, ocp_qp_gpu_ipm_memory **mems, const unsigned char *skip, int *fresh)
{
    ocp_qp_gpu_ipm_memory *m0 = mems[0];
    gpu_group *g = mem_group(m0) && m0->group->owner == m0 ? m0->group : NULL;
    int need = 0, first = -1;
    for (int i = 0; i < n && first < 0; i++) if (!(skip && skip[i])) first = i;
    *fresh = 0;
    if (first < 0) return g; /* nobody takes part */
    if (g && g->n == n)
    {
        /* every capsule's structure re-read and compared on every call (dims, idxb, idxs_rev, idxe: ~300 scattered cache lines per
         * C2-shaped capsule, cold) -- on the host threads: serial it was 3 of the 4.7 ms a 4,096-capsule call spent before the device */
        int same = 1;
#if defined(_OPENMP)
#pragma omp parallel for schedule(static) reduction(&& : same) num_threads(g->scratch_nt)
#endif
        for (int i = 0; i < n; i++)
        {
            if (skip && skip[i]) continue;
            const gpu_bucket *bk = g->bk + g->bucket_of[i];
#if defined(_OPENMP)
            int *sc = g->scratch + (size_t) omp_get_thread_num() * (size_t) g->scratch_cap;
#else
            int *sc = g->scratch;
#endif
            const int l = sig_len(ins[i]->dim);
            same = same && l == bk->sig_len && l <= g->scratch_cap && fill_sig(ins[i], sc) == l && memcmp(bk->sig, sc, sizeof(int) * l) == 0;
        }
        if (same) return g;
    }
    for (int i = 0; i < n; i++)
    {
        if (skip && skip[i]) continue;
        const int l = sig_len(ins[i]->dim);
        if (l > need) need = l;
    }
    group_release(g);
    *fresh = 1;
    pthread_mutex_lock(&g_group_pool_mu);
    g = g_group_pool;
    if (g) g_group_pool = g->next_free;
    pthread_mutex_unlock(&g_group_pool_mu);
    if (g) { const unsigned gen = g->gen + 1; memset(g, 0, sizeof(*g)); g->gen = gen; }
    else { g = (gpu_group *) xcalloc(1, sizeof(gpu_group)); g->gen = 1; }
    g->owner = m0; g->n = n;
    g->bucket_of = (int *) xcalloc(n, sizeof(int)); g->pos_of = (int *) xcalloc(n, sizeof(int));
#if defined(_OPENMP)
    g->scratch_nt = omp_get_max_threads() > 0 ? omp_get_max_threads() : 1;
#else
    g->scratch_nt = 1;
#endif
    g->scratch = (int *) xcalloc((size_t) need * (size_t) g->scratch_nt, sizeof(int)); g->scratch_cap = need;
    int cap_bk = 4;
    g->bk = (gpu_bucket *) xcalloc(cap_bk, sizeof(gpu_bucket));
    for (int i = 0; i < n; i++)
    {
        /* an instance that does not take part while the group is built rides in the class of the first one that does */
        const ocp_qp_in *in = skip && skip[i] ? ins[first] : ins[i];
        const int len = fill_sig(in, g->scratch);
        int q = 0;
        for (; q < g->nbk; q++)
            if (g->bk[q].sig_len == len && memcmp(g->bk[q].sig, g->scratch, sizeof(int) * len) == 0) break;
        if (q == g->nbk)
        {
            if (g->nbk == cap_bk)
            {
                cap_bk *= 2;
                g->bk = (gpu_bucket *) realloc(g->bk, sizeof(gpu_bucket) * cap_bk);
                if (!g->bk) { printf("\nerror: ocp_qp_gpu_ipm: out of host memory\n"); exit(1); }
                memset(g->bk + g->nbk, 0, sizeof(gpu_bucket) * (cap_bk - g->nbk));
            }
            gpu_bucket *bk = g->bk + g->nbk++;
            bk->sig = (int *) xcalloc(len, sizeof(int)); bk->sig_cap = len; bk->sig_len = len;
            memcpy(bk->sig, g->scratch, sizeof(int) * len);
            bk->members = (int *) xcalloc(n, sizeof(int));
            pthread_mutex_init(&bk->mu, NULL); bk->mu_live = 1;
        }
        gpu_bucket *bk = g->bk + q;
        g->bucket_of[i] = q; g->pos_of[i] = bk->n;
        bk->members[bk->n++] = i;
    }
    for (int q = 0; q < g->nbk; q++)
    {
        gpu_bucket *bk = g->bk + q;
        int rep = bk->members[0];
        for (int e = 0; e < bk->n; e++) if (!(skip && skip[bk->members[e]])) { rep = bk->members[e]; break; }
        const ocp_qp_in *in0 = skip && skip[rep] ? ins[first] : ins[rep];
        const int nst = in0->dim->N + 1;
        bk->seg_cap_in = nst * SEGS_IN_PER_STAGE; bk->seg_cap_out = nst * SEGS_OUT_PER_STAGE; bk->seg_cap_seed = nst * SEGS_SEED_PER_STAGE;
        bk->seg_in = (gpu_seg *) xcalloc(bk->seg_cap_in, sizeof(gpu_seg));
        bk->seg_out = (gpu_seg *) xcalloc(bk->seg_cap_out, sizeof(gpu_seg));
        bk->seg_seed = (gpu_seg *) xcalloc(bk->seg_cap_seed, sizeof(gpu_seg));
        bk->seg_cap_vec = bk->seg_cap_in;
        bk->seg_vec = (gpu_seg *) xcalloc(bk->seg_cap_vec, sizeof(gpu_seg));
        bk->st = (int *) xcalloc(bk->n, sizeof(int)); bk->it = (int *) xcalloc(bk->n, sizeof(int));
        memcpy(g->scratch, bk->sig, sizeof(int) * bk->sig_len);
        (void) bucket_build(bk, in0, g->scratch, bk->sig_len); /* on a device failure the bucket has no batch: bucket_solve fails its QPs */
        /* pinned staging: the input blob also stages the seeds (never longer than the QP data) */
        int per = bk->L_in > bk->L_seed ? bk->L_in : bk->L_seed;
        for (int k = 0; k < nst; k++) /* solver_get stages ric_L and ric_l of the whole bucket */
        {
            const int nv = in0->dim->nu[k] + in0->dim->nx[k];
            if (nv * (nv + 1) > per) per = nv * (nv + 1);
        }
        bk->cap_in = (size_t) bk->n * (size_t) per;
        bk->cap_out = (size_t) bk->n * (size_t) (bk->L_out > 0 ? bk->L_out : 1);
        bk->blob_in = (double *) ocp_qp_gpu_host_alloc(sizeof(double) * bk->cap_in);
        bk->blob_out = (double *) ocp_qp_gpu_host_alloc(sizeof(double) * bk->cap_out);
        if (!bk->blob_in || !bk->blob_out) exit(1);
    }
    m0->group = g; m0->group_gen = g->gen;
    return g;
}

/*
 * n QPs in acados structs -> one device batch per structure class.  `mem[i]` is the qp solver memory of capsule i (each
 * from this plugin's memory_assign); afterwards memory_get(mem[i], "status" / "iter" / "time_qp_solver_call") and
 * qp_out[i]->misc answer for QP i exactly as after a single evaluate, and the sensitivity / solver_get slots called with
 * mem[i] address instance i of the shared batch.  Returns the worst status (0, else the first that is not MAXITER).
 */
static int evaluate_batch_masked(void *config, int n, void **qp_in_, void **qp_out_, void *opts_, void **mem_, void *work,
                                 const unsigned char *skip, int mode)
{
    /* skip[i] != 0: capsule i does not take part in this call (its SQP loop has ended or it waits elsewhere): its
     * qp_in / qp_out are not touched, its slot of the device batch re-solves the data it holds -- the batch keeps its
     * size, nothing is re-created while capsules drop out one by one */
    if (n <= 0) return ACADOS_SUCCESS;
    const double t_start = now_s();
    ocp_qp_in **ins = (ocp_qp_in **) qp_in_;
    ocp_qp_out **outs = (ocp_qp_out **) qp_out_;
    ocp_qp_gpu_ipm_memory **mems = (ocp_qp_gpu_ipm_memory **) mem_;
    ocp_qp_gpu_ipm_opts *o = (ocp_qp_gpu_ipm_opts *) opts_;
    int fresh = 0;
    gpu_group *g = group_for(n, ins, mems, skip, &fresh);
    if (!g) return ACADOS_SUCCESS;
    const int ws = mode != BATCH_LHS && o->warm_start >= 2 ? o->warm_start : 0; /* (the preparation half has no qp_out) */
    /* RTI feedback: only the VECTOR members are read and sent where the class's matrices are resident (BATCH_LHS ran on this group);
     * anything else -- a new group, a class without its preparation -- is a full evaluate */
    int vec_only = mode == BATCH_RHS_SOLVE && !skip && !fresh;
    for (int q = 0; q < g->nbk && vec_only; q++) if (!g->bk[q].lhs_resident || g->bk[q].L_vec <= 0) vec_only = 0;
    if (mode == BATCH_RHS_SOLVE && !vec_only) for (int q = 0; q < g->nbk; q++) g->bk[q].lhs_resident = 0;

    /* host threads: every member array of every qp_in, panel-major -> the bucket's pinned blob.  Every capsule present (the batch
     * entry; a rendezvous round may have absent ones): bucket by bucket and, inside a large bucket, in CHUNKS of instances -- a
     * finished chunk goes to the device at once (asynchronous copy on the batch's stream), so the host->device copy of the QP data
     * (85 KB per C2-shaped QP: as long as the unpacking itself) runs while the host threads unpack the next chunk */
    int staged = 0;
    const double t_grouped = now_s();
    const int zc_on = !skip && zc_prepare(g, n, ins, vec_only);
    if (getenv("ACADOS_AMD_TRACE_HOST")) fprintf(stderr, "host: group_for %.3f ms, zc_prepare %.3f ms\n", (t_grouped - t_start) * 1e3, (now_s() - t_grouped) * 1e3);
    if (zc_on)
    {
        /* zero-copy: nothing of the QP data is read here -- the device gathers it from the capsules' memory (bucket_solve) */
        staged = 2;
        if (ws >= 2)
        {
#pragma omp parallel for schedule(static)
            for (int i = 0; i < n; i++)
            {
                gpu_bucket *bk = g->bk + g->bucket_of[i];
                unpack_qp_out_duals(&bk->lay, outs[i], bk->blob_out + (size_t) g->pos_of[i] * (size_t) bk->L_out);
            }
        }
    }
    else if (vec_only)
    {
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; i++)
        {
            gpu_bucket *bk = g->bk + g->bucket_of[i];
            if (i + 1 < n) prefetch_qp_vec(ins[i + 1]);
            unpack_qp_vec(&bk->lay, ins[i], bk->blob_in + (size_t) g->pos_of[i] * (size_t) bk->L_vec);
            if (ws >= 2) unpack_qp_out_duals(&bk->lay, outs[i], bk->blob_out + (size_t) g->pos_of[i] * (size_t) bk->L_out);
        }
        for (int q = 0; q < g->nbk; q++) g->bk[q].blob_clean = 0; /* (the vector blob has its own stride: the full blob's zeros are gone) */
    }
    else if (!skip && !getenv("ACADOS_AMD_NO_CHUNKS"))
    {
        staged = 1;
        for (int q = 0; q < g->nbk; q++)
        {
            gpu_bucket *bk = g->bk + q;
            const int nch = bk->n >= 512 ? 8 : (bk->n >= 128 ? 4 : 1), per = (bk->n + nch - 1) / nch;
            for (int c0 = 0; c0 < bk->n; c0 += per)
            {
                const int c1 = c0 + per < bk->n ? c0 + per : bk->n;
#pragma omp parallel for schedule(static)
                for (int e = c0; e < c1; e++)
                {
                    const int i = bk->members[e];
                    double *blob = bk->blob_in + (size_t) e * (size_t) bk->L_in;
                    if (!bk->blob_clean) memset(blob, 0, sizeof(double) * (size_t) bk->L_in);
                    if (e + 1 < c1) prefetch_qp_in(ins[bk->members[e + 1]]);
                    unpack_qp_in(&bk->lay, ins[i], blob);
                    if (ws >= 2) unpack_qp_out_duals(&bk->lay, outs[i], bk->blob_out + (size_t) e * (size_t) bk->L_out);
                }
                /* (no device batch: the device failed while it was built, bucket_solve fails the bucket's QPs; a failed copy:
                 * the whole blob once more through _set_bulk, which reports it) */
                if (!bk->batch || ocp_qp_gpu_batch_set_bulk_chunk(bk->batch, bk->blob_in + (size_t) c0 * (size_t) bk->L_in, c0, c1 - c0) != 0)
                    staged = 0;
            }
        }
    }
    else
    {
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; i++)
        {
            if (skip && skip[i]) continue;
            const gpu_bucket *bk = g->bk + g->bucket_of[i];
            double *blob = bk->blob_in + (size_t) g->pos_of[i] * (size_t) bk->L_in;
            if (!bk->blob_clean) memset(blob, 0, sizeof(double) * (size_t) bk->L_in);
            unpack_qp_in(&bk->lay, ins[i], blob);
            if (ws >= 2) unpack_qp_out_duals(&bk->lay, outs[i], bk->blob_out + (size_t) g->pos_of[i] * (size_t) bk->L_out);
        }
    }
    if (skip && (fresh || ws >= 2))
        for (int q = 0; q < g->nbk; q++)
        {
            /* a slot whose capsule is absent: in a new group it has no data yet -- it gets a copy of a present
             * neighbour's QP; on a hot start its iterate is whatever the last gather left in the staging */
            const gpu_bucket *bk = g->bk + q;
            int src = -1;
            for (int e = 0; e < bk->n && src < 0; e++) if (!skip[bk->members[e]]) src = e;
            for (int e = 0; e < bk->n && src >= 0 && fresh; e++)
                if (skip[bk->members[e]])
                    memcpy(bk->blob_in + (size_t) e * (size_t) bk->L_in, bk->blob_in + (size_t) src * (size_t) bk->L_in, sizeof(double) * (size_t) bk->L_in);
        }
    if (!skip && !vec_only && staged != 2) for (int q = 0; q < g->nbk; q++) g->bk[q].blob_clean = 1; /* every slot was cleared (or was clean) and holds QP data only */
    const double t_packed = now_s();

    /* one copy + one scatter launch, the solve, one gather launch + one copy per bucket; buckets run side by side
     * (each device batch has its own stream) */
#pragma omp parallel for schedule(dynamic, 1) if (g->nbk > 1)
    for (int q = 0; q < g->nbk; q++) bucket_solve(g->bk + q, o, ws, staged, mode);
    const double t_solved = now_s();
    if (mode == BATCH_LHS)
    {
        /* nothing to hand back but the verdict: qp_out is the feedback half's */
        int worst_lhs = 0;
        for (int q = 0; q < g->nbk; q++) if (g->bk[q].status != ACADOS_SUCCESS) worst_lhs = g->bk[q].status;
        for (int i = 0; i < n; i++)
        {
            ocp_qp_gpu_ipm_memory *mi = mems[i];
            if (mem_group(mi) && mi->group != g && mi->group->owner == mi) group_release(mi->group);
            mi->group = g; mi->group_gen = g->gen; mi->g_bucket = g->bucket_of[i]; mi->g_pos = g->pos_of[i];
            mi->time_unpack_in = t_packed - t_start; mi->time_pack_out = 0.0; mi->time_qp_solver_call = t_solved - t_packed;
            mi->upload_doubles = g->bk[g->bucket_of[i]].L_in; mi->zero_copy = staged == 2;
        }
        return worst_lhs;
    }

#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++)
    {
        if (skip && skip[i]) continue;
        const gpu_bucket *bk = g->bk + g->bucket_of[i];
        pack_qp_out(&bk->lay, bk->blob_out + (size_t) g->pos_of[i] * (size_t) bk->L_out, outs[i]);
    }
    const double t_end = now_s();

    int worst = 0;
    for (int i = 0; i < n; i++)
    {
        if (skip && skip[i]) continue;
        const gpu_bucket *bk = g->bk + g->bucket_of[i];
        const int st = bk->st[g->pos_of[i]], it = bk->it[g->pos_of[i]];
        qp_info *info = (qp_info *) outs[i]->misc;
        if (info)
        {
            info->solve_QP_time = ocp_qp_gpu_batch_get_scalar(bk->batch, "time_tot"); /* of the whole bucket */
            info->condensing_time = 0.0;
            info->interface_time = (t_packed - t_start) + (t_end - t_solved);
            info->total_time = t_end - t_start;
            info->num_iter = it;
            info->t_computed = 1;
        }
        ocp_qp_gpu_ipm_memory *mi = mems[i];
        if (mem_group(mi) && mi->group != g && mi->group->owner == mi) group_release(mi->group); /* it led another batch before */
        mi->group = g; mi->group_gen = g->gen; mi->g_bucket = g->bucket_of[i]; mi->g_pos = g->pos_of[i];
        mi->iter = it; mi->status = st; mi->time_qp_solver_call = t_solved - t_packed;
        mi->time_unpack_in = t_packed - t_start; mi->time_pack_out = t_end - t_solved;
        mi->upload_doubles = vec_only ? bk->L_vec : bk->L_in; mi->zero_copy = staged == 2;
        if (st != ACADOS_SUCCESS && (worst == 0 || worst == ACADOS_MAXITER)) worst = st;
    }
    return worst;
}

int ocp_qp_gpu_ipm_acados_evaluate_batch(void *config, int n, void **qp_in, void **qp_out, void *opts, void **mem, void *work)
{
    return evaluate_batch_masked(config, n, qp_in, qp_out, opts, mem, work, NULL, BATCH_SOLVE);
}

/* the two halves of an RTI step for n capsules (the batch counterparts of the condense_lhs / condense_rhs_and_solve slots,
 * ocp_qp_xcond_solver.c:591-669): PREPARATION -- every member array of every qp_in to the device, the matrix part of the condensing
 * there (nothing is solved, qp_out is not touched); FEEDBACK -- only the VECTOR members (b, rqz, d, d_mask: what ocp_nlp rewrites
 * between the two, ocp_nlp_common.c:3119-3138) are read and sent, 12.6 KB instead of 85 KB per C2-shaped QP, then the vector part of the
 * condensing, the IPM and the expansion.  A feedback call whose preparation did not run on the same set of capsules is a full evaluate. */
int ocp_qp_gpu_ipm_acados_condense_lhs_batch(void *config, int n, void **qp_in, void *opts, void **mem, void *work)
{
    return evaluate_batch_masked(config, n, qp_in, NULL, opts, mem, work, NULL, BATCH_LHS);
}

int ocp_qp_gpu_ipm_acados_condense_rhs_and_solve_batch(void *config, int n, void **qp_in, void **qp_out, void *opts, void **mem, void *work)
{
    return evaluate_batch_masked(config, n, qp_in, qp_out, opts, mem, work, NULL, BATCH_RHS_SOLVE);
}

/*
 * The same batch WITHOUT touching ocp_nlp: the generated `_acados_batch_solve` keeps its loop
 *     #pragma omp parallel for  ->  ocp_nlp_solve(capsule[i])                    (acados_solver.in.c:3232-3236)
 * with one thread per capsule, every capsule's SQP loop reaches qp_solver->evaluate on its own (ocp_nlp_common.c:4517), and
 * `evaluate` -- with a rendezvous in its opts -- parks the calling thread until every capsule still iterating has arrived;
 * the last arrival sends all their QPs to the GPU as ONE batch, every thread returns with its own status and continues its
 * own globalisation / termination test.  A capsule whose ocp_nlp_solve has returned leaves the rendezvous; its slot of the
 * device batch rides along.  (Capsules share their solver options: the batch runs with the opts of the last arrival.)
 */
typedef struct ocp_qp_gpu_ipm_rendezvous_
{
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int n, next_index, active, arrived;
    unsigned long round;
    void **ins, **outs, **mems;
    unsigned char *skip;     /* 1: not in this round */
    void *config, *opts;
} ocp_qp_gpu_ipm_rendezvous;

ocp_qp_gpu_ipm_rendezvous *ocp_qp_gpu_ipm_acados_rendezvous_create(int n_capsules)
{
    ocp_qp_gpu_ipm_rendezvous *r = (ocp_qp_gpu_ipm_rendezvous *) xcalloc(1, sizeof(*r));
    pthread_mutex_init(&r->mu, NULL);
    pthread_cond_init(&r->cv, NULL);
    r->n = r->active = n_capsules;
    r->ins = (void **) xcalloc(n_capsules, sizeof(void *)); r->outs = (void **) xcalloc(n_capsules, sizeof(void *));
    r->mems = (void **) xcalloc(n_capsules, sizeof(void *));
    r->skip = (unsigned char *) xcalloc(n_capsules, 1);
    memset(r->skip, 1, n_capsules);
    return r;
}

void ocp_qp_gpu_ipm_acados_rendezvous_destroy(ocp_qp_gpu_ipm_rendezvous *r)
{
    if (!r) return;
    pthread_mutex_destroy(&r->mu); pthread_cond_destroy(&r->cv);
    free(r->ins); free(r->outs); free(r->mems); free(r->skip);
    free(r);
}

/* every capsule takes part again (before the next `_acados_batch_solve`) */
void ocp_qp_gpu_ipm_acados_rendezvous_reset(ocp_qp_gpu_ipm_rendezvous *r)
{
    pthread_mutex_lock(&r->mu);
    r->active = r->n; r->arrived = 0;
    memset(r->skip, 1, r->n);
    pthread_mutex_unlock(&r->mu);
}

static void rendezvous_round(ocp_qp_gpu_ipm_rendezvous *r) /* mutex held */
{
    if (!r->mems[0])
    {
        /* slot 0 owns the device batches (group_for): it must have been filled once */
        printf("\nerror: ocp_qp_gpu_ipm rendezvous: capsule 0 left before its first QP\n");
        exit(1);
    }
    evaluate_batch_masked(r->config, r->n, r->ins, r->outs, r->opts, r->mems, NULL, r->skip, BATCH_SOLVE);
    memset(r->skip, 1, r->n);
    r->arrived = 0;
    r->round++;
    pthread_cond_broadcast(&r->cv);
}

static int rendezvous_evaluate(ocp_qp_gpu_ipm_rendezvous *r, void *config, void *qp_in, void *qp_out, void *opts, ocp_qp_gpu_ipm_memory *m)
{
    pthread_mutex_lock(&r->mu);
    if (m->rv_index < 0)
    {
        if (r->next_index >= r->n) { printf("\nerror: ocp_qp_gpu_ipm rendezvous: more capsules than it was created for (%d)\n", r->n); exit(1); }
        m->rv_index = r->next_index++;
    }
    const int idx = m->rv_index;
    r->ins[idx] = qp_in; r->outs[idx] = qp_out; r->mems[idx] = m; r->skip[idx] = 0;
    r->config = config; r->opts = opts;
    r->arrived++;
    if (r->arrived >= r->active) rendezvous_round(r);
    else
    {
        const unsigned long my = r->round;
        while (r->round == my) pthread_cond_wait(&r->cv, &r->mu);
    }
    pthread_mutex_unlock(&r->mu);
    return m->status;
}

/* capsule done (its ocp_nlp_solve returned): the others no longer wait for it */
void ocp_qp_gpu_ipm_acados_rendezvous_leave(ocp_qp_gpu_ipm_rendezvous *r)
{
    pthread_mutex_lock(&r->mu);
    r->active--;
    if (r->arrived > 0 && r->arrived >= r->active) rendezvous_round(r);
    pthread_mutex_unlock(&r->mu);
}

/* where the QP of this memory was solved last */
static gpu_bucket *bucket_of_mem(ocp_qp_gpu_ipm_memory *m, int *pos)
{
    gpu_group *g = mem_group(m);
    if (g) { *pos = m->g_pos; return g->bk + m->g_bucket; }
    *pos = 0;
    return &m->own;
}

/* ------------------------------------------------------------------ getters, sensitivities */

static void gpu_solver_get(void *config_, void *qp_in_, void *qp_out_, void *opts_, void *mem_, const char *field, int stage,
                           void *value, int size1, int size2)
{
    /* ocp_qp_hpipm.c:417-478: P p K k Lr from the factor of the last factorisation held in HBM */
    ocp_qp_in *in = (ocp_qp_in *) qp_in_;
    ocp_qp_gpu_ipm_memory *m = (ocp_qp_gpu_ipm_memory *) mem_;
    const int nx = in->dim->nx[stage], nu = in->dim->nu[stage], nv = nu + nx;
    double *out = (double *) value;
    int pos = 0;
    gpu_bucket *bk = bucket_of_mem(m, &pos);
    if (!bk->batch) { printf("\nocp_qp_gpu_ipm_solver_get: no factorisation available (solve first)\n"); exit(1); }
    /* staging is idle between evaluates: n * ((nu+nx)^2 + nu+nx) fit (blob_in_cap); shared by the capsules of a class ->
     * locked, and this capsule's block is copied out before the lock is dropped */
    bucket_lock(bk);
    double *Lb = bk->blob_in, *lb = bk->blob_in + (size_t) bk->n * nv * nv;
    bk->blob_clean = 0;
    ocp_qp_gpu_batch_get(bk->batch, "ric_L", stage, Lb, 0);
    ocp_qp_gpu_batch_get(bk->batch, "ric_l", stage, lb, 0);
    double *mine = (double *) malloc(sizeof(double) * (size_t) (nv * nv + nv + 1));
    if (!mine) { printf("\nerror: ocp_qp_gpu_ipm_solver_get: out of host memory\n"); exit(1); }
    memcpy(mine, Lb + (size_t) pos * nv * nv, sizeof(double) * (size_t) (nv * nv));
    memcpy(mine + nv * nv, lb + (size_t) pos * nv, sizeof(double) * (size_t) nv);
    bucket_unlock(bk);
    const double *L = mine, *l = mine + nv * nv;
    if (!strcmp(field, "P"))
        for (int c = 0; c < nx; c++) for (int r = 0; r < nx; r++)
        {
            double a = 0.0;
            for (int q = 0; q <= (r < c ? r : c); q++) a += L[(nu + r) + nv * (nu + q)] * L[(nu + c) + nv * (nu + q)];
            out[r + nx * c] = a;
        }
    else if (!strcmp(field, "p"))
        for (int r = 0; r < nx; r++)
        {
            double a = 0.0;
            for (int q = 0; q <= r; q++) a += L[(nu + r) + nv * (nu + q)] * l[nu + q];
            out[r] = a;
        }
    else if (!strcmp(field, "K") || !strcmp(field, "k"))
    {
        const int isK = field[0] == 'K', ncol = isK ? nx : 1;
        for (int c = 0; c < ncol; c++)
            for (int r = nu - 1; r >= 0; r--)
            {
                double a = isK ? -L[(nu + c) + nv * r] : -l[r];
                for (int q = r + 1; q < nu; q++) a -= L[q + nv * r] * out[q + nu * c];
                out[r + nu * c] = a / L[r + nv * r];
            }
    }
    else if (!strcmp(field, "Lr"))
        for (int c = 0; c < nu; c++) for (int r = 0; r < nu; r++) out[r + nu * c] = r >= c ? L[r + nv * c] : 0.0;
    else
        printf("\nocp_qp_gpu_ipm_solver_get: field %s not supported", field);
    free(mine);
}

static void gpu_memory_reset(void *config, void *qp_in, void *qp_out, void *opts, void *mem_, void *work)
{
    ocp_qp_gpu_ipm_memory *m = (ocp_qp_gpu_ipm_memory *) mem_;
    if (m->own.batch) ocp_qp_gpu_batch_destroy(m->own.batch);
    m->own.batch = NULL;
    m->own.sig_len = 0;
    if (mem_group(m) && m->group->owner == m) group_release(m->group); /* the other members see the generation change (mem_group) */
    m->group = NULL;
}

/*
 * ocp_qp_hpipm.c:481-506 -> d_ocp_qp_ipm_sens_frw / _sens_adj: d(solution)/d(parameter) for the seed = derivative of the
 * problem data, with the factorisation at the solution.  `seed` is HPIPM's d_ocp_qp_seed (BLASFEO vectors seed_g /
 * seed_b / seed_d laid out like rqz / b / d); the result lands in sens_qp_out (ux, pi, lam, t -- what
 * ocp_nlp_common.c:4095-4104 copies).  The KKT matrix of the Newton system is symmetric, so the adjoint solve of a seed
 * is the forward solve of the same seed (acados seeds only seed_g there and reads ux, pi: ocp_nlp_common.c:4128-4160).
 */
static void bucket_sens(gpu_bucket *bk)
{
    if (ocp_qp_gpu_batch_sens_set_bulk(bk->batch, bk->blob_in, 0) != 0 || ocp_qp_gpu_batch_sens_solve(bk->batch) != 0
        || ocp_qp_gpu_batch_sens_get_bulk(bk->batch, bk->blob_out, 0) != 0)
    {
        printf("\nerror: ocp_qp_gpu_ipm: sensitivity solve failed\n");
        exit(1);
    }
}

static void gpu_eval_sens(void *config, void *qp_in, void *seed_, void *sens_qp_out_, void *opts, void *mem_, void *work)
{
    ocp_qp_gpu_ipm_memory *m = (ocp_qp_gpu_ipm_memory *) mem_;
    int pos = 0;
    gpu_bucket *bk = bucket_of_mem(m, &pos);
    if (!bk->batch) { printf("\nerror: ocp_qp_gpu_ipm: eval_forw_sens / eval_adj_sens before the first evaluate\n"); exit(1); }
    /* the seed belongs to this memory's instance; the other instances of a shared batch get zero seeds.  One capsule at a
     * time per class (the per-capsule slot of a SHARED batch costs a pass over the whole batch: n capsules calling it are
     * n passes -- ocp_qp_gpu_ipm_acados_eval_sens_batch below does the n seeds in ONE pass and is what a batched caller
     * should use; this slot stays correct, not fast, under the generated OpenMP loops) */
    bucket_lock(bk);
    bk->blob_clean = 0;
    memset(bk->blob_in, 0, sizeof(double) * (size_t) bk->n * (size_t) bk->L_seed);
    unpack_seed(&bk->lay, (ocp_qp_seed *) seed_, bk->blob_in + (size_t) pos * (size_t) bk->L_seed);
    bucket_sens(bk);
    pack_qp_out(&bk->lay, bk->blob_out + (size_t) pos * (size_t) bk->L_out, (ocp_qp_out *) sens_qp_out_);
    bucket_unlock(bk);
}

/* the same for all n capsules of the last ocp_qp_gpu_ipm_acados_evaluate_batch at once: one seed each, one device pass
 * per structure class (replaces the loops of acados_solver.in.c:3292-3337) */
void ocp_qp_gpu_ipm_acados_eval_sens_batch(void *config, int n, void **qp_in, void **seed_, void **sens_qp_out_, void *opts, void **mem_, void *work)
{
    if (n <= 0) return;
    ocp_qp_gpu_ipm_memory **mems = (ocp_qp_gpu_ipm_memory **) mem_;
    gpu_group *g = mem_group(mems[0]);
    int ok = g != NULL && g->n == n;
    for (int i = 0; i < n && ok; i++) ok = mem_group(mems[i]) == g;
    if (!ok)
    {
        printf("\nerror: ocp_qp_gpu_ipm_acados_eval_sens_batch: the %d memories are not those of the last evaluate_batch\n", n);
        exit(1);
    }
    for (int q = 0; q < g->nbk; q++) g->bk[q].blob_clean = 0;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++)
    {
        const gpu_bucket *bk = g->bk + mems[i]->g_bucket;
        double *blob = bk->blob_in + (size_t) mems[i]->g_pos * (size_t) bk->L_seed;
        memset(blob, 0, sizeof(double) * (size_t) bk->L_seed);
        unpack_seed(&bk->lay, (ocp_qp_seed *) seed_[i], blob);
    }
#pragma omp parallel for schedule(dynamic, 1) if (g->nbk > 1)
    for (int q = 0; q < g->nbk; q++) bucket_sens(g->bk + q);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++)
    {
        const gpu_bucket *bk = g->bk + mems[i]->g_bucket;
        pack_qp_out(&bk->lay, bk->blob_out + (size_t) mems[i]->g_pos * (size_t) bk->L_out, (ocp_qp_out *) sens_qp_out_[i]);
    }
}

static void gpu_terminate(void *config, void *mem, void *work) { gpu_memory_reset(config, NULL, NULL, NULL, mem, work); }

/* ocp_qp_hpipm.c:517-540 */
void ocp_qp_gpu_ipm_acados_config_initialize_default(void *config_)
{
    qp_solver_config *config = (qp_solver_config *) config_;
    config->dims_set = &gpu_dims_set;
    config->opts_calculate_size = &gpu_opts_calculate_size;
    config->opts_assign = &gpu_opts_assign;
    config->opts_initialize_default = &gpu_opts_initialize_default;
    config->opts_update = &gpu_opts_update;
    config->opts_set = &gpu_opts_set;
    config->opts_get = &gpu_opts_get;
    config->memory_calculate_size = &gpu_memory_calculate_size;
    config->memory_assign = &gpu_memory_assign;
    config->memory_get = &gpu_memory_get;
    config->workspace_calculate_size = &gpu_workspace_calculate_size;
    config->evaluate = &ocp_qp_gpu_ipm_acados;
    config->solver_get = &gpu_solver_get;
    config->memory_reset = &gpu_memory_reset;
    config->eval_forw_sens = &gpu_eval_sens;
    config->eval_adj_sens = &gpu_eval_sens;
    config->terminate = &gpu_terminate;
}
