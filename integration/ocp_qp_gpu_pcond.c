/*
 * ocp_qp_gpu_pcond.c -- partial condensing ON THE DEVICE behind acados' OWN types: the 20 slots of ocp_qp_xcond_config
 * (acados/ocp_qp/ocp_qp_common.h:84-107), the file a maintainer drops into acados/ocp_qp/ next to
 * ocp_qp_partial_condensing.c and registers with
 *
 *     ocp_qp_gpu_pcond_acados_config_initialize_default(solver_config->xcond);
 *
 * in the `case PARTIAL_CONDENSING_GPU_IPM:` of ocp_qp_xcond_solver_config_initialize_from_plan
 * (interfaces/acados_c/ocp_qp_interface.c:91-182; integration/acados.patch).  qp_in / xcond_qp_in / qp_out / seeds are HPIPM's
 * d_ocp_qp / d_ocp_qp_sol / d_ocp_qp_seed holding BLASFEO objects; the arithmetic -- the Gamma products, Z'HZ, the condensed
 * bounds -> general rows, the expansion with its adjoint sweep for pi -- runs in the condensing kernels of libacados_amd_qp.so
 * (km_pcond / kz_pcond / kw_pcond / k_pexpand) through the device-batch C-ABI (include/acados_amd/ocp_qp_gpu_batch.h).
 *
 * What each slot does (reference: acados/ocp_qp/ocp_qp_partial_condensing.c):
 *   condensing        :523-556   qp_in -> device (one blob, one copy, one scatter launch), condensing kernel, the condensed QP
 *                                back in ONE gather launch + one copy, packed into xcond_qp_in (BLASFEO, panel-major)
 *   condense_lhs      :575-598   matrix part only (RTI preparation), the condensed matrices packed
 *   condense_rhs      :602-630   vector part only (RTI feedback), the condensed vectors packed
 *   condense_qp_out   :559-571   an iterate of the original QP restated in the condensed variables (warm start)
 *   expansion         :664-689   condensed solution -> device, expansion kernel, full solution (u x sl su pi lam t) packed
 *   condense_rhs_seed :634-662   seeds of a sensitivity solve through the (linear, homogeneous) vector condensing
 *   expand_sol_seed   :691-717   condensed sensitivities expanded
 *   dims_get("xcond_dims"), memory_get("xcond_qp_in" / "xcond_qp_out" / "xcond_seed" / "qp_out_info" / "time_qp_xcond")
 *                     :138-155, 467-504
 *   opts_set "N", "N_bkp", "ric_alg", "block_size" (the outer solver strips "cond_", ocp_qp_xcond_solver.c:283-311) :283-323
 *
 * Dims of the condensed QP: computed by the device library (the ONE place that decides which classes are condensable and how
 * rows are ordered), asked of a one-instance probe batch when the memory is sized.  Two deliberate differences to
 * d_part_cond_qp_compute_dim, neither of which changes the solution of the ORIGINAL QP:
 *   - (round 6: no longer a difference) x0 IS eliminated before condensing as HPIPM's d_ocp_qp_reduce_eq_dof does (:542): the
 *     states of stage 0 that equality-flagged bounds fix (idxe) leave the QP on the host -- A_0 x0 goes into b_0, H_0[., F] x0 into
 *     [r_0; q_0], C_0 x0 into the general bounds -- the device condenses the REDUCED QP (nx[0] smaller by nbxe[0], the rows gone), and
 *     the expansion puts x0 back and recovers the multipliers of its rows from stationarity (d_ocp_qp_restore_eq_dof, :683).  xcond
 *     dims / pcond_* getters show what HPIPM's show at stage 0.  Option "reduce_eq_dof" = 0 keeps the rows (the device masks the
 *     fixed variables itself: what the fused batch route does);
 *   - a block counts the class's kernel NU inputs per stage it holds: HPIPM's sum of nu wherever every stage of the block has that many
 *     inputs (uneven user blocks included), larger where a stage inside a block has fewer;
 *   - a non-zero LAST block size (block_size[N2] > 0, e.g. [6,5,4,2,2,1] of pcond_getters_test.py:200) gives one more stage
 *     with inputs in front of an input-free terminal stage (N2 + 1 stages with inputs) instead of inputs in the terminal stage.
 * A class the device does not condense (more than 64 rows in a condensed stage, ...) is handed through with N2 = N -- the
 * reference's own default (:243-265) -- and a message; the solution is identical.
 *
 * Memory rule: dims / opts / memory are carved from the caller's blocks (containers of the condensed QP created by the
 * reference's own ocp_qp_in_assign / ocp_qp_out_assign / ocp_qp_seed_assign); the device batch and its stream live outside
 * and are released by ocp_qp_gpu_pcond_acados_memory_release(mem) -- the xcond vtable has no terminate slot
 * (ocp_qp_common.h:84-107), the patch calls it from ocp_qp_xcond_solver_terminate (ocp_qp_xcond_solver.c).
 *
 * The BATCH route does not pass through here: n capsules' QPs with cond_N < N go through
 * ocp_qp_gpu_xcond_solver_acados_evaluate_batch (below), which hands the ORIGINAL QPs to the QP solver's batch entry with the
 * condensing options attached -- condensing, IPM and expansion then run back to back on the device, nothing returns to the host
 * in between (the fused route of SURVEY 8b "Outer vtable ... also legal").
 */
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "acados/ocp_qp/ocp_qp_common.h"
#include "acados/ocp_qp/ocp_qp_xcond_solver.h"
#include "acados/utils/types.h"
#include "blasfeo/include/blasfeo_d_aux.h"

#include "acados_amd/ocp_qp_gpu_batch.h"
#include "ocp_qp_gpu_segments.h"

/* the QP solver's batch entry and its extension options "cond_N" / "cond_block_size" (integration/ocp_qp_gpu_ipm.c) */
int ocp_qp_gpu_ipm_acados_evaluate_batch(void *config, int n, void **qp_in, void **qp_out, void *opts, void **mem, void *work);
int ocp_qp_gpu_ipm_acados_condense_lhs_batch(void *config, int n, void **qp_in, void *opts, void **mem, void *work);
int ocp_qp_gpu_ipm_acados_condense_rhs_and_solve_batch(void *config, int n, void **qp_in, void **qp_out, void *opts, void **mem, void *work);
void ocp_qp_gpu_ipm_acados_config_initialize_default(void *config_);

typedef struct
{
    ocp_qp_dims *orig_dims;
    ocp_qp_dims *red_dims;       /* orig_dims with the equality-flagged states of stage 0 eliminated (d_ocp_qp_dim_reduce_eq_dof) */
    int reduced;                 /* red_dims differs from orig_dims */
    ocp_qp_dims *pcond_dims;
    int *block_size;             /* N + 1 entries */
    int condensed;               /* 0: handed through (N2 = N, or a class the device does not condense) */
    int n_blocks;                /* stages of the condensed QP with inputs */
    unsigned long long probe_key;
    int probe_valid;
} ocp_qp_gpu_pcond_dims;

typedef struct
{
    int N2, N2_bkp;
    int ric_alg;                 /* accepted as the reference does; the device has one condensing algorithm (same result) */
    int *block_size;             /* N + 1 entries */
    bool block_size_was_set;
    int mem_qp_in;
    int reduce_eq_dof;           /* 1 (default, as the reference: :542): x0 is eliminated on the host before the device condenses */
    int batch_owned;             /* the capsule's QPs are condensed by the BATCH entries (ocp_qp_gpu_xcond_solver_acados_condense_lhs_batch ...): the
                                    per-capsule condense_lhs slot of an RTI preparation step has nothing to do (option "cond_batch_owned") */
} ocp_qp_gpu_pcond_opts;

typedef struct
{
    ocp_qp_in *pcond_qp_in;
    ocp_qp_out *pcond_qp_out;
    ocp_qp_seed *pcond_seed;
    qp_info *qp_out_info;        /* = pcond_qp_out->misc: what the QP solver fills */
    double time_qp_xcond;
    ocp_qp_gpu_pcond_dims *dims;
    gpu_layout par;              /* original QP on the device: par.batch is OWNED */
    gpu_layout chd;              /* condensed QP: chd.batch belongs to par.batch */
    int *sig, sig_len, sig_cap, *sig_scratch;
    double *blob;                /* staging, carved: the longest of the six blobs */
    size_t blob_cap;
    ocp_qp_in *ptr_qp_in;        /* last condensed (the seed pair needs it) */
    ocp_qp_seed *ptr_seed;
    int seeds_resident;          /* the parent batch holds the seeds as its vector fields (between the two seed slots) */
    /* x0 elimination (d_ocp_qp_reduce_eq_dof / _restore_eq_dof, :542 / :683), host side: containers of the REDUCED original QP.  Stages
     * 1..N alias the caller's member structs (same dims); stage 0 is this memory's own */
    ocp_qp_in *red_in;
    ocp_qp_out *red_out, *red_sens;
    ocp_qp_seed *red_seed;
    ocp_qp_in *src_in;           /* what the device holds: red_in, or ptr_qp_in where nothing is eliminated */
    int nF, *f_ib, *f_iv, *f_sign; /* fixed states of stage 0: box row, variable (index in [u; x]), sign of its multiplier at the solution */
    int *map_var, *map_row;      /* stage 0: variable / box row -> index in the reduced QP, -1: eliminated */
    double *xbar, *dxbar;        /* value of every eliminated variable (indexed like map_var), and its seed in a sensitivity solve */
} ocp_qp_gpu_pcond_memory;

static double pc_now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

static void copy_dims(const ocp_qp_dims *s, ocp_qp_dims *d)
{
    d->N = s->N;
    for (int k = 0; k <= s->N; k++)
    {
        d->nx[k] = s->nx[k]; d->nu[k] = s->nu[k]; d->nb[k] = s->nb[k]; d->nbx[k] = s->nbx[k]; d->nbu[k] = s->nbu[k];
        d->ng[k] = s->ng[k]; d->ns[k] = s->ns[k]; d->nbxe[k] = s->nbxe[k]; d->nbue[k] = s->nbue[k]; d->nge[k] = s->nge[k];
    }
}

/* ------------------------------------------------------------------ dims (:56-155) */

static acados_size_t pc_dims_calculate_size(void *config, int N)
{
    return size8(sizeof(ocp_qp_gpu_pcond_dims) + 3 * ocp_qp_dims_calculate_size(N) + sizeof(int) * (size_t) (N + 1) + 5 * 8);
}

static void *pc_dims_assign(void *config, int N, void *raw_memory)
{
    char *c = align8((char *) raw_memory);
    ocp_qp_gpu_pcond_dims *dims = (ocp_qp_gpu_pcond_dims *) c;
    memset(dims, 0, sizeof(*dims));
    c = align8(c + sizeof(*dims));
    dims->orig_dims = ocp_qp_dims_assign(N, c); c = align8(c + ocp_qp_dims_calculate_size(N));
    dims->red_dims = ocp_qp_dims_assign(N, c); c = align8(c + ocp_qp_dims_calculate_size(N));
    dims->pcond_dims = ocp_qp_dims_assign(N, c); c = align8(c + ocp_qp_dims_calculate_size(N)); /* worst case: N2 = N */
    dims->block_size = (int *) c;
    for (int i = 0; i <= N; i++) dims->block_size[i] = i < N ? 1 : 0;
    return dims;
}

static void pc_dims_set(void *config, void *dims_, int stage, const char *field, int *value)
{
    ocp_qp_gpu_pcond_dims *dims = (ocp_qp_gpu_pcond_dims *) dims_;
    ocp_qp_dims_set(config, dims->orig_dims, stage, field, value);
    dims->probe_valid = 0;
}

static void pc_dims_get(void *config, void *dims_, const char *field, void *value)
{
    ocp_qp_gpu_pcond_dims *dims = (ocp_qp_gpu_pcond_dims *) dims_;
    if (!strcmp(field, "xcond_dims")) *(ocp_qp_dims **) value = dims->pcond_dims;
    else { printf("\nerror: ocp_qp_partial_condensing_dims_get: field %s not available\n", field); exit(1); }
}

/* d_ocp_qp_dim_reduce_eq_dof (:171): the states of stage 0 fixed by equality-flagged bounds leave the QP */
static void pc_fill_red_dims(ocp_qp_gpu_pcond_dims *dims, int reduce)
{
    copy_dims(dims->orig_dims, dims->red_dims);
    ocp_qp_dims *r = dims->red_dims;
    dims->reduced = 0;
    if (!reduce || r->nbxe[0] <= 0) return;
    const int nf = r->nbxe[0];
    r->nx[0] -= nf; r->nbx[0] -= nf; r->nb[0] -= nf;
    r->nbxe[0] = 0; r->nbue[0] = 0; r->nge[0] = 0; /* (flags of input bounds / general rows carry no meaning on the device: lb = ub pairs) */
    dims->reduced = 1;
}

/* dims of the condensed QP for (orig_dims, N2, block sizes): asked of the device library (see the header comment) */
static void pc_compute_dims(ocp_qp_gpu_pcond_dims *dims, const ocp_qp_gpu_pcond_opts *opts)
{
    pc_fill_red_dims(dims, opts->reduce_eq_dof);
    const ocp_qp_dims *d = dims->red_dims;
    const int N = d->N, N2 = opts->N2;
    {
        unsigned long long h = 1469598103934665603ull; /* FNV-1a over everything the result depends on */
#define MIX(v) h = (h ^ (unsigned long long) (unsigned) (v)) * 1099511628211ull
        MIX(N); MIX(N2); MIX(opts->block_size_was_set ? 1 : 0); MIX(opts->reduce_eq_dof); MIX(dims->orig_dims->nbxe[0]);
        for (int k = 0; k <= N; k++) { MIX(d->nx[k]); MIX(d->nu[k]); MIX(d->nbx[k]); MIX(d->nbu[k]); MIX(d->ng[k]); MIX(d->ns[k]); MIX(d->nbxe[k]); }
        if (opts->block_size_was_set && N2 > 0 && N2 < N) for (int i = 0; i <= N2; i++) MIX(opts->block_size[i]);
#undef MIX
        if (dims->probe_valid && dims->probe_key == h) return;
        dims->probe_valid = 0;
        dims->probe_key = h;
    }
    dims->condensed = 0;
    dims->n_blocks = N;
    copy_dims(d, dims->pcond_dims);
    for (int i = 0; i <= N; i++) dims->block_size[i] = i < N ? 1 : 0;
    if (N2 <= 0 || N2 >= N) { dims->probe_valid = 1; return; }
    if (opts->block_size_was_set)
    {
        int sum = 0;
        for (int i = 0; i <= N2; i++) sum += opts->block_size[i];
        if (sum != N)
        {
            printf("partial condensing: sum of block_size should match N, got %d != N = %d\n", sum, N);
            exit(1); /* :346-356 */
        }
    }
    ocp_qp_gpu_batch *probe = ocp_qp_gpu_batch_create(N, d->nx, d->nu, d->nbx, d->nbu, d->ng, d->ns, 1, -1);
    if (!probe)
    {
        printf("\nerror: ocp_qp_gpu_pcond: no GPU batch could be created (no device or unsupported shape)\n");
        exit(1);
    }
    /* x0 equality rows as counted by nbxe (the dims depend on counts only); default index sets otherwise */
    int ie_cap = 1;
    for (int k = 0; k <= N; k++) if (d->nbxe[k] > ie_cap) ie_cap = d->nbxe[k];
    int *ie = (int *) calloc((size_t) ie_cap, sizeof(int));
    for (int k = 0; k <= N; k++)
        if (d->nbxe[k] > 0)
        {
            const int cnt = d->nbxe[k];
            for (int r = 0; r < cnt; r++) ie[r] = d->nbu[k] + r;
            ocp_qp_gpu_batch_set_int(probe, "idxe", k, ie, cnt);
        }
    free(ie);
    ocp_qp_gpu_batch_opts_set(probe, "cond_N", &N2);
    if (opts->block_size_was_set && ocp_qp_gpu_batch_opts_set(probe, "cond_block_size", opts->block_size) != 0) exit(1);
    ocp_qp_gpu_batch *c = ocp_qp_gpu_batch_condense(probe);
    if (c)
    {
        ocp_qp_dims *x = dims->pcond_dims;
        ocp_qp_gpu_batch_get_dims(c, "N", &x->N);
        const char *names[] = {"nx", "nu", "nb", "nbx", "nbu", "ng", "ns", "nbxe"};
        int *dst[] = {x->nx, x->nu, x->nb, x->nbx, x->nbu, x->ng, x->ns, x->nbxe};
        for (int q = 0; q < 8; q++) ocp_qp_gpu_batch_get_dims(c, names[q], dst[q]);
        for (int k = 0; k <= x->N; k++) { x->nbue[k] = 0; x->nge[k] = 0; }
        dims->condensed = 1;
        dims->n_blocks = x->N;
        for (int i = 0; i <= N2; i++)
            dims->block_size[i] = opts->block_size_was_set ? opts->block_size[i] : (i < N2 ? N / N2 + (i < N % N2 ? 1 : 0) : 0);
    }
    ocp_qp_gpu_batch_destroy(probe);
    dims->probe_valid = 1;
}

/* ------------------------------------------------------------------ opts (:163-323) */

static acados_size_t pc_opts_calculate_size(void *dims_)
{
    ocp_qp_gpu_pcond_dims *dims = (ocp_qp_gpu_pcond_dims *) dims_;
    /* "(temporarily) populate dimensions of new ocp_qp based on N2 == N" (:175-181): the outer solver sizes the QP solver's
     * opts from xcond_dims right after this call (ocp_qp_xcond_solver.c:195-203) */
    pc_fill_red_dims(dims, 1);
    copy_dims(dims->red_dims, dims->pcond_dims);
    dims->probe_valid = 0;
    return size8(sizeof(ocp_qp_gpu_pcond_opts) + sizeof(int) * (size_t) (dims->orig_dims->N + 1) + 3 * 8);
}

static void *pc_opts_assign(void *dims_, void *raw_memory)
{
    ocp_qp_gpu_pcond_dims *dims = (ocp_qp_gpu_pcond_dims *) dims_;
    char *c = align8((char *) raw_memory);
    ocp_qp_gpu_pcond_opts *opts = (ocp_qp_gpu_pcond_opts *) c;
    memset(opts, 0, sizeof(*opts));
    c = align8(c + sizeof(*opts));
    opts->block_size = (int *) c;
    for (int i = 0; i <= dims->orig_dims->N; i++) opts->block_size[i] = 0;
    return opts;
}

static void pc_opts_initialize_default(void *dims_, void *opts_)
{
    ocp_qp_gpu_pcond_dims *dims = (ocp_qp_gpu_pcond_dims *) dims_;
    ocp_qp_gpu_pcond_opts *opts = (ocp_qp_gpu_pcond_opts *) opts_;
    opts->N2 = dims->orig_dims->N; /* no partial condensing by default (:243-265) */
    opts->N2_bkp = opts->N2;
    opts->ric_alg = 0;
    opts->block_size_was_set = false;
    opts->mem_qp_in = 1;
    opts->reduce_eq_dof = 1;
    opts->batch_owned = 0;
    dims->pcond_dims->N = opts->N2;
}

static void pc_opts_update(void *dims_, void *opts_)
{
    ocp_qp_gpu_pcond_dims *dims = (ocp_qp_gpu_pcond_dims *) dims_;
    ocp_qp_gpu_pcond_opts *opts = (ocp_qp_gpu_pcond_opts *) opts_;
    opts->N2_bkp = opts->N2; /* :267-281 */
    pc_compute_dims(dims, opts);
}

static void pc_opts_set(void *opts_, const char *field, void *value)
{
    ocp_qp_gpu_pcond_opts *opts = (ocp_qp_gpu_pcond_opts *) opts_;
    if (!strcmp(field, "N")) opts->N2 = *(int *) value;
    else if (!strcmp(field, "N_bkp")) opts->N2_bkp = *(int *) value;
    else if (!strcmp(field, "ric_alg")) opts->ric_alg = *(int *) value;
    else if (!strcmp(field, "batch_owned")) opts->batch_owned = *(int *) value;
    else if (!strcmp(field, "reduce_eq_dof")) opts->reduce_eq_dof = *(int *) value; /* (before memory_calculate_size: the dims follow) */
    else if (!strcmp(field, "block_size"))
    {
        const int *v = (const int *) value;
        for (int i = 0; i < opts->N2 + 1; i++) opts->block_size[i] = v[i]; /* :305-313: N2 + 1 entries, N set first */
        opts->block_size_was_set = true;
    }
    else { printf("\nerror: field %s not available in ocp_qp_partial_condensing_opts_set\n", field); exit(1); }
}

/* ------------------------------------------------------------------ memory (:330-504) */

static size_t pc_blob_cap(const ocp_qp_dims *a, const ocp_qp_dims *b)
{
    int m = blob_in_cap(a), t;
    if ((t = blob_out_cap(a)) > m) m = t;
    if ((t = blob_in_cap(b)) > m) m = t;
    if ((t = blob_out_cap(b)) > m) m = t;
    return (size_t) m;
}

static acados_size_t pc_memory_calculate_size(void *dims_, void *opts_)
{
    ocp_qp_gpu_pcond_dims *dims = (ocp_qp_gpu_pcond_dims *) dims_;
    ocp_qp_gpu_pcond_opts *opts = (ocp_qp_gpu_pcond_opts *) opts_;
    pc_compute_dims(dims, opts);
    const size_t nst = (size_t) dims->orig_dims->N + 1;
    const int nv0 = dims->orig_dims->nu[0] + dims->orig_dims->nx[0], nr0 = dims->orig_dims->nb[0] + dims->orig_dims->ng[0];
    return size8(sizeof(ocp_qp_gpu_pcond_memory) + ocp_qp_in_calculate_size(dims->pcond_dims) + ocp_qp_out_calculate_size(dims->pcond_dims)
                 + ocp_qp_seed_calculate_size(dims->pcond_dims) + sizeof(double) * pc_blob_cap(dims->orig_dims, dims->pcond_dims)
                 + ocp_qp_in_calculate_size(dims->red_dims) + 2 * ocp_qp_out_calculate_size(dims->red_dims) + ocp_qp_seed_calculate_size(dims->red_dims)
                 + sizeof(int) * (size_t) (4 * nv0 + nr0 + 8) + 2 * sizeof(double) * (size_t) (nv0 + 1) + 10 * 8
                 + 2 * sizeof(gpu_seg) * nst * (SEGS_IN_PER_STAGE + SEGS_OUT_PER_STAGE + SEGS_SEED_PER_STAGE)
                 + 2 * sizeof(int) * (size_t) sig_len(dims->orig_dims) + 10 * 8);
}

static void pc_carve_layout(gpu_layout *l, int nst, char **c)
{
    l->seg_cap_in = nst * SEGS_IN_PER_STAGE; l->seg_cap_out = nst * SEGS_OUT_PER_STAGE; l->seg_cap_seed = nst * SEGS_SEED_PER_STAGE;
    l->seg_in = (gpu_seg *) *c; *c += sizeof(gpu_seg) * (size_t) l->seg_cap_in;
    l->seg_out = (gpu_seg *) *c; *c += sizeof(gpu_seg) * (size_t) l->seg_cap_out;
    l->seg_seed = (gpu_seg *) *c; *c += sizeof(gpu_seg) * (size_t) l->seg_cap_seed;
}

static void *pc_memory_assign(void *dims_, void *opts_, void *raw_memory)
{
    ocp_qp_gpu_pcond_dims *dims = (ocp_qp_gpu_pcond_dims *) dims_;
    const int nst = dims->orig_dims->N + 1;
    char *c = align8((char *) raw_memory);
    ocp_qp_gpu_pcond_memory *mem = (ocp_qp_gpu_pcond_memory *) c;
    memset(mem, 0, sizeof(*mem));
    c = align8(c + sizeof(*mem));
    mem->pcond_qp_in = ocp_qp_in_assign(dims->pcond_dims, c); c = align8(c + ocp_qp_in_calculate_size(dims->pcond_dims));
    mem->pcond_qp_out = ocp_qp_out_assign(dims->pcond_dims, c); c = align8(c + ocp_qp_out_calculate_size(dims->pcond_dims));
    mem->pcond_seed = ocp_qp_seed_assign(dims->pcond_dims, c); c = align8(c + ocp_qp_seed_calculate_size(dims->pcond_dims));
    mem->qp_out_info = (qp_info *) mem->pcond_qp_out->misc;
    mem->blob_cap = pc_blob_cap(dims->orig_dims, dims->pcond_dims);
    mem->blob = (double *) c; c = align8(c + sizeof(double) * mem->blob_cap);
    pc_carve_layout(&mem->par, nst, &c);
    pc_carve_layout(&mem->chd, nst, &c);
    c = align8(c);
    mem->sig_cap = sig_len(dims->orig_dims);
    mem->sig = (int *) c; c += sizeof(int) * (size_t) mem->sig_cap;
    mem->sig_scratch = (int *) c; c += sizeof(int) * (size_t) mem->sig_cap;
    c = align8(c);
    mem->red_in = ocp_qp_in_assign(dims->red_dims, c); c = align8(c + ocp_qp_in_calculate_size(dims->red_dims));
    mem->red_out = ocp_qp_out_assign(dims->red_dims, c); c = align8(c + ocp_qp_out_calculate_size(dims->red_dims));
    mem->red_sens = ocp_qp_out_assign(dims->red_dims, c); c = align8(c + ocp_qp_out_calculate_size(dims->red_dims));
    mem->red_seed = ocp_qp_seed_assign(dims->red_dims, c); c = align8(c + ocp_qp_seed_calculate_size(dims->red_dims));
    {
        const int nv0 = dims->orig_dims->nu[0] + dims->orig_dims->nx[0], nr0 = dims->orig_dims->nb[0] + dims->orig_dims->ng[0];
        mem->xbar = (double *) c; c = align8(c + sizeof(double) * (size_t) (nv0 + 1));
        mem->dxbar = (double *) c; c = align8(c + sizeof(double) * (size_t) (nv0 + 1));
        mem->f_ib = (int *) c; c += sizeof(int) * (size_t) nv0;
        mem->f_iv = (int *) c; c += sizeof(int) * (size_t) nv0;
        mem->f_sign = (int *) c; c += sizeof(int) * (size_t) nv0;
        mem->map_var = (int *) c; c += sizeof(int) * (size_t) nv0;
        mem->map_row = (int *) c; c += sizeof(int) * (size_t) (nr0 + 1);
    }
    mem->dims = dims;
    return mem;
}

static void pc_memory_get(void *config, void *mem_, const char *field, void *value)
{
    ocp_qp_gpu_pcond_memory *mem = (ocp_qp_gpu_pcond_memory *) mem_;
    if (!strcmp(field, "xcond_qp_in")) *(ocp_qp_in **) value = mem->pcond_qp_in;
    else if (!strcmp(field, "xcond_qp_out")) *(ocp_qp_out **) value = mem->pcond_qp_out;
    else if (!strcmp(field, "xcond_seed")) *(ocp_qp_seed **) value = mem->pcond_seed;
    else if (!strcmp(field, "qp_out_info")) *(qp_info **) value = mem->qp_out_info;
    else if (!strcmp(field, "time_qp_xcond")) *(double *) value = mem->time_qp_xcond;
    else { printf("\nerror: ocp_qp_partial_condensing_memory_get: field %s not available\n", field); exit(1); }
}

static acados_size_t pc_workspace_calculate_size(void *dims, void *opts) { return 0; }

void ocp_qp_gpu_pcond_acados_memory_release(void *mem_)
{
    ocp_qp_gpu_pcond_memory *mem = (ocp_qp_gpu_pcond_memory *) mem_;
    if (!mem) return;
    if (mem->par.batch) ocp_qp_gpu_batch_destroy(mem->par.batch); /* the condensed batch goes with it */
    mem->par.batch = NULL; mem->chd.batch = NULL; mem->sig_len = 0;
}

/* ------------------------------------------------------------------ containers <-> device */

/* blob -> BLASFEO containers (inverse of unpack_segs); `what`: 1 matrices (SEG_MAT / SEG_MAT_T and Z), 2 vectors, 3 both */
static void pack_segs_in(const gpu_layout *l, double *blob, ocp_qp_in *in, int what)
{
    struct blasfeo_dmat *mats[3] = {in->BAbt, in->RSQrq, in->DCt};
    struct blasfeo_dvec *vecs[8] = {NULL, NULL, NULL, in->b, in->rqz, in->d, in->d_mask, in->Z};
    for (int s = 0; s < l->n_in; s++)
    {
        const gpu_seg *g = l->seg_in + s;
        double *p = blob + g->off;
        /* "lbx#value" (the value of an equality-flagged x, read from the ITERATE of the batch) shares its place in d with "lbx",
         * which precedes it: the bound is what the container holds */
        if (s > 0 && g->kind == SEG_VEC && g[-1].kind == SEG_VEC && g[-1].src == g->src && g[-1].k == g->k && g[-1].ai == g->ai) continue;
        if (g->kind == SEG_VEC)
        {
            const int is_matrix_part = g->src == SRC_Z; /* Z belongs to the Hessian */
            if (!(what & (is_matrix_part ? 1 : 2))) continue;
            if (g->neg) for (int e = 0; e < g->len; e++) p[e] = -p[e];
            blasfeo_pack_dvec(g->m, p, 1, vecs[g->src] + g->k, g->ai);
        }
        else if (what & 1)
        {
            if (g->kind == SEG_MAT) blasfeo_pack_dmat(g->m, g->n, p, g->m, mats[g->src] + g->k, g->ai, g->aj);
            else blasfeo_pack_tran_dmat(g->n, g->m, p, g->n, mats[g->src] + g->k, g->ai, g->aj); /* blob holds the n x m transpose */
        }
    }
}

static void unpack_qp_out_full(const gpu_layout *l, ocp_qp_out *out, double *blob)
{
    struct blasfeo_dvec *vecs[12] = {NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, out->ux, out->pi, out->lam, out->t};
    for (int s = 0; s < l->n_out; s++)
    {
        const gpu_seg *g = l->seg_out + s;
        blasfeo_unpack_dvec(g->m, vecs[g->src] + g->k, g->ai, blob + g->off, 1);
    }
}

/* what the device does not carry: the copies of r, q, b in the last rows of RSQrq / BAbt (HPIPM's containers hold both; a
 * CPU solver paired with this module may read the rows), the index sets, m = 0, diag_H_flag = 0 */
static void finish_xcond_qp_in(ocp_qp_gpu_batch *c, ocp_qp_in *x, int what)
{
    const ocp_qp_dims *d = x->dim;
    for (int k = 0; k <= d->N; k++)
    {
        const int nv = d->nu[k] + d->nx[k], nx1 = k < d->N ? d->nx[k + 1] : 0, nct = 2 * (d->nb[k] + d->ng[k] + d->ns[k]);
        if (what & 2)
        {
            for (int j = 0; j < nv; j++) BLASFEO_DMATEL(x->RSQrq + k, nv, j) = BLASFEO_DVECEL(x->rqz + k, j);
            for (int j = 0; j < nx1; j++) BLASFEO_DMATEL(x->BAbt + k, nv, j) = BLASFEO_DVECEL(x->b + k, j);
            for (int i = 0; i < nct; i++) BLASFEO_DVECEL(x->m + k, i) = 0.0;
        }
        if (what & 1)
        {
            ocp_qp_gpu_batch_get_int(c, "idxb", k, x->idxb[k]);
            ocp_qp_gpu_batch_get_int(c, "idxs_rev", k, x->idxs_rev[k]);
            ocp_qp_gpu_batch_get_int(c, "idxe", k, x->idxe[k]);
            x->diag_H_flag[k] = 0;
        }
    }
}

/* original QP -> device (parent batch (re)created when the structure changes; condensing options sent once per batch) */
static ocp_qp_gpu_batch *pc_load(ocp_qp_gpu_pcond_memory *mem, const ocp_qp_gpu_pcond_opts *opts, ocp_qp_in *in)
{
    const ocp_qp_dims *d = in->dim;
    if (opts->N2 != opts->N2_bkp)
    {
        printf("\nerror: partial condensing: cond_N changed after the memory was sized (N2 = %d, at creation %d)\n", opts->N2, opts->N2_bkp);
        exit(1); /* assert(opts->N2 == opts->N2_bkp), :533 */
    }
    if (sig_len(d) > mem->sig_cap) { printf("\nerror: ocp_qp_gpu_pcond: dims of qp_in grew after memory_assign\n"); exit(1); }
    const int len = fill_sig(in, mem->sig_scratch);
    gpu_layout *p = &mem->par;
    if (!p->batch || mem->sig_len != len || memcmp(mem->sig, mem->sig_scratch, sizeof(int) * (size_t) len) != 0)
    {
        if (p->batch) ocp_qp_gpu_batch_destroy(p->batch);
        mem->chd.batch = NULL;
        mem->sig_len = 0;
        p->batch = ocp_qp_gpu_batch_create(d->N, d->nx, d->nu, d->nbx, d->nbu, d->ng, d->ns, 1, -1);
        if (!p->batch) return NULL;
        for (int k = 0; k <= d->N; k++)
        {
            ocp_qp_gpu_batch_set_int(p->batch, "idxb", k, in->idxb[k], d->nb[k]);
            ocp_qp_gpu_batch_set_int(p->batch, "idxs_rev", k, in->idxs_rev[k], d->nb[k] + d->ng[k]);
            ocp_qp_gpu_batch_set_int(p->batch, "idxe", k, in->idxe[k] + d->nbue[k], d->nbxe[k]); /* [bue | bxe | ge]: the bxe part */
        }
        if (gpu_layout_build(p, d) != 0 || (size_t) p->L_in > mem->blob_cap || (size_t) p->L_out > mem->blob_cap)
        {
            ocp_qp_gpu_batch_destroy(p->batch); p->batch = NULL;
            return NULL;
        }
        ocp_qp_gpu_batch_opts_set(p->batch, "cond_N", &opts->N2);
        if (opts->block_size_was_set && ocp_qp_gpu_batch_opts_set(p->batch, "cond_block_size", opts->block_size) != 0) exit(1);
        memcpy(mem->sig, mem->sig_scratch, sizeof(int) * (size_t) len);
        mem->sig_len = len;
    }
    memset(mem->blob, 0, sizeof(double) * (size_t) p->L_in);
    unpack_qp_in(p, in, mem->blob);
    if (ocp_qp_gpu_batch_set_bulk(p->batch, mem->blob, 0) != 0) return NULL;
    return p->batch;
}

/* segment tables of the condensed batch (once per parent batch: the child lives as long as the parent's condensing options) */
static int pc_child_layout(ocp_qp_gpu_pcond_memory *mem, ocp_qp_gpu_batch *c)
{
    if (mem->chd.batch == c) return 0;
    mem->chd.batch = c;
    if (gpu_layout_build(&mem->chd, mem->dims->pcond_dims) != 0 || (size_t) mem->chd.L_in > mem->blob_cap || (size_t) mem->chd.L_out > mem->blob_cap)
    {
        mem->chd.batch = NULL;
        return -1;
    }
    return 0;
}

static int pc_read_child(ocp_qp_gpu_pcond_memory *mem, ocp_qp_gpu_batch *c, ocp_qp_in *x, int what)
{
    if (pc_child_layout(mem, c) != 0) return -1;
    if (ocp_qp_gpu_batch_get_bulk_in(c, mem->blob, 0) != 0) return -1;
    pack_segs_in(&mem->chd, mem->blob, x, what);
    finish_xcond_qp_in(c, x, what);
    return 0;
}

/* handed through (N2 = N or a class the device does not condense): xcond_qp_in := qp_in, as the reference's module amounts to
 * with blocks of one stage */
static void pc_copy_qp_in(ocp_qp_in *a, ocp_qp_in *b, int what, double *tmp)
{
    const ocp_qp_dims *d = a->dim;
    for (int k = 0; k <= d->N; k++)
    {
        const int nv = d->nu[k] + d->nx[k], nx1 = k < d->N ? d->nx[k + 1] : 0, nb = d->nb[k], ng = d->ng[k], ns = d->ns[k];
        const int nct = 2 * (nb + ng + ns);
        if (what & 1)
        {
            /* (through the carved staging blob: no allocation inside a condensing call; it holds a whole QP, a block fits) */
            blasfeo_unpack_dmat(nv, nx1, a->BAbt + k, 0, 0, tmp, nv); blasfeo_pack_dmat(nv, nx1, tmp, nv, b->BAbt + k, 0, 0);
            blasfeo_unpack_dmat(nv, nv, a->RSQrq + k, 0, 0, tmp, nv); blasfeo_pack_dmat(nv, nv, tmp, nv, b->RSQrq + k, 0, 0);
            blasfeo_unpack_dmat(nv, ng, a->DCt + k, 0, 0, tmp, nv); blasfeo_pack_dmat(nv, ng, tmp, nv, b->DCt + k, 0, 0);
            for (int i = 0; i < 2 * ns; i++) BLASFEO_DVECEL(b->Z + k, i) = BLASFEO_DVECEL(a->Z + k, i);
            memcpy(b->idxb[k], a->idxb[k], sizeof(int) * (size_t) nb);
            memcpy(b->idxs_rev[k], a->idxs_rev[k], sizeof(int) * (size_t) (nb + ng));
            memcpy(b->idxe[k], a->idxe[k], sizeof(int) * (size_t) (d->nbxe[k] + d->nbue[k] + d->nge[k]));
            b->diag_H_flag[k] = a->diag_H_flag[k];
        }
        if (what & 2)
        {
            for (int i = 0; i < nx1; i++) BLASFEO_DVECEL(b->b + k, i) = BLASFEO_DVECEL(a->b + k, i);
            for (int i = 0; i < nv + 2 * ns; i++) BLASFEO_DVECEL(b->rqz + k, i) = BLASFEO_DVECEL(a->rqz + k, i);
            for (int i = 0; i < nct; i++)
            {
                BLASFEO_DVECEL(b->d + k, i) = BLASFEO_DVECEL(a->d + k, i);
                BLASFEO_DVECEL(b->d_mask + k, i) = BLASFEO_DVECEL(a->d_mask + k, i);
                BLASFEO_DVECEL(b->m + k, i) = BLASFEO_DVECEL(a->m + k, i);
            }
            for (int j = 0; j < nv; j++) BLASFEO_DMATEL(b->RSQrq + k, nv, j) = BLASFEO_DVECEL(a->rqz + k, j);
            for (int j = 0; j < nx1; j++) BLASFEO_DMATEL(b->BAbt + k, nv, j) = BLASFEO_DVECEL(a->b + k, j);
        }
    }
}

static void pc_copy_seed(ocp_qp_seed *a, ocp_qp_seed *b)
{
    const ocp_qp_dims *d = a->dim;
    for (int k = 0; k <= d->N; k++)
    {
        const int nct = 2 * (d->nb[k] + d->ng[k] + d->ns[k]), nx1 = k < d->N ? d->nx[k + 1] : 0;
        for (int i = 0; i < d->nu[k] + d->nx[k] + 2 * d->ns[k]; i++) BLASFEO_DVECEL(b->seed_g + k, i) = BLASFEO_DVECEL(a->seed_g + k, i);
        for (int i = 0; i < nx1; i++) BLASFEO_DVECEL(b->seed_b + k, i) = BLASFEO_DVECEL(a->seed_b + k, i);
        for (int i = 0; i < nct; i++) BLASFEO_DVECEL(b->seed_d + k, i) = BLASFEO_DVECEL(a->seed_d + k, i);
        for (int i = 0; i < nct; i++) BLASFEO_DVECEL(b->seed_m + k, i) = BLASFEO_DVECEL(a->seed_m + k, i);
    }
}


/* ------------------------------------------------------------------ x0 elimination on the host (:542, :683)
 *
 * d_ocp_qp_reduce_eq_dof / d_ocp_qp_restore_eq_dof restated for what acados flags: states of stage 0 fixed by equality-flagged
 * bounds (idxe, the [bxe] part).  With F the fixed states, xbar their values (the lower bound) and [u; x_keep] the variables left:
 *     b_0 += A_0[:, F] xbar,   [r_0; q_0]_keep += H_0[keep, F] xbar,   lg_0 -= C_0[:, F] xbar,  ug_0 likewise,   rows of F leave.
 * Stages 1..N of the reduced containers ALIAS the caller's member structs (same dims, read on every call).  Element access through
 * BLASFEO_DMATEL / BLASFEO_DVECEL: one stage, a few hundred elements. */
#define RSQ_SYM(in_, i_, j_) ((i_) >= (j_) ? BLASFEO_DMATEL((in_)->RSQrq, (i_), (j_)) : BLASFEO_DMATEL((in_)->RSQrq, (j_), (i_)))

static void pc_alias_in_tail(const ocp_qp_in *in, ocp_qp_in *r)
{
    const int N = in->dim->N;
    for (int k = 1; k <= N; k++)
    {
        if (k < N) { r->BAbt[k] = in->BAbt[k]; r->b[k] = in->b[k]; }
        r->RSQrq[k] = in->RSQrq[k]; r->DCt[k] = in->DCt[k]; r->rqz[k] = in->rqz[k]; r->d[k] = in->d[k]; r->d_mask[k] = in->d_mask[k];
        r->m[k] = in->m[k]; r->Z[k] = in->Z[k];
        r->idxb[k] = in->idxb[k]; r->idxs_rev[k] = in->idxs_rev[k]; r->idxe[k] = in->idxe[k];
        r->diag_H_flag[k] = in->diag_H_flag[k];
    }
}

/* the reduced original QP (or qp_in itself where nothing is eliminated); fills the maps the other helpers use */
static ocp_qp_in *pc_reduce_in(ocp_qp_gpu_pcond_memory *mem, ocp_qp_in *in)
{
    if (!mem->dims->reduced) { mem->nF = 0; mem->src_in = in; return in; }
    ocp_qp_in *r = mem->red_in;
    const ocp_qp_dims *d = in->dim, *rd = r->dim;
    const int nu = d->nu[0], nx = d->nx[0], nv = nu + nx, nb = d->nb[0], ng = d->ng[0], ns = d->ns[0], nx1 = d->N > 0 ? d->nx[1] : 0;
    const int nvr = rd->nu[0] + rd->nx[0], nbr = rd->nb[0];
    pc_alias_in_tail(in, r);
    /* who leaves */
    for (int i = 0; i < nv; i++) { mem->map_var[i] = 0; mem->xbar[i] = 0.0; }
    for (int i = 0; i < nb; i++) mem->map_row[i] = 0;
    mem->nF = d->nbxe[0];
    for (int e = 0; e < mem->nF; e++)
    {
        const int ib = in->idxe[0][d->nbue[0] + e], iv = in->idxb[0][ib];
        if (iv < nu || in->idxs_rev[0][ib] >= 0)
        {
            printf("\nerror: ocp_qp_gpu_pcond: reduce_eq_dof: an equality-flagged STATE bound without slack is expected at stage 0 (row %d)\n", ib);
            exit(1);
        }
        mem->f_ib[e] = ib; mem->f_iv[e] = iv;
        mem->map_var[iv] = -1; mem->map_row[ib] = -1;
        mem->xbar[iv] = BLASFEO_DVECEL(in->d, ib); /* the lower bound (natural sign) = the value */
    }
    for (int i = 0, p = 0; i < nv; i++) if (mem->map_var[i] == 0) mem->map_var[i] = p++;
    for (int i = 0, p = 0; i < nb; i++) if (mem->map_row[i] == 0) mem->map_row[i] = p++;
    /* dynamics: rows of [B'; A'] of the variables left; b' = b + A[:, F] xbar (vector AND last row) */
    for (int c = 0; c < nx1; c++)
    {
        double bb = BLASFEO_DVECEL(in->b, c);
        for (int i = 0; i < nv; i++)
        {
            if (mem->map_var[i] >= 0) BLASFEO_DMATEL(r->BAbt, mem->map_var[i], c) = BLASFEO_DMATEL(in->BAbt, i, c);
            else bb += BLASFEO_DMATEL(in->BAbt, i, c) * mem->xbar[i];
        }
        BLASFEO_DVECEL(r->b, c) = bb;
        BLASFEO_DMATEL(r->BAbt, nvr, c) = bb;
    }
    /* Hessian: principal block (lower triangle); gradient += H[keep, F] xbar (vector AND last row); zl, zu */
    for (int i = 0; i < nv; i++)
    {
        const int mi = mem->map_var[i];
        if (mi < 0) continue;
        double gg = BLASFEO_DVECEL(in->rqz, i);
        for (int j = 0; j < nv; j++)
        {
            const int mj = mem->map_var[j];
            if (mj < 0) gg += RSQ_SYM(in, i, j) * mem->xbar[j];
            else if (mj <= mi) BLASFEO_DMATEL(r->RSQrq, mi, mj) = RSQ_SYM(in, i, j);
        }
        BLASFEO_DVECEL(r->rqz, mi) = gg;
        BLASFEO_DMATEL(r->RSQrq, nvr, mi) = gg;
    }
    for (int q = 0; q < 2 * ns; q++)
    {
        BLASFEO_DVECEL(r->rqz, nvr + q) = BLASFEO_DVECEL(in->rqz, nv + q);
        BLASFEO_DVECEL(r->Z, q) = BLASFEO_DVECEL(in->Z, q);
    }
    /* general rows: [D'; C'] of the variables left; bounds shifted by C[:, F] xbar.  d = [lb; lg; -ub; -ug; ls; us] */
    for (int g = 0; g < ng; g++)
    {
        double cc = 0.0;
        for (int i = 0; i < nv; i++)
        {
            if (mem->map_var[i] >= 0) BLASFEO_DMATEL(r->DCt, mem->map_var[i], g) = BLASFEO_DMATEL(in->DCt, i, g);
            else cc += BLASFEO_DMATEL(in->DCt, i, g) * mem->xbar[i];
        }
        BLASFEO_DVECEL(r->d, nbr + g) = BLASFEO_DVECEL(in->d, nb + g) - cc;
        BLASFEO_DVECEL(r->d, 2 * nbr + ng + g) = BLASFEO_DVECEL(in->d, 2 * nb + ng + g) + cc; /* (-ug)' = -ug + cc */
        BLASFEO_DVECEL(r->d_mask, nbr + g) = BLASFEO_DVECEL(in->d_mask, nb + g);
        BLASFEO_DVECEL(r->d_mask, 2 * nbr + ng + g) = BLASFEO_DVECEL(in->d_mask, 2 * nb + ng + g);
        BLASFEO_DVECEL(r->m, nbr + g) = BLASFEO_DVECEL(in->m, nb + g);
        BLASFEO_DVECEL(r->m, 2 * nbr + ng + g) = BLASFEO_DVECEL(in->m, 2 * nb + ng + g);
        r->idxs_rev[0][nbr + g] = in->idxs_rev[0][nb + g];
    }
    /* box rows left */
    for (int ib = 0; ib < nb; ib++)
    {
        const int mr = mem->map_row[ib];
        if (mr < 0) continue;
        BLASFEO_DVECEL(r->d, mr) = BLASFEO_DVECEL(in->d, ib);
        BLASFEO_DVECEL(r->d, nbr + ng + mr) = BLASFEO_DVECEL(in->d, nb + ng + ib);
        BLASFEO_DVECEL(r->d_mask, mr) = BLASFEO_DVECEL(in->d_mask, ib);
        BLASFEO_DVECEL(r->d_mask, nbr + ng + mr) = BLASFEO_DVECEL(in->d_mask, nb + ng + ib);
        BLASFEO_DVECEL(r->m, mr) = BLASFEO_DVECEL(in->m, ib);
        BLASFEO_DVECEL(r->m, nbr + ng + mr) = BLASFEO_DVECEL(in->m, nb + ng + ib);
        r->idxb[0][mr] = mem->map_var[in->idxb[0][ib]];
        r->idxs_rev[0][mr] = in->idxs_rev[0][ib];
    }
    for (int q = 0; q < 2 * ns; q++)
    {
        BLASFEO_DVECEL(r->d, 2 * nbr + 2 * ng + q) = BLASFEO_DVECEL(in->d, 2 * nb + 2 * ng + q);
        BLASFEO_DVECEL(r->d_mask, 2 * nbr + 2 * ng + q) = BLASFEO_DVECEL(in->d_mask, 2 * nb + 2 * ng + q);
        BLASFEO_DVECEL(r->m, 2 * nbr + 2 * ng + q) = BLASFEO_DVECEL(in->m, 2 * nb + 2 * ng + q);
    }
    r->diag_H_flag[0] = 0;
    mem->src_in = r;
    return r;
}

/* stages 1..N (and pi of stage 0) of a reduced solution container alias the caller's */
static ocp_qp_out *pc_alias_out(ocp_qp_out *r, ocp_qp_out *out)
{
    const int N = out->dim->N;
    for (int k = 1; k <= N; k++) { r->ux[k] = out->ux[k]; r->lam[k] = out->lam[k]; r->t[k] = out->t[k]; if (k < N) r->pi[k] = out->pi[k]; }
    if (N > 0) r->pi[0] = out->pi[0];
    return r;
}

/* an iterate of the original QP in the variables / rows of the reduced one (warm start: condense_qp_out) */
static ocp_qp_out *pc_reduce_out(ocp_qp_gpu_pcond_memory *mem, ocp_qp_out *out)
{
    if (!mem->dims->reduced) return out;
    ocp_qp_out *r = pc_alias_out(mem->red_out, out);
    const ocp_qp_dims *d = out->dim, *rd = r->dim;
    const int nv = d->nu[0] + d->nx[0], nb = d->nb[0], ng = d->ng[0], ns = d->ns[0], nvr = rd->nu[0] + rd->nx[0], nbr = rd->nb[0];
    for (int i = 0; i < nv; i++) if (mem->map_var[i] >= 0) BLASFEO_DVECEL(r->ux, mem->map_var[i]) = BLASFEO_DVECEL(out->ux, i);
    for (int q = 0; q < 2 * ns; q++) BLASFEO_DVECEL(r->ux, nvr + q) = BLASFEO_DVECEL(out->ux, nv + q);
    for (int side = 0; side < 2; side++)
    {
        const int o = side * (nb + ng), ro = side * (nbr + ng);
        for (int ib = 0; ib < nb; ib++)
            if (mem->map_row[ib] >= 0)
            {
                BLASFEO_DVECEL(r->lam, ro + mem->map_row[ib]) = BLASFEO_DVECEL(out->lam, o + ib);
                BLASFEO_DVECEL(r->t, ro + mem->map_row[ib]) = BLASFEO_DVECEL(out->t, o + ib);
            }
        for (int g = 0; g < ng; g++)
        {
            BLASFEO_DVECEL(r->lam, ro + nbr + g) = BLASFEO_DVECEL(out->lam, o + nb + g);
            BLASFEO_DVECEL(r->t, ro + nbr + g) = BLASFEO_DVECEL(out->t, o + nb + g);
        }
    }
    for (int q = 0; q < 2 * ns; q++)
    {
        BLASFEO_DVECEL(r->lam, 2 * nbr + 2 * ng + q) = BLASFEO_DVECEL(out->lam, 2 * nb + 2 * ng + q);
        BLASFEO_DVECEL(r->t, 2 * nbr + 2 * ng + q) = BLASFEO_DVECEL(out->t, 2 * nb + 2 * ng + q);
    }
    return r;
}

/* d_ocp_qp_restore_eq_dof (:683): stage 0 of the reduced solution `r` back into `out` (stages 1..N are already there: aliased).
 * x[F] = xbar (a sensitivity: the seed of the value); the multipliers of the rows of F from stationarity of x[F] in the ORIGINAL QP,
 *     a = g_F + H[F, :] w_0 + A[:, F]' pi_1 - C[:, F]'(lam_lg - lam_ug),   lam_lb = max(a, 0), lam_ub = max(-a, 0), t = 0;
 * for a sensitivity (vec = the seed's seed_g, fval = the seed of the value) the same expression is the derivative of `a`, assigned
 * to the side the multiplier of the SOLUTION lives on (f_sign, kept from the last restore of a solution). */
static void pc_restore_out(ocp_qp_gpu_pcond_memory *mem, const ocp_qp_in *in, const struct blasfeo_dvec *vec, const double *fval, int is_sens,
                           ocp_qp_out *r, ocp_qp_out *out)
{
    const ocp_qp_dims *d = out->dim, *rd = r->dim;
    const int nv = d->nu[0] + d->nx[0], nb = d->nb[0], ng = d->ng[0], ns = d->ns[0], nx1 = d->N > 0 ? d->nx[1] : 0;
    const int nvr = rd->nu[0] + rd->nx[0], nbr = rd->nb[0];
    for (int i = 0; i < nv; i++) BLASFEO_DVECEL(out->ux, i) = mem->map_var[i] >= 0 ? BLASFEO_DVECEL(r->ux, mem->map_var[i]) : fval[i];
    for (int q = 0; q < 2 * ns; q++) BLASFEO_DVECEL(out->ux, nv + q) = BLASFEO_DVECEL(r->ux, nvr + q);
    for (int side = 0; side < 2; side++)
    {
        const int o = side * (nb + ng), ro = side * (nbr + ng);
        for (int ib = 0; ib < nb; ib++)
            if (mem->map_row[ib] >= 0)
            {
                BLASFEO_DVECEL(out->lam, o + ib) = BLASFEO_DVECEL(r->lam, ro + mem->map_row[ib]);
                BLASFEO_DVECEL(out->t, o + ib) = BLASFEO_DVECEL(r->t, ro + mem->map_row[ib]);
            }
        for (int g = 0; g < ng; g++)
        {
            BLASFEO_DVECEL(out->lam, o + nb + g) = BLASFEO_DVECEL(r->lam, ro + nbr + g);
            BLASFEO_DVECEL(out->t, o + nb + g) = BLASFEO_DVECEL(r->t, ro + nbr + g);
        }
    }
    for (int q = 0; q < 2 * ns; q++)
    {
        BLASFEO_DVECEL(out->lam, 2 * nb + 2 * ng + q) = BLASFEO_DVECEL(r->lam, 2 * nbr + 2 * ng + q);
        BLASFEO_DVECEL(out->t, 2 * nb + 2 * ng + q) = BLASFEO_DVECEL(r->t, 2 * nbr + 2 * ng + q);
    }
    for (int e = 0; e < mem->nF; e++)
    {
        const int ib = mem->f_ib[e], iv = mem->f_iv[e];
        double a = BLASFEO_DVECEL(vec, iv);
        for (int j = 0; j < nv; j++) a += RSQ_SYM(in, iv, j) * BLASFEO_DVECEL(out->ux, j);
        for (int c = 0; c < nx1; c++) a += BLASFEO_DMATEL(in->BAbt, iv, c) * BLASFEO_DVECEL(out->pi, c);
        for (int g = 0; g < ng; g++)
            a -= BLASFEO_DMATEL(in->DCt, iv, g) * (BLASFEO_DVECEL(out->lam, nb + g) - BLASFEO_DVECEL(out->lam, 2 * nb + ng + g));
        if (!is_sens) mem->f_sign[e] = a > 0.0;
        const int pos = is_sens ? mem->f_sign[e] : a > 0.0;
        BLASFEO_DVECEL(out->lam, ib) = pos ? a : 0.0;
        BLASFEO_DVECEL(out->lam, nb + ng + ib) = pos ? 0.0 : -a;
        BLASFEO_DVECEL(out->t, ib) = 0.0;
        BLASFEO_DVECEL(out->t, nb + ng + ib) = 0.0;
    }
}

/* d_ocp_qp_reduce_eq_dof_seed: the seeds of a sensitivity solve in the reduced QP.  The seed of a fixed state's VALUE is the seed of its
 * lower bound (seed_d is laid out like d; acados seeds both sides of the x0 rows, ocp_nlp_common.c:4057-4064); it enters like xbar. */
static ocp_qp_seed *pc_reduce_seed(ocp_qp_gpu_pcond_memory *mem, const ocp_qp_in *in, ocp_qp_seed *seed)
{
    if (!mem->dims->reduced) return seed;
    ocp_qp_seed *r = mem->red_seed;
    const ocp_qp_dims *d = in->dim, *rd = r->dim;
    const int N = d->N, nv = d->nu[0] + d->nx[0], nb = d->nb[0], ng = d->ng[0], ns = d->ns[0], nx1 = N > 0 ? d->nx[1] : 0;
    const int nvr = rd->nu[0] + rd->nx[0], nbr = rd->nb[0];
    for (int k = 1; k <= N; k++) { r->seed_g[k] = seed->seed_g[k]; r->seed_d[k] = seed->seed_d[k]; r->seed_m[k] = seed->seed_m[k]; if (k < N) r->seed_b[k] = seed->seed_b[k]; }
    double *dx = mem->dxbar;
    for (int i = 0; i < nv; i++) dx[i] = 0.0;
    for (int e = 0; e < mem->nF; e++) dx[mem->f_iv[e]] = BLASFEO_DVECEL(seed->seed_d, mem->f_ib[e]);
    for (int c = 0; c < nx1; c++)
    {
        double bb = BLASFEO_DVECEL(seed->seed_b, c);
        for (int e = 0; e < mem->nF; e++) bb += BLASFEO_DMATEL(in->BAbt, mem->f_iv[e], c) * dx[mem->f_iv[e]];
        BLASFEO_DVECEL(r->seed_b, c) = bb;
    }
    for (int i = 0; i < nv; i++)
    {
        if (mem->map_var[i] < 0) continue;
        double gg = BLASFEO_DVECEL(seed->seed_g, i);
        for (int e = 0; e < mem->nF; e++) gg += RSQ_SYM(in, i, mem->f_iv[e]) * dx[mem->f_iv[e]];
        BLASFEO_DVECEL(r->seed_g, mem->map_var[i]) = gg;
    }
    for (int q = 0; q < 2 * ns; q++) BLASFEO_DVECEL(r->seed_g, nvr + q) = BLASFEO_DVECEL(seed->seed_g, nv + q);
    for (int g = 0; g < ng; g++)
    {
        double cc = 0.0;
        for (int e = 0; e < mem->nF; e++) cc += BLASFEO_DMATEL(in->DCt, mem->f_iv[e], g) * dx[mem->f_iv[e]];
        BLASFEO_DVECEL(r->seed_d, nbr + g) = BLASFEO_DVECEL(seed->seed_d, nb + g) - cc;
        BLASFEO_DVECEL(r->seed_d, 2 * nbr + ng + g) = BLASFEO_DVECEL(seed->seed_d, 2 * nb + ng + g) + cc;
        BLASFEO_DVECEL(r->seed_m, nbr + g) = BLASFEO_DVECEL(seed->seed_m, nb + g);
        BLASFEO_DVECEL(r->seed_m, 2 * nbr + ng + g) = BLASFEO_DVECEL(seed->seed_m, 2 * nb + ng + g);
    }
    for (int ib = 0; ib < nb; ib++)
    {
        const int mr = mem->map_row[ib];
        if (mr < 0) continue;
        BLASFEO_DVECEL(r->seed_d, mr) = BLASFEO_DVECEL(seed->seed_d, ib);
        BLASFEO_DVECEL(r->seed_d, nbr + ng + mr) = BLASFEO_DVECEL(seed->seed_d, nb + ng + ib);
        BLASFEO_DVECEL(r->seed_m, mr) = BLASFEO_DVECEL(seed->seed_m, ib);
        BLASFEO_DVECEL(r->seed_m, nbr + ng + mr) = BLASFEO_DVECEL(seed->seed_m, nb + ng + ib);
    }
    for (int q = 0; q < 2 * ns; q++)
    {
        BLASFEO_DVECEL(r->seed_d, 2 * nbr + 2 * ng + q) = BLASFEO_DVECEL(seed->seed_d, 2 * nb + 2 * ng + q);
        BLASFEO_DVECEL(r->seed_m, 2 * nbr + 2 * ng + q) = BLASFEO_DVECEL(seed->seed_m, 2 * nb + 2 * ng + q);
    }
    return r;
}

/* ------------------------------------------------------------------ the condensing slots */

static int pc_condense_any(void *qp_in_, void *xin_, void *opts_, void *mem_, int what)
{
    ocp_qp_in *qp_in = (ocp_qp_in *) qp_in_, *x = (ocp_qp_in *) xin_;
    ocp_qp_gpu_pcond_opts *opts = (ocp_qp_gpu_pcond_opts *) opts_;
    ocp_qp_gpu_pcond_memory *mem = (ocp_qp_gpu_pcond_memory *) mem_;
    const double t0 = pc_now_s();
    mem->ptr_qp_in = qp_in;
    int rc = ACADOS_SUCCESS;
    ocp_qp_in *src = pc_reduce_in(mem, qp_in); /* x0 eliminated (:542), or qp_in itself */
    if (!mem->dims->condensed) pc_copy_qp_in(src, x, what, mem->blob);
    else
    {
        ocp_qp_gpu_batch *b = pc_load(mem, opts, src), *c = NULL;
        if (b)
        {
            if (what == 3) c = ocp_qp_gpu_batch_condense(b);
            else if (what == 1) { if (ocp_qp_gpu_batch_condense_lhs(b) == 0) c = ocp_qp_gpu_batch_condensed(b); }
            else c = ocp_qp_gpu_batch_condense_rhs(b);
        }
        if (!c || pc_read_child(mem, c, x, what) != 0) rc = ACADOS_QP_FAILURE;
    }
    if (what == 2) mem->time_qp_xcond += pc_now_s() - t0; /* :602-630 adds the rhs part to the lhs part's time */
    else mem->time_qp_xcond = pc_now_s() - t0;
    return rc;
}

static int pc_condensing(void *qp_in, void *xin, void *opts, void *mem, void *work) { return pc_condense_any(qp_in, xin, opts, mem, 3); }
static int pc_condense_lhs(void *qp_in, void *xin, void *opts, void *mem, void *work)
{
    if (((ocp_qp_gpu_pcond_opts *) opts)->batch_owned)
    {
        /* lock-step batch: the preparation half of ALL capsules runs as one device call right behind the per-capsule preparation steps */
        ((ocp_qp_gpu_pcond_memory *) mem)->time_qp_xcond = 0.0;
        ((ocp_qp_gpu_pcond_memory *) mem)->ptr_qp_in = (ocp_qp_in *) qp_in;
        return ACADOS_SUCCESS;
    }
    return pc_condense_any(qp_in, xin, opts, mem, 1);
}
static int pc_condense_rhs(void *qp_in, void *xin, void *opts, void *mem, void *work) { return pc_condense_any(qp_in, xin, opts, mem, 2); }

/* :559-571 */
static int pc_condense_qp_out(void *qp_in_, void *xin_, void *qp_out_, void *xout_, void *opts_, void *mem_, void *work)
{
    ocp_qp_out *out = (ocp_qp_out *) qp_out_, *xo = (ocp_qp_out *) xout_;
    ocp_qp_gpu_pcond_memory *mem = (ocp_qp_gpu_pcond_memory *) mem_;
    if (mem->dims->reduced && mem->ptr_qp_in != (ocp_qp_in *) qp_in_) pc_reduce_in(mem, (ocp_qp_in *) qp_in_); /* (the maps of stage 0) */
    out = pc_reduce_out(mem, out);
    if (!mem->dims->condensed) { ocp_qp_out_copy(out, xo); return ACADOS_SUCCESS; }
    ocp_qp_gpu_batch *b = mem->par.batch, *c = b ? ocp_qp_gpu_batch_condensed(b) : NULL;
    if (!c || pc_child_layout(mem, c) != 0) return ACADOS_QP_FAILURE; /* condensing has to run first, as in ocp_qp_xcond_solve */
    unpack_qp_out_full(&mem->par, out, mem->blob);
    if (ocp_qp_gpu_batch_set_bulk_out(b, mem->blob, 0) != 0 || ocp_qp_gpu_batch_condense_sol(b) != 0) return ACADOS_QP_FAILURE;
    if (ocp_qp_gpu_batch_get_bulk(c, mem->blob, 0) != 0) return ACADOS_QP_FAILURE;
    pack_qp_out(&mem->chd, mem->blob, xo);
    return ACADOS_SUCCESS;
}

/* the parent batch's vector fields back to the QP's own (they hold the seeds between the two seed slots) */
static int pc_restore_qp_vectors(ocp_qp_gpu_pcond_memory *mem)
{
    mem->seeds_resident = 0;
    if (!mem->par.batch || !mem->src_in) return -1;
    memset(mem->blob, 0, sizeof(double) * (size_t) mem->par.L_in);
    unpack_qp_in(&mem->par, mem->src_in, mem->blob);
    return ocp_qp_gpu_batch_set_bulk(mem->par.batch, mem->blob, 0);
}

static int pc_expand_any(void *xout_, void *qp_out_, void *mem_, int seeds)
{
    ocp_qp_out *xo = (ocp_qp_out *) xout_, *out = (ocp_qp_out *) qp_out_;
    ocp_qp_gpu_pcond_memory *mem = (ocp_qp_gpu_pcond_memory *) mem_;
    const double t0 = pc_now_s();
    int rc = ACADOS_SUCCESS;
    /* x0 eliminated: the expansion lands in the reduced container (stages 1..N of it ARE the caller's), stage 0 is restored below */
    ocp_qp_out *full = out;
    if (mem->dims->reduced) out = pc_alias_out(seeds ? mem->red_sens : mem->red_out, full);
    if (!mem->dims->condensed) ocp_qp_out_copy(xo, out);
    else
    {
        ocp_qp_gpu_batch *b = mem->par.batch, *c = b ? ocp_qp_gpu_batch_condensed(b) : NULL;
        /* a plain expansion while the seeds of an unfinished seed pair are still resident runs on the QP's own vectors */
        if (!seeds && mem->seeds_resident && pc_restore_qp_vectors(mem) != 0) return ACADOS_QP_FAILURE;
        if (!c || pc_child_layout(mem, c) != 0)
        {
            if (mem->seeds_resident) pc_restore_qp_vectors(mem);
            return ACADOS_QP_FAILURE;
        }
        /* seeds: the expansion kernel runs on the QP whose VECTORS are the seeds -- what pc_condense_rhs_seed left on the device */
        if (seeds && (!mem->ptr_qp_in || !mem->ptr_seed || !mem->seeds_resident))
        {
            printf("\nerror: partial condensing: expand_sol_seed before condense_rhs_seed\n");
            return ACADOS_QP_FAILURE;
        }
        unpack_qp_out_full(&mem->chd, xo, mem->blob);
        if (ocp_qp_gpu_batch_set_bulk_out(c, mem->blob, 0) != 0 || ocp_qp_gpu_batch_expand(b) != 0
            || ocp_qp_gpu_batch_get_bulk(b, mem->blob, 0) != 0) rc = ACADOS_QP_FAILURE;
        else pack_qp_out(&mem->par, mem->blob, out);
        if (seeds && pc_restore_qp_vectors(mem) != 0) rc = ACADOS_QP_FAILURE; /* the QP's own vectors back */
    }
    if (mem->dims->reduced && rc == ACADOS_SUCCESS)
    {
        if (seeds) pc_restore_out(mem, mem->ptr_qp_in, mem->ptr_seed->seed_g, mem->dxbar, 1, out, full);
        else pc_restore_out(mem, mem->ptr_qp_in, mem->ptr_qp_in->rqz, mem->xbar, 0, out, full);
    }
    out = full;
    if (!seeds && out->misc) ((qp_info *) out->misc)->t_computed = 1; /* t comes from the expansion kernel, every row */
    mem->time_qp_xcond += pc_now_s() - t0;
    return rc;
}

/* :664-689 */
static int pc_expansion(void *xout, void *qp_out, void *opts, void *mem, void *work) { return pc_expand_any(xout, qp_out, mem, 0); }

/*
 * Seeds through the condensing (N2 < N).  Vector condensing is LINEAR and homogeneous in the vector data (b, r, q, zl, zu,
 * bounds): the condensed seed is the vector condensing (condense_rhs) of a QP with the SAME matrices whose vectors are the
 * seeds, and the expansion of the condensed sensitivities is the expansion kernel run on that QP.  The parent batch holds the
 * original QP: its vector fields are the seeds from condense_rhs_seed until expand_sol_seed has run, which restores them.
 * (HPIPM: d_part_cond_qp_cond_seed / d_part_cond_qp_expand_sol_seed, :634-662, 691-717.)
 */
static int pc_condense_rhs_seed(void *qp_in_, void *seed_, void *xseed_, void *opts_, void *mem_, void *work)
{
    ocp_qp_in *qp_in = (ocp_qp_in *) qp_in_;
    ocp_qp_seed *seed = (ocp_qp_seed *) seed_, *xs = (ocp_qp_seed *) xseed_;
    ocp_qp_gpu_pcond_opts *opts = (ocp_qp_gpu_pcond_opts *) opts_;
    ocp_qp_gpu_pcond_memory *mem = (ocp_qp_gpu_pcond_memory *) mem_;
    const double t0 = pc_now_s();
    mem->ptr_qp_in = qp_in;
    mem->ptr_seed = seed;
    ocp_qp_in *src = pc_reduce_in(mem, qp_in);          /* (the matrices of the reduced QP; its vectors are replaced by the seeds below) */
    ocp_qp_seed *rs = pc_reduce_seed(mem, qp_in, seed); /* d_ocp_qp_reduce_eq_dof_seed */
    if (!mem->dims->condensed) { pc_copy_seed(rs, xs); mem->time_qp_xcond += pc_now_s() - t0; return ACADOS_SUCCESS; }
    /* seed_m (a seed on the complementarity rhs) has no counterpart in the vector condensing: acados leaves it zero
     * (d_ocp_qp_seed_set_zero, then seed_g / seed_b / seed_d: ocp_nlp_common.c:4057-4081); anything else is refused, not dropped */
    for (int k = 0; k <= qp_in->dim->N; k++)
        for (int i = 0; i < 2 * (qp_in->dim->nb[k] + qp_in->dim->ng[k] + qp_in->dim->ns[k]); i++)
            if (BLASFEO_DVECEL(seed->seed_m + k, i) != 0.0)
            {
                printf("\nerror: ocp_qp_gpu_pcond: condense_rhs_seed with a non-zero seed_m (stage %d) is not supported\n", k);
                return ACADOS_QP_FAILURE;
            }
    /* a SHELL of qp_in whose vector members are the seed's: seed_g = [r; q; zl; zu] is laid out like rqz, seed_b like b, seed_d
     * like d (upper halves negated, ocp_nlp_common.c:4078-4081) -- the segment tables of the input blob read them as they are;
     * matrices, masks and index sets stay qp_in's */
    ocp_qp_in shell = *src;
    shell.b = rs->seed_b; shell.rqz = rs->seed_g; shell.d = rs->seed_d;
    ocp_qp_gpu_batch *b = pc_load(mem, opts, &shell);
    ocp_qp_gpu_batch *c = b ? ocp_qp_gpu_batch_condense(b) : NULL;
    if (!c || pc_child_layout(mem, c) != 0 || ocp_qp_gpu_batch_get_bulk_in(c, mem->blob, 0) != 0)
    {
        /* pc_load has replaced the parent batch's vectors with the seeds: a later expansion must not run on them */
        if (b) pc_restore_qp_vectors(mem);
        return ACADOS_QP_FAILURE;
    }
    mem->seeds_resident = 1; /* until expand_sol_seed has run (it restores the QP's own vectors) */
    /* the condensed QP's vector fields ARE the condensed seeds: [r q zl zu] -> seed_g, b -> seed_b, bounds -> seed_d (upper
     * halves negated like d, ocp_nlp_common.c:4078-4081) */
    {
        struct blasfeo_dvec *vecs[8] = {NULL, NULL, NULL, xs->seed_b, xs->seed_g, xs->seed_d, NULL, NULL};
        const gpu_layout *l = &mem->chd;
        for (int s = 0; s < l->n_in; s++)
        {
            const gpu_seg *g = l->seg_in + s;
            if (g->kind != SEG_VEC || !vecs[g->src]) continue;
            double *p = mem->blob + g->off;
            if (g->neg) for (int e = 0; e < g->len; e++) p[e] = -p[e];
            blasfeo_pack_dvec(g->m, p, 1, vecs[g->src] + g->k, g->ai);
        }
        const ocp_qp_dims *xd = xs->dim;
        for (int k = 0; k <= xd->N; k++)
            for (int i = 0; i < 2 * (xd->nb[k] + xd->ng[k] + xd->ns[k]); i++) BLASFEO_DVECEL(xs->seed_m + k, i) = 0.0;
    }
    mem->time_qp_xcond += pc_now_s() - t0;
    return ACADOS_SUCCESS;
}

/* :691-717 */
static int pc_expand_sol_seed(void *xout, void *qp_out, void *opts, void *mem, void *work) { return pc_expand_any(xout, qp_out, mem, 1); }

/* ocp_qp_partial_condensing.c:720-750 */
void ocp_qp_gpu_pcond_acados_config_initialize_default(void *config_)
{
    ocp_qp_xcond_config *config = (ocp_qp_xcond_config *) config_;
    config->dims_calculate_size = &pc_dims_calculate_size;
    config->dims_assign = &pc_dims_assign;
    config->dims_set = &pc_dims_set;
    config->dims_get = &pc_dims_get;
    config->opts_calculate_size = &pc_opts_calculate_size;
    config->opts_assign = &pc_opts_assign;
    config->opts_initialize_default = &pc_opts_initialize_default;
    config->opts_update = &pc_opts_update;
    config->opts_set = &pc_opts_set;
    config->memory_calculate_size = &pc_memory_calculate_size;
    config->memory_assign = &pc_memory_assign;
    config->memory_get = &pc_memory_get;
    config->workspace_calculate_size = &pc_workspace_calculate_size;
    config->condensing = &pc_condensing;
    config->condense_rhs = &pc_condense_rhs;
    config->condense_rhs_seed = &pc_condense_rhs_seed;
    config->condense_lhs = &pc_condense_lhs;
    config->condense_qp_out = &pc_condense_qp_out;
    config->expansion = &pc_expansion;
    config->expand_sol_seed = &pc_expand_sol_seed;
}

/* does this outer config carry the device condensing module?  (ocp_qp_xcond_solver_terminate of the patched tree asks) */
int ocp_qp_gpu_pcond_acados_is_module(const void *xcond_config_)
{
    const ocp_qp_xcond_config *c = (const ocp_qp_xcond_config *) xcond_config_;
    return c && c->condensing == &pc_condensing;
}

/* ------------------------------------------------------------------ the batch route (fused on the device)
 *
 * n capsules, each with the 22-slot solver the reference's ocp_qp_xcond_solver.c built around { this module, the GPU QP solver }:
 *     opts  = the capsules' (shared) ocp_qp_xcond_solver_opts   (xcond_opts: this file's, qp_solver_opts: ocp_qp_gpu_ipm.c's)
 *     mem[i] = capsule i's ocp_qp_xcond_solver_memory           (solver_memory: ocp_qp_gpu_ipm.c's)
 * The ORIGINAL QPs qp_in[i] go to the QP solver's batch entry with cond_N / cond_block_size attached: one device batch per
 * structure class, condensing (km_pcond ...), IPM sweeps and expansion back to back on the device -- where the per-capsule path
 * through ocp_qp_xcond_solve (ocp_qp_xcond_solver.c:529-587) returns to the host twice per QP.  Afterwards every capsule
 * answers memory_get("iter" / "status" / "time_qp_solver_call") and qp_out[i]->misc as after its own evaluate.
 * Replaces the loop of acados_solver.in.c:3222-3243 at the QP level (integration/acados.patch adds the caller).
 */
/* which: 0 evaluate (condense + solve + expand), 1 RTI preparation (condense_lhs: :591-620), 2 RTI feedback (condense_rhs_and_solve: :623-669) */
static int xcond_batch_call(void *config_, ocp_qp_xcond_solver_dims *dims, int n, ocp_qp_in **qp_in, ocp_qp_out **qp_out, void *opts_, void **mem_, int which)
{
    ocp_qp_xcond_solver_config *config = (ocp_qp_xcond_solver_config *) config_;
    ocp_qp_xcond_solver_opts *opts = (ocp_qp_xcond_solver_opts *) opts_;
    if (n <= 0) return ACADOS_SUCCESS;
    if (!ocp_qp_gpu_pcond_acados_is_module(config->xcond))
    {
        printf("\nerror: ocp_qp_gpu_xcond_solver_acados_evaluate_batch: the condensing module of this solver is not ocp_qp_gpu_pcond\n");
        exit(1);
    }
    ocp_qp_gpu_pcond_opts *po = (ocp_qp_gpu_pcond_opts *) opts->xcond_opts;
    ocp_qp_gpu_pcond_dims *pd = (ocp_qp_gpu_pcond_dims *) dims->xcond_dims;
    const int N = dims->orig_dims->N;
    int condN = pd->condensed && po->N2 > 0 && po->N2 < N ? po->N2 : N;
    config->qp_solver->opts_set(config->qp_solver, opts->qp_solver_opts, "cond_N", &condN);
    if (condN < N && po->block_size_was_set) config->qp_solver->opts_set(config->qp_solver, opts->qp_solver_opts, "cond_block_size", po->block_size);
    void **inner = (void **) malloc(sizeof(void *) * (size_t) n);
    if (!inner) { printf("\nerror: ocp_qp_gpu_xcond_solver_acados_evaluate_batch: out of host memory\n"); exit(1); }
    for (int i = 0; i < n; i++) inner[i] = ((ocp_qp_xcond_solver_memory *) mem_[i])->solver_memory;
    int rc;
    if (which == 1) rc = ocp_qp_gpu_ipm_acados_condense_lhs_batch(config->qp_solver, n, (void **) qp_in, opts->qp_solver_opts, inner, NULL);
    else if (which == 2) rc = ocp_qp_gpu_ipm_acados_condense_rhs_and_solve_batch(config->qp_solver, n, (void **) qp_in, (void **) qp_out, opts->qp_solver_opts, inner, NULL);
    else rc = ocp_qp_gpu_ipm_acados_evaluate_batch(config->qp_solver, n, (void **) qp_in, (void **) qp_out, opts->qp_solver_opts, inner, NULL);
    free(inner);
    /* the per-capsule path solves the CONDENSED QP it is handed: the option is this call's only */
    condN = 0;
    config->qp_solver->opts_set(config->qp_solver, opts->qp_solver_opts, "cond_N", &condN);
    return rc;
}

int ocp_qp_gpu_xcond_solver_acados_evaluate_batch(void *config_, ocp_qp_xcond_solver_dims *dims, int n, ocp_qp_in **qp_in, ocp_qp_out **qp_out,
                                                  void *opts_, void **mem_, void *work_)
{
    return xcond_batch_call(config_, dims, n, qp_in, qp_out, opts_, mem_, 0);
}

/* the two halves of an RTI step for n capsules, the matrices resident on the device in between (the batch counterparts of
 * ocp_qp_xcond_solver_condense_lhs / _condense_rhs_and_solve): the feedback half reads and sends only the vector members of every qp_in */
int ocp_qp_gpu_xcond_solver_acados_condense_lhs_batch(void *config_, ocp_qp_xcond_solver_dims *dims, int n, ocp_qp_in **qp_in, void *opts_, void **mem_,
                                                      void *work_)
{
    return xcond_batch_call(config_, dims, n, qp_in, NULL, opts_, mem_, 1);
}

int ocp_qp_gpu_xcond_solver_acados_condense_rhs_and_solve_batch(void *config_, ocp_qp_xcond_solver_dims *dims, int n, ocp_qp_in **qp_in,
                                                                ocp_qp_out **qp_out, void *opts_, void **mem_, void *work_)
{
    return xcond_batch_call(config_, dims, n, qp_in, qp_out, opts_, mem_, 2);
}
