/*
 * ocp_qp_xcond.cpp -- the two OUTER vtables of the plugin surface and the acados_c-shaped convenience layer.
 *
 *   ocp_qp_xcond_config         20 slots, acados/ocp_qp/ocp_qp_common.h:84-107, filled by
 *                               ocp_qp_gpu_pcond_config_initialize_default: the DEVICE partial condensing
 *                               (pcond_kernels.hpp) behind the slots ocp_qp_partial_condensing.c:725-750 fills with
 *                               HPIPM's CPU one -- condensing, condense_lhs, condense_rhs, condense_qp_out, expansion,
 *                               dims_get("xcond_dims"), memory_get("xcond_qp_in" | "xcond_qp_out" | "xcond_seed" |
 *                               "qp_out_info" | "time_qp_xcond").  Each slot works on the plain host containers: it is
 *                               what acados composes itself in ocp_qp_xcond_solve (ocp_qp_xcond_solver.c:529-587) and what
 *                               interfaces/acados_c/condensing_interface.c drives.
 *   ocp_qp_xcond_solver_config  22 slots + the two sub-vtables, acados/ocp_qp/ocp_qp_xcond_solver.h:81-107, filled by
 *                               ocp_qp_gpu_xcond_solver_config_initialize_default: what ocp_nlp holds.  "Replacing at this
 *                               level (own condensing + own solve) is also legal" (SURVEY 8b): evaluate is the FUSED
 *                               device path -- pack the original QP once, condense, solve, expand, all in HBM, unpack once --
 *                               with the condensing request read from the xcond opts of THIS call (no state outside
 *                               opts / mem / work).
 *
 * Differences to the reference's module that a maintainer should know (also INTEGRATION.md):
 *   - x0 is not eliminated before condensing (d_ocp_qp_reduce_eq_dof): equality-flagged bounds stay box rows of the first
 *     condensed stage (nbxe travels along) and the IPM kernels mask the fixed variables; same solution, other xcond dims.
 *   - stage dims are padded to the kernel's (NX, NU): a condensed block carries block_size * NU inputs, padded ones with
 *     unit Hessian and no coupling.  xcond dims therefore come from the device library (probe in memory_calculate_size).
 *   - a QP whose condensed stage would exceed the kernel limits (nx + bs*nu > 64, > 64 rows) is NOT condensed: xcond dims =
 *     original dims, condensing / expansion are copies (the reference's default N2 = N, :243-265), one-line notice.
 *   - condense_rhs_seed / expand_sol_seed exist for N2 = N only; with N2 < N the fused solver computes sensitivities in the
 *     full space at the expanded solution (eval_forw_sens / eval_adj_sens of the 22-slot vtable), the 20-slot module refuses.
 */
#include "acados_amd/ocp_qp_interface.h"

#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "acados_amd/ocp_qp_gpu_batch.h"
#include "ocp_qp_host_internal.h"

using gqp_host::align8;
using gqp_host::cond_request;

namespace
{

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct pcond_device
{
    gqp_host::single_batch sb;           /* the ORIGINAL QP as a one-instance device batch */
    int configured_gen;                  /* generation of `sb.batch` the condensing options have been sent to (-1: none) */
    int N2_sent;
};

void copy_dims(const ocp_qp_dims *src, ocp_qp_dims *dst)
{
    dst->N = src->N;
    int *const s[] = {src->nx, src->nu, src->nb, src->nbx, src->nbu, src->ng, src->ns, src->nbxe, src->nbue, src->nge};
    int *const d[] = {dst->nx, dst->nu, dst->nb, dst->nbx, dst->nbu, dst->ng, dst->ns, dst->nbxe, dst->nbue, dst->nge};
    for (int q = 0; q < 10; q++) memcpy(d[q], s[q], sizeof(int) * (src->N + 1));
}

void copy_qp_in(const ocp_qp_in *a, ocp_qp_in *b)
{
    const ocp_qp_dims *d = a->dim;
    for (int k = 0; k <= d->N; k++)
    {
        const int nx = d->nx[k], nu = d->nu[k], nx1 = k < d->N ? d->nx[k + 1] : 0, nb = d->nb[k], ng = d->ng[k], ns = d->ns[k];
        double *const src[] = {a->A[k], a->B[k], a->b[k], a->Q[k], a->S[k], a->R[k], a->q[k], a->r[k], a->lb[k], a->ub[k], a->lb_mask[k],
                               a->ub_mask[k], a->C[k], a->D[k], a->lg[k], a->ug[k], a->lg_mask[k], a->ug_mask[k], a->Zl[k], a->Zu[k],
                               a->zl[k], a->zu[k], a->lls[k], a->lus[k], a->lls_mask[k], a->lus_mask[k]};
        double *const dst[] = {b->A[k], b->B[k], b->b[k], b->Q[k], b->S[k], b->R[k], b->q[k], b->r[k], b->lb[k], b->ub[k], b->lb_mask[k],
                               b->ub_mask[k], b->C[k], b->D[k], b->lg[k], b->ug[k], b->lg_mask[k], b->ug_mask[k], b->Zl[k], b->Zu[k],
                               b->zl[k], b->zu[k], b->lls[k], b->lus[k], b->lls_mask[k], b->lus_mask[k]};
        const int len[] = {nx1 * nx, nx1 * nu, nx1, nx * nx, nu * nx, nu * nu, nx, nu, nb, nb, nb, nb, ng * nx, ng * nu,
                           ng, ng, ng, ng, ns, ns, ns, ns, ns, ns, ns, ns};
        for (int q = 0; q < 26; q++) memcpy(dst[q], src[q], sizeof(double) * len[q]);
        memcpy(b->idxb[k], a->idxb[k], sizeof(int) * nb);
        memcpy(b->idxs_rev[k], a->idxs_rev[k], sizeof(int) * (nb + ng));
        memcpy(b->idxe[k], a->idxe[k], sizeof(int) * d->nbxe[k]);
    }
}

void copy_qp_out(const ocp_qp_out *a, ocp_qp_out *b)
{
    const ocp_qp_dims *d = a->dim;
    for (int k = 0; k <= d->N; k++)
    {
        const int nct = 2 * (d->nb[k] + d->ng[k] + d->ns[k]);
        memcpy(b->ux[k], a->ux[k], sizeof(double) * (d->nu[k] + d->nx[k] + 2 * d->ns[k]));
        if (k < d->N) memcpy(b->pi[k], a->pi[k], sizeof(double) * d->nx[k + 1]);
        memcpy(b->lam[k], a->lam[k], sizeof(double) * nct);
        memcpy(b->t[k], a->t[k], sizeof(double) * nct);
    }
}

} // namespace

extern "C" {

/* ===================================================================== the condensing module (20 slots) */

/* ocp_qp_partial_condensing.c:52-118 */
acados_size_t ocp_qp_gpu_pcond_dims_calculate_size(void *config, int N)
{
    return sizeof(ocp_qp_partial_condensing_dims) + 2 * ocp_qp_dims_calculate_size(N) + (N + 1) * sizeof(int) + 3 * 8;
}

void *ocp_qp_gpu_pcond_dims_assign(void *config, int N, void *raw_memory)
{
    char *c = align8((char *) raw_memory);
    ocp_qp_partial_condensing_dims *dims = (ocp_qp_partial_condensing_dims *) c;
    c = align8(c + sizeof(ocp_qp_partial_condensing_dims));
    dims->orig_dims = ocp_qp_dims_assign(N, c); c = align8(c + ocp_qp_dims_calculate_size(N));
    dims->pcond_dims = ocp_qp_dims_assign(N, c); c = align8(c + ocp_qp_dims_calculate_size(N)); /* worst case N2 = N */
    dims->block_size = (int *) c;
    memset(dims->block_size, 0, sizeof(int) * (N + 1));
    dims->condensed = 0;
    dims->probe_valid = 0;
    dims->probe_key = 0;
    return dims;
}

void ocp_qp_gpu_pcond_dims_set(void *config, void *dims_, int stage, const char *field, int *value)
{
    ocp_qp_partial_condensing_dims *dims = (ocp_qp_partial_condensing_dims *) dims_;
    ocp_qp_dims_set(config, dims->orig_dims, stage, field, value);
}

/* ocp_qp_partial_condensing.c:138-155 */
void ocp_qp_gpu_pcond_dims_get(void *config, void *dims_, const char *field, void *value)
{
    ocp_qp_partial_condensing_dims *dims = (ocp_qp_partial_condensing_dims *) dims_;
    if (!strcmp(field, "xcond_dims")) *(ocp_qp_dims **) value = dims->pcond_dims;
    else
    {
        printf("\nerror: ocp_qp_partial_condensing_dims_get: field %s not available\n", field);
        exit(1);
    }
}

/* :164-246 */
acados_size_t ocp_qp_gpu_pcond_opts_calculate_size(void *dims_)
{
    ocp_qp_partial_condensing_dims *dims = (ocp_qp_partial_condensing_dims *) dims_;
    return sizeof(ocp_qp_partial_condensing_opts) + (dims->orig_dims->N + 1) * sizeof(int) + 2 * 8;
}

void *ocp_qp_gpu_pcond_opts_assign(void *dims_, void *raw_memory)
{
    ocp_qp_partial_condensing_dims *dims = (ocp_qp_partial_condensing_dims *) dims_;
    char *c = align8((char *) raw_memory);
    ocp_qp_partial_condensing_opts *opts = (ocp_qp_partial_condensing_opts *) c;
    c = align8(c + sizeof(ocp_qp_partial_condensing_opts));
    opts->block_size = (int *) c;
    memset(opts->block_size, 0, sizeof(int) * (dims->orig_dims->N + 1));
    opts->block_size_was_set = false;
    opts->full_condensing = 0;
    return opts;
}

void ocp_qp_gpu_pcond_opts_initialize_default(void *dims_, void *opts_)
{
    ocp_qp_partial_condensing_dims *dims = (ocp_qp_partial_condensing_dims *) dims_;
    ocp_qp_partial_condensing_opts *opts = (ocp_qp_partial_condensing_opts *) opts_;
    opts->N2 = dims->orig_dims->N; /* no partial condensing by default (:243-265) */
    opts->N2_bkp = opts->N2;
    opts->ric_alg = 0;
    opts->mem_qp_in = 1;
    opts->full_condensing = 0;
}

/* full condensing through the same kernels: ONE block holding every stage (FULL_CONDENSING_GPU_IPM; the reference's
 * module is ocp_qp_full_condensing.c:468-556, which hands a dense QP to a dense solver) */
void ocp_qp_gpu_fcond_opts_initialize_default(void *dims_, void *opts_)
{
    ocp_qp_gpu_pcond_opts_initialize_default(dims_, opts_);
    ocp_qp_partial_condensing_opts *opts = (ocp_qp_partial_condensing_opts *) opts_;
    opts->N2 = 1;
    opts->N2_bkp = 1;
    opts->full_condensing = 1;
}

void ocp_qp_gpu_pcond_opts_update(void *dims_, void *opts_)
{
    ocp_qp_partial_condensing_opts *opts = (ocp_qp_partial_condensing_opts *) opts_;
    opts->N2_bkp = opts->N2;
}

/* :269-312 */
void ocp_qp_gpu_pcond_opts_set(void *opts_, const char *field, void *value)
{
    ocp_qp_partial_condensing_opts *opts = (ocp_qp_partial_condensing_opts *) opts_;
    if (!strcmp(field, "N"))
    {
        if (opts->full_condensing && *(int *) value != 1)
        {
            printf("\nerror: cond_N = %d: the full condensing module condenses to ONE block\n", *(int *) value);
            exit(1);
        }
        opts->N2 = *(int *) value;
    }
    else if (!strcmp(field, "N_bkp")) opts->N2_bkp = *(int *) value;
    else if (!strcmp(field, "ric_alg")) opts->ric_alg = *(int *) value; /* one condensing algorithm on the device */
    else if (!strcmp(field, "block_size"))
    {
        for (int i = 0; i < opts->N2 + 1; i++) opts->block_size[i] = ((int *) value)[i];
        opts->block_size_was_set = true;
    }
    else
    {
        printf("\nerror: field %s not available in ocp_qp_partial_condensing_opts_set\n", field);
        exit(1);
    }
}

/* dims of the condensed QP for (orig_dims, N2, block sizes): asked of the device library, which pads stage dims to its
 * kernel shapes and decides whether the class is condensable.  :330-380 computes them with d_part_cond_qp_compute_dim. */
static void pcond_compute_dims(ocp_qp_partial_condensing_dims *dims, ocp_qp_partial_condensing_opts *opts)
{
    const ocp_qp_dims *d = dims->orig_dims;
    const int N = d->N, N2 = opts->N2;
    {
        /* FNV-1a over everything the result depends on */
        unsigned long long h = 1469598103934665603ull;
        auto mix = [&](int v) { h = (h ^ (unsigned long long) (unsigned) v) * 1099511628211ull; };
        mix(N); mix(N2); mix(opts->full_condensing ? 1 : 0); mix(opts->block_size_was_set ? 1 : 0);
        int *const arrs[] = {d->nx, d->nu, d->nbx, d->nbu, d->ng, d->ns, d->nbxe};
        for (int *a : arrs) for (int k = 0; k <= N; k++) mix(a[k]);
        if (opts->block_size_was_set) for (int i = 0; i < N2 + 1 && N2 > 0 && N2 < N; i++) mix(opts->block_size[i]);
        if (dims->probe_valid && dims->probe_key == h) return;
        /* valid only once the dims below have been computed for THIS key (set at both successful ends of the function;
         * the refusals exit) */
        dims->probe_valid = 0;
        dims->probe_key = h;
    }
    dims->condensed = 0;
    copy_dims(d, dims->pcond_dims);
    for (int i = 0; i <= N; i++) dims->block_size[i] = i < N ? 1 : 0;
    if (N2 <= 0 || N2 >= N) { dims->probe_valid = 1; return; }
    if (opts->block_size_was_set)
    {
        int sum = 0;
        for (int i = 0; i < N2 + 1; i++) sum += opts->block_size[i];
        if (sum != N)
        {
            printf("partial condensing: sum of block_size should match N, got %d != N = %d\n", sum, N);
            exit(1); /* :346-356 */
        }
    }
    ocp_qp_gpu_batch *probe = ocp_qp_gpu_batch_create(N, d->nx, d->nu, d->nbx, d->nbu, d->ng, d->ns, 1, -1);
    if (!probe)
    {
        printf("\nerror: partial condensing on the device: no GPU batch could be created (no device or unsupported shape)\n");
        exit(1);
    }
    /* default index sets; x0 equality rows as counted by nbxe (the dims depend on counts only) */
    for (int k = 0; k <= N; k++)
        if (d->nbxe[k] > 0)
        {
            std::vector<int> ie(d->nbxe[k]);
            for (int r = 0; r < d->nbxe[k]; r++) ie[r] = d->nbu[k] + r;
            ocp_qp_gpu_batch_set_int(probe, "idxe", k, ie.data(), d->nbxe[k]);
        }
    if (opts->full_condensing) { const int one = 1; ocp_qp_gpu_batch_opts_set(probe, "full_dense", &one); } /* (no fall-back notice: the dense path takes over) */
    ocp_qp_gpu_batch_opts_set(probe, "cond_N", &N2);
    if (opts->block_size_was_set && ocp_qp_gpu_batch_opts_set(probe, "cond_block_size", opts->block_size) != 0) exit(1);
    ocp_qp_gpu_batch *c = ocp_qp_gpu_batch_condense(probe);
    if (c)
    {
        ocp_qp_dims *x = dims->pcond_dims;
        /* N2, or N2 + 1 when the user's last block size is not 0 (the device condenses that block in front of an input-free
         * terminal stage: gpu_batch.hip "cond_block_size") */
        ocp_qp_gpu_batch_get_dims(c, "N", &x->N);
        const char *names[] = {"nx", "nu", "nb", "nbx", "nbu", "ng", "ns", "nbxe"};
        int *dst[] = {x->nx, x->nu, x->nb, x->nbx, x->nbu, x->ng, x->ns, x->nbxe};
        for (int q = 0; q < 8; q++) ocp_qp_gpu_batch_get_dims(c, names[q], dst[q]);
        for (int k = 0; k <= x->N; k++) { x->nbue[k] = 0; x->nge[k] = 0; }
        dims->condensed = 1;
        for (int i = 0; i <= N2; i++)
            dims->block_size[i] = opts->block_size_was_set ? opts->block_size[i] : (i < N2 ? N / N2 + (i < N % N2 ? 1 : 0) : 0);
    }
    /* FULL_CONDENSING_GPU_IPM past what one condensed STAGE may carry (64 variables, 128 inequality sides): the problem is condensed to
     * ONE dense problem inside the solve (dense_kernels.hpp: every state but x0 eliminated, dense Cholesky; correctness first -- the
     * stage-wise solver is faster on every shape measured).  The module's own slots hand the QP through (xcond dims = original dims):
     * the dense problem lives in the solver's workspace on the device and is not read back. */
    ocp_qp_gpu_batch_destroy(probe);
    dims->probe_valid = 1;
}

/* :330-465 */
acados_size_t ocp_qp_gpu_pcond_memory_calculate_size(void *dims_, void *opts_)
{
    ocp_qp_partial_condensing_dims *dims = (ocp_qp_partial_condensing_dims *) dims_;
    ocp_qp_partial_condensing_opts *opts = (ocp_qp_partial_condensing_opts *) opts_;
    pcond_compute_dims(dims, opts);
    return sizeof(ocp_qp_partial_condensing_memory) + sizeof(pcond_device) + ocp_qp_in_calculate_size(dims->pcond_dims)
           + ocp_qp_out_calculate_size(dims->pcond_dims) + ocp_qp_seed_calculate_size(dims->pcond_dims) + 6 * 8;
}

void *ocp_qp_gpu_pcond_memory_assign(void *dims_, void *opts_, void *raw_memory)
{
    ocp_qp_partial_condensing_dims *dims = (ocp_qp_partial_condensing_dims *) dims_;
    char *c = align8((char *) raw_memory);
    ocp_qp_partial_condensing_memory *mem = (ocp_qp_partial_condensing_memory *) c;
    c = align8(c + sizeof(ocp_qp_partial_condensing_memory));
    mem->device = c;
    memset(mem->device, 0, sizeof(pcond_device));
    c = align8(c + sizeof(pcond_device));
    mem->pcond_qp_in = ocp_qp_in_assign(dims->pcond_dims, c); c = align8(c + ocp_qp_in_calculate_size(dims->pcond_dims));
    mem->pcond_qp_out = ocp_qp_out_assign(dims->pcond_dims, c); c = align8(c + ocp_qp_out_calculate_size(dims->pcond_dims));
    mem->pcond_qp_seed = ocp_qp_seed_assign(dims->pcond_dims, c);
    mem->qp_out_info = (qp_info *) mem->pcond_qp_out->misc;
    mem->time_qp_xcond = 0.0;
    mem->ptr_qp_in = nullptr;
    mem->ptr_pcond_qp_in = nullptr;
    mem->ptr_qp_seed = nullptr;
    mem->dims = dims;
    return mem;
}

/* :467-504 */
void ocp_qp_gpu_pcond_memory_get(void *config, void *mem_, const char *field, void *value)
{
    ocp_qp_partial_condensing_memory *mem = (ocp_qp_partial_condensing_memory *) mem_;
    if (!strcmp(field, "xcond_qp_in")) *(ocp_qp_in **) value = mem->pcond_qp_in;
    else if (!strcmp(field, "xcond_qp_out")) *(ocp_qp_out **) value = mem->pcond_qp_out;
    else if (!strcmp(field, "xcond_seed")) *(ocp_qp_seed **) value = mem->pcond_qp_seed;
    else if (!strcmp(field, "qp_out_info")) *(qp_info **) value = mem->qp_out_info;
    else if (!strcmp(field, "time_qp_xcond")) *(double *) value = mem->time_qp_xcond;
    else
    {
        printf("\nerror: ocp_qp_partial_condensing_memory_get: field %s not available\n", field);
        exit(1);
    }
}

acados_size_t ocp_qp_gpu_pcond_workspace_calculate_size(void *dims, void *opts) { return 0; }

/* original QP -> device; condensing options sent whenever the device batch is new */
static ocp_qp_gpu_batch *pcond_load(ocp_qp_partial_condensing_memory *mem, ocp_qp_partial_condensing_opts *opts, ocp_qp_in *qp_in)
{
    pcond_device *dev = (pcond_device *) mem->device;
    if (opts->N2 != opts->N2_bkp)
    {
        printf("\nerror: partial condensing: cond_N changed after the memory was sized (N2 = %d, at creation %d)\n", opts->N2, opts->N2_bkp);
        exit(1); /* assert(opts->N2 == opts->N2_bkp), :533 */
    }
    if (gqp_host::single_batch_load_in(&dev->sb, qp_in) != 0)
    {
        printf("\nerror: partial condensing on the device: no GPU batch could be created (no device or unsupported shape)\n");
        exit(1);
    }
    ocp_qp_gpu_batch *b = dev->sb.batch;
    /* keyed on the batch's generation, not on its address: a batch re-created for a changed structure (idxb, idxs_rev,
     * idxe) often gets the address of the one just destroyed and would never be told cond_N */
    if (dev->configured_gen != dev->sb.generation || dev->N2_sent != opts->N2)
    {
        ocp_qp_gpu_batch_opts_set(b, "cond_N", &opts->N2);
        if (opts->block_size_was_set && ocp_qp_gpu_batch_opts_set(b, "cond_block_size", opts->block_size) != 0) exit(1);
        dev->configured_gen = dev->sb.generation;
        dev->N2_sent = opts->N2;
    }
    return b;
}

/* :523-556 */
int ocp_qp_gpu_pcond_condensing(void *qp_in_, void *pcond_qp_in_, void *opts_, void *mem_, void *work)
{
    ocp_qp_in *qp_in = (ocp_qp_in *) qp_in_, *xc = (ocp_qp_in *) pcond_qp_in_;
    ocp_qp_partial_condensing_opts *opts = (ocp_qp_partial_condensing_opts *) opts_;
    ocp_qp_partial_condensing_memory *mem = (ocp_qp_partial_condensing_memory *) mem_;
    const double t0 = now_s();
    mem->ptr_qp_in = qp_in;
    mem->ptr_pcond_qp_in = xc;
    if (!mem->dims->condensed) copy_qp_in(qp_in, xc);
    else
    {
        ocp_qp_gpu_batch *c = ocp_qp_gpu_batch_condense(pcond_load(mem, opts, qp_in));
        if (!c) return ACADOS_QP_FAILURE;
        gqp_host::single_batch_read_in(c, xc, 3);
    }
    mem->time_qp_xcond = now_s() - t0;
    return ACADOS_SUCCESS;
}

/* :575-598 */
int ocp_qp_gpu_pcond_condense_lhs(void *qp_in_, void *pcond_qp_in_, void *opts_, void *mem_, void *work)
{
    ocp_qp_in *qp_in = (ocp_qp_in *) qp_in_, *xc = (ocp_qp_in *) pcond_qp_in_;
    ocp_qp_partial_condensing_opts *opts = (ocp_qp_partial_condensing_opts *) opts_;
    ocp_qp_partial_condensing_memory *mem = (ocp_qp_partial_condensing_memory *) mem_;
    const double t0 = now_s();
    mem->ptr_qp_in = qp_in;
    mem->ptr_pcond_qp_in = xc;
    if (!mem->dims->condensed) copy_qp_in(qp_in, xc);
    else
    {
        ocp_qp_gpu_batch *b = pcond_load(mem, opts, qp_in);
        ocp_qp_gpu_batch_condense_lhs(b);
        ocp_qp_gpu_batch *c = ocp_qp_gpu_batch_condensed(b);
        if (!c) return ACADOS_QP_FAILURE;
        gqp_host::single_batch_read_in(c, xc, 1);
    }
    mem->time_qp_xcond = now_s() - t0;
    return ACADOS_SUCCESS;
}

/* :602-630 */
int ocp_qp_gpu_pcond_condense_rhs(void *qp_in_, void *pcond_qp_in_, void *opts_, void *mem_, void *work)
{
    ocp_qp_in *qp_in = (ocp_qp_in *) qp_in_, *xc = (ocp_qp_in *) pcond_qp_in_;
    ocp_qp_partial_condensing_opts *opts = (ocp_qp_partial_condensing_opts *) opts_;
    ocp_qp_partial_condensing_memory *mem = (ocp_qp_partial_condensing_memory *) mem_;
    const double t0 = now_s();
    mem->ptr_qp_in = qp_in;
    mem->ptr_pcond_qp_in = xc;
    if (!mem->dims->condensed) copy_qp_in(qp_in, xc);
    else
    {
        ocp_qp_gpu_batch *c = ocp_qp_gpu_batch_condense_rhs(pcond_load(mem, opts, qp_in));
        if (!c) return ACADOS_QP_FAILURE;
        gqp_host::single_batch_read_in(c, xc, 2);
    }
    mem->time_qp_xcond += now_s() - t0;
    return ACADOS_SUCCESS;
}

/* :559-571 */
int ocp_qp_gpu_pcond_condense_qp_out(void *qp_in_, void *pcond_qp_in_, void *qp_out_, void *pcond_qp_out_, void *opts_,
                                     void *mem_, void *work)
{
    ocp_qp_in *qp_in = (ocp_qp_in *) qp_in_;
    ocp_qp_out *qp_out = (ocp_qp_out *) qp_out_, *xo = (ocp_qp_out *) pcond_qp_out_;
    ocp_qp_partial_condensing_memory *mem = (ocp_qp_partial_condensing_memory *) mem_;
    if (!mem->dims->condensed) { copy_qp_out(qp_out, xo); return ACADOS_SUCCESS; }
    pcond_device *dev = (pcond_device *) mem->device;
    ocp_qp_gpu_batch *b = dev->sb.batch, *c = b ? ocp_qp_gpu_batch_condensed(b) : nullptr;
    if (!c) return ACADOS_QP_FAILURE; /* condensing has to run first, as in ocp_qp_xcond_solve */
    gqp_host::single_batch_push_out(b, qp_in->dim, qp_out);
    if (ocp_qp_gpu_batch_condense_sol(b) != 0) return ACADOS_QP_FAILURE;
    gqp_host::single_batch_pull_out(c, xo->dim, xo);
    return ACADOS_SUCCESS;
}

/* :664-689 */
int ocp_qp_gpu_pcond_expansion(void *pcond_qp_out_, void *qp_out_, void *opts_, void *mem_, void *work)
{
    ocp_qp_out *xo = (ocp_qp_out *) pcond_qp_out_, *out = (ocp_qp_out *) qp_out_;
    ocp_qp_partial_condensing_memory *mem = (ocp_qp_partial_condensing_memory *) mem_;
    const double t0 = now_s();
    if (!mem->dims->condensed) copy_qp_out(xo, out);
    else
    {
        pcond_device *dev = (pcond_device *) mem->device;
        ocp_qp_gpu_batch *b = dev->sb.batch, *c = b ? ocp_qp_gpu_batch_condensed(b) : nullptr;
        if (!c) return ACADOS_QP_FAILURE;
        gqp_host::single_batch_push_out(c, xo->dim, xo);
        if (ocp_qp_gpu_batch_expand(b) != 0) return ACADOS_QP_FAILURE;
        gqp_host::single_batch_pull_out(b, out->dim, out);
    }
    if (out->misc) ((qp_info *) out->misc)->t_computed = 1;
    mem->time_qp_xcond += now_s() - t0;
    return ACADOS_SUCCESS;
}

static void copy_seed(const ocp_qp_seed *a, ocp_qp_seed *b)
{
    const ocp_qp_dims *d = a->dim;
    for (int k = 0; k <= d->N; k++)
    {
        const int nct = 2 * (d->nb[k] + d->ng[k] + d->ns[k]);
        memcpy(b->seed_g[k], a->seed_g[k], sizeof(double) * (d->nu[k] + d->nx[k] + 2 * d->ns[k]));
        if (k < d->N) memcpy(b->seed_b[k], a->seed_b[k], sizeof(double) * d->nx[k + 1]);
        memcpy(b->seed_d[k], a->seed_d[k], sizeof(double) * nct);
        memcpy(b->seed_m[k], a->seed_m[k], sizeof(double) * nct);
    }
}

/* Seeds through the condensing (N2 < N).  Vector condensing is LINEAR and homogeneous in the vector data (b, r, q, zl, zu,
 * bounds): the condensed gradient is Gamma'(H c + g) with c the block's accumulated offsets, the condensed bounds are
 * d - a_x'c, the condensed dynamics offset is the block's c -- no constant term.  So the condensed seed is the vector
 * condensing (condense_rhs) of a QP with the SAME matrices whose vectors are the seeds, and the expansion of the condensed
 * sensitivities is the expansion kernel run on that QP: dx+ = A dx + B du + seed_b, d pi from stationarity with seed_q,
 * dt = d(row value) - seed_d.  The device batch of the module holds the original QP: its vector fields are overwritten
 * by the seeds for the duration of the call and restored from qp_in afterwards.  (HPIPM: d_part_cond_qp_cond_seed /
 * d_part_cond_qp_expand_sol_seed, ocp_qp_partial_condensing.c:634-662, 691-717.) */
static void seed_vectors_to_device(ocp_qp_gpu_batch *b, const ocp_qp_dims *d, const ocp_qp_seed *sd)
{
    for (int k = 0; k <= d->N; k++)
    {
        const int nu = d->nu[k], nx = d->nx[k], ns = d->ns[k], nbu = d->nbu[k], nbx = d->nbx[k], nb = d->nb[k], ng = d->ng[k];
        double *g = sd->seed_g[k], *dd = sd->seed_d[k];
        struct { const char *name; double *p; int n; } f[] = {
            {"r", g, nu}, {"q", g + nu, nx}, {"zl", g + nu + nx, ns}, {"zu", g + nu + nx + ns, ns},
            {"b", k < d->N ? sd->seed_b[k] : nullptr, k < d->N ? d->nx[k + 1] : 0},
            {"lbu", dd, nbu}, {"lbx", dd + nbu, nbx}, {"lg", dd + nb, ng},
            {"ubu", dd + nb + ng, nbu}, {"ubx", dd + nb + ng + nbu, nbx}, {"ug", dd + 2 * nb + ng, ng},
            {"lls", dd + 2 * nb + 2 * ng, ns}, {"lus", dd + 2 * nb + 2 * ng + ns, ns}};
        for (auto &e : f)
            if (e.n > 0) ocp_qp_gpu_batch_set(b, e.name, k, e.p, 0);
    }
}

static void seed_vectors_from_device(ocp_qp_gpu_batch *c, const ocp_qp_dims *d, ocp_qp_seed *sd)
{
    for (int k = 0; k <= d->N; k++)
    {
        const int nu = d->nu[k], nx = d->nx[k], ns = d->ns[k], nbu = d->nbu[k], nbx = d->nbx[k], nb = d->nb[k], ng = d->ng[k];
        double *g = sd->seed_g[k], *dd = sd->seed_d[k];
        struct { const char *name; double *p; int n; } f[] = {
            {"r", g, nu}, {"q", g + nu, nx}, {"zl", g + nu + nx, ns}, {"zu", g + nu + nx + ns, ns},
            {"b", k < d->N ? sd->seed_b[k] : nullptr, k < d->N ? d->nx[k + 1] : 0},
            {"lbu", dd, nbu}, {"lbx", dd + nbu, nbx}, {"lg", dd + nb, ng},
            {"ubu", dd + nb + ng, nbu}, {"ubx", dd + nb + ng + nbu, nbx}, {"ug", dd + 2 * nb + ng, ng},
            {"lls", dd + 2 * nb + 2 * ng, ns}, {"lus", dd + 2 * nb + 2 * ng + ns, ns}};
        for (auto &e : f)
            if (e.n > 0) ocp_qp_gpu_batch_get(c, e.name, k, e.p, 0);
        memset(sd->seed_m[k], 0, sizeof(double) * 2 * (nb + ng + ns));
    }
}

/* :634-660 */
int ocp_qp_gpu_pcond_condense_rhs_seed(void *qp_in_, void *seed_, void *pcond_seed_, void *opts_, void *mem_, void *work)
{
    ocp_qp_in *qp_in = (ocp_qp_in *) qp_in_;
    ocp_qp_seed *seed = (ocp_qp_seed *) seed_, *xs = (ocp_qp_seed *) pcond_seed_;
    ocp_qp_partial_condensing_opts *opts = (ocp_qp_partial_condensing_opts *) opts_;
    ocp_qp_partial_condensing_memory *mem = (ocp_qp_partial_condensing_memory *) mem_;
    const double t0 = now_s();
    mem->ptr_qp_in = qp_in;
    mem->ptr_qp_seed = seed;
    if (!mem->dims->condensed) copy_seed(seed, xs);
    else
    {
        ocp_qp_gpu_batch *b = pcond_load(mem, opts, qp_in);   /* matrices (and, for now, the vectors) of qp_in */
        if (ocp_qp_gpu_batch_condense_lhs(b) != 0) return ACADOS_QP_FAILURE;
        seed_vectors_to_device(b, qp_in->dim, seed);
        ocp_qp_gpu_batch *c = ocp_qp_gpu_batch_condense_rhs(b);
        if (!c) return ACADOS_QP_FAILURE;
        seed_vectors_from_device(c, xs->dim, xs);
        if (gqp_host::single_batch_load_in(&((pcond_device *) mem->device)->sb, qp_in) != 0) return ACADOS_QP_FAILURE; /* vectors back */
    }
    mem->time_qp_xcond += now_s() - t0;
    return ACADOS_SUCCESS;
}

/* :691-716 */
int ocp_qp_gpu_pcond_expand_sol_seed(void *pcond_qp_out_, void *qp_out_, void *opts_, void *mem_, void *work)
{
    ocp_qp_out *xo = (ocp_qp_out *) pcond_qp_out_, *out = (ocp_qp_out *) qp_out_;
    ocp_qp_partial_condensing_memory *mem = (ocp_qp_partial_condensing_memory *) mem_;
    const double t0 = now_s();
    if (!mem->dims->condensed) copy_qp_out(xo, out);
    else
    {
        pcond_device *dev = (pcond_device *) mem->device;
        ocp_qp_gpu_batch *b = dev->sb.batch, *c = b ? ocp_qp_gpu_batch_condensed(b) : nullptr;
        if (!c || !mem->ptr_qp_in || !mem->ptr_qp_seed)
        {
            printf("\nerror: partial condensing: expand_sol_seed before condense_rhs_seed\n");
            return ACADOS_QP_FAILURE;
        }
        seed_vectors_to_device(b, mem->ptr_qp_in->dim, mem->ptr_qp_seed);
        gqp_host::single_batch_push_out(c, xo->dim, xo);
        if (ocp_qp_gpu_batch_expand(b) != 0) return ACADOS_QP_FAILURE;
        gqp_host::single_batch_pull_out(b, out->dim, out);
        if (gqp_host::single_batch_load_in(&dev->sb, mem->ptr_qp_in) != 0) return ACADOS_QP_FAILURE;
    }
    mem->time_qp_xcond += now_s() - t0;
    return ACADOS_SUCCESS;
}

/* device-side resources of a module memory (the reference's module has none; called by the xcond-solver's terminate and
 * by ocp_qp_condensing_free) */
void ocp_qp_gpu_pcond_memory_release(void *mem_)
{
    ocp_qp_partial_condensing_memory *mem = (ocp_qp_partial_condensing_memory *) mem_;
    if (!mem || !mem->device) return;
    pcond_device *dev = (pcond_device *) mem->device;
    gqp_host::single_batch_free(&dev->sb);
    dev->configured_gen = -1;
}

/* ocp_qp_partial_condensing.c:720-750 */
void ocp_qp_gpu_pcond_config_initialize_default(void *config_)
{
    ocp_qp_xcond_config *config = (ocp_qp_xcond_config *) config_;
    config->dims_calculate_size = &ocp_qp_gpu_pcond_dims_calculate_size;
    config->dims_assign = &ocp_qp_gpu_pcond_dims_assign;
    config->dims_set = &ocp_qp_gpu_pcond_dims_set;
    config->dims_get = &ocp_qp_gpu_pcond_dims_get;
    config->opts_calculate_size = &ocp_qp_gpu_pcond_opts_calculate_size;
    config->opts_assign = &ocp_qp_gpu_pcond_opts_assign;
    config->opts_initialize_default = &ocp_qp_gpu_pcond_opts_initialize_default;
    config->opts_update = &ocp_qp_gpu_pcond_opts_update;
    config->opts_set = &ocp_qp_gpu_pcond_opts_set;
    config->memory_calculate_size = &ocp_qp_gpu_pcond_memory_calculate_size;
    config->memory_assign = &ocp_qp_gpu_pcond_memory_assign;
    config->memory_get = &ocp_qp_gpu_pcond_memory_get;
    config->workspace_calculate_size = &ocp_qp_gpu_pcond_workspace_calculate_size;
    config->condensing = &ocp_qp_gpu_pcond_condensing;
    config->condense_rhs = &ocp_qp_gpu_pcond_condense_rhs;
    config->condense_rhs_seed = &ocp_qp_gpu_pcond_condense_rhs_seed;
    config->condense_lhs = &ocp_qp_gpu_pcond_condense_lhs;
    config->condense_qp_out = &ocp_qp_gpu_pcond_condense_qp_out;
    config->expansion = &ocp_qp_gpu_pcond_expansion;
    config->expand_sol_seed = &ocp_qp_gpu_pcond_expand_sol_seed;
}

void ocp_qp_gpu_fcond_config_initialize_default(void *config_)
{
    ocp_qp_gpu_pcond_config_initialize_default(config_);
    ((ocp_qp_xcond_config *) config_)->opts_initialize_default = &ocp_qp_gpu_fcond_opts_initialize_default;
}

/* ---- interfaces/acados_c/condensing_interface.c, same names and call sequence ---- */

ocp_qp_xcond_config *ocp_qp_condensing_config_create(condensing_plan *plan)
{
    ocp_qp_xcond_config *config = (ocp_qp_xcond_config *) calloc(1, sizeof(ocp_qp_xcond_config));
    switch (plan->condensing_type)
    {
        case PARTIAL_CONDENSING: ocp_qp_gpu_pcond_config_initialize_default(config); break;
        case FULL_CONDENSING: ocp_qp_gpu_fcond_config_initialize_default(config); break;
    }
    return config;
}

void *ocp_qp_condensing_dims_create(ocp_qp_xcond_config *config, int N)
{
    return config->dims_assign(config, N, calloc(1, config->dims_calculate_size(config, N)));
}

void *ocp_qp_condensing_opts_create(ocp_qp_xcond_config *config, void *dims_)
{
    void *opts = config->opts_assign(dims_, calloc(1, config->opts_calculate_size(dims_)));
    config->opts_initialize_default(dims_, opts);
    return opts;
}

acados_size_t ocp_qp_condensing_calculate_size(ocp_qp_xcond_config *config, void *dims_, void *opts_)
{
    return sizeof(condensing_module) + config->memory_calculate_size(dims_, opts_) + config->workspace_calculate_size(dims_, opts_) + 8;
}

condensing_module *ocp_qp_condensing_assign(ocp_qp_xcond_config *config, void *dims_, void *opts_, void *raw_memory)
{
    char *c = (char *) raw_memory;
    condensing_module *module = (condensing_module *) c;
    c = align8(c + sizeof(condensing_module));
    module->config = config;
    module->dims = dims_;
    module->opts = opts_;
    module->mem = config->memory_assign(dims_, opts_, c);
    c += config->memory_calculate_size(dims_, opts_);
    module->work = (void *) c;
    return module;
}

condensing_module *ocp_qp_condensing_create(ocp_qp_xcond_config *config, void *dims_, void *opts_)
{
    config->opts_update(dims_, opts_);
    return ocp_qp_condensing_assign(config, dims_, opts_, calloc(1, ocp_qp_condensing_calculate_size(config, dims_, opts_)));
}

/* releases the device batch of the module and the block ocp_qp_condensing_create allocated */
void ocp_qp_condensing_free(condensing_module *module)
{
    if (!module) return;
    ocp_qp_gpu_pcond_memory_release(module->mem);
    free(module);
}

int ocp_qp_condense(condensing_module *module, void *qp_in, void *qp_out)
{
    return module->config->condensing(qp_in, qp_out, module->opts, module->mem, module->work);
}

int ocp_qp_expand(condensing_module *module, void *qp_in, void *qp_out)
{
    return module->config->expansion(qp_in, qp_out, module->opts, module->mem, module->work);
}

/* ===================================================================== the xcond solver (22 slots) */
/* acados/ocp_qp/ocp_qp_xcond_solver.c, slot by slot */

/* :84-136 */
acados_size_t ocp_qp_gpu_xcond_solver_dims_calculate_size(void *config_, int N)
{
    ocp_qp_xcond_solver_config *config = (ocp_qp_xcond_solver_config *) config_;
    return sizeof(ocp_qp_xcond_solver_dims) + ocp_qp_dims_calculate_size(N) + config->xcond->dims_calculate_size(config->xcond, N) + 3 * 8;
}

ocp_qp_xcond_solver_dims *ocp_qp_gpu_xcond_solver_dims_assign(void *config_, int N, void *raw_memory)
{
    ocp_qp_xcond_solver_config *config = (ocp_qp_xcond_solver_config *) config_;
    char *c = align8((char *) raw_memory);
    ocp_qp_xcond_solver_dims *dims = (ocp_qp_xcond_solver_dims *) c;
    c = align8(c + sizeof(ocp_qp_xcond_solver_dims));
    dims->orig_dims = ocp_qp_dims_assign(N, c);
    c = align8(c + ocp_qp_dims_calculate_size(N));
    dims->xcond_dims = config->xcond->dims_assign(config->xcond, N, c);
    return dims;
}

/* :140-153 */
void ocp_qp_gpu_xcond_solver_dims_set_(void *config_, ocp_qp_xcond_solver_dims *dims, int stage, const char *field, int *value)
{
    ocp_qp_xcond_solver_config *config = (ocp_qp_xcond_solver_config *) config_;
    ocp_qp_dims_set(config_, dims->orig_dims, stage, field, value);
    config->xcond->dims_set(config->xcond, dims->xcond_dims, stage, field, value);
}

/* :157-192: "pcond_<field>" answers from the condensed dims */
void ocp_qp_gpu_xcond_solver_dims_get_(void *config_, ocp_qp_xcond_solver_dims *dims, int stage, const char *field, int *value)
{
    ocp_qp_xcond_solver_config *config = (ocp_qp_xcond_solver_config *) config_;
    if (!strncmp(field, "pcond_", 6) || !strncmp(field, "fcond_", 6))
    {
        void *xcond_qp_dims;
        config->xcond->dims_get(config->xcond, dims->xcond_dims, "xcond_dims", &xcond_qp_dims);
        ocp_qp_dims_get(config_, xcond_qp_dims, stage, field + 6, value);
        return;
    }
    ocp_qp_dims_get(config_, dims->orig_dims, stage, field, value);
}

/* :200-246.  The inner opts do not depend on dims. */
acados_size_t ocp_qp_gpu_xcond_solver_opts_calculate_size(void *config_, ocp_qp_xcond_solver_dims *dims)
{
    ocp_qp_xcond_solver_config *config = (ocp_qp_xcond_solver_config *) config_;
    return sizeof(ocp_qp_xcond_solver_opts) + config->xcond->opts_calculate_size(dims->xcond_dims)
           + config->qp_solver->opts_calculate_size(config->qp_solver, dims->orig_dims) + 3 * 8;
}

void *ocp_qp_gpu_xcond_solver_opts_assign(void *config_, ocp_qp_xcond_solver_dims *dims, void *raw_memory)
{
    ocp_qp_xcond_solver_config *config = (ocp_qp_xcond_solver_config *) config_;
    char *c = align8((char *) raw_memory);
    ocp_qp_xcond_solver_opts *opts = (ocp_qp_xcond_solver_opts *) c;
    c = align8(c + sizeof(ocp_qp_xcond_solver_opts));
    opts->xcond_opts = config->xcond->opts_assign(dims->xcond_dims, c);
    c = align8(c + config->xcond->opts_calculate_size(dims->xcond_dims));
    opts->qp_solver_opts = config->qp_solver->opts_assign(config->qp_solver, dims->orig_dims, c);
    return opts;
}

/* :250-267 */
void ocp_qp_gpu_xcond_solver_opts_initialize_default(void *config_, ocp_qp_xcond_solver_dims *dims, void *opts_)
{
    ocp_qp_xcond_solver_config *config = (ocp_qp_xcond_solver_config *) config_;
    ocp_qp_xcond_solver_opts *opts = (ocp_qp_xcond_solver_opts *) opts_;
    config->xcond->opts_initialize_default(dims->xcond_dims, opts->xcond_opts);
    config->qp_solver->opts_initialize_default(config->qp_solver, dims->orig_dims, opts->qp_solver_opts);
    opts->initialize_next_xcond_qp_from_qp_out = false;
}

/* :271-287 */
void ocp_qp_gpu_xcond_solver_opts_update(void *config_, ocp_qp_xcond_solver_dims *dims, void *opts_)
{
    ocp_qp_xcond_solver_config *config = (ocp_qp_xcond_solver_config *) config_;
    ocp_qp_xcond_solver_opts *opts = (ocp_qp_xcond_solver_opts *) opts_;
    config->xcond->opts_update(dims->xcond_dims, opts->xcond_opts);
    config->qp_solver->opts_update(config->qp_solver, dims->orig_dims, opts->qp_solver_opts);
}

/* :291-317: "cond_" prefix goes to the condensing module */
void ocp_qp_gpu_xcond_solver_opts_set_(void *config_, void *opts_, const char *field, void *value)
{
    ocp_qp_xcond_solver_config *config = (ocp_qp_xcond_solver_config *) config_;
    ocp_qp_xcond_solver_opts *opts = (ocp_qp_xcond_solver_opts *) opts_;
    if (!strncmp(field, "cond_", 5)) config->xcond->opts_set(opts->xcond_opts, field + 5, value);
    else if (!strcmp(field, "initialize_next_xcond_qp_from_qp_out")) opts->initialize_next_xcond_qp_from_qp_out = *(bool *) value;
    else config->qp_solver->opts_set(config->qp_solver, opts->qp_solver_opts, field, value);
}

/* :321-331 */
void ocp_qp_gpu_xcond_solver_opts_get_(void *config_, void *opts_, const char *field, void *value)
{
    ocp_qp_xcond_solver_config *config = (ocp_qp_xcond_solver_config *) config_;
    ocp_qp_xcond_solver_opts *opts = (ocp_qp_xcond_solver_opts *) opts_;
    config->qp_solver->opts_get(config->qp_solver, opts->qp_solver_opts, field, value);
}

/* :338-395.  The inner memory is sized for the ORIGINAL dims: the fused evaluate hands the original QP to the device. */
acados_size_t ocp_qp_gpu_xcond_solver_memory_calculate_size(void *config_, ocp_qp_xcond_solver_dims *dims, void *opts_)
{
    ocp_qp_xcond_solver_config *config = (ocp_qp_xcond_solver_config *) config_;
    ocp_qp_xcond_solver_opts *opts = (ocp_qp_xcond_solver_opts *) opts_;
    return sizeof(ocp_qp_xcond_solver_memory) + config->xcond->memory_calculate_size(dims->xcond_dims, opts->xcond_opts)
           + config->qp_solver->memory_calculate_size(config->qp_solver, dims->orig_dims, opts->qp_solver_opts) + 3 * 8;
}

void *ocp_qp_gpu_xcond_solver_memory_assign(void *config_, ocp_qp_xcond_solver_dims *dims, void *opts_, void *raw_memory)
{
    ocp_qp_xcond_solver_config *config = (ocp_qp_xcond_solver_config *) config_;
    ocp_qp_xcond_solver_opts *opts = (ocp_qp_xcond_solver_opts *) opts_;
    ocp_qp_xcond_config *xcond = config->xcond;
    char *c = align8((char *) raw_memory);
    ocp_qp_xcond_solver_memory *mem = (ocp_qp_xcond_solver_memory *) c;
    c = align8(c + sizeof(ocp_qp_xcond_solver_memory));
    /* the condensed dims are (re)computed by memory_calculate_size of the module, as in the reference (:343-347) */
    const acados_size_t xsz = xcond->memory_calculate_size(dims->xcond_dims, opts->xcond_opts);
    mem->xcond_memory = xcond->memory_assign(dims->xcond_dims, opts->xcond_opts, c);
    c = align8(c + xsz);
    mem->solver_memory = config->qp_solver->memory_assign(config->qp_solver, dims->orig_dims, opts->qp_solver_opts, c);
    xcond->memory_get(xcond, mem->xcond_memory, "xcond_qp_in", &mem->xcond_qp_in);
    xcond->memory_get(xcond, mem->xcond_memory, "xcond_qp_out", &mem->xcond_qp_out);
    xcond->memory_get(xcond, mem->xcond_memory, "xcond_seed", &mem->xcond_seed);
    return mem;
}

/* :399-417 */
void ocp_qp_gpu_xcond_solver_memory_reset(void *config_, ocp_qp_xcond_solver_dims *dims, ocp_qp_in *qp_in, ocp_qp_out *qp_out,
                                          void *opts_, void *mem_, void *work_)
{
    ocp_qp_xcond_solver_config *config = (ocp_qp_xcond_solver_config *) config_;
    ocp_qp_xcond_solver_opts *opts = (ocp_qp_xcond_solver_opts *) opts_;
    ocp_qp_xcond_solver_memory *mem = (ocp_qp_xcond_solver_memory *) mem_;
    config->qp_solver->memory_reset(config->qp_solver, qp_in, qp_out, opts->qp_solver_opts, mem->solver_memory, nullptr);
}

/* :420-432 */
void ocp_qp_gpu_xcond_solver_get(void *config_, ocp_qp_in *qp_in, ocp_qp_out *qp_out, void *opts_, void *mem_, const char *field,
                                 int stage, void *value, int size1, int size2)
{
    ocp_qp_xcond_solver_config *config = (ocp_qp_xcond_solver_config *) config_;
    ocp_qp_xcond_solver_opts *opts = (ocp_qp_xcond_solver_opts *) opts_;
    ocp_qp_xcond_solver_memory *mem = (ocp_qp_xcond_solver_memory *) mem_;
    config->qp_solver->solver_get(config->qp_solver, qp_in, qp_out, opts->qp_solver_opts, mem->solver_memory, field, stage, value, size1, size2);
}

/* :436-470 */
void ocp_qp_gpu_xcond_solver_memory_get(void *config_, void *mem_, const char *field, void *value)
{
    ocp_qp_xcond_solver_config *config = (ocp_qp_xcond_solver_config *) config_;
    ocp_qp_xcond_solver_memory *mem = (ocp_qp_xcond_solver_memory *) mem_;
    if (!strcmp(field, "time_qp_solver_call") || !strcmp(field, "tau_iter") || !strcmp(field, "iter") || !strcmp(field, "status")
        || !strcmp(field, "stat") || !strcmp(field, "stat_m") || !strcmp(field, "stat_rows"))
        config->qp_solver->memory_get(config->qp_solver, mem->solver_memory, field, value);
    else if (!strcmp(field, "time_qp_xcond"))
        config->xcond->memory_get(config->xcond, mem->xcond_memory, field, value);
    else
    {
        printf("\nerror: ocp_qp_xcond_solver_memory_get: field %s not available\n", field);
        exit(1);
    }
}

/* :478-496 */
acados_size_t ocp_qp_gpu_xcond_solver_workspace_calculate_size(void *config_, ocp_qp_xcond_solver_dims *dims, void *opts_)
{
    return sizeof(ocp_qp_xcond_solver_workspace);
}

static int xcond_fused(void *config_, ocp_qp_in *qp_in, ocp_qp_out *qp_out, void *opts_, void *mem_, int phase)
{
    ocp_qp_xcond_solver_config *config = (ocp_qp_xcond_solver_config *) config_;
    ocp_qp_xcond_solver_opts *opts = (ocp_qp_xcond_solver_opts *) opts_;
    ocp_qp_xcond_solver_memory *mem = (ocp_qp_xcond_solver_memory *) mem_;
    ocp_qp_partial_condensing_opts *xo = (ocp_qp_partial_condensing_opts *) opts->xcond_opts;
    ocp_qp_partial_condensing_memory *xm = (ocp_qp_partial_condensing_memory *) mem->xcond_memory;
    /* the condensing request of THIS call, read from the opts / memory of THIS solver (no state outside them); a class the
     * module sized as "not condensed" goes to the device as a full-space QP */
    cond_request cr = {xm->dims->condensed ? xo->N2 : 0, xm->dims->condensed && xo->block_size_was_set ? xo->block_size : nullptr,
                       opts->initialize_next_xcond_qp_from_qp_out ? 1 : 0, phase, xo->full_condensing && !xm->dims->condensed ? 1 : 0};
    void *ins[1] = {qp_in}, *outs[1] = {qp_out}, *mems[1] = {mem->solver_memory};
    int status = 0;
    gqp_host::gpu_ipm_evaluate_impl(config->qp_solver, 1, ins, outs, opts->qp_solver_opts, mems, nullptr, &status, &cr);
    if (phase != 1) opts->initialize_next_xcond_qp_from_qp_out = false; /* consumed, :565 */
    qp_info *info = (qp_info *) qp_out->misc;
    if (phase == 2) xm->time_qp_xcond += info->condensing_time;
    else if (phase == 0) xm->time_qp_xcond = info->condensing_time;
    return status;
}

/* :529-587 ocp_qp_xcond_solve: condense -> (warm-start the condensed QP) -> solve -> expand, fused on the device */
int ocp_qp_gpu_xcond_solve(void *config_, ocp_qp_xcond_solver_dims *dims, ocp_qp_in *qp_in, ocp_qp_out *qp_out, void *opts_,
                           void *mem_, void *work_)
{
    return xcond_fused(config_, qp_in, qp_out, opts_, mem_, 0);
}

/* :591-620: the matrix part of the condensing, kept resident in HBM for the feedback phase */
int ocp_qp_gpu_xcond_condense_lhs(void *config_, ocp_qp_xcond_solver_dims *dims, ocp_qp_in *qp_in, ocp_qp_out *qp_out, void *opts_,
                                  void *mem_, void *work_)
{
    const double t0 = now_s();
    xcond_fused(config_, qp_in, qp_out, opts_, mem_, 1);
    qp_info *info = (qp_info *) qp_out->misc;
    info->condensing_time = now_s() - t0;
    info->total_time = info->condensing_time;
    ((ocp_qp_partial_condensing_memory *) ((ocp_qp_xcond_solver_memory *) mem_)->xcond_memory)->time_qp_xcond = info->condensing_time;
    return ACADOS_SUCCESS;
}

/* :623-669 */
int ocp_qp_gpu_xcond_condense_rhs_and_solve(void *config_, ocp_qp_xcond_solver_dims *dims, ocp_qp_in *qp_in, ocp_qp_out *qp_out,
                                            void *opts_, void *mem_, void *work_)
{
    return xcond_fused(config_, qp_in, qp_out, opts_, mem_, 2);
}

/* :673-727.  In the full space at the (expanded) solution: the device batch of the inner memory holds the original QP. */
void ocp_qp_gpu_xcond_solver_eval_forw_sens(void *config_, ocp_qp_xcond_solver_dims *dims, ocp_qp_in *param_qp_in, ocp_qp_seed *seed,
                                            ocp_qp_out *sens_qp_out, void *opts_, void *mem_, void *work_)
{
    ocp_qp_xcond_solver_config *config = (ocp_qp_xcond_solver_config *) config_;
    ocp_qp_xcond_solver_opts *opts = (ocp_qp_xcond_solver_opts *) opts_;
    ocp_qp_xcond_solver_memory *mem = (ocp_qp_xcond_solver_memory *) mem_;
    config->qp_solver->eval_forw_sens(config->qp_solver, param_qp_in, seed, sens_qp_out, opts->qp_solver_opts, mem->solver_memory, nullptr);
}

void ocp_qp_gpu_xcond_solver_eval_adj_sens(void *config_, ocp_qp_xcond_solver_dims *dims, ocp_qp_in *param_qp_in, ocp_qp_seed *seed,
                                           ocp_qp_out *sens_qp_out, void *opts_, void *mem_, void *work_)
{
    ocp_qp_xcond_solver_config *config = (ocp_qp_xcond_solver_config *) config_;
    ocp_qp_xcond_solver_opts *opts = (ocp_qp_xcond_solver_opts *) opts_;
    ocp_qp_xcond_solver_memory *mem = (ocp_qp_xcond_solver_memory *) mem_;
    config->qp_solver->eval_adj_sens(config->qp_solver, param_qp_in, seed, sens_qp_out, opts->qp_solver_opts, mem->solver_memory, nullptr);
}

/* :731-740 (+ the device batch of the condensing module, which the reference's module does not have) */
void ocp_qp_gpu_xcond_solver_terminate(void *config_, void *mem_, void *work_)
{
    ocp_qp_xcond_solver_config *config = (ocp_qp_xcond_solver_config *) config_;
    ocp_qp_xcond_solver_memory *mem = (ocp_qp_xcond_solver_memory *) mem_;
    config->qp_solver->terminate(config->qp_solver, mem->solver_memory, nullptr);
    ocp_qp_gpu_pcond_memory_release(mem->xcond_memory);
}

/* :744-770 */
void ocp_qp_gpu_xcond_solver_config_initialize_default(void *config_)
{
    ocp_qp_xcond_solver_config *config = (ocp_qp_xcond_solver_config *) config_;
    config->dims_calculate_size = &ocp_qp_gpu_xcond_solver_dims_calculate_size;
    config->dims_assign = &ocp_qp_gpu_xcond_solver_dims_assign;
    config->dims_set = &ocp_qp_gpu_xcond_solver_dims_set_;
    config->dims_get = &ocp_qp_gpu_xcond_solver_dims_get_;
    config->opts_calculate_size = &ocp_qp_gpu_xcond_solver_opts_calculate_size;
    config->opts_assign = &ocp_qp_gpu_xcond_solver_opts_assign;
    config->opts_initialize_default = &ocp_qp_gpu_xcond_solver_opts_initialize_default;
    config->opts_update = &ocp_qp_gpu_xcond_solver_opts_update;
    config->opts_set = &ocp_qp_gpu_xcond_solver_opts_set_;
    config->opts_get = &ocp_qp_gpu_xcond_solver_opts_get_;
    config->memory_calculate_size = &ocp_qp_gpu_xcond_solver_memory_calculate_size;
    config->memory_assign = &ocp_qp_gpu_xcond_solver_memory_assign;
    config->memory_get = &ocp_qp_gpu_xcond_solver_memory_get;
    config->solver_get = &ocp_qp_gpu_xcond_solver_get;
    config->memory_reset = &ocp_qp_gpu_xcond_solver_memory_reset;
    config->workspace_calculate_size = &ocp_qp_gpu_xcond_solver_workspace_calculate_size;
    config->evaluate = &ocp_qp_gpu_xcond_solve;
    config->condense_lhs = &ocp_qp_gpu_xcond_condense_lhs;
    config->condense_rhs_and_solve = &ocp_qp_gpu_xcond_condense_rhs_and_solve;
    config->eval_forw_sens = &ocp_qp_gpu_xcond_solver_eval_forw_sens;
    config->eval_adj_sens = &ocp_qp_gpu_xcond_solver_eval_adj_sens;
    config->terminate = &ocp_qp_gpu_xcond_solver_terminate;
}

/* ===================================================================== acados_c-shaped convenience layer */
/* interfaces/acados_c/ocp_qp_interface.c */

/* :61-83 config block = the 22 slots + the two sub-vtables (ocp_qp_xcond_solver.c:48-83) */
acados_size_t ocp_qp_xcond_solver_config_calculate_size()
{
    return sizeof(ocp_qp_xcond_solver_config) + sizeof(qp_solver_config) + sizeof(ocp_qp_xcond_config) + 3 * 8;
}

ocp_qp_xcond_solver_config *ocp_qp_xcond_solver_config_assign(void *raw_memory)
{
    char *c = align8((char *) raw_memory);
    ocp_qp_xcond_solver_config *config = (ocp_qp_xcond_solver_config *) c;
    c = align8(c + sizeof(ocp_qp_xcond_solver_config));
    config->qp_solver = (qp_solver_config *) c;
    c = align8(c + sizeof(qp_solver_config));
    config->xcond = (ocp_qp_xcond_config *) c;
    return config;
}

/* :185-259 */
ocp_qp_xcond_solver_config *ocp_qp_xcond_solver_config_create_from_name(const char *name)
{
    const bool partial = !strcmp(name, "PARTIAL_CONDENSING_GPU_IPM") || !strcmp(name, "PARTIAL_CONDENSING_HPIPM");
    const bool full = !strcmp(name, "FULL_CONDENSING_GPU_IPM");
    if (!partial && !full)
    {
        printf("\nerror: ocp_qp_xcond_solver_config_create_from_name: QP solver %s not supported by acados_amd\n", name);
        return nullptr;
    }
    ocp_qp_xcond_solver_config *c = ocp_qp_xcond_solver_config_assign(calloc(1, ocp_qp_xcond_solver_config_calculate_size()));
    ocp_qp_gpu_xcond_solver_config_initialize_default(c);
    ocp_qp_gpu_ipm_config_initialize_default(c->qp_solver);
    if (full) ocp_qp_gpu_fcond_config_initialize_default(c->xcond);
    else ocp_qp_gpu_pcond_config_initialize_default(c->xcond);
    return c;
}

void ocp_qp_xcond_solver_config_free(ocp_qp_xcond_solver_config *c) { free(c); }

/* :306-316 */
ocp_qp_xcond_solver_dims *ocp_qp_xcond_solver_dims_create(ocp_qp_xcond_solver_config *config, int N)
{
    return config->dims_assign(config, N, calloc(1, config->dims_calculate_size(config, N)));
}

void ocp_qp_xcond_solver_dims_set(void *config_, ocp_qp_xcond_solver_dims *dims, int stage, const char *field, int *value)
{
    ((ocp_qp_xcond_solver_config *) config_)->dims_set(config_, dims, stage, field, value);
}

void ocp_qp_xcond_solver_dims_free(ocp_qp_xcond_solver_dims *d) { free(d); }

/* :483-511 */
void *ocp_qp_xcond_solver_opts_create(ocp_qp_xcond_solver_config *config, ocp_qp_xcond_solver_dims *dims)
{
    void *opts = config->opts_assign(config, dims, calloc(1, config->opts_calculate_size(config, dims)));
    config->opts_initialize_default(config, dims, opts);
    return opts;
}

void ocp_qp_xcond_solver_opts_set(ocp_qp_xcond_solver_config *config, void *opts, const char *field, void *value)
{
    config->opts_set(config, opts, field, value);
}

void ocp_qp_xcond_solver_opts_free(void *opts) { free(opts); }

ocp_qp_in *ocp_qp_in_create_from_xcond_dims(ocp_qp_xcond_solver_dims *dims) { return ocp_qp_in_create(dims->orig_dims); }
ocp_qp_out *ocp_qp_out_create_from_xcond_dims(ocp_qp_xcond_solver_dims *dims) { return ocp_qp_out_create(dims->orig_dims); }

/* :513-563 */
acados_size_t ocp_qp_calculate_size(ocp_qp_xcond_solver_config *config, ocp_qp_xcond_solver_dims *dims, void *opts_)
{
    return sizeof(ocp_qp_solver) + config->memory_calculate_size(config, dims, opts_)
           + config->workspace_calculate_size(config, dims, opts_) + 2 * 8;
}

ocp_qp_solver *ocp_qp_assign(ocp_qp_xcond_solver_config *config, ocp_qp_xcond_solver_dims *dims, void *opts_, void *raw_memory)
{
    char *c = (char *) raw_memory;
    ocp_qp_solver *solver = (ocp_qp_solver *) c;
    c = align8(c + sizeof(ocp_qp_solver));
    solver->config = config;
    solver->dims = dims;
    solver->opts = (ocp_qp_xcond_solver_opts *) opts_;
    const acados_size_t msz = config->memory_calculate_size(config, dims, opts_);
    solver->mem = (ocp_qp_xcond_solver_memory *) config->memory_assign(config, dims, opts_, c);
    c = align8(c + msz);
    solver->work = (ocp_qp_xcond_solver_workspace *) c;
    return solver;
}

ocp_qp_solver *ocp_qp_create(ocp_qp_xcond_solver_config *config, ocp_qp_xcond_solver_dims *dims, void *opts_)
{
    config->opts_update(config, dims, opts_);
    return ocp_qp_assign(config, dims, opts_, calloc(1, ocp_qp_calculate_size(config, dims, opts_)));
}

/* releases the device-side resources (the `terminate` slot, ocp_nlp_sqp.c:1009) and the block of ocp_qp_create */
void ocp_qp_solver_destroy(ocp_qp_solver *s)
{
    if (!s) return;
    s->config->terminate(s->config, s->mem, s->work);
    free(s);
}

/* :567-571 */
int ocp_qp_solve(ocp_qp_solver *s, ocp_qp_in *qp_in, ocp_qp_out *qp_out)
{
    return s->config->evaluate(s->config, s->dims, qp_in, qp_out, s->opts, s->mem, s->work);
}

/* RTI split at the solver level (ocp_nlp_sqp_rti.c:509, 1115 reach these two slots) */
int ocp_qp_condense_lhs(ocp_qp_solver *s, ocp_qp_in *qp_in, ocp_qp_out *qp_out)
{
    return s->config->condense_lhs(s->config, s->dims, qp_in, qp_out, s->opts, s->mem, s->work);
}

int ocp_qp_condense_rhs_and_solve(ocp_qp_solver *s, ocp_qp_in *qp_in, ocp_qp_out *qp_out)
{
    return s->config->condense_rhs_and_solve(s->config, s->dims, qp_in, qp_out, s->opts, s->mem, s->work);
}

/* batch extension: n (qp_in, qp_out) pairs of identical structure as ONE device batch (replaces the OpenMP loop of
 * c_templates_tera/acados_solver.in.c:3222-3243 for the QP part) */
int ocp_qp_solve_batch(ocp_qp_solver *s, int n, ocp_qp_in **qp_in, ocp_qp_out **qp_out, int *status)
{
    if (n <= 0) return ACADOS_SUCCESS;
    ocp_qp_partial_condensing_opts *xo = (ocp_qp_partial_condensing_opts *) s->opts->xcond_opts;
    ocp_qp_partial_condensing_memory *xm = (ocp_qp_partial_condensing_memory *) s->mem->xcond_memory;
    cond_request cr = {xm->dims->condensed ? xo->N2 : 0, xm->dims->condensed && xo->block_size_was_set ? xo->block_size : nullptr, 0, 0,
                       xo->full_condensing && !xm->dims->condensed ? 1 : 0};
    std::vector<void *> mems(n, nullptr);
    mems[0] = s->mem->solver_memory;
    return gqp_host::gpu_ipm_evaluate_impl(s->config->qp_solver, n, (void **) qp_in, (void **) qp_out, s->opts->qp_solver_opts,
                                           mems.data(), nullptr, status, &cr);
}

/* :573-595 */
void ocp_qp_xcond_solver_get_scalar(ocp_qp_solver *s, ocp_qp_out *qp_out, const char *field, void *value)
{
    qp_info *info = (qp_info *) qp_out->misc;
    if (!strcmp(field, "time_tot")) *(double *) value = info->total_time;
    else if (!strcmp(field, "time_cond")) *(double *) value = info->condensing_time;
    else s->config->memory_get(s->config, s->mem, field, value);
}

void ocp_qp_solver_eval_forw_sens(ocp_qp_solver *s, ocp_qp_in *qp_in, ocp_qp_seed *seed, ocp_qp_out *sens_out)
{
    s->config->eval_forw_sens(s->config, s->dims, qp_in, seed, sens_out, s->opts, s->mem, s->work);
}

void ocp_qp_solver_eval_adj_sens(ocp_qp_solver *s, ocp_qp_in *qp_in, ocp_qp_seed *seed, ocp_qp_out *sens_out)
{
    s->config->eval_adj_sens(s->config, s->dims, qp_in, seed, sens_out, s->opts, s->mem, s->work);
}

/* outer-level access to the solver_get slot (what ocp_nlp_ddp.c:373-377 does through the xcond vtable) */
void ocp_qp_solver_get_ric(ocp_qp_solver *s, ocp_qp_in *qp_in, ocp_qp_out *qp_out, const char *field, int stage,
                           void *value, int size1, int size2)
{
    s->config->solver_get(s->config, qp_in, qp_out, s->opts, s->mem, field, stage, value, size1, size2);
}

/* :597-610 */
void ocp_qp_solver_get_stats(ocp_qp_solver *s, double *stat_out, const char *qp_solver_name)
{
    int iter, stat_m, rows;
    double *stat;
    s->config->memory_get(s->config, s->mem, "stat_rows", &rows);
    s->config->memory_get(s->config, s->mem, "iter", &iter);
    s->config->memory_get(s->config, s->mem, "stat", &stat);
    s->config->memory_get(s->config, s->mem, "stat_m", &stat_m);
    if (!stat) return;
    if (iter + 1 > rows) iter = rows - 1;
    for (int i = 0; i < stat_m * (iter + 1); i++) stat_out[i] = stat[i];
}

} /* extern "C" */
