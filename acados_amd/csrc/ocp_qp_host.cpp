/*
 * ocp_qp_host.cpp -- acados-shaped host API (include/acados_amd/ocp_qp_interface.h) on top
 * of the device batch C-ABI: the plain containers (dims / in / out / seed / res) and the INNER plugin, i.e. the 17
 * slots of qp_solver_config.  The condensing module, the outer xcond-solver vtable and the acados_c-shaped
 * convenience layer live in ocp_qp_xcond.cpp.  Host code only: no kernels here.  Each function cites the
 * reference function whose contract it follows (paths relative to /root/reference).
 *
 * Memory rule of the reference (ocp_qp_interface.c:513-563: the caller computes sizes and allocates ONE block, the
 * callee carves it; no malloc inside `evaluate`): every host array `evaluate` touches on the single-QP path --
 * structure signature, statistics table, status / iteration slots -- is carved from the block handed to
 * memory_assign.  What is NOT in that block are device-side resources (the HBM batch, its stream, the pinned staging
 * buffers), created on the first evaluate and released by `terminate`; and the per-batch tables of the batch
 * EXTENSION (n > 1, not part of the reference ABI), which are sized once per batch size.
 */
#include "acados_amd/ocp_qp_interface.h"

#include <hip/hip_runtime.h> /* pinned staging buffers only */

#include <algorithm>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>

#include "acados_amd/ocp_qp_gpu_batch.h"
#include "ocp_qp_host_internal.h"

using gqp_host::align8;
using gqp_host::cond_request;

namespace
{

struct gpu_ipm_opts
{
    /* names as d_ocp_qp_ipm_arg_set / ocp_qp_hpipm_opts_set (ocp_qp_hpipm.c:142-183) */
    double mu0, tol_stat, tol_eq, tol_ineq, tol_comp, alpha_min, tau_min, reg_prim, t0_min, lam0_min;
    double tol_comp_soft_scale; /* exit tolerance on complementarity of soft-constrained classes = tol_comp * this (gpu_batch.hip effective_opts) */
    int iter_max, warm_start, cond_pred_corr, print_level, ric_alg, t0_init, update_fact_exit;
    int noticed; /* bit set: options that are accepted without effect and have been announced once */
};

/* one member array of ocp_qp_in / ocp_qp_out and its place in the per-instance bulk blob */
enum { F_A, F_B, F_b, F_Q, F_S, F_R, F_q, F_r, F_lb, F_ub, F_lbm, F_ubm, F_C, F_D, F_lg, F_ug, F_lgm, F_ugm,
       F_Zl, F_Zu, F_zl, F_zu, F_lls, F_lus, F_llsm, F_lusm, O_ux, O_pi, O_lam, O_t };
struct blob_seg { int off, len, fid, k, shift; };

/* device-side resources of one solver memory + the tables of the batch extension */
struct batch_cache
{
    ocp_qp_gpu_batch *batch = nullptr;
    int n = 0;
    int cond_N_sent = -1;            /* condensing request the device batch has been configured for */
    bool cond_solved = false;        /* the condensed child holds the iterate of a previous solve (hot start without qp_out) */
    std::vector<int> blocks_sent;
    std::vector<blob_seg> seg_in, seg_out; /* bulk pack / unpack segment tables (built once per batch) */
    int L_in = 0, L_out = 0;
    double *blob_in = nullptr, *blob_out = nullptr; /* PINNED staging [n][bulk_len] */
    size_t cap_in = 0, cap_out = 0;
    std::vector<int> st, it;         /* batch extension: per-instance status / iterations (n > 1) */
    ~batch_cache()
    {
        if (blob_in) (void) hipHostFree(blob_in);
        if (blob_out) (void) hipHostFree(blob_out);
    }
};

inline const double *in_field(const ocp_qp_in *q, int fid, int k)
{
    switch (fid)
    {
        case F_A: return q->A[k]; case F_B: return q->B[k]; case F_b: return q->b[k];
        case F_Q: return q->Q[k]; case F_S: return q->S[k]; case F_R: return q->R[k];
        case F_q: return q->q[k]; case F_r: return q->r[k];
        case F_lb: return q->lb[k]; case F_ub: return q->ub[k]; case F_lbm: return q->lb_mask[k]; case F_ubm: return q->ub_mask[k];
        case F_C: return q->C[k]; case F_D: return q->D[k];
        case F_lg: return q->lg[k]; case F_ug: return q->ug[k]; case F_lgm: return q->lg_mask[k]; case F_ugm: return q->ug_mask[k];
        case F_Zl: return q->Zl[k]; case F_Zu: return q->Zu[k]; case F_zl: return q->zl[k]; case F_zu: return q->zu[k];
        case F_lls: return q->lls[k]; case F_lus: return q->lus[k]; case F_llsm: return q->lls_mask[k]; case F_lusm: return q->lus_mask[k];
    }
    return nullptr;
}

inline double *out_field(ocp_qp_out *q, int fid, int k)
{
    switch (fid)
    {
        case O_ux: return q->ux[k]; case O_pi: return q->pi[k]; case O_lam: return q->lam[k]; case O_t: return q->t[k];
    }
    return nullptr;
}

/* instances [lo, hi) in parallel on host threads (member arrays of n QPs are copied one by one: the copies of
 * different instances are independent); batch extension only -- one QP is copied by the calling thread */
template <class F>
void par_instances(int n, F f)
{
    int T = (int) std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
    T = std::min(T, (n + 127) / 128);
    if (T <= 1) { f(0, n); return; }
    std::vector<std::thread> th;
    const int chunk = (n + T - 1) / T;
    for (int t = 0; t < T; t++)
    {
        const int lo = t * chunk, hi = std::min(n, lo + chunk);
        if (lo < hi) th.emplace_back(f, lo, hi);
    }
    for (auto &t : th) t.join();
}

void pinned_reserve(double *&p, size_t &cap, size_t cnt)
{
    if (cnt <= cap) return;
    if (p) (void) hipHostFree(p);
    if (hipHostMalloc((void **) &p, cnt * sizeof(double)) != hipSuccess)
    {
        printf("\nerror: ocp_qp_gpu_ipm: cannot allocate %zu bytes of pinned host memory\n", cnt * sizeof(double));
        exit(1);
    }
    memset(p, 0, cnt * sizeof(double));
    cap = cnt;
}

#define GPU_IPM_STAT_M 20

struct gpu_ipm_memory
{
    batch_cache *cache;         /* device-side resources, see the header comment */
    double time_qp_solver_call;
    int iter;
    int status;
    int stat_m;
    /* carved from the caller's block (memory_assign) */
    int *sig, *sig_scratch;     /* structure signature of the QP the device batch was built for / of the incoming one */
    int sig_cap, sig_len;
    double *stat;               /* stat_m x stat_rows, HPIPM-shaped (ocp_qp_hpipm.c:255-297 "stat") */
    int stat_rows;
    int st_it[2];               /* status / iterations of the single-QP path */
};

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int vlen(const ocp_qp_dims *d, const char *f, int k)
{
    const int N = d->N, nx = d->nx[k], nu = d->nu[k], nx1 = k < N ? d->nx[k + 1] : 0;
    if (!strcmp(f, "A")) return nx1 * nx;
    if (!strcmp(f, "B")) return nx1 * nu;
    if (!strcmp(f, "b")) return nx1;
    if (!strcmp(f, "Q")) return nx * nx;
    if (!strcmp(f, "S")) return nu * nx;
    if (!strcmp(f, "R")) return nu * nu;
    if (!strcmp(f, "q")) return nx;
    if (!strcmp(f, "r")) return nu;
    if (!strcmp(f, "C")) return d->ng[k] * nx;
    if (!strcmp(f, "D")) return d->ng[k] * nu;
    return -1;
}

} // namespace

namespace gqp_host
{

int structure_sig_len(const ocp_qp_dims *d)
{
    int len = 1;
    for (int k = 0; k <= d->N; k++) len += 7 + 2 * d->nb[k] + d->ng[k] + d->nbxe[k];
    return len;
}

int structure_sig_fill(const ocp_qp_in *in, int *s)
{
    const ocp_qp_dims *d = in->dim;
    int p = 0;
    s[p++] = d->N;
    for (int k = 0; k <= d->N; k++)
    {
        const int v[] = {d->nx[k], d->nu[k], d->nbx[k], d->nbu[k], d->ng[k], d->ns[k], d->nbxe[k]};
        memcpy(s + p, v, sizeof(v)); p += 7;
        memcpy(s + p, in->idxb[k], sizeof(int) * d->nb[k]); p += d->nb[k];
        memcpy(s + p, in->idxs_rev[k], sizeof(int) * (d->nb[k] + d->ng[k])); p += d->nb[k] + d->ng[k];
        memcpy(s + p, in->idxe[k], sizeof(int) * d->nbxe[k]); p += d->nbxe[k];
    }
    return p;
}

} // namespace gqp_host

extern "C" {

/* ------------------------------------------------------------------ dims */
/* ocp_qp_common.c:100-180 */

acados_size_t ocp_qp_dims_calculate_size(int N) { return sizeof(ocp_qp_dims) + 10 * (N + 1) * sizeof(int) + 16; }

ocp_qp_dims *ocp_qp_dims_assign(int N, void *raw_memory)
{
    char *c = (char *) raw_memory;
    ocp_qp_dims *d = (ocp_qp_dims *) c;
    c = align8(c + sizeof(ocp_qp_dims));
    int **arr[] = {&d->nx, &d->nu, &d->nb, &d->nbx, &d->nbu, &d->ng, &d->ns, &d->nbxe, &d->nbue, &d->nge};
    for (int q = 0; q < 10; q++)
    {
        *arr[q] = (int *) c;
        memset(c, 0, sizeof(int) * (N + 1));
        c += sizeof(int) * (N + 1);
    }
    d->N = N;
    return d;
}

ocp_qp_dims *ocp_qp_dims_create(int N) { return ocp_qp_dims_assign(N, calloc(1, ocp_qp_dims_calculate_size(N))); }
void ocp_qp_dims_free(void *d) { free(d); }

void ocp_qp_dims_set(void *config_, void *dims_, int stage, const char *field, int *value)
{
    ocp_qp_dims *d = (ocp_qp_dims *) dims_;
    int *dst = nullptr;
    if (!strcmp(field, "nx")) dst = d->nx;
    else if (!strcmp(field, "nu")) dst = d->nu;
    else if (!strcmp(field, "nbx")) dst = d->nbx;
    else if (!strcmp(field, "nbu")) dst = d->nbu;
    else if (!strcmp(field, "ng")) dst = d->ng;
    else if (!strcmp(field, "ns")) dst = d->ns;
    else if (!strcmp(field, "nbxe")) dst = d->nbxe;
    else if (!strcmp(field, "nbue")) dst = d->nbue;
    else if (!strcmp(field, "nge")) dst = d->nge;
    else
    {
        printf("\nerror: ocp_qp_dims_set: field %s not available\n", field);
        exit(1);
    }
    dst[stage] = *value;
    d->nb[stage] = d->nbx[stage] + d->nbu[stage];
}

void ocp_qp_dims_get(void *config_, void *dims_, int stage, const char *field, int *value)
{
    ocp_qp_dims *d = (ocp_qp_dims *) dims_;
    if (!strcmp(field, "nx")) *value = d->nx[stage];
    else if (!strcmp(field, "nu")) *value = d->nu[stage];
    else if (!strcmp(field, "nbx")) *value = d->nbx[stage];
    else if (!strcmp(field, "nbu")) *value = d->nbu[stage];
    else if (!strcmp(field, "nb")) *value = d->nb[stage];
    else if (!strcmp(field, "ng")) *value = d->ng[stage];
    else if (!strcmp(field, "ns")) *value = d->ns[stage];
    else if (!strcmp(field, "nbxe")) *value = d->nbxe[stage];
    else if (!strcmp(field, "nbue")) *value = d->nbue[stage];
    else if (!strcmp(field, "nge")) *value = d->nge[stage];
    else
    {
        printf("\nerror: ocp_qp_dims_get: field %s not available\n", field);
        exit(1);
    }
}

/* -------------------------------------------------------------------- in */
/* ocp_qp_common.c:189-211 (one block, bump assignment, 8-byte alignment asserted) */

static size_t in_walk(ocp_qp_dims *d, ocp_qp_in *in_or_null, char *base)
{
    const int N = d->N, NS = N + 1;
    char *c = base;
    ocp_qp_in sizing_only;              /* size computation walks a scratch struct */
    const bool in = in_or_null != nullptr;
    ocp_qp_in *w = in ? in_or_null : &sizing_only;
    auto tabs_d = [&](double ***t) { if (in) *t = (double **) c; c += sizeof(double *) * NS; };
    auto tabs_i = [&](int ***t) { if (in) *t = (int **) c; c += sizeof(int *) * NS; };
    double ***dt[] = {&w->A, &w->B, &w->b, &w->Q, &w->S, &w->R, &w->q, &w->r, &w->lb, &w->ub, &w->lb_mask,
                      &w->ub_mask, &w->C, &w->D, &w->lg, &w->ug, &w->lg_mask, &w->ug_mask, &w->Zl, &w->Zu,
                      &w->zl, &w->zu, &w->lls, &w->lus, &w->lls_mask, &w->lus_mask};
    const int ndt = sizeof(dt) / sizeof(dt[0]);
    int ***it[] = {&w->idxb, &w->idxs_rev, &w->idxe};
    for (int q = 0; q < ndt; q++) tabs_d(dt[q]);
    for (int q = 0; q < 3; q++) tabs_i(it[q]);
    c = align8(c);
    for (int k = 0; k <= N; k++)
    {
        const int nx = d->nx[k], nu = d->nu[k], nx1 = k < N ? d->nx[k + 1] : 0, nb = d->nb[k], ng = d->ng[k], ns = d->ns[k];
        const int len[] = {nx1 * nx, nx1 * nu, nx1, nx * nx, nu * nx, nu * nu, nx, nu, nb, nb, nb, nb, ng * nx, ng * nu,
                           ng, ng, ng, ng, ns, ns, ns, ns, ns, ns, ns, ns};
        for (int q = 0; q < ndt; q++)
        {
            if (in)
            {
                (*dt[q])[k] = (double *) c;
                const bool is_mask = q == 10 || q == 11 || q == 16 || q == 17 || q == 24 || q == 25;
                for (int e = 0; e < len[q]; e++) ((double *) c)[e] = is_mask ? 1.0 : 0.0;
            }
            c += sizeof(double) * len[q];
        }
        const int ilen[] = {nb, nb + ng, nb};
        for (int q = 0; q < 3; q++)
        {
            if (in) (*it[q])[k] = (int *) c;
            c = align8(c + sizeof(int) * ilen[q]);
        }
        if (in)
        {
            for (int e = 0; e < d->nbu[k]; e++) w->idxb[k][e] = e;
            for (int e = 0; e < d->nbx[k]; e++) w->idxb[k][d->nbu[k] + e] = nu + e;
            for (int e = 0; e < nb + ng; e++) w->idxs_rev[k][e] = -1;
            for (int e = 0; e < nb; e++) w->idxe[k][e] = 0;
        }
    }
    return (size_t) (c - base);
}

acados_size_t ocp_qp_in_calculate_size(ocp_qp_dims *dims)
{
    return sizeof(ocp_qp_in) + 8 + in_walk(dims, nullptr, nullptr) + 8;
}

ocp_qp_in *ocp_qp_in_assign(ocp_qp_dims *dims, void *raw_memory)
{
    ocp_qp_in *in = (ocp_qp_in *) raw_memory;
    in->dim = dims;
    in_walk(dims, in, align8((char *) raw_memory + sizeof(ocp_qp_in)));
    return in;
}

ocp_qp_in *ocp_qp_in_create(ocp_qp_dims *dims) { return ocp_qp_in_assign(dims, calloc(1, ocp_qp_in_calculate_size(dims))); }
void ocp_qp_in_free(void *in) { free(in); }

/* ocp_qp_interface.c:405-409 -> d_ocp_qp_set(field, stage, value, in) */
void ocp_qp_in_set(void *config, ocp_qp_in *in, int k, char *field, void *value)
{
    const ocp_qp_dims *d = in->dim;
    const char *f = field;
    const int nbu = d->nbu[k], nbx = d->nbx[k], nb = d->nb[k], ng = d->ng[k], ns = d->ns[k], nu = d->nu[k];
    auto cpd = [&](double *dst, int n) { memcpy(dst, value, sizeof(double) * n); };
    auto cpi = [&](int *dst, int n) { memcpy(dst, value, sizeof(int) * n); };
    int n;
    if ((n = vlen(d, f, k)) >= 0)
    {
        double **tab = !strcmp(f, "A") ? in->A : !strcmp(f, "B") ? in->B : !strcmp(f, "b") ? in->b : !strcmp(f, "Q") ? in->Q
                     : !strcmp(f, "S") ? in->S : !strcmp(f, "R") ? in->R : !strcmp(f, "q") ? in->q : !strcmp(f, "r") ? in->r
                     : !strcmp(f, "C") ? in->C : in->D;
        cpd(tab[k], n);
    }
    else if (!strcmp(f, "idxb")) cpi(in->idxb[k], nb);
    else if (!strcmp(f, "idxbu")) cpi(in->idxb[k], nbu);
    else if (!strcmp(f, "idxbx")) { const int *v = (const int *) value; for (int e = 0; e < nbx; e++) in->idxb[k][nbu + e] = nu + v[e]; }
    else if (!strcmp(f, "lb")) cpd(in->lb[k], nb);
    else if (!strcmp(f, "ub")) cpd(in->ub[k], nb);
    else if (!strcmp(f, "lbu")) cpd(in->lb[k], nbu);
    else if (!strcmp(f, "ubu")) cpd(in->ub[k], nbu);
    else if (!strcmp(f, "lbx")) cpd(in->lb[k] + nbu, nbx);
    else if (!strcmp(f, "ubx")) cpd(in->ub[k] + nbu, nbx);
    else if (!strcmp(f, "lbu_mask")) cpd(in->lb_mask[k], nbu);
    else if (!strcmp(f, "ubu_mask")) cpd(in->ub_mask[k], nbu);
    else if (!strcmp(f, "lbx_mask")) cpd(in->lb_mask[k] + nbu, nbx);
    else if (!strcmp(f, "ubx_mask")) cpd(in->ub_mask[k] + nbu, nbx);
    else if (!strcmp(f, "lg")) cpd(in->lg[k], ng);
    else if (!strcmp(f, "ug")) cpd(in->ug[k], ng);
    else if (!strcmp(f, "lg_mask")) cpd(in->lg_mask[k], ng);
    else if (!strcmp(f, "ug_mask")) cpd(in->ug_mask[k], ng);
    else if (!strcmp(f, "Zl")) cpd(in->Zl[k], ns);
    else if (!strcmp(f, "Zu")) cpd(in->Zu[k], ns);
    else if (!strcmp(f, "zl")) cpd(in->zl[k], ns);
    else if (!strcmp(f, "zu")) cpd(in->zu[k], ns);
    else if (!strcmp(f, "lls")) cpd(in->lls[k], ns);
    else if (!strcmp(f, "lus")) cpd(in->lus[k], ns);
    else if (!strcmp(f, "lls_mask")) cpd(in->lls_mask[k], ns);
    else if (!strcmp(f, "lus_mask")) cpd(in->lus_mask[k], ns);
    else if (!strcmp(f, "idxs_rev")) cpi(in->idxs_rev[k], nb + ng);
    else if (!strcmp(f, "idxe") || !strcmp(f, "idxbxe")) cpi(in->idxe[k], d->nbxe[k]);
    else
    {
        printf("\nerror: ocp_qp_in_set: field %s not available\n", f);
        exit(1);
    }
}

/* ------------------------------------------------------------------- out */
/* ocp_qp_common.c:234-260 */

static size_t out_walk(ocp_qp_dims *d, ocp_qp_out *out, char *base)
{
    const int N = d->N;
    char *c = base;
    if (out) { out->ux = (double **) c; } c += sizeof(double *) * (N + 1);
    if (out) { out->pi = (double **) c; } c += sizeof(double *) * (N + 1);
    if (out) { out->lam = (double **) c; } c += sizeof(double *) * (N + 1);
    if (out) { out->t = (double **) c; } c += sizeof(double *) * (N + 1);
    c = align8(c);
    for (int k = 0; k <= N; k++)
    {
        const int nv = d->nu[k] + d->nx[k] + 2 * d->ns[k], nx1 = k < N ? d->nx[k + 1] : 0;
        const int nct = 2 * (d->nb[k] + d->ng[k] + d->ns[k]);
        if (out) { out->ux[k] = (double *) c; out->pi[k] = (double *) (c + sizeof(double) * nv); }
        c += sizeof(double) * (nv + nx1);
        if (out) { out->lam[k] = (double *) c; out->t[k] = (double *) (c + sizeof(double) * nct); }
        c += sizeof(double) * 2 * nct;
    }
    c = align8(c);
    if (out) out->misc = c;
    c += sizeof(qp_info);
    return (size_t) (c - base);
}

acados_size_t ocp_qp_out_calculate_size(ocp_qp_dims *dims) { return sizeof(ocp_qp_out) + 8 + out_walk(dims, nullptr, nullptr) + 8; }

ocp_qp_out *ocp_qp_out_assign(ocp_qp_dims *dims, void *raw_memory)
{
    ocp_qp_out *out = (ocp_qp_out *) raw_memory;
    out->dim = dims;
    out_walk(dims, out, align8((char *) raw_memory + sizeof(ocp_qp_out)));
    return out;
}

ocp_qp_out *ocp_qp_out_create(ocp_qp_dims *dims) { return ocp_qp_out_assign(dims, calloc(1, ocp_qp_out_calculate_size(dims))); }
void ocp_qp_out_free(void *out) { free(out); }

/* ocp_qp_interface.c:435-480 */
void ocp_qp_out_get(ocp_qp_out *out, int k, const char *field, void *value)
{
    const ocp_qp_dims *d = out->dim;
    const int nu = d->nu[k], nx = d->nx[k], ns = d->ns[k];
    double *v = (double *) value;
    if (!strcmp(field, "qp_info")) *(qp_info **) value = (qp_info *) out->misc;
    else if (!strcmp(field, "x")) memcpy(v, out->ux[k] + nu, sizeof(double) * nx);
    else if (!strcmp(field, "u")) memcpy(v, out->ux[k], sizeof(double) * nu);
    else if (!strcmp(field, "sl")) memcpy(v, out->ux[k] + nu + nx, sizeof(double) * ns);
    else if (!strcmp(field, "su")) memcpy(v, out->ux[k] + nu + nx + ns, sizeof(double) * ns);
    else if (!strcmp(field, "pi")) memcpy(v, out->pi[k], sizeof(double) * (k < d->N ? d->nx[k + 1] : 0));
    else if (!strcmp(field, "lam")) memcpy(v, out->lam[k], sizeof(double) * 2 * (d->nb[k] + d->ng[k] + ns));
    else if (!strcmp(field, "t")) memcpy(v, out->t[k], sizeof(double) * 2 * (d->nb[k] + d->ng[k] + ns));
    else
    {
        printf("\nerror: ocp_qp_out_get: field %s not available\n", field);
        exit(1);
    }
}

/* d_ocp_qp_seed (acados/ocp_qp/ocp_qp_common.h: ocp_qp_seed; ocp_qp_common.c:263-300): seed_g [r; q; zl; zu],
 * seed_b, seed_d [lb lg ub ug ls us] (natural sign), seed_m; one block, zero-initialised */
static size_t seed_doubles(const ocp_qp_dims *dims)
{
    const int N = dims->N;
    size_t cnt = 0;
    for (int k = 0; k <= N; k++)
        cnt += dims->nu[k] + dims->nx[k] + 2 * dims->ns[k] + (k < N ? dims->nx[k + 1] : 0) + 4 * (size_t) (dims->nb[k] + dims->ng[k] + dims->ns[k]);
    return cnt;
}

acados_size_t ocp_qp_seed_calculate_size(ocp_qp_dims *dims)
{
    return sizeof(ocp_qp_seed) + 4 * (dims->N + 1) * sizeof(double *) + seed_doubles(dims) * sizeof(double) + 64;
}

ocp_qp_seed *ocp_qp_seed_assign(ocp_qp_dims *dims, void *raw_memory)
{
    const int N = dims->N;
    char *raw = (char *) raw_memory;
    ocp_qp_seed *sd = (ocp_qp_seed *) raw;
    raw = align8(raw + sizeof(ocp_qp_seed));
    sd->dim = dims;
    sd->seed_g = (double **) raw; raw += (N + 1) * sizeof(double *);
    sd->seed_b = (double **) raw; raw += (N + 1) * sizeof(double *);
    sd->seed_d = (double **) raw; raw += (N + 1) * sizeof(double *);
    sd->seed_m = (double **) raw; raw += (N + 1) * sizeof(double *);
    double *p = (double *) align8(raw);
    memset(p, 0, seed_doubles(dims) * sizeof(double));
    for (int k = 0; k <= N; k++)
    {
        const int nct = 2 * (dims->nb[k] + dims->ng[k] + dims->ns[k]);
        sd->seed_g[k] = p; p += dims->nu[k] + dims->nx[k] + 2 * dims->ns[k];
        sd->seed_b[k] = p; p += k < N ? dims->nx[k + 1] : 0;
        sd->seed_d[k] = p; p += nct;
        sd->seed_m[k] = p; p += nct;
    }
    return sd;
}

ocp_qp_seed *ocp_qp_seed_create(ocp_qp_dims *dims) { return ocp_qp_seed_assign(dims, calloc(1, ocp_qp_seed_calculate_size(dims))); }

void ocp_qp_seed_free(void *seed) { free(seed); }

/* ocp_qp_common.c:874-921, line by line on the plain containers */
void ocp_qp_compute_t(ocp_qp_in *in, ocp_qp_out *out)
{
    const ocp_qp_dims *d = in->dim;
    for (int k = 0; k <= d->N; k++)
    {
        const int nu = d->nu[k], nx = d->nx[k], nb = d->nb[k], ng = d->ng[k], ns = d->ns[k], nbg = nb + ng;
        const double *ux = out->ux[k];
        double *t = out->t[k];
        for (int i = 0; i < nb; i++)
        {
            const double c = ux[in->idxb[k][i]];
            t[i] = c - in->lb[k][i];
            t[nbg + i] = in->ub[k][i] - c;
        }
        for (int g = 0; g < ng; g++)
        {
            double c = 0.0;
            for (int j = 0; j < nu; j++) c += in->D[k][g + ng * j] * ux[j];
            for (int j = 0; j < nx; j++) c += in->C[k][g + ng * j] * ux[nu + j];
            t[nb + g] = c - in->lg[k][g];
            t[nbg + nb + g] = in->ug[k][g] - c;
        }
        for (int i = 0; i < nbg; i++)
        {
            const int idx = in->idxs_rev[k][i];
            if (idx != -1)
            {
                t[i] += ux[nu + nx + idx];
                t[nbg + i] += ux[nu + nx + ns + idx];
            }
        }
        for (int j = 0; j < ns; j++)
        {
            t[2 * nbg + j] = ux[nu + nx + j] - in->lls[k][j];
            t[2 * nbg + ns + j] = ux[nu + nx + ns + j] - in->lus[k][j];
        }
    }
}


/* ------------------------------------------------------ KKT residuals */
/* ocp_qp_res / ocp_qp_res_ws and ocp_qp_res_compute / _nrm_inf (ocp_qp_common.c:497-667): the residual vectors of an
 * arbitrary (qp_in, qp_out) pair, evaluated ON THE DEVICE by a kernel that shares nothing with the IPM sweeps
 * (res_kernels.hpp).  The workspace owns a one-instance device batch (device-side resource: release it with
 * ocp_qp_res_workspace_free, or use ocp_qp_inf_norm_residuals, which creates and releases one per call like
 * ocp_qp_interface.c:642-650 does with its malloc'ed pair). */
struct ocp_qp_res_ws_
{
    gqp_host::single_batch sb;
};

static size_t res_doubles(const ocp_qp_dims *d)
{
    size_t cnt = 0;
    for (int k = 0; k <= d->N; k++)
        cnt += d->nu[k] + d->nx[k] + 2 * d->ns[k] + (k < d->N ? d->nx[k + 1] : 0) + 4 * (size_t) (d->nb[k] + d->ng[k] + d->ns[k]);
    return cnt;
}

acados_size_t ocp_qp_res_calculate_size(ocp_qp_dims *dims)
{
    return sizeof(ocp_qp_res) + 4 * (dims->N + 1) * sizeof(double *) + res_doubles(dims) * sizeof(double) + 64;
}

ocp_qp_res *ocp_qp_res_assign(ocp_qp_dims *dims, void *raw_memory)
{
    const int N = dims->N;
    char *raw = (char *) raw_memory;
    ocp_qp_res *r = (ocp_qp_res *) raw;
    raw = align8(raw + sizeof(ocp_qp_res));
    r->dim = dims;
    r->res_g = (double **) raw; raw += (N + 1) * sizeof(double *);
    r->res_b = (double **) raw; raw += (N + 1) * sizeof(double *);
    r->res_d = (double **) raw; raw += (N + 1) * sizeof(double *);
    r->res_m = (double **) raw; raw += (N + 1) * sizeof(double *);
    double *p = (double *) align8(raw);
    for (int k = 0; k <= N; k++)
    {
        const int nct = 2 * (dims->nb[k] + dims->ng[k] + dims->ns[k]);
        r->res_g[k] = p; p += dims->nu[k] + dims->nx[k] + 2 * dims->ns[k];
        r->res_b[k] = p; p += k < N ? dims->nx[k + 1] : 0;
        r->res_d[k] = p; p += nct;
        r->res_m[k] = p; p += nct;
    }
    return r;
}

ocp_qp_res *ocp_qp_res_create(ocp_qp_dims *dims) { return ocp_qp_res_assign(dims, calloc(1, ocp_qp_res_calculate_size(dims))); }
void ocp_qp_res_free(void *res) { free(res); }

acados_size_t ocp_qp_res_workspace_calculate_size(ocp_qp_dims *dims) { return sizeof(ocp_qp_res_ws) + 8; }

ocp_qp_res_ws *ocp_qp_res_workspace_assign(ocp_qp_dims *dims, void *raw_memory)
{
    ocp_qp_res_ws *ws = (ocp_qp_res_ws *) align8((char *) raw_memory);
    memset(ws, 0, sizeof(*ws));
    return ws;
}

ocp_qp_res_ws *ocp_qp_res_workspace_create(ocp_qp_dims *dims)
{
    return ocp_qp_res_workspace_assign(dims, calloc(1, ocp_qp_res_workspace_calculate_size(dims)));
}

/* releases the device batch and the block obtained from ocp_qp_res_workspace_create */
void ocp_qp_res_workspace_free(ocp_qp_res_ws *ws)
{
    if (!ws) return;
    gqp_host::single_batch_free(&ws->sb);
    free(ws);
}

/* ocp_qp_common.c:559-594 */
void ocp_qp_res_compute(ocp_qp_in *qp_in, ocp_qp_out *qp_out, ocp_qp_res *qp_res, ocp_qp_res_ws *res_ws)
{
    qp_info *info = (qp_info *) qp_out->misc;
    if (info && info->t_computed == 0)
    {
        ocp_qp_compute_t(qp_in, qp_out);
        info->t_computed = 1;
    }
    if (gqp_host::single_batch_load_in(&res_ws->sb, qp_in) != 0)
    {
        printf("\nerror: ocp_qp_res_compute: no GPU batch could be created (no device or unsupported shape)\n");
        exit(1);
    }
    ocp_qp_gpu_batch *b = res_ws->sb.batch;
    const ocp_qp_dims *d = qp_in->dim;
    gqp_host::single_batch_push_out(b, d, qp_out);
    if (ocp_qp_gpu_batch_res_compute(b) != 0) exit(1);
    for (int k = 0; k <= d->N; k++)
    {
        const int nv = d->nu[k] + d->nx[k];
        if (nv) ocp_qp_gpu_batch_get(b, "res_g", k, qp_res->res_g[k], 0);
        if (d->ns[k]) ocp_qp_gpu_batch_get(b, "res_gs", k, qp_res->res_g[k] + nv, 0);
        if (k < d->N && d->nx[k + 1]) ocp_qp_gpu_batch_get(b, "res_b", k, qp_res->res_b[k], 0);
        if (d->nb[k] + d->ng[k] + d->ns[k])
        {
            ocp_qp_gpu_batch_get(b, "res_d", k, qp_res->res_d[k], 0);
            ocp_qp_gpu_batch_get(b, "res_m", k, qp_res->res_m[k], 0);
        }
    }
}

/* ocp_qp_common.c:598-667 */
void ocp_qp_res_compute_nrm_inf(ocp_qp_res *qp_res, double res[4])
{
    const ocp_qp_dims *d = qp_res->dim;
    auto nrm = [](const double *v, int n, double &acc) {
        for (int e = 0; e < n; e++)
        {
            const double a = v[e] < 0.0 ? -v[e] : v[e];
            if (a > acc || v[e] != v[e]) acc = a;
        }
    };
    res[0] = res[1] = res[2] = res[3] = 0.0;
    for (int k = 0; k <= d->N; k++)
    {
        const int nct = 2 * (d->nb[k] + d->ng[k] + d->ns[k]);
        nrm(qp_res->res_g[k], d->nu[k] + d->nx[k] + 2 * d->ns[k], res[0]);
        if (k < d->N) nrm(qp_res->res_b[k], d->nx[k + 1], res[1]);
        nrm(qp_res->res_d[k], nct, res[2]);
        nrm(qp_res->res_m[k], nct, res[3]);
    }
}

/* ocp_qp_interface.c:642-650 */
void ocp_qp_inf_norm_residuals(ocp_qp_dims *dims, ocp_qp_in *qp_in, ocp_qp_out *qp_out, double *res)
{
    ocp_qp_res *qp_res = ocp_qp_res_create(dims);
    ocp_qp_res_ws *res_ws = ocp_qp_res_workspace_create(dims);
    ocp_qp_res_compute(qp_in, qp_out, qp_res, res_ws);
    ocp_qp_res_compute_nrm_inf(qp_res, res);
    ocp_qp_res_free(qp_res);
    ocp_qp_res_workspace_free(res_ws);
}

} /* extern "C" */

/* ------------------------------------ one QP <-> a one-instance device batch */
namespace gqp_host
{

struct cfield { const char *name; int fid; int dyn; int matrix; };
static const cfield k_cfields[] = {
    {"A", F_A, 1, 1}, {"B", F_B, 1, 1}, {"b", F_b, 1, 0}, {"Q", F_Q, 0, 1}, {"S", F_S, 0, 1}, {"R", F_R, 0, 1}, {"q", F_q, 0, 0},
    {"r", F_r, 0, 0}, {"C", F_C, 0, 1}, {"D", F_D, 0, 1}, {"lg", F_lg, 0, 0}, {"ug", F_ug, 0, 0}, {"lg_mask", F_lgm, 0, 0},
    {"ug_mask", F_ugm, 0, 0}, {"Zl", F_Zl, 0, 1}, {"Zu", F_Zu, 0, 1}, {"zl", F_zl, 0, 0}, {"zu", F_zu, 0, 0}, {"lls", F_lls, 0, 0},
    {"lus", F_lus, 0, 0}, {"lls_mask", F_llsm, 0, 0}, {"lus_mask", F_lusm, 0, 0}};

static int clen(const ocp_qp_dims *d, const char *f, int k)
{
    const int n = vlen(d, f, k);
    if (n >= 0) return n;
    return (f[0] == 'l' || f[0] == 'u') && f[1] == 'g' ? d->ng[k] : d->ns[k]; /* lg ug (+ masks) | Zl Zu zl zu lls lus (+ masks) */
}

/* bounds and their masks: the containers keep [bu; bx] in one array, the device batch takes them per kind */
static void bounds_xfer(ocp_qp_gpu_batch *b, const ocp_qp_in *q, int k, bool to_device)
{
    const ocp_qp_dims *d = q->dim;
    const int nbu = d->nbu[k];
    struct { const char *name; double *p; int n; } f[] = {
        {"lbu", q->lb[k], nbu}, {"ubu", q->ub[k], nbu}, {"lbu_mask", q->lb_mask[k], nbu}, {"ubu_mask", q->ub_mask[k], nbu},
        {"lbx", q->lb[k] + nbu, d->nbx[k]}, {"ubx", q->ub[k] + nbu, d->nbx[k]},
        {"lbx_mask", q->lb_mask[k] + nbu, d->nbx[k]}, {"ubx_mask", q->ub_mask[k] + nbu, d->nbx[k]}};
    for (auto &e : f)
    {
        if (e.n <= 0) continue;
        if (to_device) ocp_qp_gpu_batch_set(b, e.name, k, e.p, 0);
        else ocp_qp_gpu_batch_get(b, e.name, k, e.p, 0);
    }
}

int single_batch_load_in(single_batch *sb, const ocp_qp_in *in)
{
    const ocp_qp_dims *d = in->dim;
    const int len = structure_sig_len(d);
    std::vector<int> sig(len);
    structure_sig_fill(in, sig.data());
    if (!sb->batch || sb->sig_len != len || memcmp(sb->sig, sig.data(), sizeof(int) * len) != 0)
    {
        single_batch_free(sb);
        sb->batch = ocp_qp_gpu_batch_create(d->N, d->nx, d->nu, d->nbx, d->nbu, d->ng, d->ns, 1, -1);
        if (!sb->batch) return -1;
        sb->generation++;
        sb->sig = (int *) malloc(sizeof(int) * len);
        memcpy(sb->sig, sig.data(), sizeof(int) * len);
        sb->sig_len = len;
        for (int k = 0; k <= d->N; k++)
        {
            ocp_qp_gpu_batch_set_int(sb->batch, "idxb", k, in->idxb[k], d->nb[k]);
            ocp_qp_gpu_batch_set_int(sb->batch, "idxs_rev", k, in->idxs_rev[k], d->nb[k] + d->ng[k]);
            ocp_qp_gpu_batch_set_int(sb->batch, "idxe", k, in->idxe[k], d->nbxe[k]);
        }
    }
    ocp_qp_gpu_batch *b = sb->batch;
    for (int k = 0; k <= d->N; k++)
    {
        for (const cfield &f : k_cfields)
            if (!(f.dyn && k == d->N) && clen(d, f.name, k) > 0) ocp_qp_gpu_batch_set(b, f.name, k, in_field(in, f.fid, k), 0);
        bounds_xfer(b, in, k, true);
    }
    return 0;
}

void single_batch_read_in(ocp_qp_gpu_batch *b, ocp_qp_in *in, int what)
{
    const ocp_qp_dims *d = in->dim;
    for (int k = 0; k <= d->N; k++)
    {
        for (const cfield &f : k_cfields)
            if (!(f.dyn && k == d->N) && clen(d, f.name, k) > 0 && (what & (f.matrix ? 1 : 2)))
                ocp_qp_gpu_batch_get(b, f.name, k, const_cast<double *>(in_field(in, f.fid, k)), 0);
        if (what & 2) bounds_xfer(b, in, k, false);
        if (what & 1)
        {
            ocp_qp_gpu_batch_get_int(b, "idxb", k, in->idxb[k]);
            ocp_qp_gpu_batch_get_int(b, "idxs_rev", k, in->idxs_rev[k]);
            ocp_qp_gpu_batch_get_int(b, "idxe", k, in->idxe[k]);
        }
    }
}

static void solution_xfer(ocp_qp_gpu_batch *b, const ocp_qp_dims *d, ocp_qp_out *o, bool to_device)
{
    for (int k = 0; k <= d->N; k++)
    {
        const int nu = d->nu[k], nx = d->nx[k], ns = d->ns[k], nct = 2 * (d->nb[k] + d->ng[k] + ns);
        struct { const char *name; double *p; int n; } f[] = {
            {"u", o->ux[k], nu}, {"x", o->ux[k] + nu, nx}, {"sl", o->ux[k] + nu + nx, ns}, {"su", o->ux[k] + nu + nx + ns, ns},
            {"pi", k < d->N ? o->pi[k] : nullptr, k < d->N ? d->nx[k + 1] : 0}, {"lam", o->lam[k], nct}, {"t", o->t[k], nct}};
        for (auto &e : f)
        {
            if (e.n <= 0) continue;
            if (to_device) ocp_qp_gpu_batch_set(b, e.name, k, e.p, 0);
            else ocp_qp_gpu_batch_get(b, e.name, k, e.p, 0);
        }
    }
}

void single_batch_push_out(ocp_qp_gpu_batch *b, const ocp_qp_dims *d, const ocp_qp_out *out)
{
    solution_xfer(b, d, const_cast<ocp_qp_out *>(out), true);
}

void single_batch_pull_out(ocp_qp_gpu_batch *b, const ocp_qp_dims *d, ocp_qp_out *out) { solution_xfer(b, d, out, false); }

void single_batch_free(single_batch *sb)
{
    if (sb->batch) ocp_qp_gpu_batch_destroy(sb->batch);
    free(sb->sig);
    sb->batch = nullptr; sb->sig = nullptr; sb->sig_len = 0;
}

} // namespace gqp_host

extern "C" {

/* ------------------------------------------------- inner plugin (vtable) */
/* ocp_qp_hpipm.c:60-540 */

acados_size_t ocp_qp_gpu_ipm_opts_calculate_size(void *config, void *dims) { return sizeof(gpu_ipm_opts) + 8; }

void *ocp_qp_gpu_ipm_opts_assign(void *config, void *dims, void *raw_memory) { return align8((char *) raw_memory); }

static void gpu_ipm_mode_defaults(gpu_ipm_opts *o)
{
    /* mode BALANCE + the acados overrides of ocp_qp_hpipm.c:101-113 (ocp_qp_hpipm_opts_overwrite_mode_opts: the four
     * tolerances, iter_max, alpha_min, mu0 are re-applied after EVERY mode change, so the modes do not differ in them) */
    o->mu0 = 1e0; o->tol_stat = 1e-6; o->tol_eq = 1e-8; o->tol_ineq = 1e-8; o->tol_comp = 1e-8;
    o->alpha_min = 1e-8; o->reg_prim = 1e-15; o->t0_min = 1e-16; o->lam0_min = 1e-16;
    o->iter_max = 50; o->warm_start = 0; o->cond_pred_corr = 1; o->ric_alg = 1;
    o->t0_init = 2; o->update_fact_exit = 0;
}

void ocp_qp_gpu_ipm_opts_initialize_default(void *config, void *dims, void *opts_)
{
    gpu_ipm_opts *o = (gpu_ipm_opts *) opts_;
    gpu_ipm_mode_defaults(o);
    o->print_level = 0;
    o->tau_min = 0.0; /* m_relax, ocp_qp_hpipm.c:126 */
    o->tol_comp_soft_scale = 1.0; /* opt-in (< 1): tighter complementarity exit for soft-constrained QPs; 1 = the reference's semantics */
    o->noticed = 0;
}

void ocp_qp_gpu_ipm_opts_update(void *config, void *dims, void *opts) {}

/* options that are part of the HPIPM-named surface but select nothing in this backend: said once per opts object */
static void notice_once(gpu_ipm_opts *o, int bit, const char *field, const char *what)
{
    if (o->noticed & (1 << bit)) return;
    o->noticed |= 1 << bit;
    printf("acados_amd: ocp_qp_gpu_ipm: option %s accepted, %s\n", field, what);
}

void ocp_qp_gpu_ipm_opts_set(void *config, void *opts_, const char *field, void *value)
{
    gpu_ipm_opts *o = (gpu_ipm_opts *) opts_;
    const double *d = (const double *) value;
    const int *i = (const int *) value;
    if (!strcmp(field, "hpipm_mode"))
    {
        const char *mode = (const char *) value;
        if (strcmp(mode, "BALANCE") && strcmp(mode, "SPEED") && strcmp(mode, "SPEED_ABS") && strcmp(mode, "ROBUST"))
        {
            printf("ocp_qp_gpu_ipm_opts_set: got non-supported mode %s\n", mode);
            exit(1);
        }
        /* a mode change re-applies the acados overrides (ocp_qp_hpipm.c:146-165); print_level and tau_min (m_relax)
         * live outside the HPIPM argument struct there and survive it */
        gpu_ipm_mode_defaults(o);
        /* What d_ocp_qp_ipm_arg_set_default(mode) changes BEHIND the acados overrides is HPIPM-internal (upstream
         * knowledge; sources absent here): SPEED_ABS = no conditional corrector (cond_pred_corr 0), absolute-form residuals;
         * SPEED = conditional corrector, no iterative refinement; BALANCE = + 2 refinement steps, LQ fallback; ROBUST = 4
         * refinement steps, LQ factorisation.  This backend has the corrector switch -- mapped -- and neither iterative
         * refinement nor an LQ factorisation of the KKT blocks: SPEED == BALANCE == ROBUST here, said once. */
        if (!strcmp(mode, "SPEED_ABS")) o->cond_pred_corr = 0;
        if (!strcmp(mode, "SPEED_ABS") || !strcmp(mode, "ROBUST"))
            notice_once(o, 0, "hpipm_mode", !strcmp(mode, "ROBUST")
                            ? "iterative refinement / LQ factorisation of ROBUST have no counterpart here: same arithmetic as BALANCE"
                            : "SPEED_ABS maps to the unconditional corrector (cond_pred_corr 0); residuals stay in relative form");
    }
    else if (!strcmp(field, "print_level")) o->print_level = *i;
    else if (!strcmp(field, "tau_min")) o->tau_min = *d;
    else if (!strcmp(field, "iter_max")) o->iter_max = *i;
    else if (!strcmp(field, "tol_stat")) o->tol_stat = *d;
    else if (!strcmp(field, "tol_eq")) o->tol_eq = *d;
    else if (!strcmp(field, "tol_ineq")) o->tol_ineq = *d;
    else if (!strcmp(field, "tol_comp")) o->tol_comp = *d;
    else if (!strcmp(field, "warm_start")) o->warm_start = *i;
    else if (!strcmp(field, "mu0")) { if (*d > 0.0) o->mu0 = *d; }
    else if (!strcmp(field, "t0_init"))
    {
        if (*i < 0 || *i > 2) { printf("\nerror: ocp_qp_gpu_ipm_opts_set: t0_init must be 0, 1 or 2, got %d\n", *i); exit(1); }
        o->t0_init = *i; /* 0: lam = t = sqrt(mu0); 1: lam = mu0, t = 1; 2: from the residuals (acados_ocp_options.py:1128-1143) */
    }
    else if (!strcmp(field, "ric_alg"))
    {
        /* ric_alg 0 (classical Riccati: P carried unfactored, only R + B'PB must be positive definite --
         * acados_ocp_options.py:1084-1101) exists for full-space Hessians that are indefinite; every kernel family here
         * carries the Cholesky factor of P (square-root form, ric_alg 1, the acados default).  Accepting 0 and running the
         * square-root form would factorise an indefinite block silently: refused like a wrong field. */
        if (*i != 1)
        {
            printf("\nerror: ocp_qp_gpu_ipm_opts_set: ric_alg = %d not available in this backend (only the square-root Riccati recursion, ric_alg = 1)\n", *i);
            exit(1);
        }
        o->ric_alg = *i;
    }
    else if (!strcmp(field, "alpha_min")) o->alpha_min = *d;
    else if (!strcmp(field, "reg_prim")) o->reg_prim = *d;
    else if (!strcmp(field, "t0_min")) o->t0_min = *d;       /* lower clip of t at a hot start */
    else if (!strcmp(field, "lam0_min")) o->lam0_min = *d;   /* lower clip of lam at a hot start */
    else if (!strcmp(field, "update_fact_exit")) o->update_fact_exit = *i; /* always honoured: the factor sweep factorises
                                                                              at the iterate it then judges */
    else if (!strcmp(field, "cond_pred_corr")) o->cond_pred_corr = *i;
    else if (!strcmp(field, "tol_comp_soft_scale")) o->tol_comp_soft_scale = *d; /* backend-specific, not an HPIPM name */
    else
    {
        printf("\nerror: ocp_qp_gpu_ipm_opts_set: wrong field: %s\n", field);
        exit(1);
    }
}

void ocp_qp_gpu_ipm_opts_get(void *config, void *opts_, const char *field, void *value)
{
    gpu_ipm_opts *o = (gpu_ipm_opts *) opts_;
    if (!strcmp(field, "t0_min")) *(double *) value = o->t0_min;
    else if (!strcmp(field, "lam0_min")) *(double *) value = o->lam0_min;
    else if (!strcmp(field, "iter_max")) *(int *) value = o->iter_max;
    else if (!strcmp(field, "warm_start")) *(int *) value = o->warm_start;
    else if (!strcmp(field, "tol_stat")) *(double *) value = o->tol_stat;
    else if (!strcmp(field, "tol_eq")) *(double *) value = o->tol_eq;
    else if (!strcmp(field, "tol_ineq")) *(double *) value = o->tol_ineq;
    else if (!strcmp(field, "tol_comp")) *(double *) value = o->tol_comp;
    else
    {
        printf("\nerror: ocp_qp_gpu_ipm_opts_get: field %s not available\n", field);
        exit(1);
    }
}

/* statistics rows kept: HPIPM's stat_max is fixed at 50 by acados (ocp_qp_hpipm.c:108); iter_max raised later is
 * served up to this many rows */
static int stat_rows_for(const gpu_ipm_opts *o) { return std::max(o ? o->iter_max : 50, 50) + 2; }

acados_size_t ocp_qp_gpu_ipm_memory_calculate_size(void *config, void *dims_, void *opts_)
{
    const ocp_qp_dims *d = (const ocp_qp_dims *) dims_;
    const int cap = gqp_host::structure_sig_len(d);
    return sizeof(gpu_ipm_memory) + 2 * sizeof(int) * (size_t) cap
           + sizeof(double) * GPU_IPM_STAT_M * (size_t) stat_rows_for((const gpu_ipm_opts *) opts_) + 4 * 8;
}

void *ocp_qp_gpu_ipm_memory_assign(void *config, void *dims_, void *opts_, void *raw_memory)
{
    const ocp_qp_dims *d = (const ocp_qp_dims *) dims_;
    char *c = align8((char *) raw_memory);
    gpu_ipm_memory *m = (gpu_ipm_memory *) c;
    memset(m, 0, sizeof(*m));
    c = align8(c + sizeof(gpu_ipm_memory));
    m->stat_m = GPU_IPM_STAT_M;
    m->stat_rows = stat_rows_for((const gpu_ipm_opts *) opts_);
    m->stat = (double *) c; c += sizeof(double) * GPU_IPM_STAT_M * (size_t) m->stat_rows;
    m->sig_cap = gqp_host::structure_sig_len(d);
    m->sig = (int *) c; c += sizeof(int) * (size_t) m->sig_cap;
    m->sig_scratch = (int *) c; c += sizeof(int) * (size_t) m->sig_cap;
    memset(m->stat, 0, sizeof(double) * GPU_IPM_STAT_M * (size_t) m->stat_rows);
    return m;
}

void ocp_qp_gpu_ipm_memory_get(void *config, void *mem_, const char *field, void *value)
{
    gpu_ipm_memory *m = (gpu_ipm_memory *) mem_;
    /* ocp_qp_hpipm.c:255-297 */
    if (!strcmp(field, "time_qp_solver_call")) *(double *) value = m->time_qp_solver_call;
    else if (!strcmp(field, "iter")) *(int *) value = m->iter;
    else if (!strcmp(field, "status")) *(int *) value = m->status;
    else if (!strcmp(field, "stat")) *(double **) value = m->stat;
    else if (!strcmp(field, "stat_m")) *(int *) value = m->stat_m;
    else if (!strcmp(field, "stat_rows")) *(int *) value = m->stat_rows; /* rows the table holds (extension) */
    else if (!strcmp(field, "tau_iter")) *(double *) value = 0.0;
    else
    {
        printf("\nerror: ocp_qp_gpu_ipm_memory_get: field %s not available\n", field);
        exit(1);
    }
}

acados_size_t ocp_qp_gpu_ipm_workspace_calculate_size(void *config, void *dims, void *opts) { return 0; }

} /* extern "C" */

static int build_segments(batch_cache *bc, ocp_qp_gpu_batch *b, const ocp_qp_dims *d)
{
    /* segment tables of the bulk blobs, once per device batch; -1: the device failed (the first device work after create) */
    const int N = d->N;
    bc->seg_in.clear(); bc->seg_out.clear();
    bc->L_in = ocp_qp_gpu_batch_bulk_len(b, 0);
    bc->L_out = ocp_qp_gpu_batch_bulk_len(b, 1);
    if (bc->L_in < 0 || bc->L_out < 0) { bc->L_in = bc->L_out = 0; return -1; }
    auto add = [&](std::vector<blob_seg> &tab, int output, const char *name, int k, int len, int fid, int shift) {
        if (len <= 0) return;
        int seg_len = 0;
        const int off = ocp_qp_gpu_batch_bulk_offset(b, output, name, k, &seg_len);
        if (off < 0 || seg_len != len)
        {
            if (output) return;
            printf("\nerror: ocp_qp_gpu_ipm: field %s at stage %d has no place in the device layout\n", name, k);
            exit(1);
        }
        tab.push_back(blob_seg{off, len, fid, k, shift});
    };
    for (int k = 0; k <= N; k++)
    {
        const int nu = d->nu[k], nx = d->nx[k], nbu = d->nbu[k], nbx = d->nbx[k], ng = d->ng[k], ns = d->ns[k];
        std::vector<blob_seg> &ti = bc->seg_in, &to = bc->seg_out;
        if (k < N)
        {
            add(ti, 0, "A", k, vlen(d, "A", k), F_A, 0);
            add(ti, 0, "B", k, vlen(d, "B", k), F_B, 0);
            add(ti, 0, "b", k, vlen(d, "b", k), F_b, 0);
        }
        add(ti, 0, "Q", k, vlen(d, "Q", k), F_Q, 0);
        add(ti, 0, "S", k, vlen(d, "S", k), F_S, 0);
        add(ti, 0, "R", k, vlen(d, "R", k), F_R, 0);
        add(ti, 0, "q", k, vlen(d, "q", k), F_q, 0);
        add(ti, 0, "r", k, vlen(d, "r", k), F_r, 0);
        add(ti, 0, "lbu", k, nbu, F_lb, 0);
        add(ti, 0, "ubu", k, nbu, F_ub, 0);
        add(ti, 0, "lbx", k, nbx, F_lb, nbu);
        if (d->nbxe[k] > 0) add(ti, 0, "lbx#value", k, nbx, F_lb, nbu);
        add(ti, 0, "ubx", k, nbx, F_ub, nbu);
        add(ti, 0, "lbu_mask", k, nbu, F_lbm, 0);
        add(ti, 0, "ubu_mask", k, nbu, F_ubm, 0);
        add(ti, 0, "lbx_mask", k, nbx, F_lbm, nbu);
        add(ti, 0, "ubx_mask", k, nbx, F_ubm, nbu);
        add(ti, 0, "C", k, vlen(d, "C", k), F_C, 0);
        add(ti, 0, "D", k, vlen(d, "D", k), F_D, 0);
        add(ti, 0, "lg", k, ng, F_lg, 0);
        add(ti, 0, "ug", k, ng, F_ug, 0);
        add(ti, 0, "lg_mask", k, ng, F_lgm, 0);
        add(ti, 0, "ug_mask", k, ng, F_ugm, 0);
        add(ti, 0, "Zl", k, ns, F_Zl, 0);
        add(ti, 0, "Zu", k, ns, F_Zu, 0);
        add(ti, 0, "zl", k, ns, F_zl, 0);
        add(ti, 0, "zu", k, ns, F_zu, 0);
        add(ti, 0, "lls", k, ns, F_lls, 0);
        add(ti, 0, "lus", k, ns, F_lus, 0);
        add(ti, 0, "lls_mask", k, ns, F_llsm, 0);
        add(ti, 0, "lus_mask", k, ns, F_lusm, 0);
        const int nct = 2 * (d->nb[k] + ng + ns);
        add(to, 1, "u", k, nu, O_ux, 0);
        add(to, 1, "x", k, nx, O_ux, nu);
        add(to, 1, "sl", k, ns, O_ux, nu + nx);
        add(to, 1, "su", k, ns, O_ux, nu + nx + ns);
        if (k < N) add(to, 1, "pi", k, d->nx[k + 1], O_pi, 0);
        add(to, 1, "lam", k, nct, O_lam, 0);
        add(to, 1, "t", k, nct, O_t, 0);
    }
    return 0;
}

/* the device failed under this call (the library has reported the HIP error): every QP of the call is a QP failure --
 * what ocp_nlp turns into ACADOS_QP_FAILURE and a clean return (ocp_nlp_sqp.c:720-751) -- and the process lives on */
static int device_failure(int n, ocp_qp_out **outs, int *status, void **mem_)
{
    for (int i = 0; i < n; i++)
    {
        if (status) status[i] = ACADOS_QP_FAILURE;
        if (outs && outs[i] && outs[i]->misc) { qp_info *info = (qp_info *) outs[i]->misc; info->num_iter = 0; info->t_computed = 0; }
        if (mem_ && mem_[i]) { gpu_ipm_memory *mi = (gpu_ipm_memory *) mem_[i]; mi->iter = 0; mi->status = ACADOS_QP_FAILURE; }
    }
    return ACADOS_QP_FAILURE;
}

int gqp_host::gpu_ipm_evaluate_impl(void *config, int n, void **qp_in_, void **qp_out_, void *opts_, void **mem_, void *work,
                                    int *status, const cond_request *cr)
{
    const double t_start = now_s();
    ocp_qp_in **ins = (ocp_qp_in **) qp_in_;
    ocp_qp_out **outs = (ocp_qp_out **) qp_out_;
    gpu_ipm_opts *o = (gpu_ipm_opts *) opts_;
    gpu_ipm_memory *m = (gpu_ipm_memory *) mem_[0];
    const ocp_qp_dims *d = ins[0]->dim;
    const int N = d->N;
    const int phase = cr ? cr->phase : 0;

    /* (re)create the device batch when the count or the structure changed; signatures are compared in the carved
     * scratch -- no allocation on this path */
    const int siglen = gqp_host::structure_sig_len(d);
    if (siglen > m->sig_cap)
    {
        printf("\nerror: ocp_qp_gpu_ipm: the dims of qp_in grew after memory_assign (signature %d > %d ints)\n", siglen, m->sig_cap);
        exit(1);
    }
    gqp_host::structure_sig_fill(ins[0], m->sig_scratch);
    if (!m->cache) m->cache = new batch_cache(); /* device-side resources: once per memory, released by terminate */
    batch_cache *bc = m->cache;
    const bool rebuild = !bc->batch || bc->n != n || m->sig_len != siglen || memcmp(m->sig, m->sig_scratch, sizeof(int) * siglen) != 0;
    if (rebuild)
    {
        memcpy(m->sig, m->sig_scratch, sizeof(int) * siglen);
        m->sig_len = siglen;
    }
    for (int i = 1; i < n; i++)
    {
        if (gqp_host::structure_sig_len(ins[i]->dim) == siglen)
        {
            gqp_host::structure_sig_fill(ins[i], m->sig_scratch);
            if (memcmp(m->sig, m->sig_scratch, sizeof(int) * siglen) == 0) continue;
        }
        printf("\nerror: ocp_qp_gpu_ipm_evaluate_batch: QP %d differs in structure from QP 0\n", i);
        exit(1);
    }
    if (rebuild)
    {
        if (bc->batch) ocp_qp_gpu_batch_destroy(bc->batch);
        bc->batch = ocp_qp_gpu_batch_create(N, d->nx, d->nu, d->nbx, d->nbu, d->ng, d->ns, n, -1);
        if (!bc->batch)
        {
            printf("\nerror: ocp_qp_gpu_ipm: no GPU batch could be created (no device or unsupported shape)\n");
            exit(1);
        }
        bc->n = n;
        bc->cond_N_sent = -1;
        bc->cond_solved = false;
        bc->blocks_sent.clear();
        for (int k = 0; k <= N; k++)
        {
            ocp_qp_gpu_batch_set_int(bc->batch, "idxb", k, ins[0]->idxb[k], d->nb[k]);
            ocp_qp_gpu_batch_set_int(bc->batch, "idxs_rev", k, ins[0]->idxs_rev[k], d->nb[k] + d->ng[k]);
            ocp_qp_gpu_batch_set_int(bc->batch, "idxe", k, ins[0]->idxe[k], d->nbxe[k]);
        }
        if (build_segments(bc, bc->batch, d) != 0)
        {
            ocp_qp_gpu_batch_destroy(bc->batch);
            bc->batch = nullptr;      /* the next call builds it again */
            return device_failure(n, outs, status, mem_);
        }
        if (n > 1) { bc->st.assign(n, 0); bc->it.assign(n, 0); }
    }
    ocp_qp_gpu_batch *b = bc->batch;

    /* options */
    ocp_qp_gpu_batch_opts_set(b, "iter_max", &o->iter_max);
    ocp_qp_gpu_batch_opts_set(b, "tol_stat", &o->tol_stat);
    ocp_qp_gpu_batch_opts_set(b, "tol_eq", &o->tol_eq);
    ocp_qp_gpu_batch_opts_set(b, "tol_ineq", &o->tol_ineq);
    ocp_qp_gpu_batch_opts_set(b, "tol_comp", &o->tol_comp);
    /* warm_start 1 = 0: "primal guess is kept ... NOTE: this is the same as 0, as acados resets the initial guess of
     * primal variables to zero" (acados_ocp_options.py:1029-1031; the reset is ocp_qp_hpipm.c:325-336) */
    const int ws = o->warm_start >= 2 ? o->warm_start : 0;
    ocp_qp_gpu_batch_opts_set(b, "warm_start", &ws);
    ocp_qp_gpu_batch_opts_set(b, "mu0", &o->mu0);
    ocp_qp_gpu_batch_opts_set(b, "t0_init", &o->t0_init);
    ocp_qp_gpu_batch_opts_set(b, "alpha_min", &o->alpha_min);
    ocp_qp_gpu_batch_opts_set(b, "tau_min", &o->tau_min);
    ocp_qp_gpu_batch_opts_set(b, "tol_comp_soft_scale", &o->tol_comp_soft_scale);
    ocp_qp_gpu_batch_opts_set(b, "reg_prim", &o->reg_prim);
    ocp_qp_gpu_batch_opts_set(b, "t0_min", &o->t0_min);
    ocp_qp_gpu_batch_opts_set(b, "lam0_min", &o->lam0_min);
    ocp_qp_gpu_batch_opts_set(b, "cond_pred_corr", &o->cond_pred_corr);
    ocp_qp_gpu_batch_opts_set(b, "print_level", &o->print_level);
    {
        /* the condensing request arrives as an argument (from the xcond level's opts); the device batch is told only
         * when it changes -- a change drops the resident condensed batch */
        const int cn = cr && cr->N2 > 0 && cr->N2 < N ? cr->N2 : N;
        bool same = cn == bc->cond_N_sent;
        if (same && cn < N)
        {
            const bool has = cr->block_size != nullptr;
            same = has ? ((int) bc->blocks_sent.size() == cn + 1 && memcmp(bc->blocks_sent.data(), cr->block_size, sizeof(int) * (cn + 1)) == 0)
                       : bc->blocks_sent.empty();
        }
        if (!same)
        {
            int reset = N; /* forces the device batch to drop a child built for other block sizes */
            ocp_qp_gpu_batch_opts_set(b, "cond_N", &reset);
            ocp_qp_gpu_batch_opts_set(b, "cond_N", &cn);
            bc->blocks_sent.clear();
            if (cn < N && cr->block_size)
            {
                if (ocp_qp_gpu_batch_opts_set(b, "cond_block_size", cr->block_size) != 0) exit(1); /* :352-356 */
                bc->blocks_sent.assign(cr->block_size, cr->block_size + cn + 1);
            }
            bc->cond_N_sent = cn;
            bc->cond_solved = false;
        }
    }
    {
        const int dense = cr && cr->dense ? 1 : 0;
        ocp_qp_gpu_batch_opts_set(b, "full_dense", &dense);
    }
    /* Hot start of a CONDENSED solve: the reference starts from the condensed iterate kept in its memory (xcond_qp_out) and
     * re-derives it from the caller's qp_out only when initialize_next_xcond_qp_from_qp_out is set
     * (ocp_qp_xcond_solver.c:554-571).  Same here: without the flag the condensed batch keeps the iterate of its last
     * solve (nothing is packed, k_pcond_sol does not run); with it -- or when there is no previous condensed solve -- the
     * caller's (pi, lam, t) are packed and condensed. */
    const bool condensed_call = cr && cr->N2 > 0 && cr->N2 < N;
    const int keep_child = (condensed_call && !cr->init_from_qp_out && bc->cond_solved) ? 1 : 0;
    ocp_qp_gpu_batch_opts_set(b, "cond_keep_iterate", &keep_child);

    if (ws >= 2 && phase != 1 && !keep_child)
    {
        /* hot start: pi, lam, t of qp_out are the starting point (acados_ocp_options.py:1029-1032).  The reference
         * zeroes the PRIMAL part of qp_out before every solve, whatever the warm start level (ocp_qp_hpipm.c:325-336: the
         * QPs of an SQP method live in delta space), so u, x, sl, su go in as zeros.  Staged through the pinned output
         * blob (same layout as the unpack); written BEFORE the pack, which then puts the equality-flagged values (x0)
         * back into the iterate. */
        const size_t L = (size_t) bc->L_out;
        pinned_reserve(bc->blob_out, bc->cap_out, (size_t) n * L);
        double *blob = bc->blob_out;
        const std::vector<blob_seg> &tab = bc->seg_out;
        par_instances(n, [&](int lo, int hi) {
            for (int i = lo; i < hi; i++)
                for (const blob_seg &g : tab)
                {
                    if (g.fid == O_ux) memset(blob + (size_t) i * L + g.off, 0, sizeof(double) * g.len);
                    else memcpy(blob + (size_t) i * L + g.off, out_field(outs[i], g.fid, g.k) + g.shift, sizeof(double) * g.len);
                }
        });
        ocp_qp_gpu_batch_set_bulk_out(b, blob, 0);
    }
    /* every member array of every qp_in is re-read on every call (they alias ocp_nlp memory:
     * ocp_nlp_common.c:2797-2894): host threads gather them into ONE pinned blob, then one host->device copy
     * and one scatter launch move the whole batch (ocp_qp_gpu_batch_set_bulk) */
    {
        const size_t L = (size_t) bc->L_in;
        pinned_reserve(bc->blob_in, bc->cap_in, (size_t) n * L);
        double *blob = bc->blob_in;
        const std::vector<blob_seg> &tab = bc->seg_in;
        par_instances(n, [&](int lo, int hi) {
            for (int i = lo; i < hi; i++)
                for (const blob_seg &g : tab)
                    memcpy(blob + (size_t) i * L + g.off, in_field(ins[i], g.fid, g.k) + g.shift, sizeof(double) * g.len);
        });
        if (ocp_qp_gpu_batch_set_bulk(b, blob, 0) != 0) return device_failure(n, outs, status, mem_);
    }
    if (phase == 1)
    {
        /* RTI preparation (ocp_qp_xcond_solver.c:591-620): matrix part of the condensing, resident on the device */
        return ocp_qp_gpu_batch_condense_lhs(b) == 0 ? ACADOS_SUCCESS : device_failure(n, outs, status, mem_);
    }
    const double t_packed = now_s();

    /* >= 0: instances with non-zero status; < 0: the device failed (HIP error reported by the library) -- every QP of the
     * call comes back as ACADOS_QP_FAILURE, nothing is read from the device, the process goes on */
    const int dev_rc = phase == 2 ? ocp_qp_gpu_batch_condense_rhs_and_solve(b) : ocp_qp_gpu_batch_solve(b);
    if (dev_rc < 0) return device_failure(n, outs, status, mem_);
    if (condensed_call) bc->cond_solved = true;
    const double t_solved = now_s();

    /* unpack: one gather launch + one device->host copy for the whole batch, then host threads scatter */
    {
        const size_t L = (size_t) bc->L_out;
        pinned_reserve(bc->blob_out, bc->cap_out, (size_t) n * L);
        if (ocp_qp_gpu_batch_get_bulk(b, bc->blob_out, 0) != 0) return device_failure(n, outs, status, mem_);
        const double *blob = bc->blob_out;
        const std::vector<blob_seg> &tab = bc->seg_out;
        par_instances(n, [&](int lo, int hi) {
            for (int i = lo; i < hi; i++)
                for (const blob_seg &g : tab)
                    memcpy(out_field(outs[i], g.fid, g.k) + g.shift, blob + (size_t) i * L + g.off, sizeof(double) * g.len);
        });
    }
    int *st = n > 1 ? bc->st.data() : &m->st_it[0], *it = n > 1 ? bc->it.data() : &m->st_it[1];
    ocp_qp_gpu_batch_get_info(b, "status", st);
    ocp_qp_gpu_batch_get_info(b, "iter", it);
    memset(m->stat, 0, sizeof(double) * GPU_IPM_STAT_M * (size_t) m->stat_rows);
    ocp_qp_gpu_batch_get_stat(b, 0, m->stat, m->stat_rows);
    const double t_end = now_s();

    int worst = 0;
    const double t_cond = ocp_qp_gpu_batch_get_scalar(b, "time_xcond"), t_tot = ocp_qp_gpu_batch_get_scalar(b, "time_tot");
    for (int i = 0; i < n; i++)
    {
        qp_info *info = (qp_info *) outs[i]->misc;
        info->condensing_time = t_cond;
        info->solve_QP_time = t_tot - t_cond;
        info->interface_time = (t_packed - t_start) + (t_end - t_solved);
        info->total_time = t_end - t_start;
        info->num_iter = it[i];
        info->t_computed = 1;
        if (status) status[i] = st[i];
        if (st[i] != ACADOS_SUCCESS && (worst == 0 || worst == ACADOS_MAXITER)) worst = st[i];
        if (mem_[i] && i > 0)
        {
            gpu_ipm_memory *mi = (gpu_ipm_memory *) mem_[i];
            mi->iter = it[i]; mi->status = st[i]; mi->time_qp_solver_call = info->solve_QP_time;
        }
    }
    m->iter = it[0];
    m->status = st[0];
    m->time_qp_solver_call = t_solved - t_packed;
    return worst;
}

extern "C" {

int ocp_qp_gpu_ipm_evaluate_batch(void *config, int n, void **qp_in, void **qp_out, void *opts, void **mem, void *work,
                                  int *status)
{
    if (n <= 0) return ACADOS_SUCCESS;
    return gqp_host::gpu_ipm_evaluate_impl(config, n, qp_in, qp_out, opts, mem, work, status, nullptr);
}

/* ocp_qp_hpipm.c:314-405 */
int ocp_qp_gpu_ipm(void *config, void *qp_in, void *qp_out, void *opts, void *mem, void *work)
{
    int status = 0;
    void *ins[1] = {qp_in}, *outs[1] = {qp_out}, *mems[1] = {mem};
    gqp_host::gpu_ipm_evaluate_impl(config, 1, ins, outs, opts, mems, work, &status, nullptr);
    /* status codes are already acados' (the map of ocp_qp_hpipm.c:398-404 is applied on device) */
    return status;
}

void ocp_qp_gpu_ipm_solver_get(void *config, void *qp_in_, void *qp_out, void *opts, void *mem_, const char *field,
                               int stage, void *value, int size1, int size2)
{
    /* ocp_qp_hpipm.c:417-478: P (nx x nx), p (nx), K (nu x nx), k (nu), Lr (nu x nu), column-major, of the
     * last factorisation; u = K x + k as ocp_nlp_ddp.c:373-377 uses them */
    ocp_qp_in *in = (ocp_qp_in *) qp_in_;
    gpu_ipm_memory *m = (gpu_ipm_memory *) mem_;
    const int nx = in->dim->nx[stage], nu = in->dim->nu[stage], nv = nu + nx;
    double *out = (double *) value;
    if (!m->cache || !m->cache->batch)
    {
        printf("\nocp_qp_gpu_ipm_solver_get: no factorisation available (solve first)\n");
        exit(1);
    }
    const int nb = m->cache->n;
    std::vector<double> Lb((size_t) nb * nv * nv), lb((size_t) nb * nv);
    ocp_qp_gpu_batch_get(m->cache->batch, "ric_L", stage, Lb.data(), 0);
    ocp_qp_gpu_batch_get(m->cache->batch, "ric_l", stage, lb.data(), 0);
    const double *L = Lb.data(), *l = lb.data(); /* instance 0: the QP this memory belongs to */
    auto bad_size = [&](int e1, int e2) {
        if (size1 != e1 || size2 != e2)
            printf("\nocp_qp_gpu_ipm_solver_get: size of field %s not as expected, got size %d %d.\n", field, size1, size2);
    };
    if (!strcmp(field, "P"))
    {
        bad_size(nx, nx);
        for (int c = 0; c < nx; c++)
            for (int r = 0; r < nx; r++)
            {
                double a = 0.0;
                for (int q = 0; q <= (r < c ? r : c); q++) a += L[(nu + r) + nv * (nu + q)] * L[(nu + c) + nv * (nu + q)];
                out[r + nx * c] = a;
            }
    }
    else if (!strcmp(field, "p"))
    {
        bad_size(nx, 1);
        for (int r = 0; r < nx; r++)
        {
            double a = 0.0;
            for (int q = 0; q <= r; q++) a += L[(nu + r) + nv * (nu + q)] * l[nu + q];
            out[r] = a;
        }
    }
    else if (!strcmp(field, "K") || !strcmp(field, "k"))
    {
        const bool isK = field[0] == 'K';
        bad_size(nu, isK ? nx : 1);
        const int ncol = isK ? nx : 1;
        /* solve Lr' X = -rhs, rhs = Ls' (K) or lr (k) */
        for (int c = 0; c < ncol; c++)
            for (int r = nu - 1; r >= 0; r--)
            {
                double a = isK ? -L[(nu + c) + nv * r] : -l[r];
                for (int q = r + 1; q < nu; q++) a -= L[q + nv * r] * out[q + nu * c];
                out[r + nu * c] = a / L[r + nv * r];
            }
    }
    else if (!strcmp(field, "Lr"))
    {
        bad_size(nu, nu);
        for (int c = 0; c < nu; c++)
            for (int r = 0; r < nu; r++) out[r + nu * c] = r >= c ? L[r + nv * c] : 0.0;
    }
    else
        printf("\nocp_qp_gpu_ipm_solver_get: field %s not supported", field);
}

void ocp_qp_gpu_ipm_memory_reset(void *config, void *qp_in, void *qp_out, void *opts, void *mem_, void *work)
{
    gpu_ipm_memory *m = (gpu_ipm_memory *) mem_;
    if (m->cache)
    {
        if (m->cache->batch) ocp_qp_gpu_batch_destroy(m->cache->batch);
        delete m->cache;
        m->cache = nullptr;
    }
    m->sig_len = 0;
}

/* ocp_qp_hpipm.c:481-506.  The seed is the derivative of the problem data of ONE qp (the one `mem` belongs to,
 * i.e. instance 0 of the device batch of the last evaluate), the result d(solution)/d(parameter) lands in the
 * ocp_qp_out container: ux = [du; dx; dsl; dsu], pi, lam, t (ocp_nlp_common.c:4095-4104 copies exactly these). */
static void gpu_ipm_eval_sens(const char *who, void *qp_in_, void *seed_, void *qp_out_, void *mem_)
{
    ocp_qp_in *in = (ocp_qp_in *) qp_in_;
    ocp_qp_seed *seed = (ocp_qp_seed *) seed_;
    ocp_qp_out *out = (ocp_qp_out *) qp_out_;
    gpu_ipm_memory *m = (gpu_ipm_memory *) mem_;
    if (!m->cache || !m->cache->batch)
    {
        printf("\n%s: no factorisation available (solve first)\n", who);
        exit(1);
    }
    ocp_qp_gpu_batch *b = m->cache->batch;
    const ocp_qp_dims *d = in->dim;
    const int nb_ = m->cache->n;
    std::vector<double> buf;
    auto push = [&](const char *name, int k, int len, const double *src) {
        if (len <= 0) return;
        buf.assign((size_t) nb_ * len, 0.0); /* the seed belongs to instance 0; the other instances get zero seeds */
        memcpy(buf.data(), src, sizeof(double) * len);
        if (ocp_qp_gpu_batch_sens_set(b, name, k, buf.data()) != 0) exit(1);
    };
    for (int k = 0; k <= d->N; k++)
    {
        const int nu = d->nu[k], nx = d->nx[k], nbu = d->nbu[k], nbx = d->nbx[k], nb = d->nb[k], ng = d->ng[k];
        push("seed_r", k, nu, seed->seed_g[k]);
        push("seed_q", k, nx, seed->seed_g[k] + nu);
        if (k < d->N) push("seed_b", k, d->nx[k + 1], seed->seed_b[k]);
        push("seed_lbu", k, nbu, seed->seed_d[k]);
        push("seed_lbx", k, nbx, seed->seed_d[k] + nbu);
        push("seed_lg", k, ng, seed->seed_d[k] + nb);
        push("seed_ubu", k, nbu, seed->seed_d[k] + nb + ng);
        push("seed_ubx", k, nbx, seed->seed_d[k] + nb + ng + nbu);
        push("seed_ug", k, ng, seed->seed_d[k] + 2 * nb + ng);
    }
    if (ocp_qp_gpu_batch_sens_solve(b) != 0) exit(1);
    auto pull = [&](const char *name, int k, int len, double *dst) {
        if (len <= 0) return;
        buf.assign((size_t) nb_ * len, 0.0);
        ocp_qp_gpu_batch_get(b, name, k, buf.data(), 0);
        memcpy(dst, buf.data(), sizeof(double) * len);
    };
    for (int k = 0; k <= d->N; k++)
    {
        const int nu = d->nu[k], nx = d->nx[k], ns = d->ns[k], nct = 2 * (d->nb[k] + d->ng[k] + ns);
        pull("sens_u", k, nu, out->ux[k]);
        pull("sens_x", k, nx, out->ux[k] + nu);
        pull("sens_sl", k, ns, out->ux[k] + nu + nx);
        pull("sens_su", k, ns, out->ux[k] + nu + nx + ns);
        if (k < d->N) pull("sens_pi", k, d->nx[k + 1], out->pi[k]);
        pull("sens_lam", k, nct, out->lam[k]);
        pull("sens_t", k, nct, out->t[k]);
    }
}

void ocp_qp_gpu_ipm_eval_forw_sens(void *config, void *qp_in, void *seed, void *qp_out, void *opts, void *mem, void *work)
{
    gpu_ipm_eval_sens("ocp_qp_gpu_ipm_eval_forw_sens", qp_in, seed, qp_out, mem);
}

/* the KKT matrix of the condensed Newton system is symmetric: the adjoint solve of a seed is the forward solve of the
 * same seed (acados seeds only seed_g here and reads ux, pi: ocp_nlp_common.c:4128-4160) */
void ocp_qp_gpu_ipm_eval_adj_sens(void *config, void *qp_in, void *seed, void *qp_out, void *opts, void *mem, void *work)
{
    gpu_ipm_eval_sens("ocp_qp_gpu_ipm_eval_adj_sens", qp_in, seed, qp_out, mem);
}

void ocp_qp_gpu_ipm_terminate(void *config, void *mem, void *work)
{
    ocp_qp_gpu_ipm_memory_reset(config, nullptr, nullptr, nullptr, mem, work);
}

/* ocp_qp_hpipm.c:517-540 */
void ocp_qp_gpu_ipm_config_initialize_default(void *config_)
{
    qp_solver_config *config = (qp_solver_config *) config_;
    config->dims_set = &ocp_qp_dims_set;
    config->opts_calculate_size = &ocp_qp_gpu_ipm_opts_calculate_size;
    config->opts_assign = &ocp_qp_gpu_ipm_opts_assign;
    config->opts_initialize_default = &ocp_qp_gpu_ipm_opts_initialize_default;
    config->opts_update = &ocp_qp_gpu_ipm_opts_update;
    config->opts_set = &ocp_qp_gpu_ipm_opts_set;
    config->opts_get = &ocp_qp_gpu_ipm_opts_get;
    config->memory_calculate_size = &ocp_qp_gpu_ipm_memory_calculate_size;
    config->memory_assign = &ocp_qp_gpu_ipm_memory_assign;
    config->memory_get = &ocp_qp_gpu_ipm_memory_get;
    config->workspace_calculate_size = &ocp_qp_gpu_ipm_workspace_calculate_size;
    config->evaluate = &ocp_qp_gpu_ipm;
    config->solver_get = &ocp_qp_gpu_ipm_solver_get;
    config->memory_reset = &ocp_qp_gpu_ipm_memory_reset;
    config->eval_forw_sens = &ocp_qp_gpu_ipm_eval_forw_sens;
    config->eval_adj_sens = &ocp_qp_gpu_ipm_eval_adj_sens;
    config->terminate = &ocp_qp_gpu_ipm_terminate;
}

} /* extern "C" */
