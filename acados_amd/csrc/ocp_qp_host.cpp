/*
 * ocp_qp_host.cpp -- acados-shaped host API (include/acados_amd/ocp_qp_interface.h) on top
 * of the device batch C-ABI.  Host code only: no kernels here.  Each function cites the
 * reference function whose contract it follows (paths relative to /root/reference).
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

namespace
{

inline char *align8(char *p) { return (char *) (((uintptr_t) p + 7) & ~(uintptr_t) 7); }

struct gpu_ipm_opts
{
    /* names as d_ocp_qp_ipm_arg_set / ocp_qp_hpipm_opts_set (ocp_qp_hpipm.c:142-183) */
    double mu0, tol_stat, tol_eq, tol_ineq, tol_comp, alpha_min, tau_min, reg_prim, t0_min, lam0_min;
    int iter_max, warm_start, cond_pred_corr, print_level, ric_alg, t0_init, update_fact_exit;
};

/* one member array of ocp_qp_in / ocp_qp_out and its place in the per-instance bulk blob */
enum { F_A, F_B, F_b, F_Q, F_S, F_R, F_q, F_r, F_lb, F_ub, F_lbm, F_ubm, F_C, F_D, F_lg, F_ug, F_lgm, F_ugm,
       F_Zl, F_Zu, F_zl, F_zu, F_lls, F_lus, F_llsm, F_lusm, O_ux, O_pi, O_lam, O_t };
struct blob_seg { int off, len, fid, k, shift; };

struct batch_cache
{
    ocp_qp_gpu_batch *batch = nullptr;
    int n = 0;
    bool blocks_sent = false; /* user block sizes handed to this device batch */
    std::vector<int> sig; /* dims + idxb + idxs_rev + idxe of the batch */
    std::vector<double> stat;
    std::vector<double> stage; /* host staging [n][len] */
    /* bulk pack / unpack: segment tables (built once per batch) and PINNED staging [n][bulk_len] */
    std::vector<blob_seg> seg_in, seg_out;
    int L_in = 0, L_out = 0;
    double *blob_in = nullptr, *blob_out = nullptr;
    size_t cap_in = 0, cap_out = 0;
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
 * different instances are independent) */
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

struct gpu_ipm_memory
{
    batch_cache *cache;
    double time_qp_solver_call;
    int iter;
    int status;
    int stat_m;
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

std::vector<int> structure_sig(const ocp_qp_in *in)
{
    const ocp_qp_dims *d = in->dim;
    std::vector<int> s;
    s.push_back(d->N);
    for (int k = 0; k <= d->N; k++)
    {
        int v[] = {d->nx[k], d->nu[k], d->nbx[k], d->nbu[k], d->ng[k], d->ns[k], d->nbxe[k]};
        s.insert(s.end(), v, v + 7);
        s.insert(s.end(), in->idxb[k], in->idxb[k] + d->nb[k]);
        s.insert(s.end(), in->idxs_rev[k], in->idxs_rev[k] + d->nb[k] + d->ng[k]);
        s.insert(s.end(), in->idxe[k], in->idxe[k] + d->nbxe[k]);
    }
    return s;
}

} // namespace

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

/* ocp_qp_common.c:874-921, line by line on the plain containers */
/* d_ocp_qp_seed (acados/ocp_qp/ocp_qp_common.h: ocp_qp_seed): seed_g [r; q; zl; zu], seed_b, seed_d
 * [lb lg ub ug ls us] (natural sign), seed_m; one calloc'ed block */
ocp_qp_seed *ocp_qp_seed_create(ocp_qp_dims *dims)
{
    const int N = dims->N;
    size_t cnt = 0;
    for (int k = 0; k <= N; k++)
        cnt += dims->nu[k] + dims->nx[k] + 2 * dims->ns[k] + (k < N ? dims->nx[k + 1] : 0) + 4 * (size_t) (dims->nb[k] + dims->ng[k] + dims->ns[k]);
    char *raw = (char *) calloc(1, sizeof(ocp_qp_seed) + 4 * (N + 1) * sizeof(double *) + cnt * sizeof(double) + 64);
    ocp_qp_seed *sd = (ocp_qp_seed *) raw;
    raw = align8(raw + sizeof(ocp_qp_seed));
    sd->dim = dims;
    sd->seed_g = (double **) raw; raw += (N + 1) * sizeof(double *);
    sd->seed_b = (double **) raw; raw += (N + 1) * sizeof(double *);
    sd->seed_d = (double **) raw; raw += (N + 1) * sizeof(double *);
    sd->seed_m = (double **) raw; raw += (N + 1) * sizeof(double *);
    double *p = (double *) align8(raw);
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

void ocp_qp_seed_free(void *seed) { free(seed); }

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

/* ------------------------------------------------- inner plugin (vtable) */
/* ocp_qp_hpipm.c:60-540 */

acados_size_t ocp_qp_gpu_ipm_opts_calculate_size(void *config, void *dims) { return sizeof(gpu_ipm_opts) + 8; }

void *ocp_qp_gpu_ipm_opts_assign(void *config, void *dims, void *raw_memory) { return align8((char *) raw_memory); }

void ocp_qp_gpu_ipm_opts_initialize_default(void *config, void *dims, void *opts_)
{
    gpu_ipm_opts *o = (gpu_ipm_opts *) opts_;
    /* mode BALANCE + the acados overrides of ocp_qp_hpipm.c:101-113 */
    o->mu0 = 1e0; o->tol_stat = 1e-6; o->tol_eq = 1e-8; o->tol_ineq = 1e-8; o->tol_comp = 1e-8;
    o->alpha_min = 1e-8; o->tau_min = 0.0; o->reg_prim = 1e-15; o->t0_min = 1e-16; o->lam0_min = 1e-16;
    o->iter_max = 50; o->warm_start = 0; o->cond_pred_corr = 1; o->print_level = 0; o->ric_alg = 1;
    o->t0_init = 2; o->update_fact_exit = 0;
}

void ocp_qp_gpu_ipm_opts_update(void *config, void *dims, void *opts) {}

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
        /* a mode change re-applies the acados overrides (ocp_qp_hpipm.c:146-165) */
        const int pl = o->print_level;
        ocp_qp_gpu_ipm_opts_initialize_default(config, nullptr, o);
        o->print_level = pl;
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
    else if (!strcmp(field, "t0_init")) o->t0_init = *i;
    else if (!strcmp(field, "ric_alg")) o->ric_alg = *i;
    else if (!strcmp(field, "alpha_min")) o->alpha_min = *d;
    else if (!strcmp(field, "reg_prim")) o->reg_prim = *d;
    else if (!strcmp(field, "t0_min")) o->t0_min = *d;
    else if (!strcmp(field, "lam0_min")) o->lam0_min = *d;
    else if (!strcmp(field, "update_fact_exit")) o->update_fact_exit = *i;
    else if (!strcmp(field, "cond_pred_corr")) o->cond_pred_corr = *i;
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
    else if (!strcmp(field, "tol_stat")) *(double *) value = o->tol_stat;
    else
    {
        printf("\nerror: ocp_qp_gpu_ipm_opts_get: field %s not available\n", field);
        exit(1);
    }
}

acados_size_t ocp_qp_gpu_ipm_memory_calculate_size(void *config, void *dims, void *opts) { return sizeof(gpu_ipm_memory) + 8; }

void *ocp_qp_gpu_ipm_memory_assign(void *config, void *dims, void *opts, void *raw_memory)
{
    gpu_ipm_memory *m = (gpu_ipm_memory *) align8((char *) raw_memory);
    memset(m, 0, sizeof(*m));
    m->stat_m = 20;
    return m;
}

void ocp_qp_gpu_ipm_memory_get(void *config, void *mem_, const char *field, void *value)
{
    gpu_ipm_memory *m = (gpu_ipm_memory *) mem_;
    /* ocp_qp_hpipm.c:255-297 */
    if (!strcmp(field, "time_qp_solver_call")) *(double *) value = m->time_qp_solver_call;
    else if (!strcmp(field, "iter")) *(int *) value = m->iter;
    else if (!strcmp(field, "status")) *(int *) value = m->status;
    else if (!strcmp(field, "stat")) *(double **) value = m->cache ? m->cache->stat.data() : nullptr;
    else if (!strcmp(field, "stat_m")) *(int *) value = m->stat_m;
    else if (!strcmp(field, "tau_iter")) *(double *) value = 0.0;
    else
    {
        printf("\nerror: ocp_qp_gpu_ipm_memory_get: field %s not available\n", field);
        exit(1);
    }
}

acados_size_t ocp_qp_gpu_ipm_workspace_calculate_size(void *config, void *dims, void *opts) { return 0; }

static thread_local int g_cond_N_request = 0; /* set by the xcond level right before evaluate (same thread) */
static thread_local const int *g_cond_blocks_request = nullptr;

int ocp_qp_gpu_ipm_evaluate_batch(void *config, int n, void **qp_in_, void **qp_out_, void *opts_, void **mem_,
                                  void *work, int *status)
{
    const double t_start = now_s();
    ocp_qp_in **ins = (ocp_qp_in **) qp_in_;
    ocp_qp_out **outs = (ocp_qp_out **) qp_out_;
    gpu_ipm_opts *o = (gpu_ipm_opts *) opts_;
    gpu_ipm_memory *m = (gpu_ipm_memory *) mem_[0];
    const ocp_qp_dims *d = ins[0]->dim;
    const int N = d->N;

    /* (re)create the device batch when the count or the structure changed */
    std::vector<int> sig = structure_sig(ins[0]);
    for (int i = 1; i < n; i++)
        if (structure_sig(ins[i]) != sig)
        {
            printf("\nerror: ocp_qp_gpu_ipm_evaluate_batch: QP %d differs in structure from QP 0\n", i);
            exit(1);
        }
    if (!m->cache) m->cache = new batch_cache();
    batch_cache *bc = m->cache;
    if (!bc->batch || bc->n != n || bc->sig != sig)
    {
        if (bc->batch) ocp_qp_gpu_batch_destroy(bc->batch);
        bc->batch = ocp_qp_gpu_batch_create(N, d->nx, d->nu, d->nbx, d->nbu, d->ng, d->ns, n, -1);
        if (!bc->batch)
        {
            printf("\nerror: ocp_qp_gpu_ipm: no GPU batch could be created (no device or unsupported shape)\n");
            exit(1);
        }
        bc->n = n;
        bc->blocks_sent = false;
        bc->sig = sig;
        bc->seg_in.clear();
        bc->seg_out.clear();
        for (int k = 0; k <= N; k++)
        {
            ocp_qp_gpu_batch_set_int(bc->batch, "idxb", k, ins[0]->idxb[k], d->nb[k]);
            ocp_qp_gpu_batch_set_int(bc->batch, "idxs_rev", k, ins[0]->idxs_rev[k], d->nb[k] + d->ng[k]);
            ocp_qp_gpu_batch_set_int(bc->batch, "idxe", k, ins[0]->idxe[k], d->nbxe[k]);
        }
    }
    ocp_qp_gpu_batch *b = bc->batch;

    /* options */
    ocp_qp_gpu_batch_opts_set(b, "iter_max", &o->iter_max);
    ocp_qp_gpu_batch_opts_set(b, "tol_stat", &o->tol_stat);
    ocp_qp_gpu_batch_opts_set(b, "tol_eq", &o->tol_eq);
    ocp_qp_gpu_batch_opts_set(b, "tol_ineq", &o->tol_ineq);
    ocp_qp_gpu_batch_opts_set(b, "tol_comp", &o->tol_comp);
    ocp_qp_gpu_batch_opts_set(b, "warm_start", &o->warm_start);
    ocp_qp_gpu_batch_opts_set(b, "mu0", &o->mu0);
    ocp_qp_gpu_batch_opts_set(b, "alpha_min", &o->alpha_min);
    ocp_qp_gpu_batch_opts_set(b, "tau_min", &o->tau_min);
    ocp_qp_gpu_batch_opts_set(b, "reg_prim", &o->reg_prim);
    ocp_qp_gpu_batch_opts_set(b, "cond_pred_corr", &o->cond_pred_corr);
    ocp_qp_gpu_batch_opts_set(b, "print_level", &o->print_level);
    {
        int cn = g_cond_N_request > 0 ? g_cond_N_request : N;
        ocp_qp_gpu_batch_opts_set(b, "cond_N", &cn);
        if (g_cond_blocks_request && cn > 0 && cn < N && !bc->blocks_sent)
        {
            if (ocp_qp_gpu_batch_opts_set(b, "cond_block_size", g_cond_blocks_request) != 0) exit(1); /* :352-356 */
            bc->blocks_sent = true;
        }
    }

    /* re-read every member array of qp_in on every call (they alias ocp_nlp memory:
     * ocp_nlp_common.c:2797-2894) and pack them: one host blob per instance, ONE host->device copy and ONE
     * launch for the whole batch (ocp_qp_gpu_batch_set_bulk) */
    std::vector<double> &stg = bc->stage;
    if (bc->seg_in.empty())
    {
        /* segment tables of the bulk blobs, once per device batch */
        bc->L_in = ocp_qp_gpu_batch_bulk_len(b, 0);
        bc->L_out = ocp_qp_gpu_batch_bulk_len(b, 1);
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
    }
    {
        /* every member array of every qp_in is re-read on every call (they alias ocp_nlp memory:
         * ocp_nlp_common.c:2797-2894): host threads gather them into ONE pinned blob, then one host->device copy
         * and one scatter launch move the whole batch (ocp_qp_gpu_batch_set_bulk) */
        const size_t L = (size_t) bc->L_in;
        pinned_reserve(bc->blob_in, bc->cap_in, (size_t) n * L);
        double *blob = bc->blob_in;
        const std::vector<blob_seg> &tab = bc->seg_in;
        par_instances(n, [&](int lo, int hi) {
            for (int i = lo; i < hi; i++)
                for (const blob_seg &g : tab)
                    memcpy(blob + (size_t) i * L + g.off, in_field(ins[i], g.fid, g.k) + g.shift, sizeof(double) * g.len);
        });
        ocp_qp_gpu_batch_set_bulk(b, blob, 0);
    }
    if (o->warm_start >= 2)
    {
        /* hot start: the iterate in qp_out is the starting point (acados_ocp_options.py:1029-1032) */
        for (int k = 0; k <= N; k++)
        {
            const int nu = d->nu[k], nx = d->nx[k], ns = d->ns[k];
            auto pusho = [&](const char *name, int len, auto getter) {
                if (len <= 0) return;
                stg.resize((size_t) n * len);
                for (int i = 0; i < n; i++) memcpy(stg.data() + (size_t) i * len, getter(outs[i]), sizeof(double) * len);
                ocp_qp_gpu_batch_set(b, name, k, stg.data(), 0);
            };
            pusho("u", nu, [&](ocp_qp_out *q) { return q->ux[k]; });
            pusho("x", nx, [&](ocp_qp_out *q) { return q->ux[k] + nu; });
            pusho("sl", ns, [&](ocp_qp_out *q) { return q->ux[k] + nu + nx; });
            pusho("su", ns, [&](ocp_qp_out *q) { return q->ux[k] + nu + nx + ns; });
            if (k < N) pusho("pi", d->nx[k + 1], [&](ocp_qp_out *q) { return q->pi[k]; });
            pusho("lam", 2 * (d->nb[k] + d->ng[k] + ns), [&](ocp_qp_out *q) { return q->lam[k]; });
            pusho("t", 2 * (d->nb[k] + d->ng[k] + ns), [&](ocp_qp_out *q) { return q->t[k]; });
        }
    }
    const double t_packed = now_s();

    ocp_qp_gpu_batch_solve(b);
    const double t_solved = now_s();

    /* unpack: one gather launch + one device->host copy for the whole batch, then host threads scatter */
    {
        const size_t L = (size_t) bc->L_out;
        pinned_reserve(bc->blob_out, bc->cap_out, (size_t) n * L);
        ocp_qp_gpu_batch_get_bulk(b, bc->blob_out, 0);
        const double *blob = bc->blob_out;
        const std::vector<blob_seg> &tab = bc->seg_out;
        par_instances(n, [&](int lo, int hi) {
            for (int i = lo; i < hi; i++)
                for (const blob_seg &g : tab)
                    memcpy(out_field(outs[i], g.fid, g.k) + g.shift, blob + (size_t) i * L + g.off, sizeof(double) * g.len);
        });
    }
    std::vector<int> st(n), it(n);
    ocp_qp_gpu_batch_get_info(b, "status", st.data());
    ocp_qp_gpu_batch_get_info(b, "iter", it.data());
    bc->stat.assign((size_t) 20 * (o->iter_max + 2), 0.0);
    ocp_qp_gpu_batch_get_stat(b, 0, bc->stat.data(), o->iter_max + 2);
    const double t_end = now_s();

    int worst = 0;
    for (int i = 0; i < n; i++)
    {
        qp_info *info = (qp_info *) outs[i]->misc;
        info->condensing_time = ocp_qp_gpu_batch_get_scalar(b, "time_xcond");
        info->solve_QP_time = ocp_qp_gpu_batch_get_scalar(b, "time_tot") - info->condensing_time;
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

/* ocp_qp_hpipm.c:314-405 */
int ocp_qp_gpu_ipm(void *config, void *qp_in, void *qp_out, void *opts, void *mem, void *work)
{
    int status = 0;
    void *ins[1] = {qp_in}, *outs[1] = {qp_out}, *mems[1] = {mem};
    ocp_qp_gpu_ipm_evaluate_batch(config, 1, ins, outs, opts, mems, work, &status);
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

/* ------------------------------------------------------ outer level */

struct ocp_qp_xcond_solver_config_
{
    qp_solver_config qp_solver;
    char name[64];
};

struct ocp_qp_xcond_solver_dims_
{
    ocp_qp_dims *orig_dims;
};

struct xcond_solver_opts
{
    void *qp_solver_opts;
    int cond_N;
    int cond_ric_alg;
    bool initialize_next_xcond_qp_from_qp_out;
    bool warned;
    int *cond_block_size; /* cond_N + 1 entries or NULL (ocp_qp_partial_condensing.c:305-313) */
};

struct ocp_qp_solver_
{
    ocp_qp_xcond_solver_config *config;
    ocp_qp_xcond_solver_dims *dims;
    xcond_solver_opts *opts;
    void *mem;
    void *mem_raw;
};

/* ocp_qp_interface.c:185-259 */
ocp_qp_xcond_solver_config *ocp_qp_xcond_solver_config_create_from_name(const char *name)
{
    if (strcmp(name, "PARTIAL_CONDENSING_GPU_IPM") && strcmp(name, "PARTIAL_CONDENSING_HPIPM"))
    {
        printf("\nerror: ocp_qp_xcond_solver_config_create_from_name: QP solver %s not supported by acados_amd\n", name);
        return nullptr;
    }
    ocp_qp_xcond_solver_config *c = (ocp_qp_xcond_solver_config *) calloc(1, sizeof(*c));
    ocp_qp_gpu_ipm_config_initialize_default(&c->qp_solver);
    snprintf(c->name, sizeof(c->name), "%s", name);
    return c;
}

void ocp_qp_xcond_solver_config_free(ocp_qp_xcond_solver_config *c) { free(c); }

ocp_qp_xcond_solver_dims *ocp_qp_xcond_solver_dims_create(ocp_qp_xcond_solver_config *config, int N)
{
    ocp_qp_xcond_solver_dims *d = (ocp_qp_xcond_solver_dims *) calloc(1, sizeof(*d));
    d->orig_dims = ocp_qp_dims_create(N);
    return d;
}

void ocp_qp_xcond_solver_dims_set(void *config, ocp_qp_xcond_solver_dims *dims, int stage, const char *field, int *value)
{
    ocp_qp_dims_set(config, dims->orig_dims, stage, field, value);
}

void ocp_qp_xcond_solver_dims_free(ocp_qp_xcond_solver_dims *d)
{
    if (!d) return;
    ocp_qp_dims_free(d->orig_dims);
    free(d);
}

void *ocp_qp_xcond_solver_opts_create(ocp_qp_xcond_solver_config *config, ocp_qp_xcond_solver_dims *dims)
{
    xcond_solver_opts *o = (xcond_solver_opts *) calloc(1, sizeof(*o));
    qp_solver_config *qs = &config->qp_solver;
    void *raw = calloc(1, qs->opts_calculate_size(qs, dims->orig_dims));
    o->qp_solver_opts = qs->opts_assign(qs, dims->orig_dims, raw);
    qs->opts_initialize_default(qs, dims->orig_dims, o->qp_solver_opts);
    o->cond_N = dims->orig_dims->N;
    o->cond_ric_alg = 1;
    return o;
}

/* ocp_qp_xcond_solver.c:280-312: "cond_" prefix goes to the condensing module */
void ocp_qp_xcond_solver_opts_set(ocp_qp_xcond_solver_config *config, void *opts_, const char *field, void *value)
{
    xcond_solver_opts *o = (xcond_solver_opts *) opts_;
    if (!strncmp(field, "cond_", 5))
    {
        const char *f = field + 5;
        if (!strcmp(f, "N")) o->cond_N = *(int *) value;
        else if (!strcmp(f, "ric_alg")) o->cond_ric_alg = *(int *) value;
        else if (!strcmp(f, "block_size"))
        {
            /* N2 + 1 entries; N2 ("cond_N") has to be set before, as in the reference (:305-313) */
            free(o->cond_block_size);
            o->cond_block_size = (int *) malloc(sizeof(int) * (o->cond_N + 1));
            for (int i = 0; i <= o->cond_N; i++) o->cond_block_size[i] = ((int *) value)[i];
        }
        else
        {
            printf("\nerror: field %s not available in ocp_qp_partial_condensing_opts_set\n", f);
            exit(1);
        }
    }
    else if (!strcmp(field, "initialize_next_xcond_qp_from_qp_out"))
        o->initialize_next_xcond_qp_from_qp_out = *(bool *) value;
    else
        config->qp_solver.opts_set(&config->qp_solver, o->qp_solver_opts, field, value);
}

void ocp_qp_xcond_solver_opts_free(void *opts_)
{
    xcond_solver_opts *o = (xcond_solver_opts *) opts_;
    if (!o) return;
    free(o->qp_solver_opts);
    free(o->cond_block_size);
    free(o);
}

ocp_qp_in *ocp_qp_in_create_from_xcond_dims(ocp_qp_xcond_solver_dims *dims) { return ocp_qp_in_create(dims->orig_dims); }
ocp_qp_out *ocp_qp_out_create_from_xcond_dims(ocp_qp_xcond_solver_dims *dims) { return ocp_qp_out_create(dims->orig_dims); }

/* ocp_qp_interface.c:513-563 */
ocp_qp_solver *ocp_qp_create(ocp_qp_xcond_solver_config *config, ocp_qp_xcond_solver_dims *dims, void *opts_)
{
    ocp_qp_solver *s = (ocp_qp_solver *) calloc(1, sizeof(*s));
    qp_solver_config *qs = &config->qp_solver;
    xcond_solver_opts *o = (xcond_solver_opts *) opts_;
    s->config = config; s->dims = dims; s->opts = o;
    s->mem_raw = calloc(1, qs->memory_calculate_size(qs, dims->orig_dims, o->qp_solver_opts));
    s->mem = qs->memory_assign(qs, dims->orig_dims, o->qp_solver_opts, s->mem_raw);
    return s;
}

void ocp_qp_solver_destroy(ocp_qp_solver *s)
{
    if (!s) return;
    s->config->qp_solver.terminate(&s->config->qp_solver, s->mem, nullptr);
    free(s->mem_raw);
    free(s);
}

static void xcond_note(ocp_qp_solver *s)
{
    /* the condensing request travels with the call (ocp_qp_xcond_solve: condense -> solve -> expand,
     * ocp_qp_xcond_solver.c:529-587); the device batch decides whether the QP class is condensable */
    g_cond_N_request = s->opts->cond_N;
    g_cond_blocks_request = s->opts->cond_block_size;
}

/* ocp_qp_interface.c:567-571 -> ocp_qp_xcond_solve (ocp_qp_xcond_solver.c:529-587) */
int ocp_qp_solve(ocp_qp_solver *s, ocp_qp_in *qp_in, ocp_qp_out *qp_out)
{
    xcond_note(s);
    qp_solver_config *qs = &s->config->qp_solver;
    return qs->evaluate(qs, qp_in, qp_out, s->opts->qp_solver_opts, s->mem, nullptr);
}

int ocp_qp_solve_batch(ocp_qp_solver *s, int n, ocp_qp_in **qp_in, ocp_qp_out **qp_out, int *status)
{
    xcond_note(s);
    std::vector<void *> mems(n, nullptr);
    mems[0] = s->mem;
    return ocp_qp_gpu_ipm_evaluate_batch(&s->config->qp_solver, n, (void **) qp_in, (void **) qp_out,
                                         s->opts->qp_solver_opts, mems.data(), nullptr, status);
}

/* ocp_qp_interface.c:573-595 */
void ocp_qp_xcond_solver_get_scalar(ocp_qp_solver *s, ocp_qp_out *qp_out, const char *field, void *value)
{
    qp_info *info = (qp_info *) qp_out->misc;
    if (!strcmp(field, "time_tot")) *(double *) value = info->total_time;
    else if (!strcmp(field, "time_cond") || !strcmp(field, "time_qp_xcond")) *(double *) value = info->condensing_time;
    else s->config->qp_solver.memory_get(&s->config->qp_solver, s->mem, field, value);
}

/* outer-level access to the solver_get slot (what ocp_nlp_ddp.c:373-377 does through the xcond vtable) */
void ocp_qp_solver_eval_forw_sens(ocp_qp_solver *s, ocp_qp_in *qp_in, ocp_qp_seed *seed, ocp_qp_out *sens_out)
{
    qp_solver_config *qs = &s->config->qp_solver;
    qs->eval_forw_sens(qs, qp_in, seed, sens_out, s->opts->qp_solver_opts, s->mem, nullptr);
}

void ocp_qp_solver_eval_adj_sens(ocp_qp_solver *s, ocp_qp_in *qp_in, ocp_qp_seed *seed, ocp_qp_out *sens_out)
{
    qp_solver_config *qs = &s->config->qp_solver;
    qs->eval_adj_sens(qs, qp_in, seed, sens_out, s->opts->qp_solver_opts, s->mem, nullptr);
}

void ocp_qp_solver_get_ric(ocp_qp_solver *s, ocp_qp_in *qp_in, ocp_qp_out *qp_out, const char *field, int stage,
                           void *value, int size1, int size2)
{
    qp_solver_config *qs = &s->config->qp_solver;
    qs->solver_get(qs, qp_in, qp_out, s->opts->qp_solver_opts, s->mem, field, stage, value, size1, size2);
}

/* ocp_qp_interface.c:597-610 */
void ocp_qp_solver_get_stats(ocp_qp_solver *s, double *stat_out, const char *qp_solver_name)
{
    int iter, stat_m;
    double *stat;
    qp_solver_config *qs = &s->config->qp_solver;
    qs->memory_get(qs, s->mem, "iter", &iter);
    qs->memory_get(qs, s->mem, "stat", &stat);
    qs->memory_get(qs, s->mem, "stat_m", &stat_m);
    if (!stat) return;
    for (int i = 0; i < stat_m * (iter + 1); i++) stat_out[i] = stat[i];
}

/* ---- condensing-only boundary (interfaces/acados_c/condensing_interface.c; the `condensing` / `expansion` slots of
 *      ocp_qp_xcond_config, ocp_qp_common.h:84-107, filled by ocp_qp_partial_condensing.c:523-556, :664-689) ----
 * A module owns a one-instance device batch; the condensed QP is handed out in a plain ocp_qp_in of the condensed
 * dims ("xcond_dims"), the solution of it comes back in a plain ocp_qp_out. */
struct ocp_qp_condensing_module_
{
    ocp_qp_dims *dims = nullptr, *xdims = nullptr;
    int cond_N = 0;
    std::vector<int> blocks;
    ocp_qp_gpu_batch *batch = nullptr, *child = nullptr;
    std::vector<int> sig;
};

struct cfield { const char *name; int fid; int dyn; };
static const cfield k_cfields[] = {
    {"A", F_A, 1}, {"B", F_B, 1}, {"b", F_b, 1}, {"Q", F_Q, 0}, {"S", F_S, 0}, {"R", F_R, 0}, {"q", F_q, 0}, {"r", F_r, 0},
    {"C", F_C, 0}, {"D", F_D, 0}, {"lg", F_lg, 0}, {"ug", F_ug, 0}, {"lg_mask", F_lgm, 0}, {"ug_mask", F_ugm, 0},
    {"Zl", F_Zl, 0}, {"Zu", F_Zu, 0}, {"zl", F_zl, 0}, {"zu", F_zu, 0}, {"lls", F_lls, 0}, {"lus", F_lus, 0},
    {"lls_mask", F_llsm, 0}, {"lus_mask", F_lusm, 0}};

static int clen(const ocp_qp_dims *d, const char *f, int k)
{
    const int n = vlen(d, f, k);
    if (n >= 0) return n;
    return (f[0] == 'l' || f[0] == 'u') && f[1] == 'g' ? d->ng[k] : d->ns[k]; /* lg ug (+ masks) | Zl Zu zl zu lls lus (+ masks) */
}

static ocp_qp_gpu_batch *condensing_batch(ocp_qp_condensing_module *m, const ocp_qp_dims *d, int *const *idxb,
                                          int *const *idxs_rev, int *const *idxe)
{
    ocp_qp_gpu_batch *b = ocp_qp_gpu_batch_create(d->N, d->nx, d->nu, d->nbx, d->nbu, d->ng, d->ns, 1, -1);
    if (!b) return nullptr;
    for (int k = 0; k <= d->N; k++)
    {
        std::vector<int> ib(d->nb[k]), rev(d->nb[k] + d->ng[k], -1), ie(d->nbxe[k]);
        for (int r = 0; r < d->nb[k]; r++) ib[r] = idxb ? idxb[k][r] : (r < d->nbu[k] ? r : d->nu[k] + r - d->nbu[k]);
        if (idxs_rev) for (size_t r = 0; r < rev.size(); r++) rev[r] = idxs_rev[k][r];
        for (int r = 0; r < d->nbxe[k]; r++) ie[r] = idxe ? idxe[k][r] : d->nbu[k] + r;
        ocp_qp_gpu_batch_set_int(b, "idxb", k, ib.data(), (int) ib.size());
        ocp_qp_gpu_batch_set_int(b, "idxs_rev", k, rev.data(), (int) rev.size());
        ocp_qp_gpu_batch_set_int(b, "idxe", k, ie.data(), (int) ie.size());
    }
    ocp_qp_gpu_batch_opts_set(b, "cond_N", &m->cond_N);
    if (!m->blocks.empty() && ocp_qp_gpu_batch_opts_set(b, "cond_block_size", m->blocks.data()) != 0)
    {
        printf("\nerror: partial condensing: sum of block_size should match N = %d\n", d->N);
        exit(1); /* ocp_qp_partial_condensing.c:352-356 */
    }
    return b;
}

ocp_qp_condensing_module *ocp_qp_condensing_create(ocp_qp_dims *dims, int cond_N, const int *block_size)
{
    ocp_qp_condensing_module *m = new ocp_qp_condensing_module_();
    m->dims = dims;
    m->cond_N = cond_N;
    if (block_size) m->blocks.assign(block_size, block_size + cond_N + 1);
    /* the condensed dims depend on the per-stage counts only: a batch with the default index sets tells them */
    ocp_qp_gpu_batch *probe = condensing_batch(m, dims, nullptr, nullptr, nullptr);
    ocp_qp_gpu_batch *c = probe ? ocp_qp_gpu_batch_condense(probe) : nullptr;
    if (!c)
    {
        printf("\nerror: ocp_qp_condensing_create: this QP class is not condensed to N2 = %d (no device, cond_N outside 1..N-1, "
               "or beyond the limits of the condensing kernels)\n", cond_N);
        if (probe) ocp_qp_gpu_batch_destroy(probe);
        delete m;
        return nullptr;
    }
    m->xdims = ocp_qp_dims_create(cond_N);
    const char *names[] = {"nx", "nu", "nb", "nbx", "nbu", "ng", "ns", "nbxe"};
    int *dst[] = {m->xdims->nx, m->xdims->nu, m->xdims->nb, m->xdims->nbx, m->xdims->nbu, m->xdims->ng, m->xdims->ns, m->xdims->nbxe};
    for (int q = 0; q < 8; q++) ocp_qp_gpu_batch_get_dims(c, names[q], dst[q]);
    ocp_qp_gpu_batch_destroy(probe);
    return m;
}

void ocp_qp_condensing_free(ocp_qp_condensing_module *m)
{
    if (!m) return;
    if (m->batch) ocp_qp_gpu_batch_destroy(m->batch);
    ocp_qp_dims_free(m->xdims);
    delete m;
}

ocp_qp_dims *ocp_qp_condensing_get_xcond_dims(ocp_qp_condensing_module *m) { return m->xdims; }

/* bounds and their masks: the containers keep [bu; bx] in one array, the device batch takes them per kind */
static void condensing_bounds(ocp_qp_gpu_batch *b, ocp_qp_in *q, int k, bool to_device)
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

int ocp_qp_condense(ocp_qp_condensing_module *m, void *qp_in_, void *xcond_qp_in_)
{
    ocp_qp_in *in = (ocp_qp_in *) qp_in_, *xc = (ocp_qp_in *) xcond_qp_in_;
    const ocp_qp_dims *d = in->dim;
    std::vector<int> sig = structure_sig(in);
    if (!m->batch || sig != m->sig)
    {
        if (m->batch) ocp_qp_gpu_batch_destroy(m->batch);
        m->batch = condensing_batch(m, d, in->idxb, in->idxs_rev, in->idxe);
        m->sig = sig;
        if (!m->batch) return ACADOS_QP_FAILURE;
    }
    ocp_qp_gpu_batch *b = m->batch;
    for (int k = 0; k <= d->N; k++)
    {
        for (const cfield &f : k_cfields)
            if (!(f.dyn && k == d->N) && clen(d, f.name, k) > 0) ocp_qp_gpu_batch_set(b, f.name, k, in_field(in, f.fid, k), 0);
        condensing_bounds(b, in, k, true);
    }
    ocp_qp_gpu_batch *c = ocp_qp_gpu_batch_condense(b);
    m->child = c;
    if (!c) return ACADOS_QP_FAILURE;
    const ocp_qp_dims *xd = xc->dim;
    for (int k = 0; k <= xd->N; k++)
    {
        for (const cfield &f : k_cfields)
            if (!(f.dyn && k == xd->N) && clen(xd, f.name, k) > 0)
                ocp_qp_gpu_batch_get(c, f.name, k, const_cast<double *>(in_field(xc, f.fid, k)), 0);
        condensing_bounds(c, xc, k, false);
        ocp_qp_gpu_batch_get_int(c, "idxb", k, xc->idxb[k]);
        ocp_qp_gpu_batch_get_int(c, "idxs_rev", k, xc->idxs_rev[k]);
        ocp_qp_gpu_batch_get_int(c, "idxe", k, xc->idxe[k]);
    }
    return ACADOS_SUCCESS;
}

/* ux = [u; x; sl; su], pi, lam, t of one container <-> the fields of a one-instance batch */
static void condensing_solution(ocp_qp_gpu_batch *b, const ocp_qp_dims *d, ocp_qp_out *o, bool to_device)
{
    for (int k = 0; k <= d->N; k++)
    {
        const int nu = d->nu[k], nx = d->nx[k], ns = d->ns[k], nct = d->nb[k] + d->ng[k] + ns;
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

int ocp_qp_expand(ocp_qp_condensing_module *m, void *xcond_qp_out_, void *qp_out_)
{
    ocp_qp_out *xo = (ocp_qp_out *) xcond_qp_out_, *out = (ocp_qp_out *) qp_out_;
    if (!m->batch || !m->child) return ACADOS_QP_FAILURE;
    condensing_solution(m->child, m->xdims, xo, true);
    if (ocp_qp_gpu_batch_expand(m->batch) != 0) return ACADOS_QP_FAILURE;
    condensing_solution(m->batch, m->dims, out, false);
    if (out->misc) ((qp_info *) out->misc)->t_computed = 1;
    return ACADOS_SUCCESS;
}

} /* extern "C" */
