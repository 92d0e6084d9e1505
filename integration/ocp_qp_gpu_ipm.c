/*
 * ocp_qp_gpu_ipm.c -- the acados-side adapter of the MI355X OCP-QP backend: the file a maintainer drops into
 * acados/ocp_qp/ next to ocp_qp_hpipm.c.  It is written against acados' OWN types -- ocp_qp_in / ocp_qp_out are HPIPM's
 * d_ocp_qp / d_ocp_qp_sol holding BLASFEO matrices (acados/ocp_qp/ocp_qp_common.h:49-54), panel-major in the default
 * build (external/CMakeLists.txt:46) -- and talks to libacados_amd_qp.so through the device-batch C-ABI
 * (include/acados_amd/ocp_qp_gpu_batch.h) only.  It fills the 17 slots of qp_solver_config (ocp_qp_common.h:60-79):
 *
 *     ocp_qp_gpu_ipm_acados_config_initialize_default(config->qp_solver);
 *
 * in the `case PARTIAL_CONDENSING_GPU_IPM:` of ocp_qp_xcond_solver_config_initialize_from_plan
 * (interfaces/acados_c/ocp_qp_interface.c:91-182), with HPIPM's partial condensing in the xcond slot
 * (INTEGRATION.md section 3).
 *
 * Data access rule followed (pattern of acados/ocp_qp/ocp_qp_clarabel.c:205-683, 1018-1072): matrices only
 * through blasfeo_unpack_dmat / blasfeo_unpack_tran_dmat, vectors through blasfeo_unpack_dvec / blasfeo_pack_dvec;
 * r, q, b are taken from the VECTORS rqz / b, never from the last rows of RSQrq / BAbt (ocp_nlp writes only the vectors,
 * ocp_nlp_common.c:3119-3138); d = [lb; lg; -ub; -ug; ls; us] (ocp_qp_common.c:897-906) -> natural-sign bounds; every
 * member array is re-read on every evaluate (they alias ocp_nlp memory, ocp_nlp_common.c:2797-2894).
 *
 * Memory rule: opts and memory are carved from the caller's block (sizes from dims); the only things kept outside are
 * the device batch and its stream (released by `terminate`, ocp_qp_common.h:78).  No malloc in evaluate.
 *
 * In this repository the file is compiled and RUN in the test tiers against tests/mock_acados/include (stand-ins for
 * the HPIPM / BLASFEO / acados declarations restated from the fields acados touches): tests/test_mock_acados.py.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "acados/ocp_qp/ocp_qp_common.h"
#include "acados/utils/types.h"
#include "blasfeo_d_aux.h"

#include "acados_amd/ocp_qp_gpu_batch.h"

typedef struct
{
    double mu0, tol_stat, tol_eq, tol_ineq, tol_comp, alpha_min, tau_min, reg_prim, t0_min, lam0_min;
    int iter_max, warm_start, print_level, ric_alg, t0_init, update_fact_exit;
} ocp_qp_gpu_ipm_opts;

typedef struct
{
    ocp_qp_gpu_batch *batch;  /* device-side resource, released by terminate */
    int *sig;                 /* structure the batch was built for (carved) */
    int sig_len, sig_cap;
    double *blob_in, *blob_out; /* host staging of the bulk pack / unpack (carved) */
    int cap_in, cap_out;
    double time_qp_solver_call;
    int iter, status;
} ocp_qp_gpu_ipm_memory;

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

static char *align8(char *p) { return (char *) (((size_t) p + 7) & ~(size_t) 7); }

/* ------------------------------------------------------------------ sizes from dims */

static int sig_len(const ocp_qp_dims *d)
{
    int len = 1;
    for (int k = 0; k <= d->N; k++) len += 7 + 2 * d->nb[k] + d->ng[k] + d->nbxe[k];
    return len;
}

static int blob_in_cap(const ocp_qp_dims *d)
{
    int len = 0;
    for (int k = 0; k <= d->N; k++)
    {
        const int nx = d->nx[k], nu = d->nu[k], nx1 = k < d->N ? d->nx[k + 1] : 0;
        len += nx1 * (nx + nu + 1) + (nu + nx) * (nu + nx) + nu + nx + 5 * d->nb[k] + d->ng[k] * (nu + nx) + 4 * d->ng[k] + 8 * d->ns[k];
    }
    return len;
}

static int blob_out_cap(const ocp_qp_dims *d)
{
    int len = 0;
    for (int k = 0; k <= d->N; k++)
        len += d->nu[k] + d->nx[k] + 2 * d->ns[k] + (k < d->N ? d->nx[k + 1] : 0) + 4 * (d->nb[k] + d->ng[k] + d->ns[k]);
    return len;
}

/* ------------------------------------------------------------------ opts (ocp_qp_hpipm.c:60-183) */

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

static acados_size_t gpu_opts_calculate_size(void *config, void *dims) { return sizeof(ocp_qp_gpu_ipm_opts) + 8; }
static void *gpu_opts_assign(void *config, void *dims, void *raw_memory) { return align8((char *) raw_memory); }

static void gpu_opts_initialize_default(void *config, void *dims, void *opts_)
{
    ocp_qp_gpu_ipm_opts *o = (ocp_qp_gpu_ipm_opts *) opts_;
    /* mode BALANCE + the acados overrides, ocp_qp_hpipm.c:101-113 */
    o->mu0 = 1e0; o->tol_stat = 1e-6; o->tol_eq = 1e-8; o->tol_ineq = 1e-8; o->tol_comp = 1e-8; o->alpha_min = 1e-8;
    o->tau_min = 0.0; o->reg_prim = 1e-15; o->t0_min = 1e-16; o->lam0_min = 1e-16;
    o->iter_max = 50; o->warm_start = 0; o->print_level = 0; o->ric_alg = 1; o->t0_init = 2; o->update_fact_exit = 0;
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
    else if (!strcmp(field, "ric_alg")) o->ric_alg = *i;
    else if (!strcmp(field, "t0_min")) o->t0_min = *d;
    else if (!strcmp(field, "lam0_min")) o->lam0_min = *d;
    else if (!strcmp(field, "update_fact_exit")) o->update_fact_exit = *i;
    else if (!strcmp(field, "hpipm_mode")) { /* one IPM variant; the acados overrides above hold for every mode */ }
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

static acados_size_t gpu_memory_calculate_size(void *config, void *dims_, void *opts)
{
    const ocp_qp_dims *d = (const ocp_qp_dims *) dims_;
    return sizeof(ocp_qp_gpu_ipm_memory) + sizeof(int) * (size_t) sig_len(d) + sizeof(double) * (size_t) (blob_in_cap(d) + blob_out_cap(d)) + 4 * 8;
}

static void *gpu_memory_assign(void *config, void *dims_, void *opts, void *raw_memory)
{
    const ocp_qp_dims *d = (const ocp_qp_dims *) dims_;
    char *c = align8((char *) raw_memory);
    ocp_qp_gpu_ipm_memory *m = (ocp_qp_gpu_ipm_memory *) c;
    memset(m, 0, sizeof(*m));
    c = align8(c + sizeof(*m));
    m->cap_in = blob_in_cap(d); m->cap_out = blob_out_cap(d);
    m->blob_in = (double *) c; c += sizeof(double) * (size_t) m->cap_in;
    m->blob_out = (double *) c; c += sizeof(double) * (size_t) m->cap_out;
    m->sig_cap = sig_len(d);
    m->sig = (int *) c;
    return m;
}

static void gpu_memory_get(void *config, void *mem_, const char *field, void *value)
{
    ocp_qp_gpu_ipm_memory *m = (ocp_qp_gpu_ipm_memory *) mem_;
    if (!strcmp(field, "time_qp_solver_call")) *(double *) value = m->time_qp_solver_call;
    else if (!strcmp(field, "iter")) *(int *) value = m->iter;
    else if (!strcmp(field, "status")) *(int *) value = m->status;
    else { printf("\nerror: ocp_qp_gpu_ipm_memory_get: field %s not available\n", field); exit(1); }
}

static acados_size_t gpu_workspace_calculate_size(void *config, void *dims, void *opts) { return 0; }

/* ------------------------------------------------------------------ evaluate (ocp_qp_hpipm.c:314-405) */

static int fill_sig(const ocp_qp_in *in, int *s)
{
    const ocp_qp_dims *d = in->dim;
    int p = 0;
    s[p++] = d->N;
    for (int k = 0; k <= d->N; k++)
    {
        const int v[7] = {d->nx[k], d->nu[k], d->nbx[k], d->nbu[k], d->ng[k], d->ns[k], d->nbxe[k]};
        memcpy(s + p, v, sizeof(v)); p += 7;
        memcpy(s + p, in->idxb[k], sizeof(int) * d->nb[k]); p += d->nb[k];
        memcpy(s + p, in->idxs_rev[k], sizeof(int) * (d->nb[k] + d->ng[k])); p += d->nb[k] + d->ng[k];
        memcpy(s + p, in->idxe[k], sizeof(int) * d->nbxe[k]); p += d->nbxe[k];
    }
    return p;
}

/* destination of one field in the input blob (NULL when the field has no entry at this stage) */
static double *slot(ocp_qp_gpu_ipm_memory *m, const char *field, int k, int expect)
{
    int len = 0;
    const int off = ocp_qp_gpu_batch_bulk_offset(m->batch, 0, field, k, &len);
    if (off < 0 || len == 0) return NULL;
    if (len != expect)
    {
        printf("\nerror: ocp_qp_gpu_ipm: field %s at stage %d has %d entries in the device layout, %d in qp_in\n", field, k, len, expect);
        exit(1);
    }
    return m->blob_in + off;
}

static int ocp_qp_gpu_ipm_acados(void *config, void *qp_in_, void *qp_out_, void *opts_, void *mem_, void *work)
{
    const double t_start = now_s();
    ocp_qp_in *in = (ocp_qp_in *) qp_in_;
    ocp_qp_out *out = (ocp_qp_out *) qp_out_;
    ocp_qp_gpu_ipm_opts *o = (ocp_qp_gpu_ipm_opts *) opts_;
    ocp_qp_gpu_ipm_memory *m = (ocp_qp_gpu_ipm_memory *) mem_;
    const ocp_qp_dims *d = in->dim;
    const int N = d->N;
    qp_info *info = (qp_info *) out->misc;

    /* device batch: (re)created when the structure changes; the signature is compared in carved memory */
    {
        int *scratch = (int *) m->blob_out; /* blob_out is idle until the unpack and large enough */
        const int len = fill_sig(in, scratch);
        if (len > m->sig_cap) { printf("\nerror: ocp_qp_gpu_ipm: dims of qp_in grew after memory_assign\n"); exit(1); }
        if (!m->batch || m->sig_len != len || memcmp(m->sig, scratch, sizeof(int) * len) != 0)
        {
            if (m->batch) ocp_qp_gpu_batch_destroy(m->batch);
            m->batch = ocp_qp_gpu_batch_create(N, d->nx, d->nu, d->nbx, d->nbu, d->ng, d->ns, 1, -1);
            if (!m->batch) { printf("\nerror: ocp_qp_gpu_ipm: no GPU batch could be created (no device or unsupported shape)\n"); exit(1); }
            memcpy(m->sig, scratch, sizeof(int) * len);
            m->sig_len = len;
            for (int k = 0; k <= N; k++)
            {
                ocp_qp_gpu_batch_set_int(m->batch, "idxb", k, in->idxb[k], d->nb[k]);
                ocp_qp_gpu_batch_set_int(m->batch, "idxs_rev", k, in->idxs_rev[k], d->nb[k] + d->ng[k]);
                ocp_qp_gpu_batch_set_int(m->batch, "idxe", k, in->idxe[k], d->nbxe[k]);
            }
            if (ocp_qp_gpu_batch_bulk_len(m->batch, 0) > m->cap_in || ocp_qp_gpu_batch_bulk_len(m->batch, 1) > m->cap_out)
            {
                printf("\nerror: ocp_qp_gpu_ipm: bulk blob larger than the carved staging\n");
                exit(1);
            }
        }
    }
    ocp_qp_gpu_batch *b = m->batch;

    /* every member array of qp_in, re-read on every call, unpacked from BLASFEO storage straight into the blob */
    memset(m->blob_in, 0, sizeof(double) * (size_t) ocp_qp_gpu_batch_bulk_len(b, 0));
    for (int k = 0; k <= N; k++)
    {
        const int nu = d->nu[k], nx = d->nx[k], nx1 = k < N ? d->nx[k + 1] : 0;
        const int nbu = d->nbu[k], nbx = d->nbx[k], nb = d->nb[k], ng = d->ng[k], ns = d->ns[k];
        double *p;
        if (k < N)
        {
            /* BAbt = [B'; A'; b'] (print.c:234-325): A (nx+ x nx) = (rows nu.. of BAbt)', B (nx+ x nu) = (rows 0..nu)' */
            if ((p = slot(m, "A", k, nx1 * nx))) blasfeo_unpack_tran_dmat(nx, nx1, in->BAbt + k, nu, 0, p, nx1);
            if ((p = slot(m, "B", k, nx1 * nu))) blasfeo_unpack_tran_dmat(nu, nx1, in->BAbt + k, 0, 0, p, nx1);
            if ((p = slot(m, "b", k, nx1))) blasfeo_unpack_dvec(nx1, in->b + k, 0, p, 1); /* the VECTOR, not the last row */
        }
        /* RSQrq: lower triangle of [[R, S], [S', Q]] -- only the lower triangle is valid */
        if ((p = slot(m, "R", k, nu * nu))) blasfeo_unpack_dmat(nu, nu, in->RSQrq + k, 0, 0, p, nu);
        if ((p = slot(m, "S", k, nu * nx))) blasfeo_unpack_tran_dmat(nx, nu, in->RSQrq + k, nu, 0, p, nu); /* S (nu x nx) = (S')' */
        if ((p = slot(m, "Q", k, nx * nx))) blasfeo_unpack_dmat(nx, nx, in->RSQrq + k, nu, nu, p, nx);
        /* rqz = [r; q; zl; zu]: the vectors ocp_nlp writes every iteration */
        if ((p = slot(m, "r", k, nu))) blasfeo_unpack_dvec(nu, in->rqz + k, 0, p, 1);
        if ((p = slot(m, "q", k, nx))) blasfeo_unpack_dvec(nx, in->rqz + k, nu, p, 1);
        if ((p = slot(m, "zl", k, ns))) blasfeo_unpack_dvec(ns, in->rqz + k, nu + nx, p, 1);
        if ((p = slot(m, "zu", k, ns))) blasfeo_unpack_dvec(ns, in->rqz + k, nu + nx + ns, p, 1);
        /* d = [lb; lg; -ub; -ug; ls; us] with lb = [lbu; lbx] (ocp_qp_common.c:897-906): natural sign for the device */
        if ((p = slot(m, "lbu", k, nbu))) blasfeo_unpack_dvec(nbu, in->d + k, 0, p, 1);
        if ((p = slot(m, "lbx", k, nbx))) blasfeo_unpack_dvec(nbx, in->d + k, nbu, p, 1);
        if ((p = slot(m, "lbx#value", k, nbx))) blasfeo_unpack_dvec(nbx, in->d + k, nbu, p, 1); /* equality-flagged: the value of x */
        if ((p = slot(m, "lg", k, ng))) blasfeo_unpack_dvec(ng, in->d + k, nb, p, 1);
        if ((p = slot(m, "ubu", k, nbu))) { blasfeo_unpack_dvec(nbu, in->d + k, nb + ng, p, 1); for (int e = 0; e < nbu; e++) p[e] = -p[e]; }
        if ((p = slot(m, "ubx", k, nbx))) { blasfeo_unpack_dvec(nbx, in->d + k, nb + ng + nbu, p, 1); for (int e = 0; e < nbx; e++) p[e] = -p[e]; }
        if ((p = slot(m, "ug", k, ng))) { blasfeo_unpack_dvec(ng, in->d + k, 2 * nb + ng, p, 1); for (int e = 0; e < ng; e++) p[e] = -p[e]; }
        if ((p = slot(m, "lls", k, ns))) blasfeo_unpack_dvec(ns, in->d + k, 2 * nb + 2 * ng, p, 1);
        if ((p = slot(m, "lus", k, ns))) blasfeo_unpack_dvec(ns, in->d + k, 2 * nb + 2 * ng + ns, p, 1);
        /* d_mask: same positions, 1.0 / 0.0 (aliased to nlp_in->dmask, ocp_nlp_common.c:2894) */
        if ((p = slot(m, "lbu_mask", k, nbu))) blasfeo_unpack_dvec(nbu, in->d_mask + k, 0, p, 1);
        if ((p = slot(m, "lbx_mask", k, nbx))) blasfeo_unpack_dvec(nbx, in->d_mask + k, nbu, p, 1);
        if ((p = slot(m, "lg_mask", k, ng))) blasfeo_unpack_dvec(ng, in->d_mask + k, nb, p, 1);
        if ((p = slot(m, "ubu_mask", k, nbu))) blasfeo_unpack_dvec(nbu, in->d_mask + k, nb + ng, p, 1);
        if ((p = slot(m, "ubx_mask", k, nbx))) blasfeo_unpack_dvec(nbx, in->d_mask + k, nb + ng + nbu, p, 1);
        if ((p = slot(m, "ug_mask", k, ng))) blasfeo_unpack_dvec(ng, in->d_mask + k, 2 * nb + ng, p, 1);
        if ((p = slot(m, "lls_mask", k, ns))) blasfeo_unpack_dvec(ns, in->d_mask + k, 2 * nb + 2 * ng, p, 1);
        if ((p = slot(m, "lus_mask", k, ns))) blasfeo_unpack_dvec(ns, in->d_mask + k, 2 * nb + 2 * ng + ns, p, 1);
        /* Z = [Zl; Zu] */
        if ((p = slot(m, "Zl", k, ns))) blasfeo_unpack_dvec(ns, in->Z + k, 0, p, 1);
        if ((p = slot(m, "Zu", k, ns))) blasfeo_unpack_dvec(ns, in->Z + k, ns, p, 1);
        /* DCt = [D'; C'] ((nu+nx) x ng): C (ng x nx) = (rows nu.. )', D (ng x nu) = (rows 0..nu)' */
        if ((p = slot(m, "C", k, ng * nx))) blasfeo_unpack_tran_dmat(nx, ng, in->DCt + k, nu, 0, p, ng);
        if ((p = slot(m, "D", k, ng * nu))) blasfeo_unpack_tran_dmat(nu, ng, in->DCt + k, 0, 0, p, ng);
    }
    /* options by name */
    ocp_qp_gpu_batch_opts_set(b, "iter_max", &o->iter_max);
    ocp_qp_gpu_batch_opts_set(b, "tol_stat", &o->tol_stat);
    ocp_qp_gpu_batch_opts_set(b, "tol_eq", &o->tol_eq);
    ocp_qp_gpu_batch_opts_set(b, "tol_ineq", &o->tol_ineq);
    ocp_qp_gpu_batch_opts_set(b, "tol_comp", &o->tol_comp);
    ocp_qp_gpu_batch_opts_set(b, "mu0", &o->mu0);
    ocp_qp_gpu_batch_opts_set(b, "tau_min", &o->tau_min);
    ocp_qp_gpu_batch_opts_set(b, "t0_min", &o->t0_min);
    ocp_qp_gpu_batch_opts_set(b, "lam0_min", &o->lam0_min);
    ocp_qp_gpu_batch_opts_set(b, "print_level", &o->print_level);
    const int ws = o->warm_start >= 2 ? o->warm_start : 0; /* 1 = 0, acados_ocp_options.py:1029-1031 */
    ocp_qp_gpu_batch_opts_set(b, "warm_start", &ws);
    if (ws >= 2)
    {
        /* hot start: pi, lam, t of qp_out; primal zeroed as ocp_qp_hpipm.c:325-336 does before every solve.  Written
         * before the pack, which then restores the equality-flagged values (x0) */
        const int Lo = ocp_qp_gpu_batch_bulk_len(b, 1);
        memset(m->blob_out, 0, sizeof(double) * (size_t) Lo);
        for (int k = 0; k <= N; k++)
        {
            const int nct = 2 * (d->nb[k] + d->ng[k] + d->ns[k]);
            int len = 0, off;
            if (k < N && (off = ocp_qp_gpu_batch_bulk_offset(b, 1, "pi", k, &len)) >= 0) blasfeo_unpack_dvec(len, out->pi + k, 0, m->blob_out + off, 1);
            if (nct && (off = ocp_qp_gpu_batch_bulk_offset(b, 1, "lam", k, &len)) >= 0) blasfeo_unpack_dvec(len, out->lam + k, 0, m->blob_out + off, 1);
            if (nct && (off = ocp_qp_gpu_batch_bulk_offset(b, 1, "t", k, &len)) >= 0) blasfeo_unpack_dvec(len, out->t + k, 0, m->blob_out + off, 1);
        }
        ocp_qp_gpu_batch_set_bulk_out(b, m->blob_out, 0);
    }
    ocp_qp_gpu_batch_set_bulk(b, m->blob_in, 0);
    const double t_packed = now_s();

    ocp_qp_gpu_batch_solve(b);
    const double t_solved = now_s();

    /* solution -> qp_out (BLASFEO vectors); lam, t ordered [lb lg ub ug ls us] as HPIPM's */
    ocp_qp_gpu_batch_get_bulk(b, m->blob_out, 0);
    for (int k = 0; k <= N; k++)
    {
        const int nu = d->nu[k], nx = d->nx[k], ns = d->ns[k];
        int len = 0, off;
        if ((off = ocp_qp_gpu_batch_bulk_offset(b, 1, "u", k, &len)) >= 0) blasfeo_pack_dvec(len, m->blob_out + off, 1, out->ux + k, 0);
        if ((off = ocp_qp_gpu_batch_bulk_offset(b, 1, "x", k, &len)) >= 0) blasfeo_pack_dvec(len, m->blob_out + off, 1, out->ux + k, nu);
        if ((off = ocp_qp_gpu_batch_bulk_offset(b, 1, "sl", k, &len)) >= 0) blasfeo_pack_dvec(len, m->blob_out + off, 1, out->ux + k, nu + nx);
        if ((off = ocp_qp_gpu_batch_bulk_offset(b, 1, "su", k, &len)) >= 0) blasfeo_pack_dvec(len, m->blob_out + off, 1, out->ux + k, nu + nx + ns);
        if (k < N && (off = ocp_qp_gpu_batch_bulk_offset(b, 1, "pi", k, &len)) >= 0) blasfeo_pack_dvec(len, m->blob_out + off, 1, out->pi + k, 0);
        if ((off = ocp_qp_gpu_batch_bulk_offset(b, 1, "lam", k, &len)) >= 0) blasfeo_pack_dvec(len, m->blob_out + off, 1, out->lam + k, 0);
        if ((off = ocp_qp_gpu_batch_bulk_offset(b, 1, "t", k, &len)) >= 0) blasfeo_pack_dvec(len, m->blob_out + off, 1, out->t + k, 0);
    }
    int st = 0, it = 0;
    ocp_qp_gpu_batch_get_info(b, "status", &st);
    ocp_qp_gpu_batch_get_info(b, "iter", &it);
    const double t_end = now_s();
    if (info)
    {
        info->solve_QP_time = ocp_qp_gpu_batch_get_scalar(b, "time_tot");
        info->condensing_time = 0.0;
        info->interface_time = (t_packed - t_start) + (t_end - t_solved);
        info->total_time = t_end - t_start;
        info->num_iter = it;
        info->t_computed = 1; /* t comes from the device (ocp_qp_compute_t restated there for the rows the IPM skipped) */
    }
    m->iter = it; m->status = st; m->time_qp_solver_call = t_solved - t_packed;
    return st; /* already return_values_t (acados/utils/types.h:74-87) */
}

static void gpu_solver_get(void *config_, void *qp_in_, void *qp_out_, void *opts_, void *mem_, const char *field, int stage,
                           void *value, int size1, int size2)
{
    /* ocp_qp_hpipm.c:417-478: P p K k Lr from the factor of the last factorisation held in HBM */
    ocp_qp_in *in = (ocp_qp_in *) qp_in_;
    ocp_qp_gpu_ipm_memory *m = (ocp_qp_gpu_ipm_memory *) mem_;
    const int nx = in->dim->nx[stage], nu = in->dim->nu[stage], nv = nu + nx;
    double *out = (double *) value;
    if (!m->batch) { printf("\nocp_qp_gpu_ipm_solver_get: no factorisation available (solve first)\n"); exit(1); }
    double *L = m->blob_in, *l = m->blob_in + nv * nv; /* staging is idle between evaluates: (nu+nx)^2 + nu+nx fit */
    ocp_qp_gpu_batch_get(m->batch, "ric_L", stage, L, 0);
    ocp_qp_gpu_batch_get(m->batch, "ric_l", stage, l, 0);
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
}

static void gpu_memory_reset(void *config, void *qp_in, void *qp_out, void *opts, void *mem_, void *work)
{
    ocp_qp_gpu_ipm_memory *m = (ocp_qp_gpu_ipm_memory *) mem_;
    if (m->batch) ocp_qp_gpu_batch_destroy(m->batch);
    m->batch = NULL;
    m->sig_len = 0;
}

static void gpu_eval_sens(void *config, void *qp_in, void *seed, void *qp_out, void *opts, void *mem, void *work)
{
    /* d_ocp_qp_seed holds BLASFEO vectors seed_g / seed_b / seed_d laid out like rqz / b / d: unpack them as above and
     * hand them to ocp_qp_gpu_batch_sens_set / _sens_solve (ocp_qp_gpu_batch.h); not part of the mock-build test */
    printf("\nerror: ocp_qp_gpu_ipm: eval_forw_sens / eval_adj_sens through the acados adapter: bind ocp_qp_gpu_batch_sens_*\n");
    exit(1);
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
