/*
 * ocp_qp_gpu_segments.h -- shared by the two acados-side files of the MI355X OCP-QP backend (integration/ocp_qp_gpu_ipm.c, the
 * 17-slot QP solver, and integration/ocp_qp_gpu_pcond.c, the 20-slot partial condensing module): where every field of the
 * device library's bulk blobs (include/acados_amd/ocp_qp_gpu_batch.h: ocp_qp_gpu_batch_bulk_offset) lives inside acados' OWN
 * containers -- ocp_qp_in / ocp_qp_out / ocp_qp_seed are HPIPM's d_ocp_qp / d_ocp_qp_sol / d_ocp_qp_seed holding BLASFEO
 * matrices and vectors (acados/ocp_qp/ocp_qp_common.h:49-54), panel-major in the default build -- and the loops that move one
 * instance between the two.  Data access rule (pattern of acados/ocp_qp/ocp_qp_clarabel.c:205-683, 1018-1072):
 * matrices only through blasfeo_unpack_dmat / blasfeo_unpack_tran_dmat / blasfeo_pack_dmat / blasfeo_pack_tran_dmat, vectors
 * through blasfeo_unpack_dvec / blasfeo_pack_dvec; r, q, b from the VECTORS rqz / b.
 * Everything here is `static`: the header is included by exactly these two translation units.
 */
#ifndef ACADOS_OCP_QP_OCP_QP_GPU_SEGMENTS_H_
#define ACADOS_OCP_QP_OCP_QP_GPU_SEGMENTS_H_

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "acados/ocp_qp/ocp_qp_common.h"
#include "acados/utils/types.h"
#include "blasfeo/include/blasfeo_d_aux.h"
#include "blasfeo/include/blasfeo_d_aux_ext_dep.h"

#include "acados_amd/ocp_qp_gpu_batch.h"

/* (not every function is used by both translation units) */
#if defined(__GNUC__)
#define GPU_SEG_FN static __attribute__((unused))
#else
#define GPU_SEG_FN static
#endif

GPU_SEG_FN char *align8(char *p) { return (char *) (((size_t) p + 7) & ~(size_t) 7); }
/* every *_calculate_size is a multiple of 8: the reference carves the next object right behind and asserts its alignment
 * (ocp_qp_xcond_solver.c:230, 379; make_int_multiple_of(8, &size) in its own modules) */
GPU_SEG_FN acados_size_t size8(size_t s) { return (acados_size_t) ((s + 7) & ~(size_t) 7); }

/* one piece of a bulk blob <-> one sub-block of a BLASFEO object of the QP */
enum { SEG_VEC = 0, SEG_MAT = 1, SEG_MAT_T = 2 };
enum { SRC_BAbt = 0, SRC_RSQrq, SRC_DCt, SRC_b, SRC_rqz, SRC_d, SRC_dmask, SRC_Z,  /* qp_in */
       SRC_ux, SRC_pi, SRC_lam, SRC_t,                                             /* qp_out */
       SRC_seed_g, SRC_seed_b, SRC_seed_d };                                       /* seed */
typedef struct
{
    int off, len;   /* position in the per-instance blob */
    int kind, src, k;
    int ai, aj;     /* first row (vector: first entry) / first column of the sub-block */
    int m, n;       /* rows, columns of the sub-block (SEG_MAT_T: the blob holds its transpose, n x m) */
    int neg;        /* stored negated in acados (upper bounds in d, ocp_qp_common.c:897-906) */
} gpu_seg;


/* the three segment tables of one device batch (input blob, output blob, seed blob) and the blob lengths */
#define GPU_LAYOUT_MEMBERS \
    ocp_qp_gpu_batch *batch; \
    gpu_seg *seg_in, *seg_out, *seg_seed, *seg_vec; \
    int n_in, n_out, n_seed, n_vec, seg_cap_in, seg_cap_out, seg_cap_seed, seg_cap_vec; \
    int L_in, L_out, L_seed, L_vec; /* doubles per instance of the blobs (L_vec: the vector part of the input blob, RTI feedback) */ \
    int ps;                  /* panel height of BLASFEO's matrix storage as PROBED (gpu_probe_panel_size); 0: every block through blasfeo_unpack_* */
typedef struct { GPU_LAYOUT_MEMBERS } gpu_layout;

/* ------------------------------------------------------------------ BLASFEO's storage, probed
 *
 * blasfeo_unpack_dmat / _tran_dmat are library calls per sub-block: ~20 of them per stage, 1,000 per C2-shaped QP, each a few dozen
 * doubles -- the call and its per-element index arithmetic, not the bytes, were the cost of reading n capsules' QPs (0.9 GB/s per host
 * thread, profiles/r05_orchestration_latency.txt).  The batch entries copy panel runs instead: element (i, j) of a panel-major matrix
 * lives at pA[(i - i % ps) * cn + j * ps + i % ps], so a column of a sub-block is <= ceil(m / ps) + 1 contiguous runs and a row is one
 * strided walk.  Nothing of BLASFEO's internals is ASSUMED: the panel height is found by packing a test matrix through the library's
 * own blasfeo_pack_dmat and looking where the values went, and a vector through blasfeo_pack_dvec; any other layout (column-major
 * builds, a panel height that is not found) returns 0 and every block keeps going through blasfeo_unpack_* (ACADOS_AMD_LA_API=1
 * forces that path: the byte-for-byte cross-check of tests/test_mock_acados.py).
 */
GPU_SEG_FN int gpu_probe_panel_size(void)
{
#if defined(MF_COLMAJ)
    return 0;
#else
    enum { PM = 37, PN = 3 };
    if (getenv("ACADOS_AMD_LA_API")) return 0;
    struct blasfeo_dmat sA;
    struct blasfeo_dvec sv;
    double a[PM * PN];
    for (int j = 0; j < PN; j++) for (int i = 0; i < PM; i++) a[i + PM * j] = 1.0 + i + 100.0 * j;
    blasfeo_allocate_dmat(PM, PN, &sA);
    blasfeo_allocate_dvec(PM, &sv);
    blasfeo_pack_dmat(PM, PN, a, PM, &sA, 0, 0);
    blasfeo_pack_dvec(PM, a, 1, &sv, 0);
    int found = 0;
    for (int ps = 2; ps <= 16 && !found; ps *= 2)
    {
        int ok = sA.pm % ps == 0 && sA.cn >= PN;
        for (int j = 0; j < PN && ok; j++)
            for (int i = 0; i < PM && ok; i++)
            {
                const long idx = (long) (i - i % ps) * sA.cn + (long) j * ps + i % ps;
                ok = idx < (long) sA.pm * sA.cn && sA.pA[idx] == a[i + PM * j];
            }
        if (ok) found = ps;
    }
    for (int i = 0; i < PM && found; i++) if (sv.pa[i] != a[i]) found = 0; /* vectors: one contiguous array */
    blasfeo_free_dmat(&sA);
    blasfeo_free_dvec(&sv);
    return found;
#endif
}

#if !defined(MF_COLMAJ)
/* sub-block (ai, aj), m x n -> column-major dst (ld m) */
GPU_SEG_FN void pm_unpack(int ps, int m, int n, const struct blasfeo_dmat *sA, int ai, int aj, double *dst)
{
    const int cn = sA->cn;
    for (int j = 0; j < n; j++)
    {
        int r = ai, left = m;
        double *d = dst + (size_t) m * j;
        while (left > 0)
        {
            const int in = r % ps, run = ps - in < left ? ps - in : left;
            const double *src = sA->pA + (size_t) (r - in) * cn + (size_t) (aj + j) * ps + in;
            for (int e = 0; e < run; e++) d[e] = src[e];
            d += run; r += run; left -= run;
        }
    }
}
/* ... -> its transpose, n x m column-major (ld n): dst[j + n * i] = A(ai + i, aj + j) */
GPU_SEG_FN void pm_unpack_tran(int ps, int m, int n, const struct blasfeo_dmat *sA, int ai, int aj, double *dst)
{
    const int cn = sA->cn;
    for (int i = 0; i < m; i++)
    {
        const int r = ai + i, in = r % ps;
        const double *src = sA->pA + (size_t) (r - in) * cn + (size_t) aj * ps + in;
        double *d = dst + (size_t) n * i;
        for (int j = 0; j < n; j++) d[j] = src[(size_t) j * ps];
    }
}
#endif

/* ------------------------------------------------------------------ sizes from dims */

GPU_SEG_FN int sig_len(const ocp_qp_dims *d)
{
    int len = 1;
    for (int k = 0; k <= d->N; k++) len += 7 + 2 * d->nb[k] + d->ng[k] + d->nbxe[k];
    return len;
}

GPU_SEG_FN int blob_in_cap(const ocp_qp_dims *d)
{
    int len = 0, getter = 0;
    for (int k = 0; k <= d->N; k++)
    {
        const int nx = d->nx[k], nu = d->nu[k], nx1 = k < d->N ? d->nx[k + 1] : 0;
        len += nx1 * (nx + nu + 1) + (nu + nx) * (nu + nx) + nu + nx + 5 * d->nb[k] + d->ng[k] * (nu + nx) + 4 * d->ng[k] + 8 * d->ns[k];
        if ((nu + nx) * (nu + nx + 1) > getter) getter = (nu + nx) * (nu + nx + 1); /* solver_get stages ric_L, ric_l here */
    }
    return len > getter ? len : getter;
}

GPU_SEG_FN int blob_out_cap(const ocp_qp_dims *d)
{
    int len = 0;
    for (int k = 0; k <= d->N; k++)
        len += d->nu[k] + d->nx[k] + 2 * d->ns[k] + (k < d->N ? d->nx[k + 1] : 0) + 4 * (d->nb[k] + d->ng[k] + d->ns[k]);
    return len;
}

#define SEGS_IN_PER_STAGE 34   /* A B b R S Q r q zl zu + 9 bound pieces + 8 masks + Zl Zu + C D (+ lbx#value) */
#define SEGS_OUT_PER_STAGE 7   /* u x sl su pi lam t */
#define SEGS_SEED_PER_STAGE 13 /* r q zl zu b lbu lbx lg ubu ubx ug lls lus */

/* ------------------------------------------------------------------ structure signature, segment tables */

GPU_SEG_FN int fill_sig(const ocp_qp_in *in, int *s)
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
        /* acados orders idxe [bue | bxe | ge] (ocp_nlp_constraints_bgh.c:637-655): the equality-flagged STATE bounds are what the device
         * eliminates (the ocp_qp interface of the path knows no others: acados_ocp_qp.py:374-379); an equality-flagged input bound
         * or general row stays the lb = ub pair it also is */
        memcpy(s + p, in->idxe[k] + d->nbue[k], sizeof(int) * d->nbxe[k]); p += d->nbxe[k];
    }
    return p;
}

GPU_SEG_FN void seg_add(ocp_qp_gpu_batch *b, gpu_seg *tab, int *cnt, int cap, int which, const char *field, int k, int expect,
                    int kind, int src, int ai, int aj, int m, int n, int neg)
{
    /* which: 0 input blob, 1 output blob, 2 seed blob, 3 vector part of the input blob */
    int len = 0;
    const int off = which == 2 ? ocp_qp_gpu_batch_sens_bulk_offset(b, 0, field, k, &len)
                               : ocp_qp_gpu_batch_bulk_offset(b, which == 3 ? 2 : which, field, k, &len);
    if (off < 0 || len == 0) return;
    if (len != expect)
    {
        printf("\nerror: ocp_qp_gpu_ipm: field %s at stage %d has %d entries in the device layout, %d in the acados struct\n", field, k, len, expect);
        exit(1);
    }
    if (*cnt >= cap) { printf("\nerror: ocp_qp_gpu_ipm: segment table too small\n"); exit(1); }
    gpu_seg *g = tab + (*cnt)++;
    g->off = off; g->len = len; g->kind = kind; g->src = src; g->k = k; g->ai = ai; g->aj = aj; g->m = m; g->n = n; g->neg = neg;
}

/* where every field of the three blobs lives in the acados structs: once per device batch */
GPU_SEG_FN int gpu_layout_build(gpu_layout *bk, const ocp_qp_dims *d)
{
    ocp_qp_gpu_batch *b = bk->batch;
    const int N = d->N;
    bk->n_in = bk->n_out = bk->n_seed = bk->n_vec = 0;
    bk->ps = gpu_probe_panel_size();
    /* the first device work after create (structure tables, out of HBM shows up here): negative = the device failed */
    bk->L_in = ocp_qp_gpu_batch_bulk_len(b, 0);
    bk->L_out = ocp_qp_gpu_batch_bulk_len(b, 1);
    bk->L_seed = ocp_qp_gpu_batch_sens_bulk_len(b, 0);
    bk->L_vec = bk->seg_vec ? ocp_qp_gpu_batch_bulk_len(b, 2) : 0;
    if (bk->L_in < 0 || bk->L_out < 0 || bk->L_seed < 0 || bk->L_vec < 0) { bk->L_in = bk->L_out = bk->L_seed = bk->L_vec = 0; return -1; }
    /* (a vector field of the input blob also goes into the table of the vector blob, where the owner of the layout carries one) */
#define IN(field, expect, kind, src, ai, aj, m, n, neg)                                                                                   \
    do {                                                                                                                                  \
        seg_add(b, bk->seg_in, &bk->n_in, bk->seg_cap_in, 0, field, k, expect, kind, src, ai, aj, m, n, neg);                             \
        if ((kind) == SEG_VEC && bk->seg_vec) seg_add(b, bk->seg_vec, &bk->n_vec, bk->seg_cap_vec, 3, field, k, expect, kind, src, ai, aj, m, n, neg); \
    } while (0)
#define OUT(field, expect, src, ai) seg_add(b, bk->seg_out, &bk->n_out, bk->seg_cap_out, 1, field, k, expect, SEG_VEC, src, ai, 0, expect, 1, 0)
#define SEED(field, expect, src, ai, neg) seg_add(b, bk->seg_seed, &bk->n_seed, bk->seg_cap_seed, 2, field, k, expect, SEG_VEC, src, ai, 0, expect, 1, neg)
    for (int k = 0; k <= N; k++)
    {
        const int nu = d->nu[k], nx = d->nx[k], nx1 = k < N ? d->nx[k + 1] : 0;
        const int nbu = d->nbu[k], nbx = d->nbx[k], nb = d->nb[k], ng = d->ng[k], ns = d->ns[k];
        if (k < N)
        {
            /* BAbt = [B'; A'; b'] (print.c:234-325): A (nx+ x nx) = (rows nu.. of BAbt)', B (nx+ x nu) = (rows 0..nu)' */
            IN("A", nx1 * nx, SEG_MAT_T, SRC_BAbt, nu, 0, nx, nx1, 0);
            IN("B", nx1 * nu, SEG_MAT_T, SRC_BAbt, 0, 0, nu, nx1, 0);
            IN("b", nx1, SEG_VEC, SRC_b, 0, 0, nx1, 1, 0); /* the VECTOR, not the last row */
        }
        /* RSQrq: lower triangle of [[R, S], [S', Q]] -- only the lower triangle is valid */
        IN("R", nu * nu, SEG_MAT, SRC_RSQrq, 0, 0, nu, nu, 0);
        IN("S", nu * nx, SEG_MAT_T, SRC_RSQrq, nu, 0, nx, nu, 0); /* S (nu x nx) = (S')' */
        IN("Q", nx * nx, SEG_MAT, SRC_RSQrq, nu, nu, nx, nx, 0);
        /* rqz = [r; q; zl; zu]: the vectors ocp_nlp writes every iteration */
        IN("r", nu, SEG_VEC, SRC_rqz, 0, 0, nu, 1, 0);
        IN("q", nx, SEG_VEC, SRC_rqz, nu, 0, nx, 1, 0);
        IN("zl", ns, SEG_VEC, SRC_rqz, nu + nx, 0, ns, 1, 0);
        IN("zu", ns, SEG_VEC, SRC_rqz, nu + nx + ns, 0, ns, 1, 0);
        /* d = [lb; lg; -ub; -ug; ls; us] with lb = [lbu; lbx] (ocp_qp_common.c:897-906): natural sign for the device */
        IN("lbu", nbu, SEG_VEC, SRC_d, 0, 0, nbu, 1, 0);
        IN("lbx", nbx, SEG_VEC, SRC_d, nbu, 0, nbx, 1, 0);
        IN("lbx#value", nbx, SEG_VEC, SRC_d, nbu, 0, nbx, 1, 0); /* equality-flagged: the value of x */
        IN("lg", ng, SEG_VEC, SRC_d, nb, 0, ng, 1, 0);
        IN("ubu", nbu, SEG_VEC, SRC_d, nb + ng, 0, nbu, 1, 1);
        IN("ubx", nbx, SEG_VEC, SRC_d, nb + ng + nbu, 0, nbx, 1, 1);
        IN("ug", ng, SEG_VEC, SRC_d, 2 * nb + ng, 0, ng, 1, 1);
        IN("lls", ns, SEG_VEC, SRC_d, 2 * nb + 2 * ng, 0, ns, 1, 0);
        IN("lus", ns, SEG_VEC, SRC_d, 2 * nb + 2 * ng + ns, 0, ns, 1, 0);
        /* d_mask: same positions, 1.0 / 0.0 (aliased to nlp_in->dmask, ocp_nlp_common.c:2894) */
        IN("lbu_mask", nbu, SEG_VEC, SRC_dmask, 0, 0, nbu, 1, 0);
        IN("lbx_mask", nbx, SEG_VEC, SRC_dmask, nbu, 0, nbx, 1, 0);
        IN("lg_mask", ng, SEG_VEC, SRC_dmask, nb, 0, ng, 1, 0);
        IN("ubu_mask", nbu, SEG_VEC, SRC_dmask, nb + ng, 0, nbu, 1, 0);
        IN("ubx_mask", nbx, SEG_VEC, SRC_dmask, nb + ng + nbu, 0, nbx, 1, 0);
        IN("ug_mask", ng, SEG_VEC, SRC_dmask, 2 * nb + ng, 0, ng, 1, 0);
        IN("lls_mask", ns, SEG_VEC, SRC_dmask, 2 * nb + 2 * ng, 0, ns, 1, 0);
        IN("lus_mask", ns, SEG_VEC, SRC_dmask, 2 * nb + 2 * ng + ns, 0, ns, 1, 0);
        /* Z = [Zl; Zu] */
        IN("Zl", ns, SEG_VEC, SRC_Z, 0, 0, ns, 1, 0);
        IN("Zu", ns, SEG_VEC, SRC_Z, ns, 0, ns, 1, 0);
        /* DCt = [D'; C'] ((nu+nx) x ng): C (ng x nx) = (rows nu.. )', D (ng x nu) = (rows 0..nu)' */
        IN("C", ng * nx, SEG_MAT_T, SRC_DCt, nu, 0, nx, ng, 0);
        IN("D", ng * nu, SEG_MAT_T, SRC_DCt, 0, 0, nu, ng, 0);

        /* solution: ux = [u; x; sl; su], lam / t ordered [lb lg ub ug ls us] as HPIPM's */
        const int nct = 2 * (nb + ng + ns);
        OUT("u", nu, SRC_ux, 0);
        OUT("x", nx, SRC_ux, nu);
        OUT("sl", ns, SRC_ux, nu + nx);
        OUT("su", ns, SRC_ux, nu + nx + ns);
        if (k < N) OUT("pi", nx1, SRC_pi, 0);
        OUT("lam", nct, SRC_lam, 0);
        OUT("t", nct, SRC_t, 0);

        /* seeds: seed_g = d[r; q; zl; zu], seed_b = d b, seed_d laid out like d -- upper part negated like d
         * (ocp_nlp_common.c:4078-4081 builds it that way for the nonlinear rows); the device takes natural signs */
        SEED("seed_r", nu, SRC_seed_g, 0, 0);
        SEED("seed_q", nx, SRC_seed_g, nu, 0);
        SEED("seed_zl", ns, SRC_seed_g, nu + nx, 0);
        SEED("seed_zu", ns, SRC_seed_g, nu + nx + ns, 0);
        if (k < N) SEED("seed_b", nx1, SRC_seed_b, 0, 0);
        SEED("seed_lbu", nbu, SRC_seed_d, 0, 0);
        SEED("seed_lbx", nbx, SRC_seed_d, nbu, 0);
        SEED("seed_lg", ng, SRC_seed_d, nb, 0);
        SEED("seed_ubu", nbu, SRC_seed_d, nb + ng, 1);
        SEED("seed_ubx", nbx, SRC_seed_d, nb + ng + nbu, 1);
        SEED("seed_ug", ng, SRC_seed_d, 2 * nb + ng, 1);
        SEED("seed_lls", ns, SRC_seed_d, 2 * nb + 2 * ng, 0);
        SEED("seed_lus", ns, SRC_seed_d, 2 * nb + 2 * ng + ns, 0);
    }
#undef IN
#undef OUT
#undef SEED
    return 0;
}

/* ------------------------------------------------------------------ zero-copy: the input blob as WORDS of the member arrays
 *
 * The batch entries let the device read the QP data from the capsules' own memory (ocp_qp_gpu_batch_gather_tables / _gather_run): one
 * table per structure class says, for every double of the blob, which member array it comes from (slot = member * (N + 1) + stage, the
 * members in the order of SRC_BAbt .. SRC_Z), at which offset of its storage -- the same index arithmetic as pm_unpack / pm_unpack_tran,
 * with the PROBED panel height and the matrices' own cn -- and whether it is stored negated.  Sorted by (slot, offset): the device walks
 * every array in storage order. */
#if !defined(MF_COLMAJ)
#define GPU_ZERO_COPY 1
typedef struct { int slot, off, pos; unsigned char neg; } gpu_word;
enum { GPU_WORD_MEMBERS = 8 };

GPU_SEG_FN int gpu_word_cmp(const void *a_, const void *b_)
{
    const gpu_word *a = (const gpu_word *) a_, *b = (const gpu_word *) b_;
    if (a->slot != b->slot) return a->slot < b->slot ? -1 : 1;
    if (a->off != b->off) return a->off < b->off ? -1 : 1;
    return a->pos < b->pos ? -1 : (a->pos > b->pos);
}

/* cn[member * (N + 1) + stage] of the three matrix members; returns the number of words, *out malloc'ed (NULL: out of memory) */
GPU_SEG_FN int gpu_words_build(const gpu_seg *tab, int cnt, int N, int ps, const int *cn, gpu_word **out)
{
    size_t total = 0;
    for (int s = 0; s < cnt; s++) total += (size_t) tab[s].len;
    gpu_word *w = (gpu_word *) malloc(sizeof(gpu_word) * (total ? total : 1));
    *out = w;
    if (!w) return 0;
    size_t q = 0;
    for (int s = 0; s < cnt; s++)
    {
        const gpu_seg *g = tab + s;
        const int slot = g->src * (N + 1) + g->k;
        if (g->kind == SEG_VEC)
        {
            for (int e = 0; e < g->len; e++, q++) { w[q].slot = slot; w[q].off = g->ai + e; w[q].pos = g->off + e; w[q].neg = (unsigned char) (g->neg != 0); }
            continue;
        }
        const int c = cn[slot];
        for (int j = 0; j < g->n; j++)
            for (int i = 0; i < g->m; i++, q++)
            {
                const int r = g->ai + i, in = r % ps;
                w[q].slot = slot;
                w[q].off = (r - in) * c + (g->aj + j) * ps + in;
                w[q].pos = g->off + (g->kind == SEG_MAT ? i + g->m * j : j + g->n * i);
                w[q].neg = 0;
            }
    }
    qsort(w, q, sizeof(gpu_word), gpu_word_cmp);
    return (int) q;
}

/* the storage of the eight members of one qp_in, in slot order (absent: NULL; vec_only: the matrix members are not looked at) */
GPU_SEG_FN void gpu_word_sources(const ocp_qp_in *in, int N, const void **p, int vec_only)
{
    for (int k = 0; k <= N; k++)
    {
        p[SRC_BAbt * (N + 1) + k] = k < N && !vec_only ? in->BAbt[k].pA : NULL;
        p[SRC_RSQrq * (N + 1) + k] = vec_only ? NULL : in->RSQrq[k].pA;
        p[SRC_DCt * (N + 1) + k] = vec_only ? NULL : in->DCt[k].pA;
        p[SRC_b * (N + 1) + k] = k < N ? in->b[k].pa : NULL;
        p[SRC_rqz * (N + 1) + k] = in->rqz[k].pa;
        p[SRC_d * (N + 1) + k] = in->d[k].pa;
        p[SRC_dmask * (N + 1) + k] = in->d_mask[k].pa;
        p[SRC_Z * (N + 1) + k] = in->Z[k].pa;
    }
}

GPU_SEG_FN void gpu_word_cn(const ocp_qp_in *in, int N, int *cn) /* 3 * (N + 1) */
{
    for (int k = 0; k <= N; k++)
    {
        cn[SRC_BAbt * (N + 1) + k] = k < N ? in->BAbt[k].cn : 0;
        cn[SRC_RSQrq * (N + 1) + k] = in->RSQrq[k].cn;
        cn[SRC_DCt * (N + 1) + k] = in->DCt[k].cn;
    }
}
#endif /* !MF_COLMAJ */

/* ------------------------------------------------------------------ blob <-> acados structs, one instance */

GPU_SEG_FN void unpack_segs(const gpu_seg *tab, int cnt, double *blob, struct blasfeo_dmat *const *mats, struct blasfeo_dvec *const *vecs, int ps)
{
#if !defined(MF_COLMAJ)
    if (ps > 0) /* probed panel-major storage: panel runs instead of a library call per sub-block */
    {
        for (int s = 0; s < cnt; s++)
        {
            const gpu_seg *g = tab + s;
            double *p = blob + g->off;
            if (g->kind == SEG_VEC)
            {
                const double *v = (vecs[g->src] + g->k)->pa + g->ai;
                if (g->neg) for (int e = 0; e < g->len; e++) p[e] = -v[e];
                else for (int e = 0; e < g->len; e++) p[e] = v[e];
            }
            else if (g->kind == SEG_MAT) pm_unpack(ps, g->m, g->n, mats[g->src] + g->k, g->ai, g->aj, p);
            else pm_unpack_tran(ps, g->m, g->n, mats[g->src] + g->k, g->ai, g->aj, p);
        }
        return;
    }
#endif
    for (int s = 0; s < cnt; s++)
    {
        const gpu_seg *g = tab + s;
        double *p = blob + g->off;
        if (g->kind == SEG_VEC)
        {
            blasfeo_unpack_dvec(g->m, vecs[g->src] + g->k, g->ai, p, 1);
            if (g->neg) for (int e = 0; e < g->len; e++) p[e] = -p[e];
        }
        else if (g->kind == SEG_MAT) blasfeo_unpack_dmat(g->m, g->n, mats[g->src] + g->k, g->ai, g->aj, p, g->m);
        else blasfeo_unpack_tran_dmat(g->m, g->n, mats[g->src] + g->k, g->ai, g->aj, p, g->n);
    }
}

/* every member array of qp_in, re-read on every call, unpacked from BLASFEO storage straight into the blob */
GPU_SEG_FN void unpack_qp_in(const gpu_layout *bk, ocp_qp_in *in, double *blob)
{
    struct blasfeo_dmat *mats[3] = {in->BAbt, in->RSQrq, in->DCt};
    struct blasfeo_dvec *vecs[8] = {NULL, NULL, NULL, in->b, in->rqz, in->d, in->d_mask, in->Z};
    unpack_segs(bk->seg_in, bk->n_in, blob, mats, vecs, bk->ps);
}

/* the vector members alone (b, rqz, d, d_mask) into the vector blob: the host side of an RTI feedback step */
GPU_SEG_FN void unpack_qp_vec(const gpu_layout *bk, ocp_qp_in *in, double *blob)
{
    struct blasfeo_dvec *vecs[8] = {NULL, NULL, NULL, in->b, in->rqz, in->d, in->d_mask, in->Z};
    unpack_segs(bk->seg_vec, bk->n_vec, blob, NULL, vecs, bk->ps);
}

GPU_SEG_FN void unpack_seed(const gpu_layout *bk, ocp_qp_seed *seed, double *blob)
{
    struct blasfeo_dvec *vecs[15] = {NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, seed->seed_g, seed->seed_b, seed->seed_d};
    unpack_segs(bk->seg_seed, bk->n_seed, blob, NULL, vecs, bk->ps);
}

/* hot start: pi, lam, t of qp_out; the primal part stays zero as ocp_qp_hpipm.c:325-336 leaves it before every solve */
GPU_SEG_FN void unpack_qp_out_duals(const gpu_layout *bk, ocp_qp_out *out, double *blob)
{
    struct blasfeo_dvec *vecs[12] = {NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, out->ux, out->pi, out->lam, out->t};
    for (int s = 0; s < bk->n_out; s++)
    {
        const gpu_seg *g = bk->seg_out + s;
        if (g->src == SRC_ux) memset(blob + g->off, 0, sizeof(double) * (size_t) g->len);
        else blasfeo_unpack_dvec(g->m, vecs[g->src] + g->k, g->ai, blob + g->off, 1);
    }
}

GPU_SEG_FN void pack_qp_out(const gpu_layout *bk, const double *blob, ocp_qp_out *out)
{
    struct blasfeo_dvec *vecs[12] = {NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, out->ux, out->pi, out->lam, out->t};
    for (int s = 0; s < bk->n_out; s++)
    {
        const gpu_seg *g = bk->seg_out + s;
        if (bk->ps > 0) memcpy((vecs[g->src] + g->k)->pa + g->ai, blob + g->off, sizeof(double) * (size_t) g->m); /* probed: one contiguous array */
        else blasfeo_pack_dvec(g->m, (double *) blob + g->off, 1, vecs[g->src] + g->k, g->ai);
    }
}


#endif /* ACADOS_OCP_QP_OCP_QP_GPU_SEGMENTS_H_ */
