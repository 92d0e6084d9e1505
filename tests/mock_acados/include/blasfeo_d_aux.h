/*
 * TEST INFRASTRUCTURE ONLY -- stand-in for the BLASFEO declarations acados' QP adapters use, restated from the call
 * sites in /root/reference (BLASFEO itself is an empty submodule there): struct blasfeo_dmat / blasfeo_dvec,
 * BLASFEO_DMATEL / BLASFEO_DVECEL and the pack / unpack routines as called in
 * acados/ocp_qp/ocp_qp_clarabel.c:299, 517, 556, 647, 1047-1069 and acados/ocp_qp/ocp_qp_common.c:874-921.
 * Storage is PANEL-MAJOR with panel size 4 (the default acados build, external/CMakeLists.txt:46): element (i, j) of a
 * matrix with `cn` padded columns lives at pA[(i / 4) * 4 * cn + j * 4 + i % 4].  An adapter that indexes the arrays as
 * column-major gets wrong numbers here, which is the point of building against this header.
 */
#ifndef MOCK_BLASFEO_D_AUX_H_
#define MOCK_BLASFEO_D_AUX_H_

#include <stdlib.h>

#define MOCK_PS 4

struct blasfeo_dmat
{
    double *mem;
    double *pA;
    double *dA;
    int m, n, pm, cn;
    int use_dA, memsize;
};

struct blasfeo_dvec
{
    double *mem;
    double *pa;
    int m, pm, memsize;
};

#define BLASFEO_DMATEL(sA, ai, aj) ((sA)->pA[((ai) - ((ai) & (MOCK_PS - 1))) * (sA)->cn + (aj) * MOCK_PS + ((ai) & (MOCK_PS - 1))])
#define BLASFEO_DVECEL(sa, ai) ((sa)->pa[ai])

static inline void blasfeo_allocate_dmat(int m, int n, struct blasfeo_dmat *sA)
{
    sA->m = m; sA->n = n;
    sA->pm = (m + MOCK_PS - 1) / MOCK_PS * MOCK_PS;
    sA->cn = (n + MOCK_PS - 1) / MOCK_PS * MOCK_PS;
    if (sA->cn == 0) sA->cn = MOCK_PS;
    sA->memsize = (int) sizeof(double) * (sA->pm * sA->cn + sA->cn + MOCK_PS);
    sA->mem = (double *) calloc(1, sA->memsize);
    sA->pA = sA->mem;
    sA->dA = sA->pA + sA->pm * sA->cn;
    sA->use_dA = 0;
}
static inline void blasfeo_free_dmat(struct blasfeo_dmat *sA) { free(sA->mem); }
static inline void blasfeo_allocate_dvec(int m, struct blasfeo_dvec *sa)
{
    sa->m = m;
    sa->pm = (m + MOCK_PS - 1) / MOCK_PS * MOCK_PS;
    sa->memsize = (int) sizeof(double) * (sa->pm + MOCK_PS);
    sa->mem = (double *) calloc(1, sa->memsize);
    sa->pa = sa->mem;
}
static inline void blasfeo_free_dvec(struct blasfeo_dvec *sa) { free(sa->mem); }

/* column-major A (lda) -> sub-block of sB starting at (bi, bj) */
static inline void blasfeo_pack_dmat(int m, int n, double *A, int lda, struct blasfeo_dmat *sB, int bi, int bj)
{
    for (int j = 0; j < n; j++) for (int i = 0; i < m; i++) BLASFEO_DMATEL(sB, bi + i, bj + j) = A[i + lda * j];
}
/* ... transposed: sB[bi + j, bj + i] = A[i, j] */
static inline void blasfeo_pack_tran_dmat(int m, int n, double *A, int lda, struct blasfeo_dmat *sB, int bi, int bj)
{
    for (int j = 0; j < n; j++) for (int i = 0; i < m; i++) BLASFEO_DMATEL(sB, bi + j, bj + i) = A[i + lda * j];
}
/* sub-block (ai, aj), m x n, of sA -> column-major B (ldb) */
static inline void blasfeo_unpack_dmat(int m, int n, struct blasfeo_dmat *sA, int ai, int aj, double *B, int ldb)
{
    for (int j = 0; j < n; j++) for (int i = 0; i < m; i++) B[i + ldb * j] = BLASFEO_DMATEL(sA, ai + i, aj + j);
}
/* ... transposed: B (n x m, ldb) = sub-block' */
static inline void blasfeo_unpack_tran_dmat(int m, int n, struct blasfeo_dmat *sA, int ai, int aj, double *B, int ldb)
{
    for (int j = 0; j < n; j++) for (int i = 0; i < m; i++) B[j + ldb * i] = BLASFEO_DMATEL(sA, ai + i, aj + j);
}
static inline void blasfeo_pack_dvec(int m, double *x, int xi, struct blasfeo_dvec *sy, int yi)
{
    for (int i = 0; i < m; i++) BLASFEO_DVECEL(sy, yi + i) = x[i * xi];
}
static inline void blasfeo_unpack_dvec(int m, struct blasfeo_dvec *sx, int xi, double *y, int incy)
{
    for (int i = 0; i < m; i++) y[i * incy] = BLASFEO_DVECEL(sx, xi + i);
}
static inline void blasfeo_dvecse(int m, double alpha, struct blasfeo_dvec *sx, int xi)
{
    for (int i = 0; i < m; i++) BLASFEO_DVECEL(sx, xi + i) = alpha;
}

#endif
