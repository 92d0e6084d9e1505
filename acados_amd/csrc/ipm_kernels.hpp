/*
 * ipm_kernels.hpp -- HIP kernels of the batched OCP-QP interior-point solver (gfx950).
 *
 * Mapping: ONE QP INSTANCE PER LANE.  A wavefront advances 64 independent instances;
 * every global access `a[e*Bp + i]` is a fully coalesced 512-byte wave access (see
 * gpu_ipm_internal.h).  Per-stage dense blocks ((NU+NX)^2 Hessian, (NU+NX) x NX
 * dynamics) live in VGPRs (loops are fully unrolled for small shapes) and the stage
 * recursion runs sequentially inside the lane; the batch supplies the parallelism.
 * The path is HBM-bound: per IPM iteration the stage data are streamed
 * four times (factor / forward / rhs-backward / forward).
 *
 * One IPM iteration = 4 launches over the whole batch:
 *   k_resfact   backward sweep: KKT residuals of the current iterate (+ norms,
 *               convergence decision per instance), Gamma/gamma condensation of the
 *               inequalities, square-root Riccati factorisation + rhs propagation
 *   k_forward<0> forward sweep of the affine direction, step length, mu_aff, sigma
 *   k_backrhs   backward rhs-only sweep for the Mehrotra corrector (factor reused)
 *   k_forward<1> forward sweep, step length, primal/dual update
 * What replaces what (reference, /root/reference): the whole loop is the body of
 * d_ocp_qp_ipm_solve as called from acados/ocp_qp/ocp_qp_hpipm.c:347 (HPIPM sources are
 * absent from the reference tree; the algorithm is restated in oracle/ocp_qp_oracle.c).
 */
#ifndef IPM_KERNELS_HPP_
#define IPM_KERNELS_HPP_

#include <hip/hip_runtime.h>
#include <math.h>

#include "gpu_ipm_internal.h"

#define PK(r, c) ((((r) * ((r) + 1)) >> 1) + (c))
/* uniform (scalar) base pointer + 32-bit per-lane offset: selects the saddr form of global_load /
 * global_store, so no per-element 64-bit address lives in VGPRs */
/* GAT defined below, after gat_ref */

#if defined(GQP_NO_UNROLL)
#define UNROLL _Pragma("unroll 1")
#else
#define UNROLL _Pragma("unroll")
#endif

namespace gqp
{

/* element e of this lane's instance in a wave-tiled array (gpu_ipm_internal.h, GArrT) */
#define GAT(arr, e) (arr).p[((size_t) (i >> 6) * (size_t) (arr).E + (size_t) (e)) * 64 + (size_t) (i & 63)]
/* layout-agnostic form for the kernels that are not on the hot path (pack/unpack, init, finalize,
 * condensing, compaction): wave-tiled or instance-major, decided by the array */
#define GATL(arr, e) (arr).p[(arr).aos ? (size_t) i * (size_t) (arr).E + (size_t) (e) \
                                       : ((size_t) (i >> 6) * (size_t) (arr).E + (size_t) (e)) * 64 + (size_t) (i & 63)]

__device__ static inline double dmax(double a, double b) { return a > b ? a : b; }
__device__ static inline double dabs(double a) { return a < 0.0 ? -a : a; }
__device__ static inline int popc64_g(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}
/* inf-norm accumulation that lets NaN through */
__device__ static inline void nacc(double &nrm, double v)
{
    double a = dabs(v);
    if (a > nrm || v != v) nrm = a;
}

/* ------------------------------------------------------------------ init */

/* Cold start: restates the initialisation the oracle pins (oracle/ocp_qp_oracle.c
 * init_var; reference contract ocp_qp_hpipm.c:333-336: primal iterate zeroed). */
template <int NX, int NU, int NG, int NS>
__global__ void __launch_bounds__(64) k_init(GqpDev D, GqpOpts O)
{
    constexpr int n = NX + NU;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.B) return;
    const double thr0 = 1e-1;
    /* t0_init 0 / 1: constant (t, lam), primal iterate and slacks stay at zero -- an infeasible start the IPM handles */
    const bool heur = O.t0_init != 0 && O.t0_init != 1;
    const double t_c = O.t0_init == 0 ? sqrt(O.mu0) : 1.0, l_c = O.t0_init == 0 ? sqrt(O.mu0) : O.mu0;
    for (int k = 0; k <= D.N; k++)
    {
        GQP_STAGE_REF S = D.st[k];
        const int nbg = S.nb + S.ng;
        const uint64_t am = GATL(D.amask, k);
        double v[n];
        UNROLL for (int j = 0; j < n; j++) v[j] = 0.0;
        /* fixed variables keep the value the caller put into ux (set at pack time) */
        UNROLL for (int j = 0; j < n; j++) if ((S.emask >> j) & 1) v[j] = GATL(D.ux, k * n + j);
        int ib = 0;
        UNROLL for (int j = 0; j < n; j++)
        {
            if (!((S.bmask >> j) & 1)) continue;
            const bool fixed = (S.emask >> j) & 1;
            const bool soft = S.srev[ib] >= 0;
            if (!fixed && !soft && heur)
            {
                const double lb = GATL(D.dvec, S.o_ct + ib), ub = GATL(D.dvec, S.o_ct + nbg + ib);
                const bool al = (am >> ib) & 1, au = (am >> (nbg + ib)) & 1;
                const double tl = v[j] - lb, tu = ub - v[j];
                if (al && au)
                {
                    if (tl < thr0) v[j] = (tu < thr0) ? 0.5 * (lb + ub) : lb + thr0;
                    else if (tu < thr0) v[j] = ub - thr0;
                }
                else if (al) { if (tl < thr0) v[j] = lb + thr0; }
                else if (au) { if (tu < thr0) v[j] = ub - thr0; }
            }
            ib++;
        }
        UNROLL for (int j = 0; j < n; j++) GATL(D.ux, k * n + j) = v[j];
        if (S.has_dyn) { UNROLL for (int c = 0; c < NX; c++) GATL(D.pi, (k + 1) * NX + c) = 0.0; }
        /* slacks */
        double sl[NS > 0 ? NS : 1], su[NS > 0 ? NS : 1];
        if (NS > 0)
        {
            UNROLL for (int j = 0; j < NS; j++)
            {
                sl[j] = 0.0; su[j] = 0.0;
                if (j < S.ns && heur)
                {
                    if ((am >> (2 * nbg + j)) & 1) sl[j] = GATL(D.dvec, S.o_ct + 2 * nbg + j) + thr0;
                    if ((am >> (2 * nbg + S.ns + j)) & 1) su[j] = GATL(D.dvec, S.o_ct + 2 * nbg + S.ns + j) + thr0;
                }
            }
        }
        /* row values c_i */
        double cval[n + NG];
        ib = 0;
        UNROLL for (int j = 0; j < n; j++) if ((S.bmask >> j) & 1) { cval[j] = v[j]; }
        UNROLL for (int g = 0; g < NG; g++)
        {
            cval[n + g] = 0.0;
            if (g < S.ng)
            {
                double a = 0.0;
                UNROLL for (int r = 0; r < n; r++) a += GATL(D.DCt, (S.o_g + g) * n + r) * v[r];
                cval[n + g] = a;
            }
        }
        if (NS > 0)
        {
            ib = 0;
            UNROLL for (int j = 0; j < n + NG; j++)
            {
                const bool is_row = j < n ? ((S.bmask >> j) & 1) : (j - n < S.ng);
                if (!is_row) continue;
                const int row = j < n ? ib : S.nb + (j - n);
                if (j < n) ib++;
                const int sj = S.srev[row];
                if (sj < 0 || !heur) continue;
                const double lo = GATL(D.dvec, S.o_ct + row), up = GATL(D.dvec, S.o_ct + nbg + row);
                const double need_l = lo - cval[j] + thr0, need_u = cval[j] - up + thr0;
                UNROLL for (int q = 0; q < NS; q++)
                    if (q == sj)
                    {
                        if (((am >> row) & 1) && need_l > sl[q]) sl[q] = need_l;
                        if (((am >> (nbg + row)) & 1) && need_u > su[q]) su[q] = need_u;
                    }
            }
            UNROLL for (int j = 0; j < NS; j++)
                if (j < S.ns)
                {
                    GATL(D.sv, S.o_s + j) = sl[j];
                    GATL(D.sv, S.o_s + S.ns + j) = su[j];
                }
        }
        /* t, lam */
        ib = 0;
        UNROLL for (int j = 0; j < n + NG; j++)
        {
            const bool is_row = j < n ? ((S.bmask >> j) & 1) : (j - n < S.ng);
            if (!is_row) continue;
            const int row = j < n ? ib : S.nb + (j - n);
            if (j < n) ib++;
            const double lo = GATL(D.dvec, S.o_ct + row), up = GATL(D.dvec, S.o_ct + nbg + row);
            double ssl = 0.0, ssu = 0.0;
            if (NS > 0)
            {
                const int sj = S.srev[row];
                UNROLL for (int q = 0; q < NS; q++) if (q == sj) { ssl = sl[q]; ssu = su[q]; }
            }
            double tl = cval[j] + ssl - lo, tu = up - cval[j] + ssu;
            if (tl < thr0) tl = thr0;
            if (tu < thr0) tu = thr0;
            if (!heur) { tl = t_c; tu = t_c; }
            const bool al = (am >> row) & 1, au = (am >> (nbg + row)) & 1;
            GATL(D.t, S.o_ct + row) = al ? tl : 0.0;
            GATL(D.t, S.o_ct + nbg + row) = au ? tu : 0.0;
            GATL(D.lam, S.o_ct + row) = al ? (heur ? O.mu0 / tl : l_c) : 0.0;
            GATL(D.lam, S.o_ct + nbg + row) = au ? (heur ? O.mu0 / tu : l_c) : 0.0;
        }
        if (NS > 0)
        {
            UNROLL for (int j = 0; j < NS; j++)
                if (j < S.ns)
                {
                    const int e0 = S.o_ct + 2 * nbg + j, e1 = e0 + S.ns;
                    double tl = sl[j] - GATL(D.dvec, e0), tu = su[j] - GATL(D.dvec, e1);
                    if (tl < thr0) tl = thr0;
                    if (tu < thr0) tu = thr0;
                    if (!heur) { tl = t_c; tu = t_c; }
                    const bool al = (am >> (2 * nbg + j)) & 1, au = (am >> (2 * nbg + S.ns + j)) & 1;
                    GATL(D.t, e0) = al ? tl : 0.0;
                    GATL(D.t, e1) = au ? tu : 0.0;
                    GATL(D.lam, e0) = al ? (heur ? O.mu0 / tl : l_c) : 0.0;
                    GATL(D.lam, e1) = au ? (heur ? O.mu0 / tu : l_c) : 0.0;
                }
        }
    }
    D.iter[i] = 0;
    D.status[i] = GQP_RUNNING;
    D.alpha[i] = 1.0;
}

/* ------------------------------------------------- shared row machinery */

/* Everything the sweeps need to know about one inequality row pair (lower, upper). */
struct RowQ
{
    double gl, gu;  /* Gamma = lam/t   (0 if side inactive) */
    double rl, ru;  /* rho   = (rm_eff + lam*rd)/t          */
};

/* ---------------------------------------------------------- k_resfact */

/*
 * Backward sweep k = N..0 of one IPM iteration.
 * FACT = true : full sweep (residuals of the current iterate, norms, convergence test,
 *               condensation with the affine complementarity rhs, Cholesky + rhs).
 * FACT = false: rhs-only sweep for the corrector: rm_eff = rm + dlam*dt - sigma*mu,
 *               factor L reused from HBM, only l = L^{-1} m is rewritten.
 */
template <int NX, int NU, int NG, int NS, bool FACT>
__global__ void __launch_bounds__(64) k_backward(GqpDev D, GqpOpts O, int redo)
{
    constexpr int n = NX + NU;
    constexpr int NP = n * (n + 1) / 2;
    constexpr int NSS = NS > 0 ? NS : 1;
    const int Bp = D.Bp;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.B) return;
    if (D.status[i] != GQP_RUNNING) return;

    /* redo pass (cond_pred_corr fallback): only instances whose corrector collapsed
     * (flagged by a negative alpha) re-solve with the centering-only rhs */
    if (redo && !(D.alpha[i] < 0.0)) return;
    const double smu = FACT ? 0.0 : D.smu[i];
    const bool center_only = !FACT && redo;

    double Lx[NX * (NX + 1) / 2]; /* x-block of the factor of stage k+1 */
    double lx[NX];
    double nrm_g = 0.0, nrm_b = 0.0, nrm_d = 0.0, nrm_m = 0.0, musum = 0.0, obj = 0.0;
    int nact = 0;

    for (int k = D.N; k >= 0; k--)
    {
        GQP_STAGE_REF S = D.st[k];
        const int nbg = S.nb + S.ng;
        const uint64_t am = GAT(D.amask, k);
        double M[NP];  /* FACT: Hessian then factor ; !FACT: factor loaded from HBM */
        double gt[n];  /* condensed gradient -> m -> l */
        double v[n];
        double rb[NX];

        if (FACT)
        {
            UNROLL for (int e = 0; e < NP; e++) M[e] = GAT(D.RSQ, k * NP + e);
            UNROLL for (int j = 0; j < n; j++) v[j] = GAT(D.ux, k * n + j);
            /* rg = H v + g */
            double hv[n];
            UNROLL for (int r = 0; r < n; r++) hv[r] = 0.0;
            UNROLL for (int r = 0; r < n; r++)
            {
                UNROLL for (int c = 0; c < r; c++)
                {
                    hv[r] += M[PK(r, c)] * v[c];
                    hv[c] += M[PK(r, c)] * v[r];
                }
                hv[r] += M[PK(r, r)] * v[r];
            }
            UNROLL for (int r = 0; r < n; r++)
            {
                const double g = GAT(D.rq, k * n + r);
                obj += (0.5 * hv[r] + g) * v[r];
                gt[r] = hv[r] + g;
            }
            if (k > 0) { UNROLL for (int c = 0; c < NX; c++) gt[NU + c] -= GAT(D.pi, k * NX + c); }
            UNROLL for (int r = 0; r < n; r++) M[PK(r, r)] += O.reg_prim;
        }
        else
        {
            UNROLL for (int e = 0; e < NP; e++) M[e] = GAT(D.Lf, k * NP + e);
            UNROLL for (int j = 0; j < n; j++) gt[j] = GAT(D.rg, k * n + j);
        }

        /* ---- dynamics part: rg += BAt pi+, rb, W = BAt Lx+, M += W W', m += W w0 ---- */
        double y[NX]; /* P+ rb + p+ = Lx+ (Lx+' rb + lx+) */
        if (S.has_dyn)
        {
            if (FACT)
            {
                double pin[NX];
                UNROLL for (int c = 0; c < NX; c++)
                {
                    pin[c] = GAT(D.pi, (k + 1) * NX + c);
                    rb[c] = GAT(D.bvec, k * NX + c) - GAT(D.ux, (k + 1) * n + NU + c);
                }
                double W[n * NX];
                UNROLL for (int r = 0; r < n; r++)
                {
                    double row[NX];
                    UNROLL for (int c = 0; c < NX; c++) row[c] = GAT(D.BAt, (k * n + r) * NX + c);
                    double a = 0.0;
                    UNROLL for (int c = 0; c < NX; c++)
                    {
                        a += row[c] * pin[c];
                        rb[c] += row[c] * v[r];
                    }
                    gt[r] += a;
                    UNROLL for (int c = 0; c < NX; c++)
                    {
                        double w = 0.0;
                        UNROLL for (int q = c; q < NX; q++) w += row[q] * Lx[PK(q, c)];
                        W[r * NX + c] = w;
                    }
                }
                UNROLL for (int r = 0; r < n; r++)
                    UNROLL for (int c = 0; c <= r; c++)
                    {
                        double a = 0.0;
                        UNROLL for (int q = 0; q < NX; q++) a += W[r * NX + q] * W[c * NX + q];
                        M[PK(r, c)] += a;
                    }
                UNROLL for (int c = 0; c < NX; c++)
                {
                    GAT(D.rb, k * NX + c) = rb[c];
                    nacc(nrm_b, rb[c]);
                }
            }
            else
            {
                UNROLL for (int c = 0; c < NX; c++) rb[c] = GAT(D.rb, k * NX + c);
            }
            double w0[NX];
            UNROLL for (int c = 0; c < NX; c++)
            {
                double a = lx[c];
                UNROLL for (int q = c; q < NX; q++) a += Lx[PK(q, c)] * rb[q];
                w0[c] = a;
            }
            UNROLL for (int r = 0; r < NX; r++)
            {
                double a = 0.0;
                UNROLL for (int c = 0; c <= r; c++) a += Lx[PK(r, c)] * w0[c];
                y[r] = a;
            }
        }

        /* ---- inequality rows ---- */
        double sDl[NSS], sDu[NSS], sRl[NSS], sRu[NSS], sPl[NSS], sPu[NSS];
        /* cancellation-free elimination of the slack block (see pass 2): what D and r~ are WITHOUT the rows */
        double sE0l[NSS], sE0u[NSS], sP0l[NSS], sP0u[NSS];
        int scnt[NSS];
        if (NS > 0)
        {
            UNROLL for (int q = 0; q < NS; q++)
            {
                sDl[q] = sDu[q] = sRl[q] = sRu[q] = sPl[q] = sPu[q] = 0.0;
                sE0l[q] = sE0u[q] = sP0l[q] = sP0u[q] = 0.0;
                scnt[q] = 0;
                if (q < S.ns)
                {
                    const int e0 = S.o_ct + 2 * nbg + q, e1 = e0 + S.ns;
                    const double Zl = GAT(D.Zz, (S.o_s + q) * 2), zl = GAT(D.Zz, (S.o_s + q) * 2 + 1);
                    const double Zu = GAT(D.Zz, (S.o_s + S.ns + q) * 2), zu = GAT(D.Zz, (S.o_s + S.ns + q) * 2 + 1);
                    const bool al = (am >> (2 * nbg + q)) & 1, au = (am >> (2 * nbg + S.ns + q)) & 1;
                    const double laml = al ? GAT(D.lam, e0) : 0.0, lamu = au ? GAT(D.lam, e1) : 0.0;
                    const double tl = GAT(D.t, e0), tu = GAT(D.t, e1);
                    double rdl, rdu, rml, rmu;
                    if (FACT)
                    {
                        const double sl = GAT(D.sv, S.o_s + q), su = GAT(D.sv, S.o_s + S.ns + q);
                        obj += (0.5 * Zl * sl + zl) * sl + (0.5 * Zu * su + zu) * su;
                        rdl = al ? sl - GAT(D.dvec, e0) - tl : 0.0;
                        rdu = au ? su - GAT(D.dvec, e1) - tu : 0.0;
                        rml = al ? laml * tl - O.tau_min : 0.0;
                        rmu = au ? lamu * tu - O.tau_min : 0.0;
                        GAT(D.rd, e0) = rdl; GAT(D.rd, e1) = rdu;
                        GAT(D.rm, e0) = rml; GAT(D.rm, e1) = rmu;
                        nacc(nrm_d, rdl); nacc(nrm_d, rdu); nacc(nrm_m, rml); nacc(nrm_m, rmu);
                        musum += laml * tl + lamu * tu;
                        nact += (int) al + (int) au;
                        /* slack stationarity residual (row part added below) */
                        sRl[q] = Zl * sl + zl - laml;
                        sRu[q] = Zu * su + zu - lamu;
                    }
                    else
                    {
                        rdl = GAT(D.rd, e0); rdu = GAT(D.rd, e1);
                        rml = GAT(D.rm, e0); rmu = GAT(D.rm, e1);
                        if (al) rml += (center_only ? 0.0 : GAT(D.dlam, e0) * GAT(D.dt, e0)) - smu;
                        if (au) rmu += (center_only ? 0.0 : GAT(D.dlam, e1) * GAT(D.dt, e1)) - smu;
                        sRl[q] = GAT(D.rgs, S.o_s + q);
                        sRu[q] = GAT(D.rgs, S.o_s + S.ns + q);
                    }
                    sDl[q] = Zl + (al ? laml / tl : 0.0);
                    sDu[q] = Zu + (au ? lamu / tu : 0.0);
                    /* rho sums are kept apart (sP) until the stationarity residual is stored */
                    sPl[q] = al ? (rml + laml * rdl) / tl : 0.0;
                    sPu[q] = au ? (rmu + lamu * rdu) / tu : 0.0;
                    sE0l[q] = sDl[q]; sE0u[q] = sDu[q];
                    sP0l[q] = sPl[q]; sP0u[q] = sPu[q];
                }
            }
        }

        /* pass 1 over the rows */
        double gadd[n]; /* hard-row contributions to the condensed gradient */
        UNROLL for (int j = 0; j < n; j++) gadd[j] = 0.0;
        {
            int ib = 0;
            UNROLL for (int j = 0; j < n + NG; j++)
            {
                const bool is_row = j < n ? ((S.bmask >> j) & 1) : (j - n < S.ng);
                if (!is_row) continue;
                const int row = j < n ? ib : S.nb + (j - n);
                if (j < n) ib++;
                const int el = S.o_ct + row, eu = el + nbg;
                const bool al = (am >> row) & 1, au = (am >> (nbg + row)) & 1;
                double arow[n];
                if (j >= n) { UNROLL for (int r = 0; r < n; r++) arow[r] = GAT(D.DCt, (S.o_g + (j - n)) * n + r); }
                const int sj = NS > 0 ? S.srev[row] : -1;
                const double laml = al ? GAT(D.lam, el) : 0.0, lamu = au ? GAT(D.lam, eu) : 0.0;
                const double tl = al ? GAT(D.t, el) : 1.0, tu = au ? GAT(D.t, eu) : 1.0;
                double rdl, rdu, rml, rmu;
                if (FACT)
                {
                    double c;
                    if (j < n) c = v[j < n ? j : 0];
                    else { c = 0.0; UNROLL for (int r = 0; r < n; r++) c += arow[r] * v[r]; }
                    double ssl = 0.0, ssu = 0.0;
                    if (NS > 0 && sj >= 0) { ssl = GAT(D.sv, S.o_s + sj); ssu = GAT(D.sv, S.o_s + S.ns + sj); }
                    rdl = al ? c + ssl - GAT(D.dvec, el) - tl : 0.0;
                    rdu = au ? GAT(D.dvec, eu) - c + ssu - tu : 0.0;
                    rml = al ? laml * tl - O.tau_min : 0.0;
                    rmu = au ? lamu * tu - O.tau_min : 0.0;
                    GAT(D.rd, el) = rdl; GAT(D.rd, eu) = rdu;
                    GAT(D.rm, el) = rml; GAT(D.rm, eu) = rmu;
                    nacc(nrm_d, rdl); nacc(nrm_d, rdu); nacc(nrm_m, rml); nacc(nrm_m, rmu);
                    musum += laml * tl + lamu * tu;
                    nact += (int) al + (int) au;
                    /* stationarity: rg -= a (lam_l - lam_u) */
                    const double nu_ = laml - lamu;
                    if (j < n) gt[j < n ? j : 0] -= nu_;
                    else { UNROLL for (int r = 0; r < n; r++) gt[r] -= arow[r] * nu_; }
                    if (NS > 0 && sj >= 0)
                    {
                        UNROLL for (int q = 0; q < NS; q++) if (q == sj) { sRl[q] -= laml; sRu[q] -= lamu; }
                    }
                }
                else
                {
                    rdl = GAT(D.rd, el); rdu = GAT(D.rd, eu);
                    rml = GAT(D.rm, el); rmu = GAT(D.rm, eu);
                    if (al) rml += (center_only ? 0.0 : GAT(D.dlam, el) * GAT(D.dt, el)) - smu;
                    if (au) rmu += (center_only ? 0.0 : GAT(D.dlam, eu) * GAT(D.dt, eu)) - smu;
                }
                const double gl = al ? laml / tl : 0.0, gu = au ? lamu / tu : 0.0;
                const double rl = al ? (rml + laml * rdl) / tl : 0.0, ru = au ? (rmu + lamu * rdu) / tu : 0.0;
                if (NS > 0 && sj >= 0)
                {
                    UNROLL for (int q = 0; q < NS; q++)
                        if (q == sj) { sDl[q] += gl; sDu[q] += gu; sPl[q] += rl; sPu[q] += ru; scnt[q]++; }
                }
                else
                {
                    const double nu_ = rl - ru, gm = gl + gu;
                    if (j < n)
                    {
                        gadd[j < n ? j : 0] += nu_;
                        if (FACT) M[PK((j < n ? j : 0), (j < n ? j : 0))] += gm;
                    }
                    else
                    {
                        UNROLL for (int r = 0; r < n; r++) gadd[r] += arow[r] * nu_;
                        if (FACT)
                        {
                            UNROLL for (int r = 0; r < n; r++)
                                UNROLL for (int c = 0; c <= r; c++) M[PK(r, c)] += gm * arow[r] * arow[c];
                        }
                    }
                }
            }
        }

        if (FACT)
        {
            /* multipliers of equality-flagged bounds from the raw stationarity value */
            int ib = 0;
            UNROLL for (int j = 0; j < n; j++)
            {
                if (!((S.bmask >> j) & 1)) continue;
                if ((S.emask >> j) & 1)
                {
                    const double a = gt[j];
                    GAT(D.lam, S.o_ct + ib) = a > 0.0 ? a : 0.0;
                    GAT(D.lam, S.o_ct + nbg + ib) = a < 0.0 ? -a : 0.0;
                    GAT(D.t, S.o_ct + ib) = 0.0;
                    GAT(D.t, S.o_ct + nbg + ib) = 0.0;
                }
                ib++;
            }
            UNROLL for (int j = 0; j < n; j++)
            {
                if ((S.emask >> j) & 1) gt[j] = 0.0;
                nacc(nrm_g, gt[j]);
                GAT(D.rg, k * n + j) = gt[j];
            }
            if (NS > 0)
            {
                UNROLL for (int q = 0; q < NS; q++)
                    if (q < S.ns)
                    {
                        nacc(nrm_g, sRl[q]); nacc(nrm_g, sRu[q]);
                        GAT(D.rgs, S.o_s + q) = sRl[q];
                        GAT(D.rgs, S.o_s + S.ns + q) = sRu[q];
                    }
            }
        }

        /* pass 2: soft rows (need the complete per-slack sums).  Schur complement of the slack block in its
         * CANCELLATION-FREE form: for row i coupled to slack q,  Gamma_i - Gamma_i^2/D = Gamma_i E_i/D  with
         * E_i = D - Gamma_i summed WITHOUT row i (an active soft row has Gamma_i >> Z: the difference would
         * lose every digit of the Z + Gamma_s it leaves behind), and
         * rho_i - Gamma_i r~/D = (rho_i E_i - Gamma_i X_i)/D with X_i = r~ - rho_i summed without row i.
         * Rows that share a slack add the cross terms -Gamma_i Gamma_k/D a_i a_k' (pass 3). */
        if (NS > 0)
        {
            double sX0l[NSS], sX0u[NSS];
            UNROLL for (int q = 0; q < NS; q++)
            {
                sX0l[q] = sX0u[q] = 0.0;
                if (q < S.ns)
                {
                    sX0l[q] = sRl[q] + sP0l[q]; sX0u[q] = sRu[q] + sP0u[q]; /* r~ without the rows */
                    sRl[q] += sPl[q]; sRu[q] += sPu[q]; /* r~ = stationarity residual + rho sums */
                    /* persist (D, r~) of each slack for the forward sweep */
                    GAT(D.sD, S.o_s + q) = sDl[q]; GAT(D.sD, S.o_s + S.ns + q) = sDu[q];
                    GAT(D.sR, S.o_s + q) = sRl[q]; GAT(D.sR, S.o_s + S.ns + q) = sRu[q];
                }
            }
            /* Gamma and rho of a row, recomputed from HBM (rows are revisited; registers do not hold them) */
            auto row_gr = [&](int row, double &gl, double &gu, double &rl, double &ru) {
                const int el = S.o_ct + row, eu = el + nbg;
                const bool al = (am >> row) & 1, au = (am >> (nbg + row)) & 1;
                gl = al ? GAT(D.lam, el) / GAT(D.t, el) : 0.0;
                gu = au ? GAT(D.lam, eu) / GAT(D.t, eu) : 0.0;
                double rml = GAT(D.rm, el), rmu = GAT(D.rm, eu);
                if (!FACT)
                {
                    if (al) rml += (center_only ? 0.0 : GAT(D.dlam, el) * GAT(D.dt, el)) - smu;
                    if (au) rmu += (center_only ? 0.0 : GAT(D.dlam, eu) * GAT(D.dt, eu)) - smu;
                }
                rl = al ? (rml + GAT(D.lam, el) * GAT(D.rd, el)) / GAT(D.t, el) : 0.0;
                ru = au ? (rmu + GAT(D.lam, eu) * GAT(D.rd, eu)) / GAT(D.t, eu) : 0.0;
            };
            int ib = 0;
            UNROLL for (int j = 0; j < n + NG; j++)
            {
                const bool is_row = j < n ? ((S.bmask >> j) & 1) : (j - n < S.ng);
                if (!is_row) continue;
                const int row = j < n ? ib : S.nb + (j - n);
                if (j < n) ib++;
                const int sj = S.srev[row];
                if (sj < 0) continue;
                double gl, gu, rl, ru;
                row_gr(row, gl, gu, rl, ru);
                double El = 0.0, Eu = 0.0, Xl = 0.0, Xu = 0.0, il = 0.0, iu = 0.0;
                int cnt = 0;
                UNROLL for (int q = 0; q < NS; q++)
                    if (q == sj)
                    {
                        El = sE0l[q]; Eu = sE0u[q]; Xl = sX0l[q]; Xu = sX0u[q]; cnt = scnt[q];
                        il = sDl[q] != 0.0 ? 1.0 / sDl[q] : 0.0; iu = sDu[q] != 0.0 ? 1.0 / sDu[q] : 0.0;
                    }
                if (cnt > 1)
                {
                    for (int k2 = 0; k2 < nbg; k2++)
                        if (k2 != row && S.srev[k2] == sj)
                        {
                            double g2l, g2u, r2l, r2u;
                            row_gr(k2, g2l, g2u, r2l, r2u);
                            El += g2l; Eu += g2u; Xl += r2l; Xu += r2u;
                        }
                }
                const double nu_ = (rl * El - gl * Xl) * il - (ru * Eu - gu * Xu) * iu;
                const double gm = gl * El * il + gu * Eu * iu;
                if (j < n)
                {
                    gadd[j < n ? j : 0] += nu_;
                    if (FACT) M[PK((j < n ? j : 0), (j < n ? j : 0))] += gm;
                }
                else
                {
                    double arow[n];
                    UNROLL for (int r = 0; r < n; r++) arow[r] = GAT(D.DCt, (S.o_g + (j - n)) * n + r);
                    UNROLL for (int r = 0; r < n; r++) gadd[r] += arow[r] * nu_;
                    if (FACT)
                    {
                        UNROLL for (int r = 0; r < n; r++)
                            UNROLL for (int c = 0; c <= r; c++) M[PK(r, c)] += gm * arow[r] * arow[c];
                    }
                }
            }
            /* pass 3: rows sharing one slack, cross terms (rare: plain loops, row vectors rebuilt on the fly) */
            if (FACT)
            {
                for (int r1 = 0; r1 < nbg; r1++)
                {
                    const int q = S.srev[r1];
                    if (q < 0) continue;
                    int cnt = 0;
                    double il = 0.0, iu = 0.0;
                    UNROLL for (int qq = 0; qq < NS; qq++)
                        if (qq == q)
                        {
                            cnt = scnt[qq];
                            il = sDl[qq] != 0.0 ? 1.0 / sDl[qq] : 0.0; iu = sDu[qq] != 0.0 ? 1.0 / sDu[qq] : 0.0;
                        }
                    if (cnt < 2) continue;
                    double g1l, g1u, r1l, r1u;
                    row_gr(r1, g1l, g1u, r1l, r1u);
                    for (int r2 = 0; r2 < nbg; r2++)
                    {
                        if (r2 == r1 || S.srev[r2] != q) continue;
                        double g2l, g2u, r2l, r2u;
                        row_gr(r2, g2l, g2u, r2l, r2u);
                        const double cf = g1l * g2l * il + g1u * g2u * iu;
                        /* a_i: unit vector of the variable of a box row, row of DCt for a general row */
                        UNROLL for (int r = 0; r < n; r++)
                        {
                            const double a1 = r1 >= S.nb ? GAT(D.DCt, (S.o_g + r1 - S.nb) * n + r)
                                                         : ((((S.bmask >> r) & 1) && popc64_g(S.bmask & (((uint64_t) 1 << r) - 1)) == r1) ? 1.0 : 0.0);
                            if (a1 == 0.0) continue;
                            UNROLL for (int c = 0; c <= r; c++)
                            {
                                const double a2 = r2 >= S.nb ? GAT(D.DCt, (S.o_g + r2 - S.nb) * n + c)
                                                             : ((((S.bmask >> c) & 1) && popc64_g(S.bmask & (((uint64_t) 1 << c) - 1)) == r2) ? 1.0 : 0.0);
                                M[PK(r, c)] -= cf * a1 * a2;
                            }
                        }
                    }
                }
            }
        }

        /* ---- m = gt + gadd + BAt y ; fixed variables ; Cholesky ; l = L^{-1} m ---- */
        UNROLL for (int j = 0; j < n; j++) gt[j] += gadd[j];
        if (S.has_dyn)
        {
            UNROLL for (int r = 0; r < n; r++)
            {
                double a = 0.0;
                UNROLL for (int c = 0; c < NX; c++) a += GAT(D.BAt, (k * n + r) * NX + c) * y[c];
                gt[r] += a;
            }
        }
        UNROLL for (int j = 0; j < n; j++) if ((S.emask >> j) & 1) gt[j] = 0.0;
        if (FACT)
        {
            if (S.emask)
            {
                UNROLL for (int r = 0; r < n; r++)
                    UNROLL for (int c = 0; c <= r; c++)
                        if (((S.emask >> r) & 1) || ((S.emask >> c) & 1)) M[PK(r, c)] = (r == c) ? 1.0 : 0.0;
            }
            /* in-place Cholesky, right-looking; non-positive pivots zero the column
             * (BLASFEO reference dpotrf behaviour, as restated in the oracle) */
            UNROLL for (int jc = 0; jc < n; jc++)
            {
                double d = M[PK(jc, jc)];
                double inv;
                if (d > 0.0) { d = sqrt(d); inv = 1.0 / d; } else { d = 0.0; inv = 0.0; }
                M[PK(jc, jc)] = d;
                UNROLL for (int r = jc + 1; r < n; r++) M[PK(r, jc)] *= inv;
                UNROLL for (int c = jc + 1; c < n; c++)
                    UNROLL for (int r = c; r < n; r++) M[PK(r, c)] -= M[PK(r, jc)] * M[PK(c, jc)];
            }
            UNROLL for (int e = 0; e < NP; e++) GAT(D.Lf, k * NP + e) = M[e];
        }
        UNROLL for (int r = 0; r < n; r++)
        {
            double a = gt[r];
            UNROLL for (int c = 0; c < r; c++) a -= M[PK(r, c)] * gt[c];
            const double d = M[PK(r, r)];
            gt[r] = d != 0.0 ? a / d : 0.0;
            GAT(D.lf, k * n + r) = gt[r];
        }
        UNROLL for (int r = 0; r < NX; r++)
        {
            lx[r] = gt[NU + r];
            UNROLL for (int c = 0; c <= r; c++) Lx[PK(r, c)] = M[PK(NU + r, NU + c)];
        }
    }

    if (FACT)
    {
        const double mu = nact > 0 ? musum / nact : 0.0;
        D.mu[i] = mu;
        D.obj[i] = obj;
        D.res[0 * Bp + i] = nrm_g; D.res[1 * Bp + i] = nrm_b; D.res[2 * Bp + i] = nrm_d; D.res[3 * Bp + i] = nrm_m;
        const int it = D.iter[i];
        if (i < D.stat_inst && it < D.stat_rows)
        {
            double *st = D.stat + (size_t) it * GQP_STAT_COLS * D.stat_inst + i;
            st[6 * D.stat_inst] = mu;
            st[7 * D.stat_inst] = nrm_g; st[8 * D.stat_inst] = nrm_b; st[9 * D.stat_inst] = nrm_d; st[10 * D.stat_inst] = nrm_m;
            st[12 * D.stat_inst] = obj;
        }
        int status = GQP_RUNNING;
        const bool bad = nrm_g != nrm_g || nrm_b != nrm_b || nrm_d != nrm_d || nrm_m != nrm_m || mu != mu;
        if (bad) status = 1;                                   /* ACADOS_NAN_DETECTED */
        else if (nrm_g <= O.tol_stat && nrm_b <= O.tol_eq && nrm_d <= O.tol_ineq && nrm_m <= O.tol_comp) status = 0;
        else if (it >= O.iter_max) status = 2;                 /* ACADOS_MAXITER */
        else if (dabs(D.alpha[i]) <= O.alpha_min) status = 3;  /* ACADOS_MINSTEP */
        if (status != GQP_RUNNING)
        {
            D.status[i] = status;
            atomicSub(D.n_active, 1);
        }
    }
}

/* ---------------------------------------------------------- k_forward */

/*
 * Forward sweep k = 0..N: dux, dpi, dsv, dt, dlam and the step length.
 * CORR = false: affine direction; ends with mu_aff, sigma (-> smu = sigma*mu).
 * CORR = true : corrected direction; ends with the update of (ux, sv, pi, lam, t).
 */
template <int NX, int NU, int NG, int NS, bool CORR>
__global__ void __launch_bounds__(64) k_forward(GqpDev D, GqpOpts O, int redo)
{
    constexpr int n = NX + NU;
    constexpr int NP = n * (n + 1) / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.B) return;
    if (D.status[i] != GQP_RUNNING) return;

    if (redo && !(D.alpha[i] < 0.0)) return;
    const double smu = CORR ? D.smu[i] : 0.0;
    const bool center_only = CORR && redo;
    double alpha = 1.0;
    double dx[NX];
    UNROLL for (int c = 0; c < NX; c++) dx[c] = 0.0;

    for (int k = 0; k <= D.N; k++)
    {
        GQP_STAGE_REF S = D.st[k];
        const int nbg = S.nb + S.ng;
        const uint64_t am = GAT(D.amask, k);
        double L[NP], l[n], dv[n];
        UNROLL for (int e = 0; e < NP; e++) L[e] = GAT(D.Lf, k * NP + e);
        UNROLL for (int j = 0; j < n; j++) l[j] = GAT(D.lf, k * n + j);
        if (k > 0)
        {
            /* dpi_k = Lx (Lx' dx + lx) with the x-block of this stage's factor */
            double w0[NX];
            UNROLL for (int c = 0; c < NX; c++)
            {
                double a = l[NU + c];
                UNROLL for (int q = c; q < NX; q++) a += L[PK(NU + q, NU + c)] * dx[q];
                w0[c] = a;
            }
            UNROLL for (int r = 0; r < NX; r++)
            {
                double a = 0.0;
                UNROLL for (int c = 0; c <= r; c++) a += L[PK(NU + r, NU + c)] * w0[c];
                GAT(D.dpi, k * NX + r) = a;
            }
            UNROLL for (int c = 0; c < NX; c++) dv[NU + c] = dx[c];
        }
        /* solve L' dv = -(l + ...) for the free block: all of [u;x] at k = 0, u otherwise */
        if (k == 0)
        {
            UNROLL for (int r = n - 1; r >= 0; r--)
            {
                double a = -l[r];
                UNROLL for (int p = r + 1; p < n; p++) a -= L[PK(p, r)] * dv[p];
                const double d = L[PK(r, r)];
                dv[r] = d != 0.0 ? a / d : 0.0;
            }
        }
        else
        {
            UNROLL for (int r = NU - 1; r >= 0; r--)
            {
                double a = -l[r];
                UNROLL for (int p = r + 1; p < n; p++) a -= L[PK(p, r)] * dv[p];
                const double d = L[PK(r, r)];
                dv[r] = d != 0.0 ? a / d : 0.0;
            }
        }
        UNROLL for (int j = 0; j < n; j++) GAT(D.dux, k * n + j) = dv[j];
        if (S.has_dyn)
        {
            UNROLL for (int c = 0; c < NX; c++) dx[c] = GAT(D.rb, k * NX + c);
            UNROLL for (int r = 0; r < n; r++)
                UNROLL for (int c = 0; c < NX; c++) dx[c] += GAT(D.BAt, (k * n + r) * NX + c) * dv[r];
        }

        /* ---- inequality rows: dc -> dsl/dsu -> dt -> dlam, step length ---- */
        constexpr int NSS = NS > 0 ? NS : 1;
        double dsl[NSS], dsu[NSS];
        /* per slack, for the cancellation-free dt of the soft rows (k_backward pass 2):
         * E0 = Z + Gamma_s, r~, 1/D, Gamma-weighted dc sums, rows coupled */
        double fE0l[NSS], fE0u[NSS], fRl[NSS], fRu[NSS], fIl[NSS], fIu[NSS], accl[NSS], accu[NSS];
        int fcnt[NSS];
        if (NS > 0)
        {
            /* dsl_j = (-r~sl_j - sum_i Gamma_l,i dc_i)/Dl_j ; dsu_j = (-r~su_j + sum_i Gamma_u,i dc_i)/Du_j */
            UNROLL for (int q = 0; q < NS; q++) { accl[q] = accu[q] = 0.0; fcnt[q] = 0; }
            int ib = 0;
            UNROLL for (int j = 0; j < n + NG; j++)
            {
                const bool is_row = j < n ? ((S.bmask >> j) & 1) : (j - n < S.ng);
                if (!is_row) continue;
                const int row = j < n ? ib : S.nb + (j - n);
                if (j < n) ib++;
                const int sj = S.srev[row];
                if (sj < 0) continue;
                double dc;
                if (j < n) dc = dv[j < n ? j : 0];
                else { dc = 0.0; UNROLL for (int r = 0; r < n; r++) dc += GAT(D.DCt, (S.o_g + (j - n)) * n + r) * dv[r]; }
                const int el = S.o_ct + row, eu = el + nbg;
                const bool al = (am >> row) & 1, au = (am >> (nbg + row)) & 1;
                const double gl = al ? GAT(D.lam, el) / GAT(D.t, el) : 0.0;
                const double gu = au ? GAT(D.lam, eu) / GAT(D.t, eu) : 0.0;
                UNROLL for (int q = 0; q < NS; q++) if (q == sj) { accl[q] += gl * dc; accu[q] += gu * dc; fcnt[q]++; }
            }
            UNROLL for (int q = 0; q < NS; q++)
            {
                dsl[q] = dsu[q] = 0.0;
                fE0l[q] = fE0u[q] = fRl[q] = fRu[q] = fIl[q] = fIu[q] = 0.0;
                if (q < S.ns)
                {
                    const double dl = GAT(D.sD, S.o_s + q), du = GAT(D.sD, S.o_s + S.ns + q);
                    const double rsl = GAT(D.sR, S.o_s + q), rsu = GAT(D.sR, S.o_s + S.ns + q);
                    dsl[q] = dl != 0.0 ? (-rsl - accl[q]) / dl : 0.0;
                    dsu[q] = du != 0.0 ? (-rsu + accu[q]) / du : 0.0;
                    GAT(D.dsv, S.o_s + q) = dsl[q];
                    GAT(D.dsv, S.o_s + S.ns + q) = dsu[q];
                    /* slack-bound rows */
                    const int e0 = S.o_ct + 2 * nbg + q, e1 = e0 + S.ns;
                    const bool al = (am >> (2 * nbg + q)) & 1, au = (am >> (2 * nbg + S.ns + q)) & 1;
                    double rml = GAT(D.rm, e0), rmu = GAT(D.rm, e1);
                    if (CORR)
                    {
                        if (al) rml += (center_only ? 0.0 : GAT(D.dlam, e0) * GAT(D.dt, e0)) - smu;
                        if (au) rmu += (center_only ? 0.0 : GAT(D.dlam, e1) * GAT(D.dt, e1)) - smu;
                    }
                    const double laml = GAT(D.lam, e0), lamu = GAT(D.lam, e1), tl = GAT(D.t, e0), tu = GAT(D.t, e1);
                    fE0l[q] = GAT(D.Zz, (S.o_s + q) * 2) + (al ? laml / tl : 0.0);
                    fE0u[q] = GAT(D.Zz, (S.o_s + S.ns + q) * 2) + (au ? lamu / tu : 0.0);
                    fRl[q] = rsl; fRu[q] = rsu;
                    fIl[q] = dl != 0.0 ? 1.0 / dl : 0.0; fIu[q] = du != 0.0 ? 1.0 / du : 0.0;
                    const double dtl = al ? dsl[q] + GAT(D.rd, e0) : 0.0, dtu = au ? dsu[q] + GAT(D.rd, e1) : 0.0;
                    const double dll = al ? -(rml + laml * dtl) / tl : 0.0, dlu = au ? -(rmu + lamu * dtu) / tu : 0.0;
                    GAT(D.dt, e0) = dtl; GAT(D.dt, e1) = dtu;
                    GAT(D.dlam, e0) = dll; GAT(D.dlam, e1) = dlu;
                    if (dll < 0.0 && -laml > alpha * dll) alpha = -laml / dll;
                    if (dlu < 0.0 && -lamu > alpha * dlu) alpha = -lamu / dlu;
                    if (dtl < 0.0 && -tl > alpha * dtl) alpha = -tl / dtl;
                    if (dtu < 0.0 && -tu > alpha * dtu) alpha = -tu / dtu;
                }
            }
        }
        {
            int ib = 0;
            UNROLL for (int j = 0; j < n + NG; j++)
            {
                const bool is_row = j < n ? ((S.bmask >> j) & 1) : (j - n < S.ng);
                if (!is_row) continue;
                const int row = j < n ? ib : S.nb + (j - n);
                if (j < n) ib++;
                const int el = S.o_ct + row, eu = el + nbg;
                const bool al = (am >> row) & 1, au = (am >> (nbg + row)) & 1;
                if (!al && !au) { GAT(D.dt, el) = 0.0; GAT(D.dt, eu) = 0.0; GAT(D.dlam, el) = 0.0; GAT(D.dlam, eu) = 0.0; continue; }
                double dc;
                if (j < n) dc = dv[j < n ? j : 0];
                else { dc = 0.0; UNROLL for (int r = 0; r < n; r++) dc += GAT(D.DCt, (S.o_g + (j - n)) * n + r) * dv[r]; }
                const double laml = al ? GAT(D.lam, el) : 0.0, lamu = au ? GAT(D.lam, eu) : 0.0;
                const double tl = al ? GAT(D.t, el) : 1.0, tu = au ? GAT(D.t, eu) : 1.0;
                /* dc + ds resp. -dc + ds; soft rows: (E dc - r~ - sum_{k != i} Gamma_k dc_k)/D, no cancellation */
                double dcl = dc, dcu = -dc;
                if (NS > 0)
                {
                    const int sj = S.srev[row];
                    if (sj >= 0)
                    {
                        double El = 0.0, Eu = 0.0, xl = 0.0, xu = 0.0, rsl = 0.0, rsu = 0.0, il = 0.0, iu = 0.0;
                        int cnt = 0;
                        UNROLL for (int q = 0; q < NS; q++)
                            if (q == sj) { El = fE0l[q]; Eu = fE0u[q]; rsl = fRl[q]; rsu = fRu[q]; il = fIl[q]; iu = fIu[q]; cnt = fcnt[q]; }
                        if (cnt > 1)
                        {
                            for (int k2 = 0; k2 < nbg; k2++)
                                if (k2 != row && S.srev[k2] == sj)
                                {
                                    const bool a2l = (am >> k2) & 1, a2u = (am >> (nbg + k2)) & 1;
                                    const double g2l = a2l ? GAT(D.lam, S.o_ct + k2) / GAT(D.t, S.o_ct + k2) : 0.0;
                                    const double g2u = a2u ? GAT(D.lam, S.o_ct + nbg + k2) / GAT(D.t, S.o_ct + nbg + k2) : 0.0;
                                    double dc2 = 0.0;
                                    if (k2 >= S.nb) { UNROLL for (int r = 0; r < n; r++) dc2 += GAT(D.DCt, (S.o_g + k2 - S.nb) * n + r) * dv[r]; }
                                    else
                                    {
                                        UNROLL for (int r = 0; r < n; r++)
                                            if (((S.bmask >> r) & 1) && popc64_g(S.bmask & (((uint64_t) 1 << r) - 1)) == k2) dc2 = dv[r];
                                    }
                                    El += g2l; Eu += g2u; xl += g2l * dc2; xu += g2u * dc2;
                                }
                        }
                        dcl = (El * dc - rsl - xl) * il;
                        dcu = (-Eu * dc - rsu + xu) * iu;
                    }
                }
                double rml = GAT(D.rm, el), rmu = GAT(D.rm, eu);
                if (CORR)
                {
                    if (al) rml += (center_only ? 0.0 : GAT(D.dlam, el) * GAT(D.dt, el)) - smu;
                    if (au) rmu += (center_only ? 0.0 : GAT(D.dlam, eu) * GAT(D.dt, eu)) - smu;
                }
                const double dtl = al ? dcl + GAT(D.rd, el) : 0.0;
                const double dtu = au ? dcu + GAT(D.rd, eu) : 0.0;
                const double dll = al ? -(rml + laml * dtl) / tl : 0.0;
                const double dlu = au ? -(rmu + lamu * dtu) / tu : 0.0;
                GAT(D.dt, el) = dtl; GAT(D.dt, eu) = dtu;
                GAT(D.dlam, el) = dll; GAT(D.dlam, eu) = dlu;
                if (dll < 0.0 && -laml > alpha * dll) alpha = -laml / dll;
                if (dlu < 0.0 && -lamu > alpha * dlu) alpha = -lamu / dlu;
                if (dtl < 0.0 && -tl > alpha * dtl) alpha = -tl / dtl;
                if (dtu < 0.0 && -tu > alpha * dtu) alpha = -tu / dtu;
            }
        }
    }

    const int it = D.iter[i];
    double *st = (i < D.stat_inst && it + 1 < D.stat_rows) ? D.stat + (size_t) (it + 1) * GQP_STAT_COLS * D.stat_inst + i : nullptr;
    /* duality measure at the end of the step of length a: sum (lam + a dlam)(t + a dt) / rows taking part */
    auto mu_after = [&](double a_) -> double
    {
        double s = 0.0;
        int nact = 0;
        for (int k = 0; k <= D.N; k++)
        {
            GQP_STAGE_REF S = D.st[k];
            const int nct = 2 * (S.nb + S.ng) + 2 * S.ns;
            const uint64_t am = GAT(D.amask, k);
            for (int e = 0; e < nct; e++)
                if ((am >> e) & 1)
                {
                    s += (GAT(D.lam, S.o_ct + e) + a_ * GAT(D.dlam, S.o_ct + e)) * (GAT(D.t, S.o_ct + e) + a_ * GAT(D.dt, S.o_ct + e));
                    nact++;
                }
        }
        return nact > 0 ? s / nact : 0.0;
    };
    if (!CORR)
    {
        /* mu_aff and sigma = (mu_aff/mu)^3 */
        const double mu = D.mu[i];
        const double mu_aff = mu_after(alpha);
        double sigma = mu > 0.0 ? mu_aff / mu : 0.0;
        sigma = sigma * sigma * sigma;
        D.smu[i] = sigma * mu;
        D.alpha[i] = alpha; /* alpha_aff, read by the cond_pred_corr test */
        if (st) { st[0] = alpha; st[1 * D.stat_inst] = alpha; st[2 * D.stat_inst] = mu_aff; st[3 * D.stat_inst] = sigma; }
    }
    else
    {
        const double alpha_aff = dabs(D.alpha[i]);
        if (O.cond_pred_corr && !redo && mu_after(alpha) > 2.0 * D.mu[i])
        {
            /* conditional corrector (GQP_COND_RULE, gpu_ipm_internal.h): the step would more than double the duality
             * measure -- ask the host loop for a centering-only re-solve (flag = negative alpha); no update this pass */
            D.alpha[i] = -alpha_aff;
            return;
        }
        /* no inequality rows (mu == 0 exactly): the Newton step solves the QP, take it fully */
        const double a = D.mu[i] > 0.0 ? gqp_step_scale(alpha) : 1.0;
        for (int k = 0; k <= D.N; k++)
        {
            GQP_STAGE_REF S = D.st[k];
            const int nct = 2 * (S.nb + S.ng) + 2 * S.ns;
            const uint64_t am = GAT(D.amask, k);
            UNROLL for (int j = 0; j < n; j++) GAT(D.ux, k * n + j) += a * GAT(D.dux, k * n + j);
            if (S.has_dyn) { UNROLL for (int c = 0; c < NX; c++) GAT(D.pi, (k + 1) * NX + c) += a * GAT(D.dpi, (k + 1) * NX + c); }
            for (int q = 0; q < 2 * S.ns; q++) GAT(D.sv, S.o_s + q) += a * GAT(D.dsv, S.o_s + q);
            for (int e = 0; e < nct; e++)
                if ((am >> e) & 1)
                {
                    double lam = GAT(D.lam, S.o_ct + e) + a * GAT(D.dlam, S.o_ct + e);
                    double t = GAT(D.t, S.o_ct + e) + a * GAT(D.dt, S.o_ct + e);
                    GAT(D.lam, S.o_ct + e) = lam < O.lam_min ? O.lam_min : lam;
                    GAT(D.t, S.o_ct + e) = t < O.t_min ? O.t_min : t;
                }
        }
        D.alpha[i] = alpha;
        D.iter[i] = it + 1;
        if (st) { st[4 * D.stat_inst] = alpha; st[5 * D.stat_inst] = alpha; }
    }
}

/* ----------------------------------------------------------- finalize */

/* natural slack value for rows that did not take part (masked sides); restates
 * ocp_qp_compute_t (acados/ocp_qp/ocp_qp_common.c:874-921) for those rows only */
template <int NX, int NU, int NG, int NS>
__global__ void __launch_bounds__(64) k_finalize(GqpDev D)
{
    constexpr int n = NX + NU;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.B) return;
    for (int k = 0; k <= D.N; k++)
    {
        GQP_STAGE_REF S = D.st[k];
        const int nbg = S.nb + S.ng;
        const uint64_t am = GATL(D.amask, k);
        double v[n];
        UNROLL for (int j = 0; j < n; j++) v[j] = GATL(D.ux, k * n + j);
        int ib = 0;
        UNROLL for (int j = 0; j < n + NG; j++)
        {
            const bool is_row = j < n ? ((S.bmask >> j) & 1) : (j - n < S.ng);
            if (!is_row) continue;
            const int row = j < n ? ib : S.nb + (j - n);
            if (j < n) ib++;
            const bool al = (am >> row) & 1, au = (am >> (nbg + row)) & 1;
            if (al && au) continue;
            double c;
            if (j < n) c = v[j < n ? j : 0];
            else { c = 0.0; UNROLL for (int r = 0; r < n; r++) c += GATL(D.DCt, (S.o_g + (j - n)) * n + r) * v[r]; }
            double ssl = 0.0, ssu = 0.0;
            if (NS > 0)
            {
                const int sj = S.srev[row];
                if (sj >= 0) { ssl = GATL(D.sv, S.o_s + sj); ssu = GATL(D.sv, S.o_s + S.ns + sj); }
            }
            const bool fixed = j < n && ((S.emask >> (j < n ? j : 0)) & 1);
            if (!al)
            {
                GATL(D.t, S.o_ct + row) = c + ssl - GATL(D.dvec, S.o_ct + row);
                if (!fixed) GATL(D.lam, S.o_ct + row) = 0.0;
            }
            if (!au)
            {
                GATL(D.t, S.o_ct + nbg + row) = GATL(D.dvec, S.o_ct + nbg + row) - c + ssu;
                if (!fixed) GATL(D.lam, S.o_ct + nbg + row) = 0.0;
            }
        }
        for (int q = 0; q < S.ns; q++)
        {
            const int e0 = S.o_ct + 2 * nbg + q, e1 = e0 + S.ns;
            if (!((am >> (2 * nbg + q)) & 1)) { GATL(D.t, e0) = GATL(D.sv, S.o_s + q) - GATL(D.dvec, e0); GATL(D.lam, e0) = 0.0; }
            if (!((am >> (2 * nbg + S.ns + q)) & 1)) { GATL(D.t, e1) = GATL(D.sv, S.o_s + S.ns + q) - GATL(D.dvec, e1); GATL(D.lam, e1) = 0.0; }
        }
    }
}

/* -------------------------------------------- layout conversion kernels */

/* element map[e] of instance i <- src[i * len + e]  (map[e] < 0: element dropped) */
static __global__ void k_scatter(const double *src, int nb, int len, const int *map, GArr dst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    for (int e = 0; e < len; e++)
    {
        const int m = map[e];
        if (m >= 0) GATL(dst, m) = src[(size_t) i * len + e];
    }
}

/* dst[i * len + e] = element map[e] of instance i  (map[e] < 0: 0) */
static __global__ void k_gather(double *dst, int nb, int len, const int *map, GArr src)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    for (int e = 0; e < len; e++)
    {
        const int m = map[e];
        dst[(size_t) i * len + e] = m >= 0 ? GATL(src, m) : 0.0;
    }
}

/* dst[i * len + e] = 1.0 / 0.0: bit bitpos[e] of amask[stage] (bitpos < 0: a row that always takes part) */
static __global__ void k_getmask(double *dst, int nb, int len, const int *bitpos, GArrU64 amask, int stage, int AW)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    for (int e = 0; e < len; e++)
    {
        const int b = bitpos[e];
        dst[(size_t) i * len + e] = (b < 0 || ((GATL(amask, stage * AW + (b >> 6)) >> (b & 63)) & 1)) ? 1.0 : 0.0;
    }
}

/* activity bit masks: for every element e of a (lower|upper|slack) mask vector handed
 * over by the caller, set or clear bit bitpos[e] of amask[stage] */
static __global__ void k_setmask(const double *src, int nb, int len, const int *bitpos, GArrU64 amask, int stage, int AW)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    for (int w = 0; w < AW; w++)
    {
        uint64_t m = GATL(amask, stage * AW + w);
        for (int e = 0; e < len; e++)
        {
            const int b = bitpos[e] - 64 * w;
            if (bitpos[e] < 0 || b < 0 || b >= 64) continue;
            if (src[(size_t) i * len + e] != 0.0) m |= (uint64_t) 1 << b;
            else m &= ~((uint64_t) 1 << b);
        }
        GATL(amask, stage * AW + w) = m;
    }
}

/* ---- bulk pack / unpack: one launch moves a whole QP (or solution) per instance ----
 * blob[i * len + e] <-> element map_elem[e] of array map_arr[e] (map_arr < 0: entry skipped) */
struct GArrTable
{
    GArr a[16];
};

static __global__ void k_bulk_scatter(const double *blob, int nb, int len, const int *map_arr, const int *map_elem,
                                      GArrTable T)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    const int e0 = blockIdx.y * 256, e1 = e0 + 256 < len ? e0 + 256 : len;
    for (int e = e0; e < e1; e++)
    {
        const int a = map_arr[e];
        if (a >= 0) GATL(T.a[a], map_elem[e]) = blob[(size_t) i * len + e];
    }
}

/* the same for INSTANCE-MAJOR destinations (the wave-per-instance families): lanes along the elements of one instance -- source and
 * destination of a wave are both runs of one instance's memory.  Grid (elements / 256, instances). */
static __global__ void __launch_bounds__(256) k_bulk_scatter_aos(const double *blob, int nb, int len, const int *map_arr, const int *map_elem,
                                                                 GArrTable T)
{
    const int i = blockIdx.y, e = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb || e >= len) return;
    const int a = map_arr[e];
    if (a >= 0) GATL(T.a[a], map_elem[e]) = blob[(size_t) i * len + e];
}

/* the same through an LDS tile, for WAVE-TILED destinations: the blob is instance-major, so a lane that walks its own instance makes
 * every load of its wave touch 64 different lines (0.44 ms for 4,096 C3-shaped QPs, 1.6 TB/s of reads + writes).  Here the wave reads
 * 64 consecutive doubles of one instance per load and writes element e of 64 consecutive instances per store.  Grid (instances / 64,
 * elements / 64), one wave per block. */
static __global__ void __launch_bounds__(64) k_bulk_scatter_tile(const double *blob, int nb, int len, const int *map_arr, const int *map_elem,
                                                                 GArrTable T)
{
    __shared__ double tile[64 * 65];
    const int lane = threadIdx.x, i0 = blockIdx.x * 64, e0 = blockIdx.y * 64;
    const int ni = nb - i0 < 64 ? nb - i0 : 64, ne = len - e0 < 64 ? len - e0 : 64;
    for (int r = 0; r < ni; r++)
        if (lane < ne) tile[lane * 65 + r] = blob[(size_t) (i0 + r) * len + e0 + lane];
    __syncthreads();
    const int i = i0 + lane;
    if (lane < ni)
        for (int c = 0; c < ne; c++)
        {
            const int a = map_arr[e0 + c];
            if (a >= 0) GATL(T.a[a], map_elem[e0 + c]) = tile[c * 65 + lane];
        }
}

/*
 * The step of an iteration applied by a launch of its own: (ux, pi, sv) += a d(.), (lam, t) of every side that takes part
 * += a d(.) floored at (lam_min, t_min) -- exactly what the update passes at the end of the corrector sweeps do, but with the
 * whole chip: those passes walk the stages of an instance one after the other inside a kernel whose waves are there for the
 * sweep's dependent chain (sixteen lanes per instance: N / 4..6 memory round trips per launch, twice that with general rows
 * -- 0.9 of the 2.5 ms of C4's corrector sweep), here every element is its own work item.
 *   instance-major arrays (the sixteen-lanes / wave-per-instance families): one workgroup of 256 per instance, threads over elements;
 *   wave-tiled arrays: one thread per instance, elements one after the other (coalesced across lanes).
 * side_map[e] = stage * 128 + bit of side e in the stage's activity word(s), or -1 for a side that takes no step (the two sides
 * of an equality-flagged box row).  a = GqpDev::apend[i], written by the corrector sweep (O.ext_update) for every instance it
 * ran; instances that are not iterating are skipped.
 */
#define GQP_UPD_THREADS 256 /* work items per instance (instance-major arrays) */
static __global__ void __launch_bounds__(GQP_UPD_THREADS) k_step_update(GqpDev D, GqpOpts O, const int *side_map, int n_sides, int n_sv)
{
    const bool aos = D.ux.aos != 0;
    const int tid = threadIdx.x;
    const int i = aos ? (int) blockIdx.x : (int) blockIdx.x * GQP_UPD_THREADS + tid;
    if (i >= D.B || D.status[i] != GQP_RUNNING) return;
    const double a = D.apend[i];
    if (a == 0.0) return;
    const int n = D.NX + D.NU, e0 = aos ? tid : 0, de = aos ? GQP_UPD_THREADS : 1;
    /* four elements per thread and round trip: all loads of a chunk are issued before its first store (a plain `x += a dx` loop makes
     * one memory round trip per element: the stores may alias the next loads as far as the compiler knows) */
#define GQP_UPD_AXPY(X, DX, FIRST, COUNT)                                                          \
    for (int e = (FIRST) + e0; e < (COUNT); e += 4 * de)                                           \
    {                                                                                              \
        double x_[4], d_[4];                                                                       \
        UNROLL for (int q = 0; q < 4; q++)                                                         \
        {                                                                                          \
            const int ee = e + q * de < (COUNT) ? e + q * de : e;                                  \
            x_[q] = GATL(X, ee); d_[q] = GATL(DX, ee);                                             \
        }                                                                                          \
        UNROLL for (int q = 0; q < 4; q++)                                                         \
            if (e + q * de < (COUNT)) GATL(X, e + q * de) = x_[q] + a * d_[q];                     \
    }
    GQP_UPD_AXPY(D.ux, D.dux, 0, (D.N + 1) * n)
    GQP_UPD_AXPY(D.pi, D.dpi, D.NX, (D.N + 1) * D.NX)
    GQP_UPD_AXPY(D.sv, D.dsv, 0, n_sv)
#undef GQP_UPD_AXPY
    for (int e = e0; e < n_sides; e += 4 * de)
    {
        double l_[4], dl_[4], t_[4], dt_[4];
        bool on[4];
        UNROLL for (int q = 0; q < 4; q++)
        {
            const int ee = e + q * de < n_sides ? e + q * de : e;
            const int sb = side_map[ee];
            const int k = sb >= 0 ? sb >> 7 : 0, bit = sb >= 0 ? sb & 127 : 0;
            on[q] = e + q * de < n_sides && sb >= 0 && ((GATL(D.amask, k * D.AW + (bit >> 6)) >> (bit & 63)) & 1);
            l_[q] = GATL(D.lam, ee); dl_[q] = GATL(D.dlam, ee); t_[q] = GATL(D.t, ee); dt_[q] = GATL(D.dt, ee);
        }
        UNROLL for (int q = 0; q < 4; q++)
            if (on[q])
            {
                const double lam = l_[q] + a * dl_[q], t = t_[q] + a * dt_[q];
                GATL(D.lam, e + q * de) = lam < O.lam_min ? O.lam_min : lam;
                GATL(D.t, e + q * de) = t < O.t_min ? O.t_min : t;
            }
    }
}

/* a launch that does nothing but carry a number in its NAME: the profile summaries (profiles/summarize.py) cut the
 * dispatch sequence of a traced run into labelled sections with it (option "marker") */
template <int ID>
static __global__ void k_marker(int *sink)
{
    if (sink && threadIdx.x == 1023) *sink = ID;
}

/* seeds of a sensitivity solve, all fields of all stages in one launch: entry e of the blob goes (with its sign: the
 * residual arrays hold lower-bound seeds negated) to array T.a[map_arr[e]] and, where the row is an equality-flagged
 * bound, also into the table's array 4 (derivative of the fixed variable itself) */
static __global__ void k_bulk_scatter_seed(const double *blob, int nb, int len, const int *map_arr, const int *map_elem,
                                           const int *map_sgn, const int *map_elem2, GArrTable T)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    const int e0 = blockIdx.y * 256, e1 = e0 + 256 < len ? e0 + 256 : len;
    for (int e = e0; e < e1; e++)
    {
        const double v = blob[(size_t) i * len + e];
        const int a = map_arr[e];
        if (a >= 0) GATL(T.a[a], map_elem[e]) = map_sgn[e] < 0 ? -v : v;
        if (map_elem2[e] >= 0) GATL(T.a[4], map_elem2[e]) = v;
    }
}

/* ZERO-COPY GATHER: instance i's QP data read by the device straight from the caller's (registered) host memory -- word w of the
 * class-wide table is element w_off[w] of the instance's source array w_slot[w] (ptrs[i * P + slot]: acados' BLASFEO storage, panel-major
 * matrices and plain vectors) and lands at position w_pos[w] of the instance's bulk blob in device memory, negated where w_neg[w].  The
 * words are sorted by (slot, offset): consecutive lanes read consecutive host addresses, the reads cross PCIe as full lines. */
static __global__ void __launch_bounds__(256) k_gather_host(const double *const *ptrs, int P, int nb, int n_words, const int *w_slot, const int *w_off,
                                                            const int *w_pos, const unsigned char *w_neg, double *blob, int len)
{
    const int i = blockIdx.y, w = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb || w >= n_words) return;
    const double v = ptrs[(size_t) i * P + w_slot[w]][w_off[w]];
    blob[(size_t) i * len + w_pos[w]] = w_neg[w] ? -v : v;
}

static __global__ void k_bulk_gather(double *blob, int nb, int len, const int *map_arr, const int *map_elem, GArrTable T)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    const int e0 = blockIdx.y * 256, e1 = e0 + 256 < len ? e0 + 256 : len;
    for (int e = e0; e < e1; e++)
    {
        const int a = map_arr[e];
        blob[(size_t) i * len + e] = a >= 0 ? GATL(T.a[a], map_elem[e]) : 0.0;
    }
}

/* ... from INSTANCE-MAJOR arrays: lanes along the elements of one instance (k_bulk_scatter_aos the other way round).  Grid (elements / 256,
 * instances). */
static __global__ void __launch_bounds__(256) k_bulk_gather_aos(double *blob, int nb, int len, const int *map_arr, const int *map_elem, GArrTable T)
{
    const int i = blockIdx.y, e = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb || e >= len) return;
    const int a = map_arr[e];
    blob[(size_t) i * len + e] = a >= 0 ? GATL(T.a[a], map_elem[e]) : 0.0;
}

/* ... from WAVE-TILED arrays, through an LDS tile (k_bulk_scatter_tile the other way round).  Grid (instances / 64, elements / 64). */
static __global__ void __launch_bounds__(64) k_bulk_gather_tile(double *blob, int nb, int len, const int *map_arr, const int *map_elem, GArrTable T)
{
    __shared__ double tile[64 * 65];
    const int lane = threadIdx.x, i0 = blockIdx.x * 64, e0 = blockIdx.y * 64;
    const int ni = nb - i0 < 64 ? nb - i0 : 64, ne = len - e0 < 64 ? len - e0 : 64;
    const int i = i0 + lane;
    if (lane < ni)
        for (int c = 0; c < ne; c++)
        {
            const int a = map_arr[e0 + c];
            tile[c * 65 + lane] = a >= 0 ? GATL(T.a[a], map_elem[e0 + c]) : 0.0;
        }
    __syncthreads();
    for (int r = 0; r < ni; r++)
        if (lane < ne) blob[(size_t) (i0 + r) * len + e0 + lane] = tile[lane * 65 + r];
}

/* mask entries of the blob: (offset in blob, stage, bit) triples */
static __global__ void k_bulk_masks(const double *__restrict__ blob, int nb, int len, const int *__restrict__ m_off, const int *__restrict__ m_stage,
                                    const int *__restrict__ m_bit, int nm, GArrU64 amask, int AW, int spc)
{
    /* grid.y: groups of `spc` stages (mask words belong to ONE stage: no two groups touch the same word); a word is read once, takes
     * all its bits in a register and is written once -- a read-modify-write of global memory per ENTRY, one thread walking all of an
     * instance's entries, was a chain of 300 dependent round trips (0.11 ms whatever the batch size) */
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    const int s0 = blockIdx.y * spc, s1 = s0 + spc;
    int wc = -1;
    uint64_t m = 0;
    for (int q = 0; q < nm; q++)
    {
        const int st = m_stage[q], bq = m_bit[q];
        if (bq < 0 || st < s0 || st >= s1) continue;
        const int w = st * AW + (bq >> 6), bit = bq & 63;
        if (w != wc)
        {
            if (wc >= 0) GATL(amask, wc) = m;
            wc = w;
            m = GATL(amask, w);
        }
        if (blob[(size_t) i * len + m_off[q]] != 0.0) m |= (uint64_t) 1 << bit;
        else m &= ~((uint64_t) 1 << bit);
    }
    if (wc >= 0) GATL(amask, wc) = m;
}

/* the reverse: mask entries of the blob <- the bits (1.0 / 0.0) */
static __global__ void k_bulk_masks_get(double *blob, int nb, int len, const int *m_off, const int *m_stage,
                                        const int *m_bit, int nm, GArrU64 amask, int AW)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    for (int q = 0; q < nm; q++)
    {
        if (m_bit[q] < 0) { blob[(size_t) i * len + m_off[q]] = 1.0; continue; } /* an equality-flagged bound: always in force */
        const int w = m_stage[q] * AW + (m_bit[q] >> 6), bit = m_bit[q] & 63;
        blob[(size_t) i * len + m_off[q]] = ((GATL(amask, w) >> bit) & 1) ? 1.0 : 0.0;
    }
}

/* gather (dir 0): slot s of the dense level `c` <- instance list[s] of level `b`; scatter (dir 1):
 * the reverse.  blockIdx.y selects a chunk of 64 elements, so the copy is parallel over elements;
 * `list` is sorted, so the strided side still touches few 512-byte lines per wave. */
template <class T>
static __global__ void k_compact_copy(GArrT<T> big, GArrT<T> small, const int *list, int cnt, int dir)
{
    const int sidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx >= cnt) return;
    const int src = list[sidx];
    const int e0 = blockIdx.y * 64, e1 = e0 + 64 < big.E ? e0 + 64 : big.E;
    const size_t sb = big.aos ? 1 : 64, ss = small.aos ? 1 : 64;
    T *pb = big.aos ? big.p + (size_t) src * big.E : big.p + ((size_t) (src >> 6) * (size_t) big.E) * 64 + (src & 63);
    T *ps = small.aos ? small.p + (size_t) sidx * small.E : small.p + ((size_t) (sidx >> 6) * (size_t) small.E) * 64 + (sidx & 63);
    if (dir == 0) for (int e = e0; e < e1; e++) ps[(size_t) e * ss] = pb[(size_t) e * sb];
    else for (int e = e0; e < e1; e++) pb[(size_t) e * sb] = ps[(size_t) e * ss];
}

/* the same copy between a wave-tiled and an instance-major level (tail switch, sensitivity slices): a 64 x 64
 * (slots x elements) tile goes through LDS so that BOTH sides see coalesced accesses -- lanes walk instances on the
 * wave-tiled side and elements on the instance-major side.  One wave per block; grid (slots/64, elements/64). */
template <class T>
static __global__ void __launch_bounds__(64) k_compact_tile(GArrT<T> big, GArrT<T> small, const int *list, int cnt, int dir)
{
    __shared__ T tile[64 * 65];
    const int lane = threadIdx.x, s0 = blockIdx.x * 64, e0 = blockIdx.y * 64;
    const int ns = cnt - s0 < 64 ? cnt - s0 : 64, ne = big.E - e0 < 64 ? big.E - e0 : 64;
    const GArrT<T> &X = dir == 0 ? big : small, &Y = dir == 0 ? small : big;
    const bool x_big = dir == 0;
    /* source -> tile[e][s] */
    if (X.aos)
        for (int s = 0; s < ns; s++)
        {
            const int inst = x_big ? list[s0 + s] : s0 + s;
            if (lane < ne) tile[lane * 65 + s] = X.p[(size_t) inst * X.E + e0 + lane];
        }
    else if (lane < ns)
    {
        const int inst = x_big ? list[s0 + lane] : s0 + lane;
        const T *px = X.p + ((size_t) (inst >> 6) * (size_t) X.E + e0) * 64 + (inst & 63);
        for (int e = 0; e < ne; e++) tile[e * 65 + lane] = px[(size_t) e * 64];
    }
    __syncthreads();
    /* tile -> destination */
    if (Y.aos)
        for (int s = 0; s < ns; s++)
        {
            const int inst = x_big ? s0 + s : list[s0 + s];
            if (lane < ne) Y.p[(size_t) inst * Y.E + e0 + lane] = tile[lane * 65 + s];
        }
    else if (lane < ns)
    {
        const int inst = x_big ? s0 + lane : list[s0 + lane];
        T *py = Y.p + ((size_t) (inst >> 6) * (size_t) Y.E + e0) * 64 + (inst & 63);
        for (int e = 0; e < ne; e++) py[(size_t) e * 64] = tile[e * 65 + lane];
    }
}

static __global__ void k_compact_scalars(GqpDev big, GqpDev small, const int *list, int cnt, int dir)
{
    const int sidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx >= cnt) return;
    const int i = list[sidx];
    if (dir == 0)
    {
        small.iter[sidx] = big.iter[i];
        small.status[sidx] = big.status[i];
        small.alpha[sidx] = big.alpha[i];
        small.mu[sidx] = big.mu[i];
        small.smu[sidx] = big.smu[i];
    }
    else
    {
        big.iter[i] = small.iter[sidx];
        big.status[i] = small.status[sidx];
        big.alpha[i] = small.alpha[sidx];
        big.mu[i] = small.mu[sidx];
        big.obj[i] = small.obj[sidx];
        for (int q = 0; q < 4; q++) big.res[(size_t) q * big.Bp + i] = small.res[(size_t) q * small.Bp + sidx];
    }
}

/* ---- solution sensitivities (forward / adjoint with the factorisation at the solution) ----
 * The seed of a parameter p is the derivative of the problem data: d g/dp (-> rg), d b/dp (-> rb), d bounds/dp in
 * natural sign (-> rd: lower sides -d lb, upper sides +d ub).  A corrector-style pass with these "residuals" and
 * zero complementarity rhs returns -K^{-1} (d residual/dp) = d solution/dp in (dux, dsv, dpi, dlam, dt).
 * k_sens_prep makes the complementarity rhs vanish (pcorr = tau - lam t, smu = 0) and wakes every instance up;
 * k_sens_fixed moves the derivative of equality-flagged variables (their value IS the parameter, e.g. x0) into the
 * seeds of the rows they touch -- dynamics, stationarity of the free variables, general rows -- and afterwards
 * (out = 1) writes it into dux. */
static __global__ void k_sens_prep(GqpDev D, double tau, int *saved_status)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.B) return;
    for (int e = 0; e < D.lam.E; e++) GATL(D.pcorr, e) = tau - GATL(D.lam, e) * GATL(D.t, e);
    D.smu[i] = 0.0;
    saved_status[i] = D.status[i];
    D.status[i] = GQP_RUNNING;
}

/* Hot start (warm_start >= 2, acados_ocp_options.py:1029-1032): the iterate (ux, pi, lam, t) in HBM is the starting
 * point.  Per-instance loop state is reset exactly as the cold-start kernels leave it (iter 0, running, alpha 1 --
 * a stale alpha of 0 would read as MINSTEP in the first factor sweep), and t / lam of the rows that take part are
 * clipped from below (warm_start 2: 0.1; 3: the t0_min / lam0_min the factorize-only path of
 * ocp_nlp_common.c:3946-3971 sets to keep the factorisation well conditioned). */
static __global__ void k_hot_start(GqpDev D, double t_min, double lam_min)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.B) return;
    D.iter[i] = 0;
    D.status[i] = GQP_RUNNING;
    D.alpha[i] = 1.0;
    D.mu[i] = 0.0;
    D.smu[i] = 0.0;
    for (int k = 0; k <= D.N; k++)
    {
        GQP_STAGE_REF S = D.st[k];
        const int nct = 2 * (S.nb + S.ng) + 2 * S.ns;
        for (int e = 0; e < nct; e++)
        {
            if (!((GATL(D.amask, k * D.AW + (e >> 6)) >> (e & 63)) & 1)) continue;
            const double tv = GATL(D.t, S.o_ct + e), lv = GATL(D.lam, S.o_ct + e);
            if (!(tv >= t_min)) GATL(D.t, S.o_ct + e) = t_min;
            if (!(lv >= lam_min)) GATL(D.lam, S.o_ct + e) = lam_min;
        }
    }
}

/* (status, iter) of every instance interleaved: the per-instance part of the multi-GPU gather payload */
static __global__ void k_pack_info(GqpDev D, int *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.B) return;
    out[2 * i] = D.status[i];
    out[2 * i + 1] = D.iter[i];
}

static __global__ void k_status_restore(GqpDev D, const int *saved_status)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < D.B) D.status[i] = saved_status[i];
}

/* ---- terminal polishing step (option "polish", opt-in; gpu_batch.hip polish_pass) ----
 * An IPM stops ON the central path: a weakly active row with multiplier lam* sits at t = mu / lam*, so at tol_comp 1e-8 the returned
 * point can be 1e-6 ... 1e-4 away from the solution although every KKT residual is below tolerance.  Such rows are
 * the ones whose pair (lam, t) is BALANCED at the exit -- min(lam, t) > ratio * max(lam, t) -- where a strictly active / inactive
 * row has one of the two at the size of mu.  k_polish_select wakes the converged instances that hold such a pair for ONE more
 * iteration of the sweeps (iteration counter 0 against iter_max 1, exit tolerances that cannot be met); k_polish_restore puts
 * (status, iter) back where the solve left them and flags the instances whose polished point no longer passes the exit test the
 * solve was run with -- k_polish_revert copies their iterate back from the copy taken before the step. */
static __global__ void k_polish_select(GqpDev D, const int *side_map, int n_sides, double ratio, double vmin, int *saved_status, int *saved_iter, double *saved_sc,
                                       int *count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.B) return;
    const int st = D.status[i];
    saved_status[i] = st;
    saved_iter[i] = D.iter[i];
    for (int q = 0; q < 4; q++) saved_sc[q * D.Bp + i] = D.res[q * D.Bp + i]; /* what the getters report for an instance that is reverted */
    saved_sc[4 * D.Bp + i] = D.mu[i];
    saved_sc[5 * D.Bp + i] = D.obj[i];
    if (st != 0) return;
    bool weak = false;
    for (int e = 0; e < n_sides && !weak; e++)
    {
        const int sb = side_map[e];
        if (sb < 0) continue;
        const int k = sb >> 7, bit = sb & 127;
        if (!((GATL(D.amask, k * D.AW + (bit >> 6)) >> (bit & 63)) & 1)) continue;
        const double l = GATL(D.lam, e), t = GATL(D.t, e);
        const double lo = l < t ? l : t, hi = l < t ? t : l;
        weak = lo > ratio * hi && lo > vmin;
    }
    if (!weak) return;
    D.status[i] = GQP_RUNNING;
    D.iter[i] = 0;
    D.alpha[i] = 1.0;
    atomicAdd(count, 1);
}

static __global__ void k_polish_restore(GqpDev D, GqpOpts O, const int *saved_status, const int *saved_iter, const double *saved_sc, int *flag, int *count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.B) return;
    const bool polished = saved_status[i] == 0 && D.status[i] != 0; /* woken by k_polish_select (left at MAXITER of the one-iteration loop) */
    flag[i] = 0;
    if (polished || D.status[i] == GQP_RUNNING)
    {
        const double g = D.res[0 * D.Bp + i], b = D.res[1 * D.Bp + i], d = D.res[2 * D.Bp + i], m = D.res[3 * D.Bp + i];
        const bool ok = g <= O.tol_stat && b <= O.tol_eq && d <= O.tol_ineq && m <= O.tol_comp; /* false for NaN */
        if (!ok)
        {
            flag[i] = 1;
            atomicAdd(count, 1);
            for (int q = 0; q < 4; q++) D.res[q * D.Bp + i] = saved_sc[q * D.Bp + i];
            D.mu[i] = saved_sc[4 * D.Bp + i];
            D.obj[i] = saved_sc[5 * D.Bp + i];
        }
    }
    D.status[i] = saved_status[i];
    D.iter[i] = saved_iter[i];
}

static __global__ void __launch_bounds__(GQP_UPD_THREADS) k_polish_revert(GqpDev D, const int *flag, GArr ux, GArr sv, GArr pi, GArr lam, GArr t)
{
    const bool aos = D.ux.aos != 0;
    const int tid = threadIdx.x;
    const int i = aos ? (int) blockIdx.x : (int) blockIdx.x * GQP_UPD_THREADS + tid;
    if (i >= D.B || !flag[i]) return;
    const int e0 = aos ? tid : 0, de = aos ? GQP_UPD_THREADS : 1;
    for (int e = e0; e < D.ux.E; e += de) GATL(D.ux, e) = GATL(ux, e);
    for (int e = e0; e < D.sv.E; e += de) GATL(D.sv, e) = GATL(sv, e);
    for (int e = e0; e < D.pi.E; e += de) GATL(D.pi, e) = GATL(pi, e);
    for (int e = e0; e < D.lam.E; e += de) GATL(D.lam, e) = GATL(lam, e);
    for (int e = e0; e < D.t.E; e += de) GATL(D.t, e) = GATL(t, e);
}

static __global__ void k_sens_fixed(GqpDev D, GArr sfix, int out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.B) return;
    const int NX = D.NX, NU = D.NU, n = NX + NU, NP = n * (n + 1) / 2;
    for (int k = 0; k <= D.N; k++)
    {
        GQP_STAGE_REF S = D.st[k];
        if (!S.emask) continue;
        for (int j = 0; j < n; j++)
        {
            if (!((S.emask >> j) & 1)) continue;
            const double dl = GATL(sfix, k * n + j);
            if (out) { GATL(D.dux, k * n + j) = dl; continue; }
            if (dl == 0.0) continue;
            if (S.has_dyn)
                for (int c = 0; c < NX; c++) GATL(D.rb, k * NX + c) += GATL(D.BAt, (k * n + j) * NX + c) * dl;
            for (int r = 0; r < n; r++)
                if (!((S.emask >> r) & 1)) GATL(D.rg, k * n + r) += GATL(D.RSQ, k * NP + (r >= j ? PK(r, j) : PK(j, r))) * dl;
            const int nbg = S.nb + S.ng;
            for (int g = 0; g < S.ng; g++)
            {
                const double a = GATL(D.DCt, (S.o_g + g) * n + j) * dl;
                GATL(D.rd, S.o_ct + S.nb + g) += a;
                GATL(D.rd, S.o_ct + nbg + S.nb + g) -= a;
            }
        }
    }
}

/* statistics rows >= row0 of the sub-level slots whose instance has a row in the parent's table
 * (the list is sorted: they are the first slots) */
static __global__ void k_stat_merge(GqpDev big, GqpDev small, const int *list, int nslots, int row0)
{
    const int sidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx >= nslots || sidx >= small.stat_inst) return;
    const int i = list[sidx];
    if (i >= big.stat_inst) return;
    const int rows = big.stat_rows < small.stat_rows ? big.stat_rows : small.stat_rows;
    const int r = row0 + blockIdx.y; /* one table row per block row */
    if (r >= rows) return;
    for (int c = 0; c < GQP_STAT_COLS; c++)
    {
        const double v = small.stat[((size_t) r * GQP_STAT_COLS + c) * small.stat_inst + sidx];
        if (v != 0.0) big.stat[((size_t) r * GQP_STAT_COLS + c) * big.stat_inst + i] = v;
    }
}

static __global__ void k_fill_u64(GArrU64 dst, uint64_t val, int nb, int e)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nb) GATL(dst, e) = val;
}

static __global__ void k_fill_strided(GArr dst, double val, int nb, int e)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nb) GATL(dst, e) = val;
}

} // namespace gqp

#endif
