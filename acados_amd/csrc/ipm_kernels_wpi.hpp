/*
 * ipm_kernels_wpi.hpp -- WAVE-PER-INSTANCE kernels of the batched OCP-QP interior-point solver.
 *
 * For stage blocks that do not fit the register file of the one-instance-per-lane mapping
 * (nu+nx > ~14: configuration C4, the nx = 12/24 classes of C5, the condensed QP of C3).  One
 * 64-lane wavefront (= one workgroup) owns one QP instance; the stage matrices live in LDS and the
 * 64 lanes split every dense operation:
 *   - lane r owns variable r of the stage ([u;x], nu+nx <= 64) for all vector work;
 *   - matrix entries are dealt round-robin over the lanes (packed lower-triangular index);
 *   - the Cholesky factorisation runs column by column in LDS with the rhs carried as an extra row
 *     (so l = L^{-1} m falls out of the same loop), two workgroup barriers per column.
 * Arrays are instance-major (GArrT::aos = 1): everything a wave reads in a stage is one contiguous
 * block, loaded with fully coalesced wave accesses.  Dimensions are RUNTIME values (dynamic LDS), so
 * one compiled kernel serves every shape -- there is no per-shape instantiation list.
 *
 * Same algorithm, same arrays and slot conventions as ipm_kernels_box.hpp (box constraints; general
 * constraints and slacks are handled by the general one-instance-per-lane kernels for now).
 * Only __syncthreads(), LDS and plain global accesses are used, so the CPU test tier can run these
 * kernels under tests/hostsim (lanes = host threads).
 */
#ifndef IPM_KERNELS_WPI_HPP_
#define IPM_KERNELS_WPI_HPP_

#include "ipm_kernels.hpp"
#include "ipm_kernels_box.hpp"

#ifndef GQP_DYN_SHARED /* the host-simulation shim of the CPU test tier brings its own */
#define GQP_DYN_SHARED(name) extern __shared__ double name[]
#endif
#ifndef GQP_LAUNCH_COOP
#define GQP_LAUNCH_COOP hipLaunchKernelGGL
#endif

/* lower bound on the waves per SIMD the register allocation must allow (template-dependent expressions are fine) */
#if defined(__HIPCC__)
#define GQP_WAVES_PER_EU(minw) __attribute__((amdgpu_waves_per_eu(minw, 8)))
#else
#define GQP_WAVES_PER_EU(minw)
#endif

/* dot-product loops over LDS operands: several reads in flight per lane */
#define GQP_DOT_UNROLL _Pragma("unroll 4")

namespace gqp
{

/* element e of instance `inst` in an instance-major array */
#define WAT(arr, e) (arr).p[(size_t) inst * (size_t) (arr).E + (size_t) (e)]

/* activity bits of one stage: up to 128 inequality sides (GqpDev::AW words) */
struct Am128
{
    uint64_t lo, hi;
};
__device__ static inline Am128 wpi_am(const GqpDev &D, int inst, int k)
{
    Am128 a;
    a.lo = WAT(D.amask, k * D.AW);
    a.hi = D.AW > 1 ? WAT(D.amask, k * D.AW + 1) : 0;
    return a;
}
__device__ static inline bool abit(const Am128 &a, int i) { return i < 64 ? (a.lo >> i) & 1 : (a.hi >> (i - 64)) & 1; }

/* LDS carve-up shared by the three sweep kernels (doubles) */
struct WpiLds
{
    double *M;   /* (n+1) x n : stage matrix / factor, row n = rhs row */
    double *B;   /* n x NX    : [B A]' */
    double *W;   /* n x NX    : [B A]' Lx+ */
    double *Lx;  /* NX x NX   : x-block of the factor of stage k+1 (lower) */
    double *lx, *v, *g, *rb, *pin, *pik, *w0, *y, *gam, *red; /* vectors */
    unsigned char *prow, *pcol; /* packed index -> (row, col) */
};

__host__ __device__ static inline size_t wpi_lds_doubles(int NX, int NU)
{
    const int n = NX + NU, NP = n * (n + 1) / 2;
    return (size_t) (n + 1) * n + 2 * (size_t) n * NX + (size_t) NX * NX + 10 * 64 + 64 + (2 * (size_t) NP + 15) / 8 + 8;
}

__device__ static inline WpiLds wpi_carve(double *sm, int NX, int NU)
{
    const int n = NX + NU;
    WpiLds L;
    double *p = sm;
    L.M = p; p += (n + 1) * n;
    L.B = p; p += n * NX;
    L.W = p; p += n * NX;
    L.Lx = p; p += NX * NX;
    L.lx = p; p += 64; L.v = p; p += 64; L.g = p; p += 64; L.rb = p; p += 64; L.pin = p; p += 64;
    L.pik = p; p += 64; L.w0 = p; p += 64; L.y = p; p += 64; L.gam = p; p += 64; L.red = p; p += 64;
    p += 64;
    L.prow = (unsigned char *) p;
    L.pcol = L.prow + n * (n + 1) / 2;
    return L;
}

/* packed lower-triangular index tables (lane r fills its row) */
__device__ static inline void wpi_tables(const WpiLds &L, int n, int lane)
{
    for (int r = lane; r < n; r += 64)
        for (int c = 0; c <= r; c++)
        {
            L.prow[PK(r, c)] = (unsigned char) r;
            L.pcol[PK(r, c)] = (unsigned char) c;
        }
    __syncthreads();
}

/* block-wide reductions through LDS (NaN-propagating max, sum, min) */
__device__ static inline double wpi_max(double v, double *red, int lane)
{
    red[lane] = v;
    __syncthreads();
    for (int s = 32; s > 0; s >>= 1)
    {
        if (lane < s)
        {
            const double a = red[lane + s], b = red[lane];
            red[lane] = (a > b || a != a) ? a : b;
        }
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}
__device__ static inline double wpi_sum(double v, double *red, int lane)
{
    red[lane] = v;
    __syncthreads();
    for (int s = 32; s > 0; s >>= 1)
    {
        if (lane < s) red[lane] += red[lane + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}
__device__ static inline double wpi_min(double v, double *red, int lane)
{
    red[lane] = v;
    __syncthreads();
    for (int s = 32; s > 0; s >>= 1)
    {
        if (lane < s && red[lane + s] < red[lane]) red[lane] = red[lane + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

/* column-by-column Cholesky of the n x n lower triangle of M with the rhs as row n:
 * afterwards M = L (lower), row n = l = L^{-1} m.  Non-positive pivots zero the column. */
__device__ static inline void wpi_chol_rhs(const WpiLds &L, int n, int lane)
{
    double *M = L.M;
    for (int j = 0; j < n; j++)
    {
        const double d = M[j * n + j];
        const bool pos = d > 0.0;
        const double inv = pos ? frsqrt(d) : 0.0;
        for (int r = j + 1 + lane; r <= n; r += 64) M[r * n + j] *= inv;
        __syncthreads();
        /* the pivot itself is rewritten only now: nobody reads it during the trailing update */
        if (lane == 63) M[j * n + j] = pos ? d * inv : 0.0;
        /* trailing update of rows j+1..n (row n = rhs), columns j+1..min(r, n-1) */
        const int m = n - j - 1;               /* size of the trailing triangle */
        const int T = m * (m + 1) / 2 + m;     /* + the rhs row (m entries) */
        for (int idx = lane; idx < T; idx += 64)
        {
            int r, c;
            if (idx < m * (m + 1) / 2) { r = j + 1 + L.prow[idx]; c = j + 1 + L.pcol[idx]; }
            else { r = n; c = j + 1 + (idx - m * (m + 1) / 2); }
            M[r * n + c] -= M[r * n + j] * M[c * n + j];
        }
        __syncthreads();
    }
}

/* forward substitution of the rhs row only (factor already in M): row n <- L^{-1} row n */
__device__ static inline void wpi_fsub(const WpiLds &L, int n, int lane)
{
    double *M = L.M;
    for (int j = 0; j < n; j++)
    {
        const double d = M[j * n + j];
        if (lane == 0) M[n * n + j] = d != 0.0 ? M[n * n + j] * frcp(d) : 0.0;
        __syncthreads();
        const double lj = M[n * n + j];
        for (int c = j + 1 + lane; c < n; c += 64) M[n * n + c] -= M[c * n + j] * lj;
        __syncthreads();
    }
}

/* ------------------------------------------------------------------ factor / rhs-only backward */

/* FACT = 1: residuals, norms, status decision, condensation, factorisation.
 * FACT = 0: corrector rhs (redo = 1: centering only), factor reused. */
template <bool FACT>
__global__ void __launch_bounds__(64) kw_backward(GqpDev D, GqpOpts O, int redo)
{
    GQP_DYN_SHARED(smem);
    const int NX = D.NX, NU = D.NU, n = NX + NU, NP = n * (n + 1) / 2;
    const int inst = blockIdx.x, lane = threadIdx.x;
    if (inst >= D.B) return;
    if (D.status[inst] != GQP_RUNNING) return;
    if (!FACT && redo && !(D.alpha[inst] < 0.0)) return;
    const WpiLds L = wpi_carve(smem, NX, NU);
    wpi_tables(L, n, lane);
    const double smu = FACT ? 0.0 : D.smu[inst];
    const double pscale = (FACT || redo) ? 0.0 : 1.0;

    for (int e = lane; e < NX * NX; e += 64) L.Lx[e] = 0.0;
    if (lane < NX) L.lx[lane] = 0.0;
    double nrm_g = 0.0, nrm_b = 0.0, nrm_d = 0.0, nrm_m = 0.0, musum = 0.0, obj = 0.0;
    int nact = 0;
    __syncthreads();

    for (int k = D.N; k >= 0; k--)
    {
        GQP_STAGE_REF S = D.st[k];
        const uint64_t imask = S.bmask & ~S.emask;
        const Am128 am = wpi_am(D, inst, k);
        const int nbg = S.nb;
        const bool mine = lane < n;
        const bool fixed = mine && ((S.emask >> lane) & 1);

        /* ---- loads into LDS (coalesced: the stage block of this instance is contiguous) ---- */
        for (int e = lane; e < n * NX; e += 64) L.B[e] = WAT(D.BAt, k * n * NX + e);
        if (FACT) { for (int p = lane; p < NP; p += 64) L.M[L.prow[p] * n + L.pcol[p]] = WAT(D.RSQ, k * NP + p); }
        else { for (int p = lane; p < NP; p += 64) L.M[L.prow[p] * n + L.pcol[p]] = WAT(D.Lf, k * NP + p); }
        double gt = 0.0; /* lane r: entry r of the gradient-like vector being built */
        if (FACT)
        {
            if (mine) { L.v[lane] = WAT(D.ux, k * n + lane); L.g[lane] = WAT(D.rq, k * n + lane); }
            if (lane < NX)
            {
                L.rb[lane] = WAT(D.bvec, k * NX + lane) - WAT(D.ux, (k + 1) * n + NU + lane);
                L.pin[lane] = WAT(D.pi, (k + 1) * NX + lane);
                L.pik[lane] = WAT(D.pi, k * NX + lane);
            }
        }
        else
        {
            if (mine) gt = WAT(D.rg, k * n + lane);
            if (lane < NX) L.rb[lane] = WAT(D.rb, k * NX + lane);
        }
        /* box row of variable `lane` */
        const bool has = mine && ((imask >> lane) & 1);
        const int ib = has ? popc64(S.bmask & (((uint64_t) 1 << lane) - 1)) : 0;
        const bool al = has && abit(am, ib), au = has && abit(am, nbg + ib);
        const int el = S.o_ct + ib, eu = el + nbg;
        const double ll = al ? WAT(D.lam, el) : 0.0, lu = au ? WAT(D.lam, eu) : 0.0;
        const double ttl = al ? WAT(D.t, el) : 1.0, ttu = au ? WAT(D.t, eu) : 1.0;
        __syncthreads();

        double gam = 0.0, gadd = 0.0;
        if (FACT)
        {
            /* rb, BAt pi+, W */
            if (lane < NX)
            {
                double a = L.rb[lane];
                GQP_DOT_UNROLL
                for (int r = 0; r < n; r++) a += L.B[r * NX + lane] * L.v[r];
                L.rb[lane] = a; /* own slot only */
            }
            if (mine)
            {
                double a = 0.0;
                GQP_DOT_UNROLL
                for (int c = 0; c < NX; c++) a += L.B[lane * NX + c] * L.pin[c];
                gt = a;
            }
            for (int e = lane; e < n * NX; e += 64)
            {
                const int r = e / NX, c = e - r * NX;
                double w = 0.0;
                GQP_DOT_UNROLL
                for (int q = c; q < NX; q++) w += L.B[r * NX + q] * L.Lx[q * NX + c];
                L.W[e] = w;
            }
            /* H v */
            double hv = 0.0;
            if (mine)
            {
                GQP_DOT_UNROLL
                for (int c = 0; c <= lane; c++) hv += L.M[lane * n + c] * L.v[c];
                GQP_DOT_UNROLL
                for (int c = lane + 1; c < n; c++) hv += L.M[c * n + lane] * L.v[c];
                obj += (0.5 * hv + L.g[lane]) * L.v[lane];
                gt += hv + L.g[lane];
                if (lane >= NU) gt -= L.pik[lane - NU];
            }
            /* box row */
            double rdl = 0.0, rdu = 0.0;
            if (has)
            {
                const double vj = L.v[lane];
                rdl = al ? vj - WAT(D.dvec, el) - ttl : 0.0;
                rdu = au ? WAT(D.dvec, eu) - vj - ttu : 0.0;
                const double rml = al ? ll * ttl - O.tau_min : 0.0, rmu = au ? lu * ttu - O.tau_min : 0.0;
                nacc(nrm_d, rdl); nacc(nrm_d, rdu); nacc(nrm_m, rml); nacc(nrm_m, rmu);
                musum += ll * ttl + lu * ttu;
                nact += (int) al + (int) au;
                gt -= ll - lu;
                const double itl = frcp(ttl), itu = frcp(ttu);
                gam = ll * itl + lu * itu;
                gadd = (rml + ll * rdl) * itl - (rmu + lu * rdu) * itu;
                WAT(D.rd, el) = rdl;
                WAT(D.rd, eu) = rdu;
            }
            if (fixed) gt = 0.0;
            if (mine) { nacc(nrm_g, gt); WAT(D.rg, k * n + lane) = gt; }
            __syncthreads(); /* rb complete, W complete */
            if (lane < NX) { nacc(nrm_b, L.rb[lane]); WAT(D.rb, k * NX + lane) = L.rb[lane]; }
        }
        else
        {
            if (has)
            {
                const double rdl = al ? WAT(D.rd, el) : 0.0, rdu = au ? WAT(D.rd, eu) : 0.0;
                const double rml = al ? ll * ttl - O.tau_min + pscale * WAT(D.pcorr, el) - smu : 0.0;
                const double rmu = au ? lu * ttu - O.tau_min + pscale * WAT(D.pcorr, eu) - smu : 0.0;
                gadd = (rml + ll * rdl) * frcp(ttl) - (rmu + lu * rdu) * frcp(ttu);
            }
        }
        /* w0 = Lx+' rb + lx+ ; y = Lx+ w0 */
        if (lane < NX)
        {
            double a = L.lx[lane];
            GQP_DOT_UNROLL
            for (int q = lane; q < NX; q++) a += L.Lx[q * NX + lane] * L.rb[q];
            L.w0[lane] = a;
        }
        if (FACT && mine) L.gam[lane] = gam;
        __syncthreads();
        double m;
        if (FACT)
        {
            /* M += W W' (+ reg and Gamma on the diagonal); m = gt + gadd + W w0 */
            for (int p = lane; p < NP; p += 64)
            {
                const int r = L.prow[p], c = L.pcol[p];
                double a = 0.0;
                GQP_DOT_UNROLL
                for (int q = 0; q < NX; q++) a += L.W[r * NX + q] * L.W[c * NX + q];
                if (r == c) a += O.reg_prim + L.gam[r];
                L.M[r * n + c] += a;
            }
            double a = 0.0;
            if (mine)
            {
                GQP_DOT_UNROLL
                for (int c = 0; c < NX; c++) a += L.W[lane * NX + c] * L.w0[c];
            }
            m = gt + gadd + a;
        }
        else
        {
            if (lane < NX)
            {
                double a = 0.0;
                GQP_DOT_UNROLL
                for (int c = 0; c <= lane; c++) a += L.Lx[lane * NX + c] * L.w0[c];
                L.y[lane] = a;
            }
            __syncthreads();
            double a = 0.0;
            if (mine)
            {
                GQP_DOT_UNROLL
                for (int c = 0; c < NX; c++) a += L.B[lane * NX + c] * L.y[c];
            }
            m = gt + gadd + a;
        }
        if (fixed) m = 0.0;
        __syncthreads();
        if (mine) L.M[n * n + lane] = m; /* rhs row */
        if (FACT && S.emask)
        {
            for (int p = lane; p < NP; p += 64)
            {
                const int r = L.prow[p], c = L.pcol[p];
                if (((S.emask >> r) & 1) || ((S.emask >> c) & 1)) L.M[r * n + c] = r == c ? 1.0 : 0.0;
            }
        }
        __syncthreads();
        if (FACT)
        {
            wpi_chol_rhs(L, n, lane);
            for (int p = lane; p < NP; p += 64) WAT(D.Lf, k * NP + p) = L.M[L.prow[p] * n + L.pcol[p]];
        }
        else
            wpi_fsub(L, n, lane);
        if (mine) WAT(D.lf, k * n + lane) = L.M[n * n + lane];
        /* x-block of the factor and of l for the next (earlier) stage */
        for (int e = lane; e < NX * NX; e += 64)
        {
            const int r = e / NX, c = e - r * NX;
            L.Lx[e] = c <= r ? L.M[(NU + r) * n + NU + c] : 0.0;
        }
        if (lane < NX) L.lx[lane] = L.M[n * n + NU + lane];
        __syncthreads();
    }

    if (FACT)
    {
        nrm_g = wpi_max(nrm_g, L.red, lane);
        nrm_b = wpi_max(nrm_b, L.red, lane);
        nrm_d = wpi_max(nrm_d, L.red, lane);
        nrm_m = wpi_max(nrm_m, L.red, lane);
        musum = wpi_sum(musum, L.red, lane);
        obj = wpi_sum(obj, L.red, lane);
        const double nact_d = wpi_sum((double) nact, L.red, lane);
        if (lane == 0)
        {
            const int Bp = D.Bp;
            const double mu = nact_d > 0.0 ? musum / nact_d : 0.0;
            D.mu[inst] = mu;
            D.obj[inst] = obj;
            D.res[0 * Bp + inst] = nrm_g; D.res[1 * Bp + inst] = nrm_b; D.res[2 * Bp + inst] = nrm_d; D.res[3 * Bp + inst] = nrm_m;
            const int it = D.iter[inst];
            if (inst < D.stat_inst && it < D.stat_rows)
            {
                double *st = D.stat + (size_t) it * GQP_STAT_COLS * D.stat_inst + inst;
                st[6 * D.stat_inst] = mu;
                st[7 * D.stat_inst] = nrm_g; st[8 * D.stat_inst] = nrm_b; st[9 * D.stat_inst] = nrm_d; st[10 * D.stat_inst] = nrm_m;
                st[12 * D.stat_inst] = obj;
            }
            int status = GQP_RUNNING;
            const bool bad = nrm_g != nrm_g || nrm_b != nrm_b || nrm_d != nrm_d || nrm_m != nrm_m || mu != mu;
            if (bad) status = 1;
            else if (nrm_g <= O.tol_stat && nrm_b <= O.tol_eq && nrm_d <= O.tol_ineq && nrm_m <= O.tol_comp) status = 0;
            else if (it >= O.iter_max) status = 2;
            else if (dabs(D.alpha[inst]) <= O.alpha_min) status = 3;
            if (status != GQP_RUNNING)
            {
                D.status[inst] = status;
                atomicSub(D.n_active, 1);
            }
        }
    }
}

/* ----------------------------------------------------------------------------------- forward */

template <bool CORR>
__global__ void __launch_bounds__(64) kw_forward(GqpDev D, GqpOpts O, int redo)
{
    GQP_DYN_SHARED(smem);
    const int NX = D.NX, NU = D.NU, n = NX + NU, NP = n * (n + 1) / 2;
    const int inst = blockIdx.x, lane = threadIdx.x;
    if (inst >= D.B) return;
    if (D.status[inst] != GQP_RUNNING) return;
    if (redo && !(D.alpha[inst] < 0.0)) return;
    const WpiLds L = wpi_carve(smem, NX, NU);
    wpi_tables(L, n, lane);
    const double smu = CORR ? D.smu[inst] : 0.0;
    const double pscale = (CORR && !redo) ? 1.0 : 0.0;
    double alpha = 1.0, S0 = 0.0, S1 = 0.0, S2 = 0.0;
    int nact = 0;
    double *dvv = L.v;  /* dv of the stage, entry per variable */
    double *dx = L.y;   /* dx of the stage being entered */
    if (lane < NX) dx[lane] = 0.0;
    __syncthreads();

    for (int k = 0; k <= D.N; k++)
    {
        GQP_STAGE_REF S = D.st[k];
        const uint64_t imask = S.bmask & ~S.emask;
        const Am128 am = wpi_am(D, inst, k);
        const int nbg = S.nb;
        const bool mine = lane < n;

        for (int p = lane; p < NP; p += 64) L.M[L.prow[p] * n + L.pcol[p]] = WAT(D.Lf, k * NP + p);
        for (int e = lane; e < n * NX; e += 64) L.B[e] = WAT(D.BAt, k * n * NX + e);
        if (mine) L.g[lane] = WAT(D.lf, k * n + lane); /* l */
        if (lane < NX) L.rb[lane] = WAT(D.rb, k * NX + lane);
        const bool has = mine && ((imask >> lane) & 1);
        const int ib = has ? popc64(S.bmask & (((uint64_t) 1 << lane) - 1)) : 0;
        const bool al = has && abit(am, ib), au = has && abit(am, nbg + ib);
        const int el = S.o_ct + ib, eu = el + nbg;
        const double ll = al ? WAT(D.lam, el) : 0.0, lu = au ? WAT(D.lam, eu) : 0.0;
        const double ttl = al ? WAT(D.t, el) : 1.0, ttu = au ? WAT(D.t, eu) : 1.0;
        const double rdl = al ? WAT(D.rd, el) : 0.0, rdu = au ? WAT(D.rd, eu) : 0.0;
        const double pl = (CORR && al) ? WAT(D.pcorr, el) : 0.0, pu = (CORR && au) ? WAT(D.pcorr, eu) : 0.0;
        __syncthreads();

        /* dpi_k = Lx (Lx' dx + lx) */
        if (CORR && k > 0)
        {
            if (lane < NX)
            {
                double a = L.g[NU + lane];
                GQP_DOT_UNROLL
                for (int q = lane; q < NX; q++) a += L.M[(NU + q) * n + NU + lane] * dx[q];
                L.w0[lane] = a;
            }
            __syncthreads();
            if (lane < NX)
            {
                double a = 0.0;
                GQP_DOT_UNROLL
                for (int c = 0; c <= lane; c++) a += L.M[(NU + lane) * n + NU + c] * L.w0[c];
                WAT(D.dpi, k * NX + lane) = a;
            }
        }
        /* solve L' dv = -l for the free block: everything at k = 0, the inputs otherwise.
         * acc[r] = -l[r] - sum_{p >= top} L[p][r] dv[p] first, then a short sequential sweep. */
        const int top = k == 0 ? n : NU;
        if (mine && lane >= top) dvv[lane] = dx[lane - NU];
        __syncthreads();
        double acc = 0.0;
        if (lane < top)
        {
            acc = -L.g[lane];
            GQP_DOT_UNROLL
            for (int p = top; p < n; p++) acc -= L.M[p * n + lane] * dvv[p];
        }
        for (int r = top - 1; r >= 0; r--)
        {
            if (lane == r)
            {
                const double d = L.M[r * n + r];
                dvv[r] = d != 0.0 ? acc * frcp(d) : 0.0;
            }
            __syncthreads();
            if (lane < r) acc -= L.M[r * n + lane] * dvv[r];
        }
        __syncthreads();
        const double dvj = mine ? dvv[lane] : 0.0;
        if (CORR && mine) WAT(D.dux, k * n + lane) = dvj;
        /* dx of the next stage */
        double dxn = 0.0;
        if (lane < NX)
        {
            dxn = L.rb[lane];
            GQP_DOT_UNROLL
            for (int r = 0; r < n; r++) dxn += L.B[r * NX + lane] * dvv[r];
        }
        /* box row: dt, dlam, ratio test */
        if (has)
        {
            const double rml = al ? ll * ttl - O.tau_min + pscale * pl - smu : 0.0;
            const double rmu = au ? lu * ttu - O.tau_min + pscale * pu - smu : 0.0;
            const double dtl = al ? dvj + rdl : 0.0, dtu = au ? -dvj + rdu : 0.0;
            const double dll = al ? -(rml + ll * dtl) * frcp(ttl) : 0.0;
            const double dlu = au ? -(rmu + lu * dtu) * frcp(ttu) : 0.0;
            const double c1 = -ll * frcp(dll), c2 = -lu * frcp(dlu), c3 = -ttl * frcp(dtl), c4 = -ttu * frcp(dtu);
            alpha = (dll < 0.0 && c1 < alpha) ? c1 : alpha;
            alpha = (dlu < 0.0 && c2 < alpha) ? c2 : alpha;
            alpha = (dtl < 0.0 && c3 < alpha) ? c3 : alpha;
            alpha = (dtu < 0.0 && c4 < alpha) ? c4 : alpha;
            /* (the sums in the corrector sweep too: the conditional corrector asks for the duality measure its step ends at) */
            S0 += ll * ttl + lu * ttu;
            S1 += ll * dtl + ttl * dll + lu * dtu + ttu * dlu;
            S2 += dll * dtl + dlu * dtu;
            nact += (int) al + (int) au;
            if (!CORR)
            {
                WAT(D.pcorr, el) = dll * dtl;
                WAT(D.pcorr, eu) = dlu * dtu;
            }
            else
            {
                WAT(D.dlam, el) = dll; WAT(D.dlam, eu) = dlu;
                WAT(D.dt, el) = dtl; WAT(D.dt, eu) = dtu;
            }
        }
        __syncthreads();
        if (lane < NX) dx[lane] = dxn;
        __syncthreads();
    }

    alpha = wpi_min(alpha, L.red, lane);
    const int it = D.iter[inst];
    double *st = (inst < D.stat_inst && it + 1 < D.stat_rows) ? D.stat + (size_t) (it + 1) * GQP_STAT_COLS * D.stat_inst + inst : nullptr;
    if (!CORR)
    {
        S0 = wpi_sum(S0, L.red, lane); S1 = wpi_sum(S1, L.red, lane); S2 = wpi_sum(S2, L.red, lane);
        const double nact_d = wpi_sum((double) nact, L.red, lane);
        if (lane == 0)
        {
            const double mu = D.mu[inst];
            const double mu_aff = nact_d > 0.0 ? (S0 + alpha * S1 + alpha * alpha * S2) / nact_d : 0.0;
            double sigma = mu > 0.0 ? mu_aff / mu : 0.0;
            sigma = sigma * sigma * sigma;
            D.smu[inst] = sigma * mu;
            D.alpha[inst] = alpha;
            if (st) { st[0] = alpha; st[1 * D.stat_inst] = alpha; st[2 * D.stat_inst] = mu_aff; st[3 * D.stat_inst] = sigma; }
        }
        return;
    }
    const double alpha_aff = dabs(D.alpha[inst]);
    if (O.cond_pred_corr && !redo)
    {
        /* conditional corrector: a step that would more than double the duality measure is taken again with the centering term
         * alone (redo pair of the host loop) */
        S0 = wpi_sum(S0, L.red, lane); S1 = wpi_sum(S1, L.red, lane); S2 = wpi_sum(S2, L.red, lane);
        const double nact_d = wpi_sum((double) nact, L.red, lane);
        if (nact_d > 0.0 && (S0 + alpha * S1 + alpha * alpha * S2) / nact_d > 2.0 * D.mu[inst])
        {
            __syncthreads(); /* everybody has read alpha[inst] */
            if (lane == 0) D.alpha[inst] = -alpha_aff;
            return;
        }
    }
    const double a = D.mu[inst] > 0.0 ? gqp_step_scale(alpha) : 1.0;
    /* update: the arrays of one instance are contiguous */
    for (int e = lane; e < (D.N + 1) * n; e += 64) WAT(D.ux, e) += a * WAT(D.dux, e);
    for (int e = NX + lane; e < (D.N + 1) * NX; e += 64) WAT(D.pi, e) += a * WAT(D.dpi, e);
    for (int k = 0; k <= D.N; k++)
    {
        GQP_STAGE_REF S = D.st[k];
        const uint64_t imask = S.bmask & ~S.emask;
        const Am128 am = wpi_am(D, inst, k);
        const int nbg = S.nb;
        if (lane < n && ((imask >> lane) & 1))
        {
            const int ib = popc64(S.bmask & (((uint64_t) 1 << lane) - 1));
            for (int side = 0; side < 2; side++)
            {
                const int e = S.o_ct + side * nbg + ib;
                if (!abit(am, side * nbg + ib)) continue;
                const double lam = WAT(D.lam, e) + a * WAT(D.dlam, e);
                const double t = WAT(D.t, e) + a * WAT(D.dt, e);
                WAT(D.lam, e) = lam < O.lam_min ? O.lam_min : lam;
                WAT(D.t, e) = t < O.t_min ? O.t_min : t;
            }
        }
    }
    __syncthreads();
    if (lane == 0)
    {
        D.alpha[inst] = alpha;
        D.iter[inst] = it + 1;
        if (st) { st[4 * D.stat_inst] = alpha; st[5 * D.stat_inst] = alpha; }
    }
}

/* ------------------------------------------------------------------ factor, register tiles */

/*
 * kw_factor<T8>: the factorisation sweep with the stage matrix held in REGISTER TILES.
 * The 64 lanes form an 8 x 8 grid (lr = lane / 8, lc = lane % 8); lane (lr, lc) owns the entries
 * (lr + 8a, lc + 8b), b <= a < T8 = ceil(n / 8), of the lower triangle (2-D cyclic distribution, so the
 * trailing matrix of every Cholesky column is spread over all lanes).  Per stage:
 *   - H (packed), [B A]' and the vectors are fetched with coalesced loads into LDS;
 *   - W = [B A]' Lx+ and M = H~ + W W' are computed as outer-product tiles: 2 T8 LDS reads feed
 *     T8 (T8+1)/2 FMAs per k-step (LDS rows have an odd stride: conflict-free);
 *   - Cholesky column j: the 8 owner lanes publish the column (one LDS round trip), every lane rescales
 *     its 2 T8 operands itself and updates its tiles; the rhs (one entry per lane, lane = variable) rides
 *     along, so l = L^{-1} m needs no extra pass.  One workgroup barrier per column (free for a
 *     single-wave workgroup; it is what the host simulation of the CPU test tier synchronises on).
 * Same arithmetic as kw_backward<true> (which stays as the plain reference of this family).
 */
/* ---------------------------------------------------- general constraints and slacks (GEN kernels)
 * Inequality rows of a stage: box rows (sorted by variable, row = rank of the variable in bmask) and general
 * rows (row = nb + g), each with a lower and an upper side; slack q couples to the rows with srev[row] == q.
 * Lane roles per stage: lane j < n owns variable j and its box row, lane g < ng owns general row g, lane
 * q < ns owns slack q (its two bound rows, its Z/z, its sums over the coupled rows).  What the roles exchange
 * goes through the small LDS block below.  Algebra as in ipm_kernels.hpp (k_backward / k_forward):
 *   hard row: M += (Gl+Gu) a a', grad += a (rho_l - rho_u)
 *   slack q:  D = Z + lam_s/t_s + sum Gamma_row, r~ = stationarity residual of the slack + rho_s + sum rho_row
 *   soft row i -> q, CANCELLATION-FREE form of the Schur complement of the slack block (an active soft row
 *   has Gamma_i >> Z, and Gamma_i - Gamma_i^2/D would lose every digit of the Z + Gamma_s it leaves behind):
 *             E_i = D - Gamma_i and X_i = r~ - rho_i are summed WITHOUT row i;
 *             M += (Gl El/Dl + Gu Eu/Du) a a',  grad += a ((rho_l El - Gl Xl)/Dl - (rho_u Eu - Gu Xu)/Du),
 *             dt_l = (El dc - r~l - sum_{k != i} Gl_k dc_k)/Dl + rd_l  in the forward sweep;
 *   rows sharing one slack add the cross terms  M -= (Gl_i Gl_k/Dl + Gu_i Gu_k/Du) a_i a_k'.
 * (oracle/ocp_qp_oracle.c stage_condense / expand_step use the same form.) */
struct WpiCon
{
    double *__restrict__ G;   /* NG x SG: general rows [D C] */
    double *__restrict__ rGl, *__restrict__ rGu, *__restrict__ rRl, *__restrict__ rRu, *__restrict__ rLl, *__restrict__ rLu; /* per row */
    double *__restrict__ sIl, *__restrict__ sIu, *__restrict__ sRl, *__restrict__ sRu, *__restrict__ dsl, *__restrict__ dsu;   /* per slack */
    double *__restrict__ sEl, *__restrict__ sEu, *__restrict__ sXl, *__restrict__ sXu; /* per slack: Z + Gamma_s, stationarity + rho_s */
    double *__restrict__ nuG, *__restrict__ gmG; /* per general row */
    int *__restrict__ rsj;  /* slack index of every row of the stage (copy of GqpStage::srev) */
    int *__restrict__ scnt; /* rows coupled to each slack */
    int SG;
};

/* sized by what the batch can hold: at most n box rows + NG general rows per stage, NS slacks (the LDS of a wave
 * decides how many waves a CU keeps: the arrays used to be sized for 64 rows / 32 slacks whatever the problem) */
__host__ __device__ static inline int wpi_con_rows(int n, int NG) { return (n + NG + 1) & ~1; }
__host__ __device__ static inline size_t wpi_con_doubles(int n, int NG, int NS)
{
    if (NG == 0 && NS == 0) return 0;
    const int SG = n | 1, NR = wpi_con_rows(n, NG), NSe = (NS + 1) & ~1, NGe = (NG + 1) & ~1;
    return (size_t) NG * SG + 6 * NR + 10 * NSe + 2 * NGe + NR / 2 + NSe / 2 + 8;
}

__device__ static inline WpiCon wpi_con_carve(double *p, int n, int NG, int NS)
{
    WpiCon C;
    const int NR = wpi_con_rows(n, NG), NSe = (NS + 1) & ~1, NGe = (NG + 1) & ~1;
    C.SG = n | 1;
    C.G = p; p += NG * C.SG;
    C.rGl = p; p += NR; C.rGu = p; p += NR; C.rRl = p; p += NR; C.rRu = p; p += NR; C.rLl = p; p += NR; C.rLu = p; p += NR;
    C.sIl = p; p += NSe; C.sIu = p; p += NSe; C.sRl = p; p += NSe; C.sRu = p; p += NSe; C.dsl = p; p += NSe; C.dsu = p; p += NSe;
    C.sEl = p; p += NSe; C.sEu = p; p += NSe; C.sXl = p; p += NSe; C.sXu = p; p += NSe;
    C.nuG = p; p += NGe; C.gmG = p; p += NGe;
    C.rsj = (int *) p; p += NR / 2;
    C.scnt = (int *) p;
    return C;
}

/* coalesced copy of the ng x n general rows of a stage into LDS rows of odd stride */
__device__ static inline void wpi_load_G(const WpiCon &C, GQP_STAGE_REF S, const GArr &DCt, int inst, int o_g, int ng, int n, int lane)
{
    if (lane < S.nb + ng) C.rsj[lane] = S.srev[lane];
    int r = lane / n, c = lane - r * n;
    const int dr = 64 / n, dc = 64 - dr * n;
    for (int e = lane; e < ng * n; e += 64)
    {
        C.G[r * C.SG + c] = WAT(DCt, o_g * n + e);
        r += dr; c += dc;
        if (c >= n) { c -= n; r++; }
    }
}

/* sums over the rows coupled to slack q, WITHOUT row `row` */
__device__ static inline void wpi_excl(const WpiCon &C, int nbg, int row, int q, const double *al_, const double *au_, double &sl, double &su)
{
    sl = 0.0; su = 0.0;
    if (C.scnt[q] < 2) return; /* the usual case: one row per slack */
    for (int k = 0; k < nbg; k++)
        if (k != row && C.rsj[k] == q) { sl += al_[k]; su += au_[k]; }
}

/* entry idx of the constraint row `row` (box row: unit vector of its variable; general row: row of G) */
__device__ static inline double wpi_arow(const WpiCon &C, GQP_STAGE_REF S, int row, int idx)
{
    if (row >= S.nb) return C.G[(row - S.nb) * C.SG + idx];
    return popc64(S.bmask & (((uint64_t) 1 << idx) - 1)) == row && ((S.bmask >> idx) & 1) ? 1.0 : 0.0;
}

/* one side pair of an inequality row as its owner lane sees it */
struct WpiRow
{
    bool al, au;
    int el, eu, sj;
    double ll, lu, tl, tu; /* lam (0 if inactive), t (1 if inactive) */
};

__device__ static inline WpiRow wpi_row(const GqpDev &D, GQP_STAGE_REF S, const Am128 &am, int inst, int row, bool exists)
{
    const int nbg = S.nb + S.ng;
    WpiRow R;
    R.al = exists && abit(am, row);
    R.au = exists && abit(am, nbg + row);
    R.el = S.o_ct + (exists ? row : 0);
    R.eu = R.el + nbg;
    R.sj = exists ? (int) S.srev[row] : -1;
    R.ll = R.al ? WAT(D.lam, R.el) : 0.0;
    R.lu = R.au ? WAT(D.lam, R.eu) : 0.0;
    R.tl = R.al ? WAT(D.t, R.el) : 1.0;
    R.tu = R.au ? WAT(D.t, R.eu) : 1.0;
    return R;
}

#if defined(GQP_WPI_TIMING)
/* development aid (not in the default build): cycles per phase of instance 0, lane 0 */
static __device__ unsigned long long gqp_wpi_cycles[16];
#endif
#if defined(GQP_WPI_TIMING) && defined(__HIP_DEVICE_COMPILE__)
#define GQP_TICK(slot)                                                             \
    do {                                                                           \
        const unsigned long long now_ = clock64();                                 \
        if (inst == 0 && lane == 0) gqp_wpi_cycles[slot] += now_ - tick_;          \
        tick_ = now_;                                                              \
    } while (0)
#define GQP_TICK_INIT() unsigned long long tick_ = clock64()
#else
#define GQP_TICK(slot) do { } while (0)
#define GQP_TICK_INIT() do { } while (0)
#endif

struct WpiLds2
{
    double *__restrict__ Hp;   /* NP: packed H, later packed L */
    double *__restrict__ Bw;   /* 8 T8 x SX: [B A]' then W (rows >= n stay zero) */
    double *__restrict__ Lx;   /* NX x SX: x-block of the factor of stage k+1, zeros above the diagonal */
    double *__restrict__ lx, *__restrict__ v, *__restrict__ rb, *__restrict__ pin, *__restrict__ w0, *__restrict__ gam;
    double *__restrict__ cb;   /* 2 x 72: published Cholesky column (double-buffered) + rhs entry */
    double *__restrict__ red;
    int SX;
};

__host__ __device__ static inline int wpi2_sx(int NX) { return (8 * ((NX + 7) / 8)) | 1; }
__host__ __device__ static inline size_t wpi2_lds_doubles(int NX, int NU)
{
    const int n = NX + NU, NP = n * (n + 1) / 2, T8 = (n + 7) / 8, SX = wpi2_sx(NX);
    return (size_t) NP + 8 + (size_t) 8 * T8 * SX + (size_t) NX * SX + 6 * 64 + 2 * 72 + 64 + 8;
}

__device__ static inline WpiLds2 wpi2_carve(double *sm, int NX, int NU)
{
    const int n = NX + NU, NP = n * (n + 1) / 2, T8 = (n + 7) / 8;
    WpiLds2 L;
    L.SX = wpi2_sx(NX);
    double *p = sm;
    L.Hp = p; p += NP + 8;
    L.Bw = p; p += 8 * T8 * L.SX;
    L.Lx = p; p += NX * L.SX;
    L.lx = p; p += 64; L.v = p; p += 64; L.rb = p; p += 64; L.pin = p; p += 64; L.w0 = p; p += 64; L.gam = p; p += 64;
    L.cb = p; p += 2 * 72;
    L.red = p; p += 64;
    return L;
}

/* CNX, CNU != 0: the state / input dimensions as COMPILE-TIME constants (the BASELINE shapes): trip counts, row strides
 * and the packed-triangle arithmetic fold; 0 = run-time dims, any shape */
template <int T8, bool GEN, int CNX = 0, int CNU = 0>
/* T8 <= 4: keep two waves per SIMD (<= 256 VGPRs; the GEN variant sits right at the line) */
__global__ void __launch_bounds__(64) GQP_WAVES_PER_EU(T8 <= 4 ? 2 : 1) kw_factor(GqpDev D, GqpOpts O, int redo)
{
    GQP_DYN_SHARED(smem);
    const int NX = CNX ? CNX : D.NX, NU = CNX ? CNU : D.NU, n = NX + NU, NP = n * (n + 1) / 2;
    const int inst = blockIdx.x, lane = threadIdx.x;
    if (inst >= D.B) return;
    if (D.status[inst] != GQP_RUNNING) return;
    const WpiLds2 L = wpi2_carve(smem, NX, NU);
    const WpiCon C = wpi_con_carve(smem + wpi2_lds_doubles(NX, NU), n, GEN ? D.NG : 0, GEN ? D.NS : 0);
    const int SX = L.SX, TX = (NX + 7) / 8;
    const int lr = lane >> 3, lc = lane & 7;
    const bool mine = lane < n;

    /* zero what is only ever partly overwritten */
    for (int e = lane; e < 8 * T8 * SX; e += 64) L.Bw[e] = 0.0;
    for (int e = lane; e < NX * SX; e += 64) L.Lx[e] = 0.0;
    for (int e = lane; e < 2 * 72; e += 64) L.cb[e] = 0.0;
    L.lx[lane] = 0.0;
    double nrm_g = 0.0, nrm_b = 0.0, nrm_d = 0.0, nrm_m = 0.0, musum = 0.0, obj = 0.0;
    int nact = 0;
    __syncthreads();

    GQP_TICK_INIT();
    for (int k = D.N; k >= 0; k--)
    {
        GQP_STAGE_REF S = D.st[k];
        const uint64_t imask = S.bmask & ~S.emask;
        const Am128 am = wpi_am(D, inst, k);
        const int nbg = S.nb + (GEN ? S.ng : 0);
        const bool fixed = mine && ((S.emask >> lane) & 1);

        /* ---- coalesced loads into LDS ---- */
        for (int p = lane; p < NP; p += 64) L.Hp[p] = WAT(D.RSQ, k * NP + p);
        {
            int r = lane / NX, c = lane - r * NX;
            const int dr = 64 / NX, dc = 64 - dr * NX;
            for (int e = lane; e < n * NX; e += 64)
            {
                L.Bw[r * SX + c] = WAT(D.BAt, k * n * NX + e);
                r += dr; c += dc;
                if (c >= NX) { c -= NX; r++; }
            }
        }
        double vj = 0.0, gj = 0.0, pik = 0.0;
        if (mine) { vj = WAT(D.ux, k * n + lane); gj = WAT(D.rq, k * n + lane); L.v[lane] = vj; }
        if (lane < NX)
        {
            L.rb[lane] = WAT(D.bvec, k * NX + lane) - WAT(D.ux, (k + 1) * n + NU + lane);
            L.pin[lane] = WAT(D.pi, (k + 1) * NX + lane);
        }
        if (mine && lane >= NU) pik = WAT(D.pi, k * NX + lane - NU);
        const bool has = mine && ((imask >> lane) & 1);
        const int ib = has ? popc64(S.bmask & (((uint64_t) 1 << lane) - 1)) : 0;
        const bool al = has && abit(am, ib), au = has && abit(am, nbg + ib);
        const int el = S.o_ct + ib, eu = el + nbg;
        const double ll = al ? WAT(D.lam, el) : 0.0, lu = au ? WAT(D.lam, eu) : 0.0;
        const double ttl = al ? WAT(D.t, el) : 1.0, ttu = au ? WAT(D.t, eu) : 1.0;
        const double lbv = al ? WAT(D.dvec, el) : 0.0, ubv = au ? WAT(D.dvec, eu) : 0.0;
        /* GEN: soft box row, general row g = lane, slack q = lane */
        const int bsj = (GEN && has) ? (int) S.srev[ib] : -1;
        double bssl = 0.0, bssu = 0.0;
        const bool isg = GEN && lane < S.ng, iss = GEN && lane < S.ns;
        WpiRow Rg;
        double gdl = 0.0, gdu = 0.0, gssl = 0.0, gssu = 0.0;
        bool sal = false, sau = false;
        int se0 = 0, se1 = 0;
        double sll = 0.0, slu = 0.0, stl = 1.0, stu = 1.0, sZl = 0.0, szl = 0.0, sZu = 0.0, szu = 0.0, ssl = 0.0, ssu = 0.0, sdl = 0.0, sdu = 0.0;
        if (GEN)
        {
            wpi_load_G(C, S, D.DCt, inst, S.o_g, S.ng, n, lane);
            if (bsj >= 0) { bssl = WAT(D.sv, S.o_s + bsj); bssu = WAT(D.sv, S.o_s + S.ns + bsj); }
            Rg = wpi_row(D, S, am, inst, S.nb + lane, isg);
            if (isg)
            {
                gdl = Rg.al ? WAT(D.dvec, Rg.el) : 0.0; gdu = Rg.au ? WAT(D.dvec, Rg.eu) : 0.0;
                if (Rg.sj >= 0) { gssl = WAT(D.sv, S.o_s + Rg.sj); gssu = WAT(D.sv, S.o_s + S.ns + Rg.sj); }
            }
            if (iss)
            {
                se0 = S.o_ct + 2 * nbg + lane; se1 = se0 + S.ns;
                sal = abit(am, 2 * nbg + lane); sau = abit(am, 2 * nbg + S.ns + lane);
                sll = sal ? WAT(D.lam, se0) : 0.0; slu = sau ? WAT(D.lam, se1) : 0.0;
                stl = sal ? WAT(D.t, se0) : 1.0; stu = sau ? WAT(D.t, se1) : 1.0;
                sdl = sal ? WAT(D.dvec, se0) : 0.0; sdu = sau ? WAT(D.dvec, se1) : 0.0;
                sZl = WAT(D.Zz, (S.o_s + lane) * 2); szl = WAT(D.Zz, (S.o_s + lane) * 2 + 1);
                sZu = WAT(D.Zz, (S.o_s + S.ns + lane) * 2); szu = WAT(D.Zz, (S.o_s + S.ns + lane) * 2 + 1);
                ssl = WAT(D.sv, S.o_s + lane); ssu = WAT(D.sv, S.o_s + S.ns + lane);
            }
        }
        __syncthreads();
        GQP_TICK(0);
        /* touch the next stage's matrix blocks (one 8-byte load per 64-byte line): by the time the stage loop
         * comes round, they wait in L2 instead of HBM.  The values are only looked at after the Cholesky.
         * Only from T8 = 3 on: the smaller kernels sit at 128 VGPRs (4 waves per SIMD) and the handful of
         * registers this costs would take one wave away (measured: nx=12 nu=3 lost 20 %). */
        double pf0 = 0.0, pf1 = 0.0, pf2 = 0.0, pf3 = 0.0;
        if (T8 >= 3 && k > 0)
        {
            const int e8 = lane * 8, nb_ = n * NX;
            if (e8 < NP) pf0 = WAT(D.RSQ, (k - 1) * NP + e8);
            if (e8 < nb_) pf1 = WAT(D.BAt, (k - 1) * nb_ + e8);
            if (e8 + 512 < nb_) pf2 = WAT(D.BAt, (k - 1) * nb_ + e8 + 512);
            if (e8 + 1024 < nb_) pf3 = WAT(D.BAt, (k - 1) * nb_ + e8 + 1024);
            if (e8 + 512 < NP) pf0 += WAT(D.RSQ, (k - 1) * NP + e8 + 512);
        }

        GQP_TICK(6);
        /* ---- vector part, lane = variable: rb, [B A] pi+, H v, box row ---- */
        double gt = 0.0, gadd = 0.0, gam = 0.0;
        if (lane < NX)
        {
            double a = L.rb[lane];
            GQP_DOT_UNROLL
            for (int r = 0; r < n; r++) a += L.Bw[r * SX + lane] * L.v[r];
            nacc(nrm_b, a);
            WAT(D.rb, k * NX + lane) = a;
            L.rb[lane] = a; /* own slot: nobody else reads rb before the next barrier */
        }
        if (mine)
        {
            double a = 0.0;
            GQP_DOT_UNROLL
            for (int c = 0; c < NX; c++) a += L.Bw[lane * SX + c] * L.pin[c];
            double hv = 0.0;
            GQP_DOT_UNROLL
            for (int c = 0; c <= lane; c++) hv += L.Hp[PK(lane, c)] * L.v[c];
            GQP_DOT_UNROLL
            for (int c = lane + 1; c < n; c++) hv += L.Hp[PK(c, lane)] * L.v[c];
            obj += (0.5 * hv + gj) * vj;
            gt = a + hv + gj - pik;
        }
        double bGl = 0.0, bGu = 0.0, bRl = 0.0, bRu = 0.0; /* Gamma, rho of this lane's box row */
        if (has)
        {
            const double rdl = al ? vj + bssl - lbv - ttl : 0.0, rdu = au ? ubv - vj + bssu - ttu : 0.0;
            const double rml = al ? ll * ttl - O.tau_min : 0.0, rmu = au ? lu * ttu - O.tau_min : 0.0;
            nacc(nrm_d, rdl); nacc(nrm_d, rdu); nacc(nrm_m, rml); nacc(nrm_m, rmu);
            musum += ll * ttl + lu * ttu;
            nact += (int) al + (int) au;
            gt -= ll - lu;
            const double itl = frcp(ttl), itu = frcp(ttu);
            bGl = ll * itl; bGu = lu * itu;
            bRl = (rml + ll * rdl) * itl; bRu = (rmu + lu * rdu) * itu;
            gam = bGl + bGu;
            gadd = bRl - bRu; /* hard row; a soft row is completed below */
            WAT(D.rd, el) = rdl;
            WAT(D.rd, eu) = rdu;
            if (GEN && bsj >= 0) { C.rGl[ib] = bGl; C.rGu[ib] = bGu; C.rRl[ib] = bRl; C.rRu[ib] = bRu; C.rLl[ib] = ll; C.rLu[ib] = lu; }
        }
        double gGl = 0.0, gGu = 0.0, gRl = 0.0, gRu = 0.0; /* general row of this lane */
        double sGl = 0.0, sGu = 0.0, sPl = 0.0, sPu = 0.0; /* slack-bound rows of this lane's slack */
        if (GEN)
        {
            if (isg)
            {
                double c = 0.0;
                GQP_DOT_UNROLL
                for (int r = 0; r < n; r++) c += C.G[lane * C.SG + r] * L.v[r];
                const double rdl = Rg.al ? c + gssl - gdl - Rg.tl : 0.0, rdu = Rg.au ? gdu - c + gssu - Rg.tu : 0.0;
                const double rml = Rg.al ? Rg.ll * Rg.tl - O.tau_min : 0.0, rmu = Rg.au ? Rg.lu * Rg.tu - O.tau_min : 0.0;
                nacc(nrm_d, rdl); nacc(nrm_d, rdu); nacc(nrm_m, rml); nacc(nrm_m, rmu);
                musum += Rg.ll * Rg.tl + Rg.lu * Rg.tu;
                nact += (int) Rg.al + (int) Rg.au;
                const double itl = frcp(Rg.tl), itu = frcp(Rg.tu);
                gGl = Rg.ll * itl; gGu = Rg.lu * itu;
                gRl = (rml + Rg.ll * rdl) * itl; gRu = (rmu + Rg.lu * rdu) * itu;
                WAT(D.rd, Rg.el) = rdl;
                WAT(D.rd, Rg.eu) = rdu;
                const int row = S.nb + lane;
                C.rGl[row] = gGl; C.rGu[row] = gGu; C.rRl[row] = gRl; C.rRu[row] = gRu; C.rLl[row] = Rg.ll; C.rLu[row] = Rg.lu;
            }
            if (iss)
            {
                obj += (0.5 * sZl * ssl + szl) * ssl + (0.5 * sZu * ssu + szu) * ssu;
                const double rdl = sal ? ssl - sdl - stl : 0.0, rdu = sau ? ssu - sdu - stu : 0.0;
                const double rml = sal ? sll * stl - O.tau_min : 0.0, rmu = sau ? slu * stu - O.tau_min : 0.0;
                nacc(nrm_d, rdl); nacc(nrm_d, rdu); nacc(nrm_m, rml); nacc(nrm_m, rmu);
                musum += sll * stl + slu * stu;
                nact += (int) sal + (int) sau;
                const double itl = frcp(stl), itu = frcp(stu);
                sGl = sll * itl; sGu = slu * itu;
                sPl = (rml + sll * rdl) * itl; sPu = (rmu + slu * rdu) * itu;
                WAT(D.rd, se0) = rdl;
                WAT(D.rd, se1) = rdu;
            }
            __syncthreads(); /* rows published */
            if (iss)
            {
                /* sums over the rows coupled to this slack */
                double Dl = sZl + sGl, Du = sZu + sGu, Pl = sPl, Pu = sPu, Rl = sZl * ssl + szl - sll, Ru = sZu * ssu + szu - slu;
                int cnt = 0;
                for (int row = 0; row < nbg; row++)
                    if (C.rsj[row] == lane)
                    {
                        Dl += C.rGl[row]; Du += C.rGu[row];
                        Pl += C.rRl[row]; Pu += C.rRu[row];
                        Rl -= C.rLl[row]; Ru -= C.rLu[row];
                        cnt++;
                    }
                C.scnt[lane] = cnt;
                nacc(nrm_g, Rl); nacc(nrm_g, Ru);
                WAT(D.rgs, S.o_s + lane) = Rl; WAT(D.rgs, S.o_s + S.ns + lane) = Ru;
                C.sEl[lane] = sZl + sGl; C.sEu[lane] = sZu + sGu; /* what D leaves behind without the rows */
                C.sXl[lane] = Rl + sPl; C.sXu[lane] = Ru + sPu;   /* what r~ is without the rows */
                Rl += Pl; Ru += Pu; /* r~ */
                WAT(D.sD, S.o_s + lane) = Dl; WAT(D.sD, S.o_s + S.ns + lane) = Du;
                WAT(D.sR, S.o_s + lane) = Rl; WAT(D.sR, S.o_s + S.ns + lane) = Ru;
                C.sIl[lane] = Dl != 0.0 ? frcp(Dl) : 0.0; C.sIu[lane] = Du != 0.0 ? frcp(Du) : 0.0;
            }
            if (mine)
            {
                /* stationarity: general rows */
                double a = 0.0;
                for (int g = 0; g < S.ng; g++) a += C.G[g * C.SG + lane] * (C.rLl[S.nb + g] - C.rLu[S.nb + g]);
                gt -= a;
            }
            __syncthreads(); /* slack sums published */
            if (has && bsj >= 0)
            {
                double El, Eu, Xl, Xu;
                wpi_excl(C, nbg, ib, bsj, C.rGl, C.rGu, El, Eu);
                wpi_excl(C, nbg, ib, bsj, C.rRl, C.rRu, Xl, Xu);
                El += C.sEl[bsj]; Eu += C.sEu[bsj]; Xl += C.sXl[bsj]; Xu += C.sXu[bsj];
                gam = bGl * El * C.sIl[bsj] + bGu * Eu * C.sIu[bsj];
                gadd = (bRl * El - bGl * Xl) * C.sIl[bsj] - (bRu * Eu - bGu * Xu) * C.sIu[bsj];
            }
            if (isg)
            {
                double nu = gRl - gRu, gm = gGl + gGu;
                if (Rg.sj >= 0)
                {
                    const int q = Rg.sj;
                    double El, Eu, Xl, Xu;
                    wpi_excl(C, nbg, S.nb + lane, q, C.rGl, C.rGu, El, Eu);
                    wpi_excl(C, nbg, S.nb + lane, q, C.rRl, C.rRu, Xl, Xu);
                    El += C.sEl[q]; Eu += C.sEu[q]; Xl += C.sXl[q]; Xu += C.sXu[q];
                    gm = gGl * El * C.sIl[q] + gGu * Eu * C.sIu[q];
                    nu = (gRl * El - gGl * Xl) * C.sIl[q] - (gRu * Eu - gGu * Xu) * C.sIu[q];
                }
                C.nuG[lane] = nu;
                C.gmG[lane] = gm;
            }
        }
        if (fixed) gt = 0.0;
        if (mine) { nacc(nrm_g, gt); WAT(D.rg, k * n + lane) = gt; }
        L.gam[lane] = gam;
        GQP_TICK(5);
        /* ---- W tiles: W = [B A]' Lx+ (Lx+ has explicit zeros above its diagonal) ---- */
        double Wt[T8][T8];
#pragma unroll
        for (int a = 0; a < T8; a++)
#pragma unroll
            for (int b = 0; b < T8; b++) Wt[a][b] = 0.0;
#pragma unroll (T8 == 4 && !GEN ? 4 : 2)
        for (int q = 0; q < NX; q++)
        {
            double bb[T8], xx[T8];
#pragma unroll
            for (int a = 0; a < T8; a++) bb[a] = L.Bw[(lr + 8 * a) * SX + q];
#pragma unroll
            for (int b = 0; b < T8; b++) xx[b] = b < TX ? L.Lx[q * SX + lc + 8 * b] : 0.0;
#pragma unroll
            for (int a = 0; a < T8; a++)
#pragma unroll
                for (int b = 0; b < T8; b++)
                    if (b < TX) Wt[a][b] += bb[a] * xx[b];
        }
        __syncthreads(); /* everybody is done with [B A]': the buffer becomes W */
        GQP_TICK(1);
#pragma unroll
        for (int a = 0; a < T8; a++)
#pragma unroll
            for (int b = 0; b < T8; b++)
            {
                const int r = lr + 8 * a, c = lc + 8 * b;
                if (b < TX && r < n && c < NX) L.Bw[r * SX + c] = Wt[a][b];
            }
        /* ---- tiles of H (only now: the W tiles are dead, the two never share registers) ---- */
        double Mt[T8][T8];
#pragma unroll
        for (int a = 0; a < T8; a++)
#pragma unroll
            for (int b = 0; b <= a; b++)
            {
                const int r = lr + 8 * a, c = lc + 8 * b;
                Mt[a][b] = (r < n && c <= r) ? L.Hp[PK(r, c)] : (r == c ? 1.0 : 0.0);
            }
        __syncthreads();
        /* w0 = Lx+' rb + lx+ */
        if (lane < NX)
        {
            double a = L.lx[lane];
            GQP_DOT_UNROLL
            for (int q = lane; q < NX; q++) a += L.Lx[q * SX + lane] * L.rb[q];
            L.w0[lane] = a;
        }
        /* ---- M = H + reg + Gamma + W W' in tiles ---- */
#pragma unroll (T8 == 4 && !GEN ? 4 : 2)
        for (int q = 0; q < NX; q++)
        {
            double wr[T8], wc[T8];
#pragma unroll
            for (int a = 0; a < T8; a++) wr[a] = L.Bw[(lr + 8 * a) * SX + q];
#pragma unroll
            for (int b = 0; b < T8; b++) wc[b] = L.Bw[(lc + 8 * b) * SX + q];
#pragma unroll
            for (int a = 0; a < T8; a++)
#pragma unroll
                for (int b = 0; b <= a; b++) Mt[a][b] += wr[a] * wc[b];
        }
        if (GEN)
        {
            if (mine)
                for (int g = 0; g < S.ng; g++) gadd += C.G[g * C.SG + lane] * C.nuG[g];
            /* M += sum_g Gamma_eff a a' in tiles */
            for (int g = 0; g < S.ng; g++)
            {
                const double gm = C.gmG[g];
                double gr[T8], gc[T8];
#pragma unroll
                for (int a = 0; a < T8; a++) gr[a] = (lr + 8 * a < n) ? C.G[g * C.SG + lr + 8 * a] * gm : 0.0;
#pragma unroll
                for (int b = 0; b < T8; b++) gc[b] = (lc + 8 * b < n) ? C.G[g * C.SG + lc + 8 * b] : 0.0;
#pragma unroll
                for (int a = 0; a < T8; a++)
#pragma unroll
                    for (int b = 0; b <= a; b++) Mt[a][b] += gr[a] * gc[b];
            }
            /* rows sharing a slack: cross terms (rare; plain loops) */
            for (int i = 0; i < nbg; i++)
            {
                const int q = C.rsj[i];
                if (q < 0 || C.scnt[q] < 2) continue;
                for (int kk = 0; kk < nbg; kk++)
                {
                    if (kk == i || C.rsj[kk] != q) continue;
                    const double cf = C.rGl[i] * C.rGl[kk] * C.sIl[q] + C.rGu[i] * C.rGu[kk] * C.sIu[q];
#pragma unroll
                    for (int a = 0; a < T8; a++)
#pragma unroll
                        for (int b = 0; b <= a; b++)
                        {
                            const int r = lr + 8 * a, c = lc + 8 * b;
                            if (r < n && c < n) Mt[a][b] -= cf * wpi_arow(C, S, i, r) * wpi_arow(C, S, kk, c);
                        }
                }
            }
        }
#pragma unroll
        for (int a = 0; a < T8; a++)
        {
            const int r = lr + 8 * a;
            if (lr == lc && r < n) Mt[a][a] += O.reg_prim + L.gam[r];
        }
        if (S.emask)
        {
#pragma unroll
            for (int a = 0; a < T8; a++)
#pragma unroll
                for (int b = 0; b <= a; b++)
                {
                    const int r = lr + 8 * a, c = lc + 8 * b;
                    if (((S.emask >> r) & 1) || ((S.emask >> c) & 1)) Mt[a][b] = r == c ? 1.0 : 0.0;
                }
        }
        __syncthreads(); /* w0 published */
        GQP_TICK(2);
        double m = 0.0;
        if (mine)
        {
            double a = 0.0;
            GQP_DOT_UNROLL
            for (int c = 0; c < NX; c++) a += L.Bw[lane * SX + c] * L.w0[c];
            m = fixed ? 0.0 : gt + gadd + a;
        }

        /* ---- Cholesky, column by column; rhs entry m of variable `lane` rides along ---- */
        int pb = 0;
#pragma unroll
        for (int jb = 0; jb < T8; jb++)
        {
            for (int jj = 0; jj < 8; jj++)
            {
                const int j = 8 * jb + jj;
                if (j >= n) break;
                double *cb = L.cb + pb * 72;
                pb ^= 1;
                if (lc == jj)
                {
#pragma unroll
                    for (int a = jb; a < T8; a++) cb[lr + 8 * a] = Mt[a][jb];
                }
                if (lane == j) cb[64] = m;
                __syncthreads();
                /* everything this lane needs from the published column in ONE round trip */
                const double d = cb[j], mj = cb[64], cmine = cb[lane];
                double rL[T8], cL[T8];
#pragma unroll
                for (int a = jb; a < T8; a++) rL[a] = cb[lr + 8 * a];
#pragma unroll
                for (int b = jb; b < T8; b++) cL[b] = cb[lc + 8 * b];
                const bool pos = d > 0.0;
                const double inv0 = frsqrt(pos ? d : 1.0);
                const double inv = pos ? inv0 : 0.0;
                const double lj = mj * inv;
#pragma unroll
                for (int a = jb; a < T8; a++) rL[a] *= inv;
#pragma unroll
                for (int b = jb; b < T8; b++) cL[b] *= inv;
                const double lmine = cmine * inv; /* L[lane][j] for the rhs update */
                /* the column itself */
                if (lc == jj)
                {
#pragma unroll
                    for (int a = jb; a < T8; a++)
                    {
                        const int r = lr + 8 * a;
                        Mt[a][jb] = r > j ? rL[a] : (r == j ? (pos ? d * inv : 0.0) : Mt[a][jb]);
                    }
                }
                /* trailing update: operands of finished rows / columns are zeroed instead of predicating */
#pragma unroll
                for (int a = jb; a < T8; a++) rL[a] = (lr + 8 * a > j) ? rL[a] : 0.0;
#pragma unroll
                for (int b = jb; b < T8; b++) cL[b] = (lc + 8 * b > j) ? cL[b] : 0.0;
#pragma unroll
                for (int a = jb; a < T8; a++)
#pragma unroll
                    for (int b = jb; b <= a; b++) Mt[a][b] -= rL[a] * cL[b];
                if (lane == j) m = lj;
                else if (lane > j) m -= lmine * lj;
            }
        }
        if (T8 >= 3 && pf0 + pf1 + pf2 + pf3 == 1.2345678e-300) L.red[lane] = pf0; /* keeps the touch loads alive; never true in practice, harmless if it were */
        __syncthreads(); /* Hp (packed H) is dead: it receives the packed factor */
        GQP_TICK(3);
#pragma unroll
        for (int a = 0; a < T8; a++)
#pragma unroll
            for (int b = 0; b < T8; b++)
            {
                const int r = lr + 8 * a, c = lc + 8 * b;
                if (b <= a && r < n && c <= r) L.Hp[PK(r, c)] = Mt[a][b];
                /* x-block for the next (earlier) stage, explicit zeros above the diagonal */
                if (r >= NU && r < n && c >= NU && c < n) L.Lx[(r - NU) * SX + c - NU] = (b <= a && c <= r) ? Mt[a][b] : 0.0;
            }
        if (mine)
        {
            WAT(D.lf, k * n + lane) = m;
            if (lane >= NU) L.lx[lane - NU] = m;
        }
        __syncthreads();
        for (int p = lane; p < NP; p += 64) WAT(D.Lf, k * NP + p) = L.Hp[p];
        __syncthreads(); /* Hp is refilled by the next stage */
        GQP_TICK(4);
    }

    nrm_g = wpi_max(nrm_g, L.red, lane);
    nrm_b = wpi_max(nrm_b, L.red, lane);
    nrm_d = wpi_max(nrm_d, L.red, lane);
    nrm_m = wpi_max(nrm_m, L.red, lane);
    musum = wpi_sum(musum, L.red, lane);
    obj = wpi_sum(obj, L.red, lane);
    const double nact_d = wpi_sum((double) nact, L.red, lane);
    if (lane == 0)
    {
        const int Bp = D.Bp;
        const double mu = nact_d > 0.0 ? musum / nact_d : 0.0;
        D.mu[inst] = mu;
        D.obj[inst] = obj;
        D.res[0 * Bp + inst] = nrm_g; D.res[1 * Bp + inst] = nrm_b; D.res[2 * Bp + inst] = nrm_d; D.res[3 * Bp + inst] = nrm_m;
        const int it = D.iter[inst];
        if (inst < D.stat_inst && it < D.stat_rows)
        {
            double *st = D.stat + (size_t) it * GQP_STAT_COLS * D.stat_inst + inst;
            st[6 * D.stat_inst] = mu;
            st[7 * D.stat_inst] = nrm_g; st[8 * D.stat_inst] = nrm_b; st[9 * D.stat_inst] = nrm_d; st[10 * D.stat_inst] = nrm_m;
            st[12 * D.stat_inst] = obj;
        }
        int status = GQP_RUNNING;
        const bool bad = nrm_g != nrm_g || nrm_b != nrm_b || nrm_d != nrm_d || nrm_m != nrm_m || mu != mu;
        if (bad) status = 1;
        else if (nrm_g <= O.tol_stat && nrm_b <= O.tol_eq && nrm_d <= O.tol_ineq && nrm_m <= O.tol_comp) status = 0;
        else if (it >= O.iter_max) status = 2;
        else if (dabs(D.alpha[inst]) <= O.alpha_min) status = 3;
        if (status != GQP_RUNNING)
        {
            D.status[inst] = status;
            atomicSub(D.n_active, 1);
        }
    }
}

/* ------------------------------------------------- rhs-only backward and forward, packed factor */

/*
 * The corrector's rhs-only backward sweep does NOT need l = L^{-1} m in full.  With L = [Lr 0; Ls Lx],
 * m = [m_u; m_x]:  l_u = Lr^{-1} m_u  and  p = Lx l_x = m_x - Ls l_u,  and everything downstream uses l_x only
 * through p (P rb + p = Lx (Lx' rb) + p in the next stage, dpi = Lx (Lx' dx) + p and dv_u = -Lr^{-T}(l_u + Ls' dx)
 * in the forward sweep).  So the sequential part of a stage is NU substitution steps instead of NU + NX;
 * the rest is matrix-vector work that the lanes share.  lf then holds [l_u; p] ("p-form"); only stage 0 of
 * the forward sweep, where the states are free, recovers l_x = Lx^{-1} p.  The factor sweep keeps writing the
 * plain l (its Cholesky carries the rhs along for free, and the Riccati getters read it).
 * The factor stays PACKED in LDS (coalesced copy from HBM, index arithmetic instead of a re-layout).
 */
struct WpiLds3
{
    double *__restrict__ Lp;  /* 2 x NPa: packed factor of this stage / of the stage handled before */
    double *__restrict__ B;   /* n x SXb: [B A]' */
    double *__restrict__ rb, *__restrict__ w0, *__restrict__ y, *__restrict__ pn, *__restrict__ dv, *__restrict__ dx, *__restrict__ bc, *__restrict__ red;
    int NPa, SXb;
};

/* nbuf: packed-factor buffers -- two for the backward sweep (this stage's and the previous one's), one for the
 * forward sweep (less LDS per wave there = more waves per CU) */
__host__ __device__ static inline size_t wpi3_lds_doubles(int NX, int NU, int nbuf = 2)
{
    const int n = NX + NU, NPa = n * (n + 1) / 2 + 8, SXb = NX | 1, nv = (n + 1) & ~1;
    return (size_t) nbuf * NPa + (size_t) n * SXb + 6 * (size_t) nv + 8 + 64 + 8;
}

__device__ static inline WpiLds3 wpi3_carve(double *sm, int NX, int NU, int nbuf = 2)
{
    const int n = NX + NU;
    WpiLds3 L;
    L.NPa = n * (n + 1) / 2 + 8;
    L.SXb = NX | 1;
    double *p = sm;
    L.Lp = p; p += nbuf * L.NPa;
    L.B = p; p += n * L.SXb;
    const int nv = (n + 1) & ~1; /* vectors over the variables: n entries, not a full wave's 64 */
    L.rb = p; p += nv; L.w0 = p; p += nv; L.y = p; p += nv; L.pn = p; p += nv;
    L.dv = p; p += nv; L.dx = p; p += nv; L.bc = p; p += 8; L.red = p; p += 64;
    return L;
}

/* touch the matrix blocks of stage `ks` (one 8-byte load per 64-byte line) so that they wait in L2 when the stage
 * loop comes round; the returned sum is only looked at much later (keeps the loads alive, costs no wait) */
__device__ static inline double wpi_touch(const GArr &M1, int NP, const GArr &BAt, int nb_, int inst, int ks, int lane)
{
    const int e8 = lane * 8;
    double a = 0.0, b = 0.0, c = 0.0, d = 0.0;
    if (e8 < NP) a = WAT(M1, ks * NP + e8);
    if (e8 + 512 < NP) b = WAT(M1, ks * NP + e8 + 512);
    if (e8 < nb_) c = WAT(BAt, ks * nb_ + e8);
    if (e8 + 512 < nb_) d = WAT(BAt, ks * nb_ + e8 + 512);
    if (e8 + 1024 < nb_) c += WAT(BAt, ks * nb_ + e8 + 1024);
    return (a + b) + (c + d);
}

/* coalesced [B A]' -> LDS rows of odd stride */
__device__ static inline void wpi_load_B(double *__restrict__ dst, int SXb, const GArr &BAt, int inst, int k, int n, int NX, int lane)
{
    int r = lane / NX, c = lane - r * NX;
    const int dr = 64 / NX, dc = 64 - dr * NX;
    for (int e = lane; e < n * NX; e += 64)
    {
        dst[r * SXb + c] = WAT(BAt, k * n * NX + e);
        r += dr; c += dc;
        if (c >= NX) { c -= NX; r++; }
    }
}

template <bool GEN, int CNX = 0, int CNU = 0>
__global__ void __launch_bounds__(64) kw_backrhs(GqpDev D, GqpOpts O, int redo)
{
    GQP_DYN_SHARED(smem);
    const int NX = CNX ? CNX : D.NX, NU = CNX ? CNU : D.NU, n = NX + NU, NP = n * (n + 1) / 2;
    const int inst = blockIdx.x, lane = threadIdx.x;
    if (inst >= D.B) return;
    if (D.status[inst] != GQP_RUNNING) return;
    if (redo == 1 && !(D.alpha[inst] < 0.0)) return; /* redo = 2: sensitivity pass (direction only, every instance) */
    const WpiLds3 L = wpi3_carve(smem, NX, NU);
    const WpiCon C = wpi_con_carve(smem + wpi3_lds_doubles(NX, NU), n, GEN ? D.NG : 0, GEN ? D.NS : 0);
    const int SXb = L.SXb;
    const double smu = D.smu[inst];
    const double pscale = redo == 1 ? 0.0 : 1.0;
    const bool mine = lane < n;
    for (int e = lane; e < 2 * L.NPa; e += 64) L.Lp[e] = 0.0;
    if (lane < n) L.pn[lane] = 0.0;
    int cur = 0;
    __syncthreads();

    for (int k = D.N; k >= 0; k--)
    {
        GQP_STAGE_REF S = D.st[k];
        const uint64_t imask = S.bmask & ~S.emask;
        const Am128 am = wpi_am(D, inst, k);
        const int nbg = S.nb + (GEN ? S.ng : 0);
        double *__restrict__ Lc = L.Lp + cur * L.NPa;
        const double *__restrict__ Ln = L.Lp + (cur ^ 1) * L.NPa;
        const bool fixed = mine && ((S.emask >> lane) & 1);

        for (int p = lane; p < NP; p += 64) Lc[p] = WAT(D.Lf, k * NP + p);
        wpi_load_B(L.B, SXb, D.BAt, inst, k, n, NX, lane);
        if (lane < NX) L.rb[lane] = WAT(D.rb, k * NX + lane);
        double m = mine ? WAT(D.rg, k * n + lane) : 0.0;
        const bool has = mine && ((imask >> lane) & 1);
        const int ib = has ? popc64(S.bmask & (((uint64_t) 1 << lane) - 1)) : 0;
        const int bsj = (GEN && has) ? (int) S.srev[ib] : -1;
        double bGl = 0.0, bGu = 0.0, bRl = 0.0, bRu = 0.0;
        if (has)
        {
            const WpiRow R = wpi_row(D, S, am, inst, ib, true);
            const double rdl = R.al ? WAT(D.rd, R.el) : 0.0, rdu = R.au ? WAT(D.rd, R.eu) : 0.0;
            const double rml = R.al ? R.ll * R.tl - O.tau_min + pscale * WAT(D.pcorr, R.el) - smu : 0.0;
            const double rmu = R.au ? R.lu * R.tu - O.tau_min + pscale * WAT(D.pcorr, R.eu) - smu : 0.0;
            const double itl = frcp(R.tl), itu = frcp(R.tu);
            bGl = R.ll * itl; bGu = R.lu * itu;
            bRl = (rml + R.ll * rdl) * itl; bRu = (rmu + R.lu * rdu) * itu;
            if (bsj < 0) m += bRl - bRu;
            else { C.rRl[ib] = bRl; C.rRu[ib] = bRu; C.rGl[ib] = bGl; C.rGu[ib] = bGu; }
        }
        if (GEN)
        {
            wpi_load_G(C, S, D.DCt, inst, S.o_g, S.ng, n, lane);
            const bool isg = lane < S.ng, iss = lane < S.ns;
            double gGl = 0.0, gGu = 0.0, gRl = 0.0, gRu = 0.0;
            int gsj = -1;
            if (isg)
            {
                const WpiRow R = wpi_row(D, S, am, inst, S.nb + lane, true);
                const double rdl = R.al ? WAT(D.rd, R.el) : 0.0, rdu = R.au ? WAT(D.rd, R.eu) : 0.0;
                const double rml = R.al ? R.ll * R.tl - O.tau_min + pscale * WAT(D.pcorr, R.el) - smu : 0.0;
                const double rmu = R.au ? R.lu * R.tu - O.tau_min + pscale * WAT(D.pcorr, R.eu) - smu : 0.0;
                const double itl = frcp(R.tl), itu = frcp(R.tu);
                gGl = R.ll * itl; gGu = R.lu * itu;
                gRl = (rml + R.ll * rdl) * itl; gRu = (rmu + R.lu * rdu) * itu;
                gsj = R.sj;
                C.rRl[S.nb + lane] = gRl; C.rRu[S.nb + lane] = gRu; C.rGl[S.nb + lane] = gGl; C.rGu[S.nb + lane] = gGu;
            }
            double sPl = 0.0, sPu = 0.0, sGl = 0.0, sGu = 0.0, sRgl = 0.0, sRgu = 0.0, sZl_ = 0.0, sZu_ = 0.0, sDl_ = 0.0, sDu_ = 0.0;
            if (iss)
            {
                const int e0 = S.o_ct + 2 * nbg + lane, e1 = e0 + S.ns;
                const bool al = abit(am, 2 * nbg + lane), au = abit(am, 2 * nbg + S.ns + lane);
                const double ll = al ? WAT(D.lam, e0) : 0.0, lu = au ? WAT(D.lam, e1) : 0.0;
                const double tl = al ? WAT(D.t, e0) : 1.0, tu = au ? WAT(D.t, e1) : 1.0;
                const double rdl = al ? WAT(D.rd, e0) : 0.0, rdu = au ? WAT(D.rd, e1) : 0.0;
                const double rml = al ? ll * tl - O.tau_min + pscale * WAT(D.pcorr, e0) - smu : 0.0;
                const double rmu = au ? lu * tu - O.tau_min + pscale * WAT(D.pcorr, e1) - smu : 0.0;
                sPl = (rml + ll * rdl) * frcp(tl); sPu = (rmu + lu * rdu) * frcp(tu);
                sGl = ll * frcp(tl); sGu = lu * frcp(tu);
                /* everything else this lane needs from HBM, issued before the barrier */
                sRgl = WAT(D.rgs, S.o_s + lane); sRgu = WAT(D.rgs, S.o_s + S.ns + lane);
                sZl_ = WAT(D.Zz, (S.o_s + lane) * 2); sZu_ = WAT(D.Zz, (S.o_s + S.ns + lane) * 2);
                sDl_ = WAT(D.sD, S.o_s + lane); sDu_ = WAT(D.sD, S.o_s + S.ns + lane);
            }
            __syncthreads(); /* rho of the soft rows published */
            if (iss)
            {
                double Rl = sRgl + sPl, Ru = sRgu + sPu;
                C.sXl[lane] = Rl; C.sXu[lane] = Ru; /* r~ without the rows */
                C.sEl[lane] = sZl_ + sGl; C.sEu[lane] = sZu_ + sGu;
                int cnt = 0;
                for (int row = 0; row < nbg; row++)
                    if (C.rsj[row] == lane) { Rl += C.rRl[row]; Ru += C.rRu[row]; cnt++; }
                C.scnt[lane] = cnt;
                WAT(D.sR, S.o_s + lane) = Rl; WAT(D.sR, S.o_s + S.ns + lane) = Ru;
                const double Dl = sDl_, Du = sDu_;
                C.sIl[lane] = Dl != 0.0 ? frcp(Dl) : 0.0;
                C.sIu[lane] = Du != 0.0 ? frcp(Du) : 0.0;
            }
            __syncthreads();
            if (has && bsj >= 0)
            {
                double El, Eu, Xl, Xu;
                wpi_excl(C, nbg, ib, bsj, C.rGl, C.rGu, El, Eu);
                wpi_excl(C, nbg, ib, bsj, C.rRl, C.rRu, Xl, Xu);
                El += C.sEl[bsj]; Eu += C.sEu[bsj]; Xl += C.sXl[bsj]; Xu += C.sXu[bsj];
                m += (bRl * El - bGl * Xl) * C.sIl[bsj] - (bRu * Eu - bGu * Xu) * C.sIu[bsj];
            }
            if (isg)
            {
                double nu = gRl - gRu;
                if (gsj >= 0)
                {
                    double El, Eu, Xl, Xu;
                    wpi_excl(C, nbg, S.nb + lane, gsj, C.rGl, C.rGu, El, Eu);
                    wpi_excl(C, nbg, S.nb + lane, gsj, C.rRl, C.rRu, Xl, Xu);
                    El += C.sEl[gsj]; Eu += C.sEu[gsj]; Xl += C.sXl[gsj]; Xu += C.sXu[gsj];
                    nu = (gRl * El - gGl * Xl) * C.sIl[gsj] - (gRu * Eu - gGu * Xu) * C.sIu[gsj];
                }
                C.nuG[lane] = nu;
            }
            __syncthreads();
            if (mine)
                for (int g = 0; g < S.ng; g++) m += C.G[g * C.SG + lane] * C.nuG[g];
        }
        __syncthreads();
        const double pf = k > 0 ? wpi_touch(D.Lf, NP, D.BAt, n * NX, inst, k - 1, lane) : 0.0;
        /* y = Lx+ (Lx+' rb) + p+ with the x-block of the factor handled one stage ago */
        if (lane < NX)
        {
            double a = 0.0;
            GQP_DOT_UNROLL
            for (int q = lane; q < NX; q++) a += Ln[PK(NU + q, NU + lane)] * L.rb[q];
            L.w0[lane] = a;
        }
        __syncthreads();
        if (lane < NX)
        {
            double a = L.pn[lane];
            GQP_DOT_UNROLL
            for (int c = 0; c <= lane; c++) a += Ln[PK(NU + lane, NU + c)] * L.w0[c];
            L.y[lane] = a;
        }
        __syncthreads();
        if (mine)
        {
            double a = 0.0;
            GQP_DOT_UNROLL
            for (int c = 0; c < NX; c++) a += L.B[lane * SXb + c] * L.y[c];
            m = fixed ? 0.0 : m + a;
        }
        /* l_u = Lr^{-1} m_u ; the lanes below keep m_r -= L[r][j] l_j, which leaves p in the state lanes */
        for (int j = 0; j < NU; j++)
        {
            const double lrj = (mine && lane > j) ? Lc[PK(lane, j)] : 0.0;
            if (lane == j)
            {
                const double d = Lc[PK(j, j)];
                m = d != 0.0 ? m * frcp(d) : 0.0;
                L.bc[j & 1] = m;
            }
            __syncthreads();
            m -= lrj * L.bc[j & 1];
        }
        if (mine)
        {
            WAT(D.lf, k * n + lane) = m;
            if (lane >= NU) L.pn[lane - NU] = m;
        }
        if (pf == 1.2345678e-300) L.bc[2] = pf; /* never true in practice: keeps the touch loads alive */
        cur ^= 1;
        __syncthreads();
    }
}

/* forward sweep; PFORM (= CORR): lf holds [l_u; p], otherwise the plain l of the factor sweep */
template <bool CORR, bool GEN, int CNX = 0, int CNU = 0>
__global__ void __launch_bounds__(64) kw_fwd(GqpDev D, GqpOpts O, int redo)
{
    GQP_DYN_SHARED(smem);
    constexpr bool PFORM = CORR;
    const int NX = CNX ? CNX : D.NX, NU = CNX ? CNU : D.NU, n = NX + NU, NP = n * (n + 1) / 2;
    const int inst = blockIdx.x, lane = threadIdx.x;
    if (inst >= D.B) return;
    if (D.status[inst] != GQP_RUNNING) return;
    if (redo == 1 && !(D.alpha[inst] < 0.0)) return; /* redo = 2: sensitivity pass (direction only, every instance) */
    const WpiLds3 L = wpi3_carve(smem, NX, NU, 1);
    const WpiCon C = wpi_con_carve(smem + wpi3_lds_doubles(NX, NU, 1), n, GEN ? D.NG : 0, GEN ? D.NS : 0);
    const int SXb = L.SXb;
    double *__restrict__ Lc = L.Lp;
    const double smu = CORR ? D.smu[inst] : 0.0;
    const double pscale = (CORR && redo != 1) ? 1.0 : 0.0;
    const bool mine = lane < n;
    double alpha = 1.0, S0 = 0.0, S1 = 0.0, S2 = 0.0;
    int nact = 0;
    if (lane < n) L.dx[lane] = 0.0;
    __syncthreads();

    for (int k = 0; k <= D.N; k++)
    {
        GQP_STAGE_REF S = D.st[k];
        const uint64_t imask = S.bmask & ~S.emask;
        const Am128 am = wpi_am(D, inst, k);
        const int nbg = S.nb + (GEN ? S.ng : 0);

        for (int p = lane; p < NP; p += 64) Lc[p] = WAT(D.Lf, k * NP + p);
        if (GEN) wpi_load_G(C, S, D.DCt, inst, S.o_g, S.ng, n, lane);
        wpi_load_B(L.B, SXb, D.BAt, inst, k, n, NX, lane);
        double lv = mine ? WAT(D.lf, k * n + lane) : 0.0; /* l_u / l_x or p */
        const double rbv = lane < NX ? WAT(D.rb, k * NX + lane) : 0.0;
        const bool has = mine && ((imask >> lane) & 1);
        const int ib = has ? popc64(S.bmask & (((uint64_t) 1 << lane) - 1)) : 0;
        const bool al = has && abit(am, ib), au = has && abit(am, nbg + ib);
        const int el = S.o_ct + ib, eu = el + nbg;
        const double ll = al ? WAT(D.lam, el) : 0.0, lu = au ? WAT(D.lam, eu) : 0.0;
        const double ttl = al ? WAT(D.t, el) : 1.0, ttu = au ? WAT(D.t, eu) : 1.0;
        const double rdl = al ? WAT(D.rd, el) : 0.0, rdu = au ? WAT(D.rd, eu) : 0.0;
        const double pl = (CORR && al) ? WAT(D.pcorr, el) : 0.0, pu = (CORR && au) ? WAT(D.pcorr, eu) : 0.0;
        /* GEN: what the slack lane and the general-row lane need from HBM, issued with the other loads */
        double fDl = 0.0, fDu = 0.0, fRl = 0.0, fRu = 0.0, fZl = 0.0, fZu = 0.0, fsll = 0.0, fslu = 0.0, fstl = 1.0, fstu = 1.0,
               fspl = 0.0, fspu = 0.0, fsrdl = 0.0, fsrdu = 0.0, fgrdl = 0.0, fgrdu = 0.0, fgpl = 0.0, fgpu = 0.0;
        bool fsal = false, fsau = false;
        WpiRow Rg;
        if (GEN)
        {
            Rg = wpi_row(D, S, am, inst, S.nb + lane, lane < S.ng);
            if (lane < S.ng)
            {
                fgrdl = Rg.al ? WAT(D.rd, Rg.el) : 0.0; fgrdu = Rg.au ? WAT(D.rd, Rg.eu) : 0.0;
                fgpl = (CORR && Rg.al) ? WAT(D.pcorr, Rg.el) : 0.0; fgpu = (CORR && Rg.au) ? WAT(D.pcorr, Rg.eu) : 0.0;
            }
            if (lane < S.ns)
            {
                const int e0 = S.o_ct + 2 * nbg + lane, e1 = e0 + S.ns;
                fsal = abit(am, 2 * nbg + lane); fsau = abit(am, 2 * nbg + S.ns + lane);
                fDl = WAT(D.sD, S.o_s + lane); fDu = WAT(D.sD, S.o_s + S.ns + lane);
                fRl = WAT(D.sR, S.o_s + lane); fRu = WAT(D.sR, S.o_s + S.ns + lane);
                fZl = WAT(D.Zz, (S.o_s + lane) * 2); fZu = WAT(D.Zz, (S.o_s + S.ns + lane) * 2);
                fsll = fsal ? WAT(D.lam, e0) : 0.0; fslu = fsau ? WAT(D.lam, e1) : 0.0;
                fstl = fsal ? WAT(D.t, e0) : 1.0; fstu = fsau ? WAT(D.t, e1) : 1.0;
                fspl = (CORR && fsal) ? WAT(D.pcorr, e0) : 0.0; fspu = (CORR && fsau) ? WAT(D.pcorr, e1) : 0.0;
                fsrdl = fsal ? WAT(D.rd, e0) : 0.0; fsrdu = fsau ? WAT(D.rd, e1) : 0.0;
            }
        }
        __syncthreads();
        const double pf = k < D.N ? wpi_touch(D.Lf, NP, D.BAt, n * NX, inst, k + 1, lane) : 0.0;

        if (PFORM && k == 0)
        {
            /* the states of stage 0 are free: recover l_x = Lx^{-1} p */
            for (int j = NU; j < n; j++)
            {
                const double lrj = (mine && lane > j) ? Lc[PK(lane, j)] : 0.0;
                if (lane == j)
                {
                    const double d = Lc[PK(j, j)];
                    lv = d != 0.0 ? lv * frcp(d) : 0.0;
                    L.bc[j & 1] = lv;
                }
                __syncthreads();
                lv -= lrj * L.bc[j & 1];
            }
        }
        /* dpi_k = Lx (Lx' dx + l_x)  resp.  Lx (Lx' dx) + p */
        if (CORR && k > 0)
        {
            if (lane < NX)
            {
                double a = 0.0;
                GQP_DOT_UNROLL
                for (int q = lane; q < NX; q++) a += Lc[PK(NU + q, NU + lane)] * L.dx[q];
                L.w0[lane] = a;
            }
            __syncthreads();
            if (lane < NX)
            {
                double a = 0.0;
                GQP_DOT_UNROLL
                for (int c = 0; c <= lane; c++) a += Lc[PK(NU + lane, NU + c)] * L.w0[c];
                L.y[lane] = a; /* + p of this lane's state, added by its owner below */
            }
            __syncthreads();
            if (mine && lane >= NU) WAT(D.dpi, k * NX + lane - NU) = L.y[lane - NU] + lv;
        }
        /* L' dv = -l for the free block: everything at k = 0, the inputs otherwise */
        const int top = k == 0 ? n : NU;
        if (mine && lane >= top) L.dv[lane] = L.dx[lane - NU];
        __syncthreads();
        double acc = 0.0;
        if (lane < top)
        {
            acc = -lv;
            GQP_DOT_UNROLL
            for (int p = top; p < n; p++) acc -= Lc[PK(p, lane)] * L.dv[p];
        }
        for (int r = top - 1; r >= 0; r--)
        {
            const double lrl = lane < r ? Lc[PK(r, lane)] : 0.0;
            if (lane == r)
            {
                const double d = Lc[PK(r, r)];
                L.dv[r] = d != 0.0 ? acc * frcp(d) : 0.0;
            }
            __syncthreads();
            acc -= lrl * L.dv[r];
        }
        __syncthreads();
        const double dvj = mine ? L.dv[lane] : 0.0;
        if (CORR && mine) WAT(D.dux, k * n + lane) = dvj;
        double dxn = 0.0;
        if (lane < NX)
        {
            dxn = rbv;
            GQP_DOT_UNROLL
            for (int r = 0; r < n; r++) dxn += L.B[r * SXb + lane] * L.dv[r];
        }
        /* ---- inequality rows: dc -> slack steps -> dt, dlam, ratio test ---- */
        double bdsl = 0.0, bdsu = 0.0; /* soft box row of this lane: dc + ds resp. -dc + ds, cancellation-free */
        bool bsoft = false;
        if (GEN)
        {
            const bool isg = lane < S.ng, iss = lane < S.ns;
            const int bsj = has ? (int) S.srev[ib] : -1;
            double gdc = 0.0;
            if (isg)
            {
                GQP_DOT_UNROLL
                for (int r = 0; r < n; r++) gdc += C.G[lane * C.SG + r] * L.dv[r];
                if (Rg.sj >= 0)
                {
                    const double gl_ = Rg.ll * frcp(Rg.tl), gu_ = Rg.lu * frcp(Rg.tu);
                    C.rGl[S.nb + lane] = gl_; C.rGu[S.nb + lane] = gu_;
                    C.rRl[S.nb + lane] = gl_ * gdc; C.rRu[S.nb + lane] = gu_ * gdc;
                }
            }
            if (bsj >= 0)
            {
                const double gl_ = ll * frcp(ttl), gu_ = lu * frcp(ttu);
                C.rGl[ib] = gl_; C.rGu[ib] = gu_;
                C.rRl[ib] = gl_ * dvj; C.rRu[ib] = gu_ * dvj;
            }
            __syncthreads(); /* Gamma dc of the soft rows published */
            if (iss)
            {
                double accl = 0.0, accu = 0.0;
                int cnt = 0;
                for (int row = 0; row < nbg; row++)
                    if (C.rsj[row] == lane) { accl += C.rRl[row]; accu += C.rRu[row]; cnt++; }
                C.scnt[lane] = cnt;
                const double Dl = fDl, Du = fDu;
                const double il = Dl != 0.0 ? frcp(Dl) : 0.0, iu = Du != 0.0 ? frcp(Du) : 0.0;
                const double rsl = fRl, rsu = fRu;
                const double dsl = (-rsl - accl) * il, dsu = (-rsu + accu) * iu;
                C.sIl[lane] = il; C.sIu[lane] = iu; C.sRl[lane] = rsl; C.sRu[lane] = rsu;
                if (CORR) { WAT(D.dsv, S.o_s + lane) = dsl; WAT(D.dsv, S.o_s + S.ns + lane) = dsu; }
                /* the two bound rows of the slack */
                const int e0 = S.o_ct + 2 * nbg + lane, e1 = e0 + S.ns;
                const bool sal = fsal, sau = fsau;
                const double sll = fsll, slu = fslu, stl = fstl, stu = fstu;
                C.sEl[lane] = fZl + sll * frcp(stl);
                C.sEu[lane] = fZu + slu * frcp(stu);
                const double spl = fspl, spu = fspu;
                const double rml = sal ? sll * stl - O.tau_min + pscale * spl - smu : 0.0;
                const double rmu = sau ? slu * stu - O.tau_min + pscale * spu - smu : 0.0;
                const double dtl = sal ? dsl + fsrdl : 0.0, dtu = sau ? dsu + fsrdu : 0.0;
                const double dll = sal ? -(rml + sll * dtl) * frcp(stl) : 0.0;
                const double dlu = sau ? -(rmu + slu * dtu) * frcp(stu) : 0.0;
                const double c1 = -sll * frcp(dll), c2 = -slu * frcp(dlu), c3 = -stl * frcp(dtl), c4 = -stu * frcp(dtu);
                alpha = (dll < 0.0 && c1 < alpha) ? c1 : alpha;
                alpha = (dlu < 0.0 && c2 < alpha) ? c2 : alpha;
                alpha = (dtl < 0.0 && c3 < alpha) ? c3 : alpha;
                alpha = (dtu < 0.0 && c4 < alpha) ? c4 : alpha;
                S0 += sll * stl + slu * stu;
                S1 += sll * dtl + stl * dll + slu * dtu + stu * dlu;
                S2 += dll * dtl + dlu * dtu;
                nact += (int) sal + (int) sau;
                if (!CORR)
                {
                    WAT(D.pcorr, e0) = dll * dtl;
                    WAT(D.pcorr, e1) = dlu * dtu;
                }
                else
                {
                    WAT(D.dlam, e0) = dll; WAT(D.dlam, e1) = dlu;
                    WAT(D.dt, e0) = dtl; WAT(D.dt, e1) = dtu;
                }
            }
            __syncthreads(); /* slack steps published */
            if (bsj >= 0)
            {
                /* dc + ds = (E dc - r~ - sum_{k != i} Gamma_k dc_k)/D */
                double El, Eu, al_, au_;
                wpi_excl(C, nbg, ib, bsj, C.rGl, C.rGu, El, Eu);
                wpi_excl(C, nbg, ib, bsj, C.rRl, C.rRu, al_, au_);
                El += C.sEl[bsj]; Eu += C.sEu[bsj];
                bsoft = true;
                bdsl = (El * dvj - C.sRl[bsj] - al_) * C.sIl[bsj];
                bdsu = (-Eu * dvj - C.sRu[bsj] + au_) * C.sIu[bsj];
            }
            if (isg)
            {
                double gsl_ = gdc, gsu_ = -gdc; /* dc + ds resp. -dc + ds */
                if (Rg.sj >= 0)
                {
                    const int q = Rg.sj;
                    double El, Eu, al_, au_;
                    wpi_excl(C, nbg, S.nb + lane, q, C.rGl, C.rGu, El, Eu);
                    wpi_excl(C, nbg, S.nb + lane, q, C.rRl, C.rRu, al_, au_);
                    El += C.sEl[q]; Eu += C.sEu[q];
                    gsl_ = (El * gdc - C.sRl[q] - al_) * C.sIl[q];
                    gsu_ = (-Eu * gdc - C.sRu[q] + au_) * C.sIu[q];
                }
                const double grdl = fgrdl, grdu = fgrdu, gpl = fgpl, gpu = fgpu;
                const double rml = Rg.al ? Rg.ll * Rg.tl - O.tau_min + pscale * gpl - smu : 0.0;
                const double rmu = Rg.au ? Rg.lu * Rg.tu - O.tau_min + pscale * gpu - smu : 0.0;
                const double dtl = Rg.al ? gsl_ + grdl : 0.0, dtu = Rg.au ? gsu_ + grdu : 0.0;
                const double dll = Rg.al ? -(rml + Rg.ll * dtl) * frcp(Rg.tl) : 0.0;
                const double dlu = Rg.au ? -(rmu + Rg.lu * dtu) * frcp(Rg.tu) : 0.0;
                const double c1 = -Rg.ll * frcp(dll), c2 = -Rg.lu * frcp(dlu), c3 = -Rg.tl * frcp(dtl), c4 = -Rg.tu * frcp(dtu);
                alpha = (dll < 0.0 && c1 < alpha) ? c1 : alpha;
                alpha = (dlu < 0.0 && c2 < alpha) ? c2 : alpha;
                alpha = (dtl < 0.0 && c3 < alpha) ? c3 : alpha;
                alpha = (dtu < 0.0 && c4 < alpha) ? c4 : alpha;
                S0 += Rg.ll * Rg.tl + Rg.lu * Rg.tu;
                S1 += Rg.ll * dtl + Rg.tl * dll + Rg.lu * dtu + Rg.tu * dlu;
                S2 += dll * dtl + dlu * dtu;
                nact += (int) Rg.al + (int) Rg.au;
                if (!CORR)
                {
                    WAT(D.pcorr, Rg.el) = dll * dtl;
                    WAT(D.pcorr, Rg.eu) = dlu * dtu;
                }
                else
                {
                    WAT(D.dlam, Rg.el) = dll; WAT(D.dlam, Rg.eu) = dlu;
                    WAT(D.dt, Rg.el) = dtl; WAT(D.dt, Rg.eu) = dtu;
                }
            }
        }
        if (has)
        {
            const double rml = al ? ll * ttl - O.tau_min + pscale * pl - smu : 0.0;
            const double rmu = au ? lu * ttu - O.tau_min + pscale * pu - smu : 0.0;
            const double dtl = al ? (bsoft ? bdsl : dvj) + rdl : 0.0, dtu = au ? (bsoft ? bdsu : -dvj) + rdu : 0.0;
            const double dll = al ? -(rml + ll * dtl) * frcp(ttl) : 0.0;
            const double dlu = au ? -(rmu + lu * dtu) * frcp(ttu) : 0.0;
            const double c1 = -ll * frcp(dll), c2 = -lu * frcp(dlu), c3 = -ttl * frcp(dtl), c4 = -ttu * frcp(dtu);
            alpha = (dll < 0.0 && c1 < alpha) ? c1 : alpha;
            alpha = (dlu < 0.0 && c2 < alpha) ? c2 : alpha;
            alpha = (dtl < 0.0 && c3 < alpha) ? c3 : alpha;
            alpha = (dtu < 0.0 && c4 < alpha) ? c4 : alpha;
            S0 += ll * ttl + lu * ttu;
            S1 += ll * dtl + ttl * dll + lu * dtu + ttu * dlu;
            S2 += dll * dtl + dlu * dtu;
            nact += (int) al + (int) au;
            if (!CORR)
            {
                WAT(D.pcorr, el) = dll * dtl;
                WAT(D.pcorr, eu) = dlu * dtu;
            }
            else
            {
                WAT(D.dlam, el) = dll; WAT(D.dlam, eu) = dlu;
                WAT(D.dt, el) = dtl; WAT(D.dt, eu) = dtu;
            }
        }
        __syncthreads();
        if (lane < NX) L.dx[lane] = dxn;
        if (pf == 1.2345678e-300) L.bc[2] = pf; /* never true in practice: keeps the touch loads alive */
        __syncthreads();
    }

    if (redo == 2) return; /* sensitivity pass: dux, dsv, dpi, dlam, dt are the result */
    alpha = wpi_min(alpha, L.red, lane);
    const int it = D.iter[inst];
    double *st = (inst < D.stat_inst && it + 1 < D.stat_rows) ? D.stat + (size_t) (it + 1) * GQP_STAT_COLS * D.stat_inst + inst : nullptr;
    if (!CORR)
    {
        S0 = wpi_sum(S0, L.red, lane); S1 = wpi_sum(S1, L.red, lane); S2 = wpi_sum(S2, L.red, lane);
        const double nact_d = wpi_sum((double) nact, L.red, lane);
        if (lane == 0)
        {
            const double mu = D.mu[inst];
            const double mu_aff = nact_d > 0.0 ? (S0 + alpha * S1 + alpha * alpha * S2) / nact_d : 0.0;
            double sigma = mu > 0.0 ? mu_aff / mu : 0.0;
            sigma = sigma * sigma * sigma;
            D.smu[inst] = sigma * mu;
            D.alpha[inst] = alpha;
            if (st) { st[0] = alpha; st[1 * D.stat_inst] = alpha; st[2 * D.stat_inst] = mu_aff; st[3 * D.stat_inst] = sigma; }
        }
        return;
    }
    const double alpha_aff = dabs(D.alpha[inst]);
    if (O.cond_pred_corr && !redo)
    {
        /* conditional corrector: a step that would more than double the duality measure is taken again with the centering term
         * alone (redo pair of the host loop) */
        S0 = wpi_sum(S0, L.red, lane); S1 = wpi_sum(S1, L.red, lane); S2 = wpi_sum(S2, L.red, lane);
        const double nact_d = wpi_sum((double) nact, L.red, lane);
        if (nact_d > 0.0 && (S0 + alpha * S1 + alpha * alpha * S2) / nact_d > 2.0 * D.mu[inst])
        {
            __syncthreads(); /* everybody has read alpha[inst] */
            if (lane == 0) D.alpha[inst] = -alpha_aff;
            return;
        }
    }
    const double a = D.mu[inst] > 0.0 ? gqp_step_scale(alpha) : 1.0;
    for (int e = lane; e < (D.N + 1) * n; e += 64) WAT(D.ux, e) += a * WAT(D.dux, e);
    for (int e = NX + lane; e < (D.N + 1) * NX; e += 64) WAT(D.pi, e) += a * WAT(D.dpi, e);
    if (GEN)
    {
        GQP_STAGE_REF SN = D.st[D.N];
        for (int e = lane; e < SN.o_s + 2 * SN.ns; e += 64) WAT(D.sv, e) += a * WAT(D.dsv, e);
    }
    for (int k = 0; k <= D.N; k++)
    {
        /* one lane per inequality side of the stage (at most 128 sides: two passes) */
        GQP_STAGE_REF S = D.st[k];
        const Am128 am = wpi_am(D, inst, k);
        const int nct = 2 * (S.nb + S.ng) + 2 * S.ns;
        for (int side = lane; side < nct; side += 64)
            if (abit(am, side))
            {
                const int e = S.o_ct + side;
                const double lam = WAT(D.lam, e) + a * WAT(D.dlam, e);
                const double t = WAT(D.t, e) + a * WAT(D.dt, e);
                WAT(D.lam, e) = lam < O.lam_min ? O.lam_min : lam;
                WAT(D.t, e) = t < O.t_min ? O.t_min : t;
            }
    }
    __syncthreads();
    if (lane == 0)
    {
        D.alpha[inst] = alpha;
        D.iter[inst] = it + 1;
        if (st) { st[4 * D.stat_inst] = alpha; st[5 * D.stat_inst] = alpha; }
    }
}

/* ------------------------------------------------------------------------ init / finalize */

/* cold start, same rules as k_init (ipm_kernels.hpp); lane = variable / general row / slack */
template <bool GEN>
__global__ void __launch_bounds__(64) kw_init(GqpDev D, GqpOpts O)
{
    GQP_DYN_SHARED(smem);
    const int NX = D.NX, NU = D.NU, n = NX + NU;
    const int inst = blockIdx.x, lane = threadIdx.x;
    if (inst >= D.B) return;
    const double thr0 = 1e-1;
    /* t0_init 0 / 1: constant (t, lam), primal iterate and slacks stay at zero (see k_init) */
    const bool heur = O.t0_init != 0 && O.t0_init != 1;
    const double t_c = O.t0_init == 0 ? sqrt(O.mu0) : 1.0, l_c = O.t0_init == 0 ? sqrt(O.mu0) : O.mu0;
    double *vv = smem, *cv = smem + 64, *ssl = smem + 128, *ssu = smem + 160; /* v, row values, slack values */
    for (int k = 0; k <= D.N; k++)
    {
        GQP_STAGE_REF S = D.st[k];
        const int nbg = S.nb + (GEN ? S.ng : 0);
        const Am128 am = wpi_am(D, inst, k);
        const bool mine = lane < n, hasb = mine && ((S.bmask >> lane) & 1);
        const bool fixed = mine && ((S.emask >> lane) & 1);
        const int ib = hasb ? popc64(S.bmask & (((uint64_t) 1 << lane) - 1)) : 0;
        const int bsj = (GEN && hasb) ? (int) S.srev[ib] : -1;
        double v = 0.0, lb = 0.0, ub = 0.0;
        bool al = false, au = false;
        if (mine)
        {
            v = fixed ? WAT(D.ux, k * n + lane) : 0.0;
            if (hasb)
            {
                lb = WAT(D.dvec, S.o_ct + ib); ub = WAT(D.dvec, S.o_ct + nbg + ib);
                al = abit(am, ib); au = abit(am, nbg + ib);
                if (!fixed && bsj < 0 && heur)
                {
                    const double tl = v - lb, tu = ub - v;
                    if (al && au)
                    {
                        if (tl < thr0) v = (tu < thr0) ? 0.5 * (lb + ub) : lb + thr0;
                        else if (tu < thr0) v = ub - thr0;
                    }
                    else if (al) { if (tl < thr0) v = lb + thr0; }
                    else if (au) { if (tu < thr0) v = ub - thr0; }
                }
            }
            WAT(D.ux, k * n + lane) = v;
        }
        if (S.has_dyn && lane < NX) WAT(D.pi, (k + 1) * NX + lane) = 0.0;
        double sl = 0.0, su = 0.0; /* slack values seen by this lane's box row */
        if (GEN)
        {
            vv[lane] = v;
            __syncthreads();
            /* row values: box rows by their owner, general rows by lane g */
            const bool isg = lane < S.ng, iss = lane < S.ns;
            double gc = 0.0;
            if (isg)
            {
                for (int r = 0; r < n; r++) gc += WAT(D.DCt, (S.o_g + lane) * n + r) * vv[r];
                cv[S.nb + lane] = gc;
            }
            if (hasb) cv[ib] = v;
            __syncthreads();
            if (iss)
            {
                double a = 0.0, c = 0.0;
                if (heur && abit(am, 2 * nbg + lane)) a = WAT(D.dvec, S.o_ct + 2 * nbg + lane) + thr0;
                if (heur && abit(am, 2 * nbg + S.ns + lane)) c = WAT(D.dvec, S.o_ct + 2 * nbg + S.ns + lane) + thr0;
                for (int row = 0; row < nbg; row++)
                    if (heur && S.srev[row] == lane)
                    {
                        const double need_l = WAT(D.dvec, S.o_ct + row) - cv[row] + thr0, need_u = cv[row] - WAT(D.dvec, S.o_ct + nbg + row) + thr0;
                        if (abit(am, row) && need_l > a) a = need_l;
                        if (abit(am, nbg + row) && need_u > c) c = need_u;
                    }
                WAT(D.sv, S.o_s + lane) = a; WAT(D.sv, S.o_s + S.ns + lane) = c;
                ssl[lane] = a; ssu[lane] = c;
                const int e0 = S.o_ct + 2 * nbg + lane, e1 = e0 + S.ns;
                double tl = a - WAT(D.dvec, e0), tu = c - WAT(D.dvec, e1);
                if (tl < thr0) tl = thr0;
                if (tu < thr0) tu = thr0;
                if (!heur) { tl = t_c; tu = t_c; }
                const bool sal = abit(am, 2 * nbg + lane), sau = abit(am, 2 * nbg + S.ns + lane);
                WAT(D.t, e0) = sal ? tl : 0.0; WAT(D.t, e1) = sau ? tu : 0.0;
                WAT(D.lam, e0) = sal ? (heur ? O.mu0 / tl : l_c) : 0.0; WAT(D.lam, e1) = sau ? (heur ? O.mu0 / tu : l_c) : 0.0;
            }
            __syncthreads();
            if (bsj >= 0) { sl = ssl[bsj]; su = ssu[bsj]; }
            if (isg)
            {
                const int row = S.nb + lane, sj = S.srev[row];
                const double gsl = sj >= 0 ? ssl[sj] : 0.0, gsu = sj >= 0 ? ssu[sj] : 0.0;
                double tl = gc + gsl - WAT(D.dvec, S.o_ct + row), tu = WAT(D.dvec, S.o_ct + nbg + row) - gc + gsu;
                if (tl < thr0) tl = thr0;
                if (tu < thr0) tu = thr0;
                if (!heur) { tl = t_c; tu = t_c; }
                const bool gal = abit(am, row), gau = abit(am, nbg + row);
                WAT(D.t, S.o_ct + row) = gal ? tl : 0.0; WAT(D.t, S.o_ct + nbg + row) = gau ? tu : 0.0;
                WAT(D.lam, S.o_ct + row) = gal ? (heur ? O.mu0 / tl : l_c) : 0.0; WAT(D.lam, S.o_ct + nbg + row) = gau ? (heur ? O.mu0 / tu : l_c) : 0.0;
            }
            __syncthreads(); /* the exchange buffers are reused by the next stage */
        }
        if (hasb)
        {
            double tl = v + sl - lb, tu = ub - v + su;
            if (tl < thr0) tl = thr0;
            if (tu < thr0) tu = thr0;
            if (!heur) { tl = t_c; tu = t_c; }
            WAT(D.t, S.o_ct + ib) = al ? tl : 0.0;
            WAT(D.t, S.o_ct + nbg + ib) = au ? tu : 0.0;
            WAT(D.lam, S.o_ct + ib) = al ? (heur ? O.mu0 / tl : l_c) : 0.0;
            WAT(D.lam, S.o_ct + nbg + ib) = au ? (heur ? O.mu0 / tu : l_c) : 0.0;
        }
    }
    if (lane == 0)
    {
        D.iter[inst] = 0;
        D.status[inst] = GQP_RUNNING;
        D.alpha[inst] = 1.0;
    }
}

/* multipliers of the fixed variables from stationarity; natural slacks and zero multipliers of the masked
 * sides (ocp_qp_compute_t, acados/ocp_qp/ocp_qp_common.c:874-921, restated for those rows), as kb_finalize /
 * k_finalize */
template <bool GEN>
__global__ void __launch_bounds__(64) kw_finalize(GqpDev D)
{
    const int NX = D.NX, NU = D.NU, n = NX + NU, NP = n * (n + 1) / 2;
    const int inst = blockIdx.x, lane = threadIdx.x;
    if (inst >= D.B) return;
    for (int k = 0; k <= D.N; k++)
    {
        GQP_STAGE_REF S = D.st[k];
        const int nbg = S.nb + (GEN ? S.ng : 0);
        const Am128 am = wpi_am(D, inst, k);
        if (lane < n && ((S.bmask >> lane) & 1))
        {
            const int j = lane;
            const int ib = popc64(S.bmask & (((uint64_t) 1 << j) - 1));
            const int el = S.o_ct + ib, eu = el + nbg;
            const double vj = WAT(D.ux, k * n + j);
            if ((S.emask >> j) & 1)
            {
                double a = WAT(D.rq, k * n + j);
                for (int c = 0; c < n; c++) a += WAT(D.RSQ, k * NP + (c <= j ? PK(j, c) : PK(c, j))) * WAT(D.ux, k * n + c);
                for (int c = 0; c < NX; c++) a += WAT(D.BAt, (k * n + j) * NX + c) * WAT(D.pi, (k + 1) * NX + c);
                if (j >= NU) a -= WAT(D.pi, k * NX + j - NU);
                if (GEN)
                    for (int g = 0; g < S.ng; g++)
                        a -= WAT(D.DCt, (S.o_g + g) * n + j) * (WAT(D.lam, S.o_ct + S.nb + g) - WAT(D.lam, S.o_ct + nbg + S.nb + g));
                WAT(D.lam, el) = a > 0.0 ? a : 0.0;
                WAT(D.lam, eu) = a < 0.0 ? -a : 0.0;
                WAT(D.t, el) = 0.0;
                WAT(D.t, eu) = 0.0;
            }
            else
            {
                const int sj = GEN ? (int) S.srev[ib] : -1;
                const double sl = sj >= 0 ? WAT(D.sv, S.o_s + sj) : 0.0, su = sj >= 0 ? WAT(D.sv, S.o_s + S.ns + sj) : 0.0;
                if (!abit(am, ib)) { WAT(D.t, el) = vj + sl - WAT(D.dvec, el); WAT(D.lam, el) = 0.0; }
                if (!abit(am, nbg + ib)) { WAT(D.t, eu) = WAT(D.dvec, eu) - vj + su; WAT(D.lam, eu) = 0.0; }
            }
        }
        if (GEN && lane < S.ng)
        {
            const int row = S.nb + lane, el = S.o_ct + row, eu = el + nbg, sj = S.srev[row];
            const bool al = abit(am, row), au = abit(am, nbg + row);
            if (!al || !au)
            {
                double c = 0.0;
                for (int r = 0; r < n; r++) c += WAT(D.DCt, (S.o_g + lane) * n + r) * WAT(D.ux, k * n + r);
                const double sl = sj >= 0 ? WAT(D.sv, S.o_s + sj) : 0.0, su = sj >= 0 ? WAT(D.sv, S.o_s + S.ns + sj) : 0.0;
                if (!al) { WAT(D.t, el) = c + sl - WAT(D.dvec, el); WAT(D.lam, el) = 0.0; }
                if (!au) { WAT(D.t, eu) = WAT(D.dvec, eu) - c + su; WAT(D.lam, eu) = 0.0; }
            }
        }
        if (GEN && lane < S.ns)
        {
            const int e0 = S.o_ct + 2 * nbg + lane, e1 = e0 + S.ns;
            if (!abit(am, 2 * nbg + lane)) { WAT(D.t, e0) = WAT(D.sv, S.o_s + lane) - WAT(D.dvec, e0); WAT(D.lam, e0) = 0.0; }
            if (!abit(am, 2 * nbg + S.ns + lane)) { WAT(D.t, e1) = WAT(D.sv, S.o_s + S.ns + lane) - WAT(D.dvec, e1); WAT(D.lam, e1) = 0.0; }
        }
    }
}

} // namespace gqp

#endif
