/*
 * ipm_kernels_wpi_mfma.hpp -- the factor sweep of the wave-per-instance family for 17 <= nu + nx <= 32 with the three
 * O(n^3) parts of a stage on the FP64 matrix pipe (v_mfma_f64_16x16x4_f64) and a BLOCKED Cholesky:
 *
 *   W = [B A]' Lx+            16x16 tiles, K = nx in steps of 4                 (2 LDS reads per MFMA)
 *   M = H~ + W W'             three lower tiles T00 T10 T11, 3 MFMAs per K-step (2 LDS reads per 3 MFMAs)
 *   Cholesky of M             panels of FOUR columns: the panel is published once, every lane factors the 4 x 4
 *                             diagonal block redundantly and solves its own row of the panel (lane = row), the rhs rides
 *                             along; the trailing update of all three tiles is ONE rank-4 MFMA each.  Two LDS round
 *                             trips per four columns instead of one per column.
 *
 * Why: the register-tile kernel kw_factor<4> (ipm_kernels_wpi.hpp) is instruction-issue bound -- ~4,200 instructions per
 * stage at n = 27, two thirds of them in the 27 Cholesky column steps (~100 each), a fifth in the two outer-product
 * loops.  Here a stage is ~1,500 instructions.  The FP64 matrix pipe itself is NOT faster than the vector pipe on this
 * chip (tools/mfma_f64_probe: 47.6 TFLOP/s v_mfma_f64_16x16x4_f64 vs 69.3 TFLOP/s v_fma_f64, dependent-accumulator
 * latency ~186 cycles): the MFMAs are used for what they save in issue slots and LDS operand reads (one instruction =
 * 1,024 multiply-adds fed by two 8-byte reads per lane), not for flops.
 *
 * Operand / result layout of v_mfma_f64_16x16x4_f64 (checked by the probe with A = I and an asymmetric B):
 *   D (16 x 16) += A (16 x 4) B (4 x 16);  lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15] and holds
 *   D[(l >> 4) + 4 r][l & 15], r = 0..3.
 * Same arithmetic contract as kw_factor: same HBM inputs / outputs (packed factor Lf, lf, residual norms, status).
 */
#ifndef IPM_KERNELS_WPI_MFMA_HPP_
#define IPM_KERNELS_WPI_MFMA_HPP_

#include "ipm_kernels_wpi.hpp"

namespace gqp
{

#if defined(__HIP_DEVICE_COMPILE__)
typedef double gqp_d4 __attribute__((ext_vector_type(4)));
__device__ static inline void gqp_mfma(double a, double b, gqp_d4 &c) { c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ static inline gqp_d4 gqp_zero4() { return (gqp_d4){0.0, 0.0, 0.0, 0.0}; }
#else
/* host pass of hipcc (never executed) and the host simulation of the CPU test tier, where the 64 lanes of a workgroup are
 * coroutines of one host thread that share `__shared__` storage: operands are exchanged through it */
struct gqp_d4
{
    double v[4];
    __host__ __device__ double &operator[](int i) { return v[i]; }
    __host__ __device__ const double &operator[](int i) const { return v[i]; }
};
__device__ static inline void gqp_mfma(double a, double b, gqp_d4 &c)
{
    __shared__ double mf_a[64], mf_b[64];
    const int l = threadIdx.x;
    mf_a[l] = a; mf_b[l] = b;
    __syncthreads();
    for (int r = 0; r < 4; r++)
    {
        const int row = (l >> 4) + 4 * r, col = l & 15;
        double s = 0.0;
        for (int kk = 0; kk < 4; kk++) s += mf_a[row + 16 * kk] * mf_b[col + 16 * kk];
        c[r] += s;
    }
    __syncthreads();
}
__device__ static inline gqp_d4 gqp_zero4() { gqp_d4 z; z[0] = z[1] = z[2] = z[3] = 0.0; return z; }
#endif

/* LDS of the MFMA factor sweep: WpiLds2 with 32 rows of [B A]' / W, the x-block of the next stage's factor padded to a
 * multiple of four rows (K-steps) plus slack for the column over-read of the last tile, and the two panel buffers */
__host__ __device__ static inline size_t wpim_lds_doubles(int NX, int NU)
{
    const int n = NX + NU, NP = n * (n + 1) / 2, SX = wpi2_sx(NX), NX4 = (NX + 3) & ~3;
    return (size_t) NP + 8 + (size_t) 32 * SX + (size_t) NX4 * SX + 32 + 6 * 64 + 2 * 32 * 5 + 8 + 64 + 8;
}

struct WpiLdsM
{
    double *__restrict__ Hp, *__restrict__ Bw, *__restrict__ Lx;
    double *__restrict__ lx, *__restrict__ v, *__restrict__ rb, *__restrict__ pin, *__restrict__ w0, *__restrict__ gam;
    double *__restrict__ P;  /* 32 x 5: the published panel (4 columns, stride 5) */
    double *__restrict__ Lp; /* 32 x 5: rows of L21 (zero for finished rows): operands of the trailing update */
    double *__restrict__ pm; /* 4 (+4): rhs entries of the panel variables */
    double *__restrict__ red;
    int SX;
};

__device__ static inline WpiLdsM wpim_carve(double *sm, int NX, int NU)
{
    const int n = NX + NU, NP = n * (n + 1) / 2, NX4 = (NX + 3) & ~3;
    WpiLdsM L;
    L.SX = wpi2_sx(NX);
    double *p = sm;
    L.Hp = p; p += NP + 8;
    L.Bw = p; p += 32 * L.SX;
    L.Lx = p; p += NX4 * L.SX + 32;
    L.lx = p; p += 64; L.v = p; p += 64; L.rb = p; p += 64; L.pin = p; p += 64; L.w0 = p; p += 64; L.gam = p; p += 64;
    L.P = p; p += 32 * 5; L.Lp = p; p += 32 * 5;
    L.pm = p; p += 8;
    L.red = p; p += 64;
    return L;
}

/* PF: software pipelining of the stage's matrix blocks through registers -- 0 none (loaded at the stage top), 1 the packed
 * Hessian block, 2 the Hessian block and [B A]' */
template <bool GEN, int PF>
__global__ void __launch_bounds__(64) GQP_WAVES_PER_EU(2) kw_factor_m(GqpDev D, GqpOpts O, int redo)
{
    GQP_DYN_SHARED(smem);
    const int NX = D.NX, NU = D.NU, n = NX + NU, NP = n * (n + 1) / 2;
    const int inst = blockIdx.x, lane = threadIdx.x;
    if (inst >= D.B) return;
    if (D.status[inst] != GQP_RUNNING) return;
    const WpiLdsM L = wpim_carve(smem, NX, NU);
    const WpiCon C = wpi_con_carve(smem + wpim_lds_doubles(NX, NU), n, GEN ? D.NG : 0, GEN ? D.NS : 0);
    const int SX = L.SX, NX4 = (NX + 3) & ~3, KS = NX4 >> 2;
    const int c16 = lane & 15, g4 = lane >> 4;
    const bool mine = lane < n;
    const bool xt1 = NX > 16; /* the x-block spans two column tiles */

    /* zero what is only ever partly overwritten: rows >= n and columns >= nx of [B A]' / W, the strict upper triangle
     * and the padding rows of Lx, the panel buffers */
    for (int e = lane; e < 32 * SX; e += 64) L.Bw[e] = 0.0;
    for (int e = lane; e < NX4 * SX + 32; e += 64) L.Lx[e] = 0.0;
    for (int e = lane; e < 2 * 32 * 5 + 8; e += 64) L.P[e] = 0.0;
    L.lx[lane] = 0.0;
    double nrm_g = 0.0, nrm_b = 0.0, nrm_d = 0.0, nrm_m = 0.0, musum = 0.0, obj = 0.0;
    int nact = 0;
    __syncthreads();

    /* register prefetch of RSQ (packed, <= 528 entries) and [B A]' (<= 31 x 31 entries) of the stage to come */
    constexpr int PFH = PF >= 1 ? 9 : 1, PFB = PF >= 2 ? 16 : 1;
    double pfH[PFH], pfB[PFB];
#pragma unroll
    for (int i = 0; i < PFH; i++) { const int p = lane + 64 * i; pfH[i] = (PF >= 1 && p < NP) ? WAT(D.RSQ, D.N * NP + p) : 0.0; }
#pragma unroll
    for (int i = 0; i < PFB; i++) { const int e = lane + 64 * i; pfB[i] = (PF >= 2 && e < n * NX) ? WAT(D.BAt, D.N * n * NX + e) : 0.0; }

    GQP_TICK_INIT();
    for (int k = D.N; k >= 0; k--)
    {
        GQP_STAGE_REF S = D.st[k];
        const uint64_t imask = S.bmask & ~S.emask;
        const Am128 am = wpi_am(D, inst, k);
        const int nbg = S.nb + (GEN ? S.ng : 0);
        const bool fixed = mine && ((S.emask >> lane) & 1);

        /* ---- the two matrix blocks of this stage were fetched into registers one stage ago: registers -> LDS ---- */
        if (PF >= 1)
        {
#pragma unroll
            for (int i = 0; i < PFH; i++) { const int p = lane + 64 * i; if (p < NP) L.Hp[p] = pfH[i]; }
        }
        else
            for (int p = lane; p < NP; p += 64) L.Hp[p] = WAT(D.RSQ, k * NP + p);
        {
            int r = lane / NX, c = lane - r * NX;
            const int dr = 64 / NX, dc = 64 - dr * NX;
            if (PF >= 2)
            {
#pragma unroll
                for (int i = 0; i < PFB; i++)
                {
                    if (lane + 64 * i < n * NX) L.Bw[r * SX + c] = pfB[i];
                    r += dr; c += dc;
                    if (c >= NX) { c -= NX; r++; }
                }
            }
            else
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
        /* software pipeline: the matrix blocks of the NEXT stage (k - 1) start their way from HBM now and are consumed at
         * the top of the next iteration -- a lone wave cannot hide an HBM round trip behind anything else */
        if (k > 0)
        {
            if (PF >= 1)
            {
#pragma unroll
                for (int i = 0; i < PFH; i++) { const int p = lane + 64 * i; if (p < NP) pfH[i] = WAT(D.RSQ, (k - 1) * NP + p); }
            }
            if (PF >= 2)
            {
#pragma unroll
                for (int i = 0; i < PFB; i++) { const int e = lane + 64 * i; if (e < n * NX) pfB[i] = WAT(D.BAt, (k - 1) * n * NX + e); }
            }
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
        GQP_TICK(8);
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
        GQP_TICK(9);
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
            GQP_TICK(10);
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
            GQP_TICK(11);
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
        /* ---- W = [B A]' Lx+ on the matrix pipe: row tiles 0 / 1 (variables 0..15 / 16..31), column tiles 0 / 1 of
         * the x-block; Lx+ is lower triangular (explicit zeros above its diagonal), so column tile 1 starts at K = 16 ---- */
        gqp_d4 w00 = gqp_zero4(), w10 = gqp_zero4(), w01 = gqp_zero4(), w11 = gqp_zero4();
        for (int s = 0; s < KS; s++)
        {
            const int q = 4 * s + g4;
            const double a0 = L.Bw[c16 * SX + q], a1 = L.Bw[(16 + c16) * SX + q];
            const double b0 = L.Lx[q * SX + c16];
            gqp_mfma(a0, b0, w00);
            gqp_mfma(a1, b0, w10);
            if (xt1 && s >= 4)
            {
                const double b1 = L.Lx[q * SX + 16 + c16];
                gqp_mfma(a0, b1, w01);
                gqp_mfma(a1, b1, w11);
            }
        }
        __syncthreads(); /* everybody is done with [B A]': the buffer becomes W */
        GQP_TICK(1);
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
            const int r0 = g4 + 4 * r, r1 = 16 + r0;
            if (c16 < NX) { L.Bw[r0 * SX + c16] = w00[r]; if (r1 < n) L.Bw[r1 * SX + c16] = w10[r]; }
            if (xt1 && 16 + c16 < NX) { L.Bw[r0 * SX + 16 + c16] = w01[r]; if (r1 < n) L.Bw[r1 * SX + 16 + c16] = w11[r]; }
        }
        /* ---- the three lower tiles of H~ (symmetric fill of the diagonal tiles; unit diagonal on the padding) ---- */
        gqp_d4 t00, t10, t11;
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
            const int r0 = g4 + 4 * r, r1 = 16 + r0, c0 = c16, c1 = 16 + c16;
            const int lo0 = r0 < c0 ? r0 : c0, hi0 = r0 < c0 ? c0 : r0;
            t00[r] = hi0 < n ? L.Hp[PK(hi0, lo0)] : (r0 == c0 ? 1.0 : 0.0);
            t10[r] = r1 < n ? L.Hp[PK(r1, c0)] : 0.0;
            const int lo1 = r1 < c1 ? r1 : c1, hi1 = r1 < c1 ? c1 : r1;
            t11[r] = hi1 < n ? L.Hp[PK(hi1, lo1)] : (r1 == c1 ? 1.0 : 0.0);
        }
        __syncthreads(); /* W published */
        /* w0 = Lx+' rb + lx+ */
        if (lane < NX)
        {
            double a = L.lx[lane];
            GQP_DOT_UNROLL
            for (int q = lane; q < NX; q++) a += L.Lx[q * SX + lane] * L.rb[q];
            L.w0[lane] = a;
        }
        /* ---- M += W W' : three MFMAs per K-step from two operand reads ---- */
        for (int s = 0; s < KS; s++)
        {
            const int q = 4 * s + g4;
            const double a0 = L.Bw[c16 * SX + q], a1 = L.Bw[(16 + c16) * SX + q];
            gqp_mfma(a0, a0, t00);
            gqp_mfma(a1, a0, t10);
            gqp_mfma(a1, a1, t11);
        }
        if (GEN)
        {
            if (mine)
                for (int g = 0; g < S.ng; g++) gadd += C.G[g * C.SG + lane] * C.nuG[g];
            /* M += sum_g Gamma_eff a a': K runs over the general rows */
            for (int s = 0; 4 * s < S.ng; s++)
            {
                const int g = 4 * s + g4;
                const bool ok = g < S.ng;
                const double gm = ok ? C.gmG[g] : 0.0;
                const double b0 = (ok && c16 < n) ? C.G[g * C.SG + c16] : 0.0, b1 = (ok && 16 + c16 < n) ? C.G[g * C.SG + 16 + c16] : 0.0;
                const double a0 = b0 * gm, a1 = b1 * gm;
                gqp_mfma(a0, b0, t00);
                gqp_mfma(a1, b0, t10);
                gqp_mfma(a1, b1, t11);
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
                    for (int r = 0; r < 4; r++)
                    {
                        const int r0 = g4 + 4 * r, r1 = 16 + r0, c0 = c16, c1 = 16 + c16;
                        if (r0 < n && c0 < n) t00[r] -= cf * wpi_arow(C, S, i, r0) * wpi_arow(C, S, kk, c0);
                        if (r1 < n && c0 < n) t10[r] -= cf * wpi_arow(C, S, i, r1) * wpi_arow(C, S, kk, c0);
                        if (r1 < n && c1 < n) t11[r] -= cf * wpi_arow(C, S, i, r1) * wpi_arow(C, S, kk, c1);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
            const int r0 = g4 + 4 * r, r1 = 16 + r0;
            if (r0 == c16 && r0 < n) t00[r] += O.reg_prim + L.gam[r0];
            if (r0 == c16 && r1 < n) t11[r] += O.reg_prim + L.gam[r1];
        }
        if (S.emask)
        {
#pragma unroll
            for (int r = 0; r < 4; r++)
            {
                const int r0 = g4 + 4 * r, r1 = 16 + r0, c0 = c16, c1 = 16 + c16;
                if (((S.emask >> r0) & 1) || ((S.emask >> c0) & 1)) t00[r] = r0 == c0 ? 1.0 : 0.0;
                if (((S.emask >> r1) & 1) || ((S.emask >> c0) & 1)) t10[r] = 0.0;
                if (((S.emask >> r1) & 1) || ((S.emask >> c1) & 1)) t11[r] = r1 == c1 ? 1.0 : 0.0;
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

        /* ---- blocked Cholesky, panels of four columns; the rhs entry m of variable `lane` rides along ---- */
        const int npan = (n + 3) >> 2;
        for (int p = 0; p < npan; p++)
        {
            const int j0 = 4 * p, cc0 = j0 & 15;
            const bool right = j0 >= 16; /* panel in tile column 1 (only T11 is left) */
            /* publish the panel columns (rows of the tiles in this tile column) and the rhs entries of its variables */
            if (c16 >= cc0 && c16 < cc0 + 4)
            {
                const int cq = c16 - cc0;
#pragma unroll
                for (int r = 0; r < 4; r++)
                {
                    const int r0 = g4 + 4 * r;
                    if (!right) { L.P[r0 * 5 + cq] = t00[r]; L.P[(16 + r0) * 5 + cq] = t10[r]; }
                    else L.P[(16 + r0) * 5 + cq] = t11[r];
                }
            }
            if (lane >= j0 && lane < j0 + 4) L.pm[lane - j0] = m;
            __syncthreads();
            GQP_TICK(12);
            /* every lane: the 4 x 4 diagonal block, factored redundantly (a non-positive pivot zeroes its column, as in
             * the column-by-column kernels) */
            const double *Pd = L.P + j0 * 5;
            const double d00 = Pd[0], d10 = Pd[5], d11 = Pd[6], d20 = Pd[10], d21 = Pd[11], d22 = Pd[12];
            const double d30 = Pd[15], d31 = Pd[16], d32 = Pd[17], d33 = Pd[18];
            const double m0 = L.pm[0], m1 = L.pm[1], m2 = L.pm[2], m3 = L.pm[3];
            const int prl = (lane & 31) * 5; /* rows exist for lane < n <= 32 */
            const double prow0 = L.P[prl + 0], prow1 = L.P[prl + 1], prow2 = L.P[prl + 2], prow3 = L.P[prl + 3];
            const bool p0 = d00 > 0.0;
            const double i0 = p0 ? frsqrt(p0 ? d00 : 1.0) : 0.0;
            const double l10 = d10 * i0, l20 = d20 * i0, l30 = d30 * i0;
            const double e11 = d11 - l10 * l10;
            const bool p1 = e11 > 0.0;
            const double i1 = p1 ? frsqrt(p1 ? e11 : 1.0) : 0.0;
            const double l21 = (d21 - l20 * l10) * i1, l31 = (d31 - l30 * l10) * i1;
            const double e22 = d22 - l20 * l20 - l21 * l21;
            const bool p2 = e22 > 0.0;
            const double i2 = p2 ? frsqrt(p2 ? e22 : 1.0) : 0.0;
            const double l32 = (d32 - l30 * l20 - l31 * l21) * i2;
            const double e33 = d33 - l30 * l30 - l31 * l31 - l32 * l32;
            const bool p3 = e33 > 0.0;
            const double i3 = p3 ? frsqrt(p3 ? e33 : 1.0) : 0.0;
            /* rhs of the panel variables: y = L11^{-1} m */
            const double y0 = m0 * i0;
            const double y1 = (m1 - l10 * y0) * i1;
            const double y2 = (m2 - l20 * y0 - l21 * y1) * i2;
            const double y3 = (m3 - l30 * y0 - l31 * y1 - l32 * y2) * i3;
            /* this lane's row of the panel: rows below it solve against L11', rows inside it are rows of L11 */
            double x0 = 0.0, x1 = 0.0, x2 = 0.0, x3 = 0.0;
            const int rel = lane - j0;
            if (lane < n && rel >= 4)
            {
                x0 = prow0 * i0;
                x1 = (prow1 - x0 * l10) * i1;
                x2 = (prow2 - x0 * l20 - x1 * l21) * i2;
                x3 = (prow3 - x0 * l30 - x1 * l31 - x2 * l32) * i3;
                m -= x0 * y0 + x1 * y1 + x2 * y2 + x3 * y3;
                L.Lp[lane * 5 + 0] = x0; L.Lp[lane * 5 + 1] = x1; L.Lp[lane * 5 + 2] = x2; L.Lp[lane * 5 + 3] = x3;
            }
            else if (lane < 32)
            {
                /* finished rows and the rows of the panel itself take no part in the trailing update */
                L.Lp[lane * 5 + 0] = 0.0; L.Lp[lane * 5 + 1] = 0.0; L.Lp[lane * 5 + 2] = 0.0; L.Lp[lane * 5 + 3] = 0.0;
                if (rel >= 0 && rel < 4)
                {
                    /* row rel of L11 (diagonal d * rsqrt(d) as in the column kernels) */
                    const double dg0 = p0 ? d00 * i0 : 0.0, dg1 = p1 ? e11 * i1 : 0.0, dg2 = p2 ? e22 * i2 : 0.0, dg3 = p3 ? e33 * i3 : 0.0;
                    x0 = rel == 0 ? dg0 : rel == 1 ? l10 : rel == 2 ? l20 : l30;
                    x1 = rel == 1 ? dg1 : rel == 2 ? l21 : rel == 3 ? l31 : 0.0;
                    x2 = rel == 2 ? dg2 : rel == 3 ? l32 : 0.0;
                    x3 = rel == 3 ? dg3 : 0.0;
                    m = rel == 0 ? y0 : rel == 1 ? y1 : rel == 2 ? y2 : y3;
                }
            }
            /* the four new entries of this row go straight into the packed factor and into the x-block the next
             * (earlier) stage multiplies with; rows above the panel have nothing there */
            if (lane < n && rel >= 0)
            {
                const int base = PK(lane, j0);
                L.Hp[base] = x0;
                if (j0 + 1 <= lane) L.Hp[base + 1] = x1;
                if (j0 + 2 <= lane) L.Hp[base + 2] = x2;
                if (j0 + 3 <= lane) L.Hp[base + 3] = x3;
                if (lane >= NU)
                {
                    double *lxr = L.Lx + (lane - NU) * SX - NU;
                    if (j0 >= NU) lxr[j0] = x0;
                    if (j0 + 1 >= NU && j0 + 1 <= lane) lxr[j0 + 1] = x1;
                    if (j0 + 2 >= NU && j0 + 2 <= lane) lxr[j0 + 2] = x2;
                    if (j0 + 3 >= NU && j0 + 3 <= lane) lxr[j0 + 3] = x3;
                }
            }
            __syncthreads();
            GQP_TICK(13);
            /* trailing update: one rank-4 MFMA per tile that still has unfinished columns */
            if (j0 + 4 < n)
            {
                const double b1 = L.Lp[(16 + c16) * 5 + g4];
                if (!right)
                {
                    const double b0 = L.Lp[c16 * 5 + g4];
                    gqp_mfma(-b0, b0, t00);
                    gqp_mfma(-b1, b0, t10);
                }
                gqp_mfma(-b1, b1, t11);
            }
            GQP_TICK(14);
        }
        __syncthreads(); /* the packed factor is complete in Hp, the x-block in Lx */
        GQP_TICK(3);
        if (mine)
        {
            WAT(D.lf, k * n + lane) = m;
            if (lane >= NU) L.lx[lane - NU] = m;
        }
        for (int pp = lane; pp < NP; pp += 64) WAT(D.Lf, k * NP + pp) = L.Hp[pp];
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

} // namespace gqp

#endif
