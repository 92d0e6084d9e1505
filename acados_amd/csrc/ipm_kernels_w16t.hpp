/*
 * ipm_kernels_w16t.hpp -- the FACTOR SWEEP of the two-rows family (17 <= nu + nx <= 32: the condensed C3 shape, the nx = 24
 * classes of C5, and with general rows + slacks C4) with its three O(n^3) parts on the FP64 matrix pipe,
 * v_mfma_f64_4x4x4_4b_f64.
 *
 * ky_factor (ipm_kernels_w16r.hpp) feeds every two multiply-adds with one DPP row broadcast: W = [B A]' Lx+, M += W W' and the
 * Cholesky are 28 k of its 58 k cycles per stage at n = 30 (profiles/r03_w16r_phase_cycles.txt), bound by the issue of
 * `v_mov_b64_dpp` + 2 x `v_fma_f64` = 14.3 cycles for 256 flops.  The 4x4x4 MFMA is FOUR INDEPENDENT 4 x 4 x 4 products per
 * instruction at 17-19 cycles for 512 flops (profiles/r04_mfma4x4x4_probe.txt), needs no broadcast at all, and its four
 * blocks are exactly the four instances a wavefront of this family carries.  So this kernel keeps ky_factor's contract --
 * same HBM arrays in and out (the packed factor Lf / lf the other three sweeps read), same LDS-DMA staging one stage ahead,
 * same per-variable arithmetic -- and changes the mapping inside the stage:
 *
 *   lanes     block b = (lane >> 2) & 3 = instance; inside it x = lane & 3, y = lane >> 4 (mfma4.hpp).  Matrices are 4 x 4
 *             tiles in the D layout (lane (x, y) holds [y][x]); gqp_mfma4(P, Q, C) = C + P' Q.
 *   order     variables inside the kernel: [pad; u; x] with PAD = 4 ceil(n/4) - n leading unit rows, so that the state block
 *             -- what the next (earlier) stage needs of this factor -- is whole tile rows at the END (the Riccati recursion
 *             eliminates u before x: x must come last).
 *   storage   the UPPER tiles MT(J,I), J <= I, of the symmetric stage matrix; the Cholesky runs as M = U'U, U = L':
 *                 U(J,J) from the 4 x 4 diagonal block (below),  U(J,I) = L_JJ^-1 MT(J,I) = gqp_mfma4(G, MT(J,I), 0) with
 *                 G the D-layout tile of L_JJ^-T,  MT(K,I) -= U(J,K)' U(J,I) = gqp_mfma4(-U(J,K), U(J,I), MT(K,I)).
 *   rhs       rides as ONE EXTRA TILE COLUMN (column 0 used): MT(J,NT)[.][0] = m, so U(J,NT) = L^-1 m = l comes out of the same
 *             panel products; W gets the matching extra column from [B A | rb] (w0 = lx+ + Lx+' rb), M's from W w0.
 *   W         WT(C,I) = W(I,C)' = sum_{Q >= C} Lx(Q,C)' BA(Q,I): natural tiles of the previous factor's state block times
 *             tiles of [B A] read straight from the LDS image of [B A]' with swapped indices; M += sum_C WT(C,J)' WT(C,I).
 *             The natural tiles Lx(Q,C) = U(C,Q)' of THIS stage for the next one: one product with the identity tile each.
 *   diagonal  4 x 4 block: four products with 0/1 selector tiles broadcast its rows across the 16-lane rows (row r of the tile
 *             in every lane, indexed by x), ten quad broadcasts give every lane of the block all of L_JJ as scalars, the
 *             4 x 4 Cholesky and the inverse are computed redundantly by the sixteen lanes (four dependent rsqrt chains per
 *             block, independent of the trailing products still in the matrix pipe), each lane then selects its element.
 *   vectors   per-variable work (residuals, box rows, norms) stays ONE VARIABLE PER LANE-SLOT as in ky_factor ("compact":
 *             lane l = x + 4 y of the block owns variables l and l + 16); H v, [B A] v, [B A]' pi+ are tile products on the
 *             vector pipe (operand by x, result by y after a quad sum); three small exchanges through LDS per stage move
 *             vectors between the compact form and the by-x / by-y forms.
 *   waves     nx = 24 (box and GEN): one wave per SIMD at ~480-510 registers, tiles of [B A] re-read from LDS by the W product,
 *             the LDS reads of tile row r + 1 issued in front of the work of row r, the next stage's vectors requested behind the
 *             per-variable work.  The C3 shape: TWO waves per SIMD at 256 registers (W16TShape), and with them HBM-bound like
 *             the three other sweeps of that shape.
 *   GEN       general rows + slacks (NG > 0, C4): the branch-free row functions of ky_factor unchanged (one inequality row per
 *             lane of the block, before the tiles of H are loaded); a general row's a'v is one more tile row of the
 *             vector-pipe product, M += A' diag(gamma) A one more chain of tile products.
 * Same results as ky_factor up to the order of the floating-point sums (ACADOS_AMD_W16T=0 / ACADOS_AMD_W16T_GEN=0 keep ky_factor;
 * both test tiers compare the two).  Measurements: DESIGN.md 4.6, profiles/NOTES.md (round 4).
 */
#ifndef IPM_KERNELS_W16T_HPP_
#define IPM_KERNELS_W16T_HPP_

#include "ipm_kernels_w16r.hpp"
#include "mfma4.hpp"

namespace gqp
{

template <int NX, int NU, int NG = 0>
struct W16TLds
{
    static constexpr int n = NX + NU, NP = n * (n + 1) / 2, NB = n * NX;
    static constexpr int NPD = (n + 3) & ~3, PAD = NPD - n, NT = NPD / 4, NXT = NX / 4, XT0 = NT - NXT;
    /* vector exchange (three vectors at a time), packed H block, [B A]' block (both by LDS-DMA as they lie in memory); GEN: the
     * general rows [D C] of the stage [g][n], four 16-entry row vectors (value in; gamma / gadd / dlam out) and the row index
     * of every lane -- the third exchange vector (b - x+, read back before the row vectors are written) shares the first */
    static constexpr int VA = 0, VB = NPD, VC_OWN = NPD + (NPD > NX ? NPD : NX);
    static constexpr int VSZ = NG > 0 ? ((VC_OWN + 1) & ~1) : ((VC_OWN + NX + 1) & ~1); /* (the block reductions at the end of the kernel reuse the first 16) */
    static constexpr int HR = VSZ, HSZ = (NP + 1) & ~1, BR = HR + HSZ, BSZ = (NB + 1) & ~1;
    static constexpr int GT = BR + BSZ, GN = NG > 0 ? ((NG * n + 1) & ~1) : 0, RW = GT + GN, RI = RW + 64;
    static constexpr int VC = NG > 0 ? RW : VC_OWN;
    static constexpr int SZ = ((NG > 0 ? RI + 8 : GT) + 1) & ~1;
};

/* max / sum over the sixteen lanes of a block (once per launch): through LDS */
__device__ static inline double w16t_bmax(double v, double *vx, int l)
{
    GQP_ROWSYNC();
    vx[l] = v;
    GQP_ROWSYNC();
    double r = vx[0];
    W16_UNROLL for (int j = 1; j < 16; j++) { const double o = vx[j]; r = (o > r || o != o) ? o : r; }
    return r;
}
__device__ static inline double w16t_bsum(double v, double *vx, int l)
{
    GQP_ROWSYNC();
    vx[l] = v;
    GQP_ROWSYNC();
    double r = 0.0;
    W16_UNROLL for (int j = 0; j < 16; j++) r += vx[j];
    return r;
}

/* Waves per SIMD.  The shapes whose tiles are few (the condensed C3 shape: 27 tiles of M, 14 of [B A]) fit 256 registers
 * without scratch when nothing is held that can be read again -- tiles of [B A] from LDS in the W product, one tile row of LDS
 * reads in flight, the next stage's vectors requested behind the per-variable work -- and then TWO waves share a SIMD: the
 * dependent chains of the diagonal blocks and the LDS / memory round trips of one wave are the other one's issue slots (what
 * ky_factor never reached: at the 256-register line it spilled and lost).  W16T_ONE_WAVE keeps one wave per SIMD with
 * everything in registers (development builds: the A/B).  nx = 24 needs ~480 registers either way. */
template <int NX, int NU, int NG = 0>
struct W16TShape
{
    static constexpr bool SMALL = NG == 0 && (NX / 4) * (((NX + NU + 3) & ~3) / 4 + 1) <= 24;
#if defined(W16T_ONE_WAVE)
    static constexpr bool TWO_WAVES = false;
#else
    static constexpr bool TWO_WAVES = SMALL;
#endif
    static constexpr int WPE = TWO_WAVES ? 2 : 1;
};

#if defined(__HIP_DEVICE_COMPILE__)
#define W16T_WPE(NX, NU, NG) __attribute__((amdgpu_waves_per_eu(W16TShape<NX, NU, NG>::WPE, W16TShape<NX, NU, NG>::WPE)))
#else
#define W16T_WPE(NX, NU, NG)
#endif
template <int NX, int NU, int NG = 0>
__global__ void __launch_bounds__(64) W16T_WPE(NX, NU, NG) kt_factor(GqpDev D, GqpOpts O, int redo)
{
    GQP_DYN_SHARED(smem);
    typedef W16TLds<NX, NU, NG> LY;
    constexpr bool GEN = NG > 0; /* general rows and slacks (one slack per row): every inequality row through w16r_row_factor, as ky_factor */
    constexpr int NGP = (NG * (NX + NU) + 15) / 16;
    static_assert(NX % 4 == 0, "kt_factor: the state block must be whole 4 x 4 tile rows");
    constexpr int n = NX + NU, R = (n + 15) / 16, NP = LY::NP, NB = LY::NB, PAD = LY::PAD, NT = LY::NT, NXT = LY::NXT, XT0 = LY::XT0;
    const int lane = threadIdx.x & 63, x = lane & 3, y = lane >> 4, bq = (lane >> 2) & 3;
    int l = x + 4 * y; /* position in the block: owner of variables l, l + 16 */
    /* liveness per block, as ky_factor: no block leaves while another one of the wave is alive (all 64 lanes take part in the
     * LDS-DMA and in every MFMA); a dead block computes on whatever its LDS tile holds and writes nothing */
    const int inst0 = blockIdx.x * 4;
    bool aq[4], any = false, alive = false;
    int iq[4], inst = 0;
    W16_UNROLL for (int q = 0; q < 4; q++)
    {
        const int ir = w16_slot_inst(D, inst0 + q);
        iq[q] = ir >= 0 ? ir : D.B - 1;
        aq[q] = ir >= 0 && D.status[iq[q]] == GQP_RUNNING;
        any = any || aq[q];
        if (q == bq) { alive = aq[q]; inst = iq[q]; }
    }
    if (!any) return;
    double *T = smem + bq * LY::SZ, *VXA = T + LY::VA, *VXB = T + LY::VB, *VXC = T + LY::VC;
    double *HRq = T + LY::HR, *BRq = T + LY::BR;
    double *GTq = T + LY::GT;     /* GEN: general rows of the stage, [g][n] */
    double *RW = T + LY::RW;      /* GEN: row vectors, 4 x 16 */
    int *RI = (int *) (T + LY::RI); /* GEN: inequality row handled by every lane */
    int row[R], cx[R];
    bool mine[R], isx[R];
    W16_UNROLL for (int s = 0; s < R; s++) row[s] = l + 16 * s;

    auto dma_h = [&](int kk)
    {
        W16R_LDS_DRAIN();
        W16_UNROLL for (int q = 0; q < 4; q++)
            if (aq[q])
                w16r_dma_region<NP / 2>(D.RSQ.p + (size_t) iq[q] * (size_t) D.RSQ.E + (size_t) kk * NP, smem + q * LY::SZ + LY::HR);
    };
    auto dma_b = [&](int kk)
    {
        W16R_LDS_DRAIN();
        W16_UNROLL for (int q = 0; q < 4; q++)
            if (aq[q])
                w16r_dma_region<NB / 2>(D.BAt.p + (size_t) iq[q] * (size_t) D.BAt.E + (size_t) kk * NB, smem + q * LY::SZ + LY::BR);
    };
    /* vectors and box rows of a stage in compact form, one stage ahead (ky_factor's prefetch) */
    double p_v[R], p_g[R], p_b[R], p_xn[R], p_pin[R], p_pik[R], p_ll[R], p_lu[R], p_tl[R], p_tu[R], p_dl[R], p_du[R];
    double p_ht = 0.0, p_bt = 0.0;
    uint64_t p_am, c_bm, c_em, n_bm, n_em;
    int c_nb, c_oct, n_nb, n_oct, c_ng = 0, c_og = 0, n_ng = 0, n_og = 0, c_ns = 0, c_os = 0, n_ns = 0, n_os = 0;
    double p_G[NGP > 0 ? NGP : 1];
    auto load_desc = [&](int kk)
    {
        GQP_STAGE_REF Sn = D.st[kk];
        n_bm = Sn.bmask; n_em = Sn.emask; n_nb = Sn.nb; n_oct = Sn.o_ct;
        if (GEN) { n_ng = Sn.ng; n_og = Sn.o_g; n_ns = Sn.ns; n_os = Sn.o_s; }
    };
    auto prefetch_v = [&](int kk)
    {
        p_am = WAT(D.amask, kk * D.AW);
        if (NP & 1) p_ht = WAT(D.RSQ, kk * NP + NP - 1);
        if (NB & 1) p_bt = WAT(D.BAt, kk * NB + NB - 1);
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            const bool mn = row[s] < n, ix = row[s] >= NU && row[s] < n;
            const int lc = mn ? row[s] : 0, xc = ix ? row[s] - NU : 0;
            p_v[s] = WAT(D.ux, kk * n + lc);
            p_g[s] = WAT(D.rq, kk * n + lc);
            p_b[s] = WAT(D.bvec, kk * NX + xc);
            p_xn[s] = WAT(D.ux, (kk + 1) * n + NU + xc);
            p_pin[s] = WAT(D.pi, (kk + 1) * NX + xc);
            p_pik[s] = WAT(D.pi, kk * NX + xc);
            if (!GEN)
            {
                const bool hs = mn && (((c_bm & ~c_em) >> row[s]) & 1);
                const int ib = hs ? popc64(c_bm & (((uint64_t) 1 << row[s]) - 1)) : 0;
                const int el = c_oct + ib, eu = el + c_nb;
                p_ll[s] = WAT(D.lam, el); p_lu[s] = WAT(D.lam, eu);
                p_tl[s] = WAT(D.t, el); p_tu[s] = WAT(D.t, eu);
                p_dl[s] = WAT(D.dvec, el); p_du[s] = WAT(D.dvec, eu);
            }
        }
        if (GEN)
        {
            W16_UNROLL for (int i = 0; i < NGP; i++)
            {
                const int e = l + 16 * i;
                p_G[i] = WAT(D.DCt, c_og * n + (e < c_ng * n ? e : 0));
            }
        }
    };
    load_desc(D.N);
    c_bm = n_bm; c_em = n_em; c_nb = n_nb; c_oct = n_oct; c_ng = n_ng; c_og = n_og; c_ns = n_ns; c_os = n_os;
    dma_h(D.N);
    dma_b(D.N);
    prefetch_v(D.N);
    load_desc(D.N > 0 ? D.N - 1 : 0);

    /* natural tiles of the state block of the factor of stage k + 1 (Q >= C) and its rhs part by y (lanes x == 0) */
    double Lx[NXT][NXT], lxy[NXT];
    W16_UNROLL for (int q = 0; q < NXT; q++)
    {
        lxy[q] = 0.0;
        W16_UNROLL for (int c = 0; c < NXT; c++) Lx[q][c] = 0.0;
    }
    /* selector tiles: row r all ones (row broadcast of a diagonal block), identity (transposition) */
    const double e4[4] = {y == 0 ? 1.0 : 0.0, y == 1 ? 1.0 : 0.0, y == 2 ? 1.0 : 0.0, y == 3 ? 1.0 : 0.0};
    const double i4 = x == y ? 1.0 : 0.0;
    double nrm_g = 0.0, nrm_b = 0.0, nrm_d = 0.0, nrm_m = 0.0, musum = 0.0, obj = 0.0, nact = 0.0;
    GQP_TICK_INIT();
    for (int k = D.N; k >= 0; k--)
    {
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            W16R_OPAQUE(row[s]);
            mine[s] = row[s] < n;
            isx[s] = row[s] >= NU && row[s] < n;
            cx[s] = isx[s] ? row[s] - NU : 0;
        }
        /* (the lane's base addresses in the ~15 arrays of the stage are loop invariant: hoisted, they live across the register
         * peak and are spilled -- recomputed per stage instead, as in the GEN instantiations of ky_factor) */
        W16R_OPAQUE(inst); W16R_OPAQUE(l);
        /* (GEN: ky_factor loads the vectors and the general rows of the stage at this point, one exposed latency per stage -- its
         * register file has no room for values that live across a stage.  Here they are requested a stage ahead like the box
         * shapes', behind the per-variable work, where the row functions' registers are free again) */
        W16R_DMA_WAIT(); /* everything issued for this stage has landed */
        W16R_TICK(0);
        const uint64_t bmask = c_bm, emask = c_em, imask = bmask & ~emask, am = p_am;
        const int nbg = c_nb, o_ct = c_oct;
        const int ng = GEN ? c_ng : 0;
        const W16Dsc dsc = {c_nb, ng, c_ns, c_oct, c_os};
        const int nbf = popc64(imask); /* box rows that take part (equality-flagged ones do not) */
        double v[R], g[R], pik[R], q_ll[R], q_lu[R], q_tl[R], q_tu[R], q_dl[R], q_du[R];
        bool fixed[R];
        /* ---- exchange 1: v by x, pi+ by x, b - x+ by y ---- */
        GQP_ROWSYNC();
        if (PAD > 0)
        {
            double zero = 0.0;
            W16R_OPAQUE(zero); /* (materialised here: as a loop-invariant constant it was the one value spilled at 256 registers) */
            if (l < PAD) VXA[l] = zero;
        }
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            fixed[s] = mine[s] && ((emask >> row[s]) & 1);
            v[s] = mine[s] ? p_v[s] : 0.0;
            g[s] = mine[s] ? p_g[s] : 0.0;
            pik[s] = isx[s] ? p_pik[s] : 0.0;
            if (!GEN) { q_ll[s] = p_ll[s]; q_lu[s] = p_lu[s]; q_tl[s] = p_tl[s]; q_tu[s] = p_tu[s]; q_dl[s] = p_dl[s]; q_du[s] = p_du[s]; }
            if (mine[s]) VXA[PAD + row[s]] = v[s];
            if (isx[s]) { VXB[cx[s]] = p_pin[s]; VXC[cx[s]] = p_b[s] - p_xn[s]; }
        }
        if ((NP & 1) || (NB & 1))
        {
            if (l == 0)
            {
                if (NP & 1) HRq[NP - 1] = p_ht;
                if (NB & 1) BRq[NB - 1] = p_bt;
            }
        }
        if (GEN)
            W16_UNROLL for (int i = 0; i < NGP; i++)
            {
                const int e = l + 16 * i;
                if (e < NG * n) GTq[e] = e < ng * n ? p_G[i] : 0.0;
            }
        GQP_ROWSYNC();
        /* the descriptor loaded a stage ago becomes the next stage's (GEN fields) */
        const int x_ng = n_ng, x_og = n_og, x_ns = n_ns, x_os = n_os;
        /* the descriptor loaded a stage ago becomes the next stage's */
        const uint64_t x_bm = n_bm, x_em = n_em;
        const int x_nb = n_nb, x_oct = n_oct;
        double vx[NT], pix[NXT], rby[NXT];
        W16_UNROLL for (int I = 0; I < NT; I++) vx[I] = VXA[4 * I + x];
        W16_UNROLL for (int Q = 0; Q < NXT; Q++) { pix[Q] = VXB[4 * Q + x]; rby[Q] = VXC[4 * Q + y]; }
        GQP_ROWSYNC(); /* exchange 2 (H v, [B A]' pi+ back to the compact form) fills the same slots tile row by tile row */
        /* ---- GEN: every inequality row -- the sorted box rows that take part, then the general rows -- by ONE lane of the block
         * with the branch-free row functions of ky_factor (residuals, norms, slack elimination), BEFORE the tiles of H are
         * loaded (the row functions keep ~40 values and as many addresses alive).  Row values travel through the row vectors
         * by row index: a box row's v_j from the slot that owns the variable, a general row's a'v as one tile row of the
         * vector-pipe product below (operand by x, quad sum, the result by y = g); the results come back the same way ---- */
        double gtr[R], gar[R], gmr[R]; /* what the inequality rows add to the stationarity residual / gradient / Hessian diagonal */
        W16_UNROLL for (int s = 0; s < R; s++) { gtr[s] = 0.0; gar[s] = 0.0; gmr[s] = 0.0; }
        if (GEN)
        {
            W16_UNROLL for (int s = 0; s < R; s++)
                if (mine[s] && ((imask >> row[s]) & 1))
                {
                    const int jb = popc64(imask & (((uint64_t) 1 << row[s]) - 1));
                    RW[jb] = v[s];
                    RI[jb] = popc64(bmask & (((uint64_t) 1 << row[s]) - 1));
                }
            {
                double acc = 0.0;
                W16_UNROLL for (int I = 0; I < NT; I++)
                {
                    const int c = 4 * I + x - PAD;
                    double a = GTq[(y < NG ? y : 0) * n + (c > 0 ? c : 0)];
                    if (y >= ng || c < 0) a = 0.0;
                    acc += a * vx[I];
                }
                acc = mfma4_qsum(acc);
                if (y < ng) { RW[nbf + y] = acc; RI[nbf + y] = nbg + y; } /* (the four lanes of the quad write the same value) */
            }
            GQP_ROWSYNC();
            const bool hr = l < nbf + ng;
            const int rr = hr ? RI[l] : 0;
            const int sjr = hr ? (int) D.st[k].srev[rr] : -1;
            const W16RowF rf = w16r_row_factor(D, O, inst, alive, dsc, am, hr, rr, sjr, RW[hr ? l : 0], nrm_g, nrm_d, nrm_m, musum, nact, obj);
            GQP_ROWSYNC();
            RW[16 + l] = rf.gam; RW[32 + l] = rf.gadd; RW[48 + l] = rf.dlam;
            GQP_ROWSYNC();
            W16_UNROLL for (int s = 0; s < R; s++)
            {
                const bool has = mine[s] && ((imask >> row[s]) & 1);
                const int jb = has ? popc64(imask & (((uint64_t) 1 << row[s]) - 1)) : 0;
                gmr[s] = has ? RW[16 + jb] : 0.0;
                gar[s] = has ? RW[32 + jb] : 0.0;
                gtr[s] = has ? RW[48 + jb] : 0.0;
            }
            W16_UNROLL for (int g_ = 0; g_ < NG; g_++)
            {
                const int rg_ = nbf + g_ < 16 ? nbf + g_ : 15;
                const double Ag = g_ < ng ? RW[32 + rg_] : 0.0, Lg = g_ < ng ? RW[48 + rg_] : 0.0;
                W16_UNROLL for (int s = 0; s < R; s++)
                {
                    const double ap = mine[s] ? GTq[g_ * n + (mine[s] ? row[s] : 0)] : 0.0;
                    gtr[s] += ap * Lg;
                    gar[s] += ap * Ag;
                }
            }
            W16_UNROLL for (int s = 0; s < R; s++) { W16R_OPAQUE(gtr[s]); W16R_OPAQUE(gar[s]); W16R_OPAQUE(gmr[s]); }
        }

        W16R_TICK(1);
        /* ---- tiles of H from the packed block: MT(J,I), J <= I, and H v by tile row (operand by x, partial sums per lane) ----
         * ROOMY (the C3 shape: 60 tiles, 380 registers): all LDS reads of the three products below may be in flight at once and
         * the quad sums are taken together at the end; at nx = 24 the register file is full -- one tile row in flight at a
         * time (fences), its quad sum and its store right behind it.  Two accumulator chains per tile row either way. */
        constexpr bool BA_REG = W16TShape<NX, NU, NG>::SMALL && !W16TShape<NX, NU, NG>::TWO_WAVES, ROOMY = BA_REG;
        /* PIPE1 (one wave per SIMD, full register file: nx = 24): the LDS reads of tile row r + 1 are issued in front of the
         * multiply-adds of tile row r -- with a fence per row and nothing in flight across it every one of the 22 rows of the three
         * products exposed an LDS round trip */
        constexpr bool PIPE1 = !ROOMY && !W16TShape<NX, NU, NG>::TWO_WAVES;
        double MT[NT][NT], MTr[NT], hacc[ROOMY ? NT : 1], bacc[ROOMY ? NT : 1], racc[ROOMY ? NXT : 1];
        auto h_tile = [&](int J, int I) -> double
        {
            const int r = 4 * J + y - PAD, c = 4 * I + x - PAD; /* natural indices; negative: padding (unit diagonal) */
            const int rr = r > 0 ? r : 0, cc = c > 0 ? c : 0;
            const int e = J < I ? PK(cc, rr) : (J > I ? PK(rr, cc) : (rr >= cc ? PK(rr, cc) : PK(cc, rr)));
            double h = HRq[e];
            if (PAD > 0 && (J == 0 || I == 0)) h = (r < 0 || c < 0) ? (r == c ? 1.0 : 0.0) : h;
            return h;
        };
        {
            double cur[PIPE1 ? NT : 1], nxt[PIPE1 ? NT : 1];
            if (PIPE1)
                W16_UNROLL for (int I = 0; I < NT; I++) cur[I] = h_tile(0, I);
            W16_UNROLL for (int J = 0; J < NT; J++)
            {
                if (PIPE1 && J + 1 < NT)
                    W16_UNROLL for (int I = 0; I < NT; I++) nxt[I] = h_tile(J + 1, I);
                if (!ROOMY) W16R_FENCE();
                double acc0 = 0.0, acc1 = 0.0;
                W16_UNROLL for (int I = 0; I < NT; I++)
                {
                    const double h = PIPE1 ? cur[I] : h_tile(J, I);
                    if (J <= I) MT[J][I] = h;
                    if (I & 1) acc1 += h * vx[I]; else acc0 += h * vx[I];
                }
                if (ROOMY) hacc[J] = acc0 + acc1;
                else
                {
                    const double t = mfma4_qsum(acc0 + acc1);
                    if (x == 0) VXA[4 * J + y] = t;
                }
                if (PIPE1)
                    W16_UNROLL for (int I = 0; I < NT; I++) cur[I] = nxt[I];
            }
        }
        if (k > 0) dma_h(k - 1);
        W16R_TICK(2);
        /* ---- tiles of [B A] (rows: next state, columns: variables) from the image of [B A]'; [B A] v by state tile row ---- */
        /* (kept in registers for the W product where they are few -- the condensed C3 shape: 14 tiles; at nx = 24 they are 54
         * and the register file is full: there the W product reads them again from LDS, row of tiles by row of tiles, and
         * the DMA of the next stage's block waits until it is done) */
        auto ba_tile = [&](int Q, int I) -> double
        {
            const int c = 4 * I + x - PAD;
            double bv = BRq[(c > 0 ? c : 0) * NX + 4 * Q + y];
            if (PAD > 0 && I == 0) bv = c < 0 ? 0.0 : bv;
            return bv;
        };
        auto bt_tile = [&](int I, int Q) -> double /* tiles of [B A]' as they lie: rows = variables */
        {
            const int r = 4 * I + y - PAD;
            double bv = BRq[(r > 0 ? r : 0) * NX + 4 * Q + x];
            if (PAD > 0 && I == 0) bv = r < 0 ? 0.0 : bv;
            return bv;
        };
        double BA[BA_REG ? NXT : 1][NT + 1];
        {
            double cur[PIPE1 ? NT : 1], nxt[PIPE1 ? NT : 1];
            if (PIPE1)
                W16_UNROLL for (int I = 0; I < NT; I++) cur[I] = ba_tile(0, I);
            W16_UNROLL for (int Q = 0; Q < NXT; Q++)
            {
                if (PIPE1 && Q + 1 < NXT)
                    W16_UNROLL for (int I = 0; I < NT; I++) nxt[I] = ba_tile(Q + 1, I);
                if (!ROOMY) W16R_FENCE();
                double acc0 = 0.0, acc1 = 0.0;
                W16_UNROLL for (int I = 0; I < NT; I++)
                {
                    const double bv = PIPE1 ? cur[I] : ba_tile(Q, I);
                    if (BA_REG) BA[Q][I] = bv;
                    if (I & 1) acc1 += bv * vx[I]; else acc0 += bv * vx[I];
                }
                if (ROOMY) racc[Q] = acc0 + acc1;
                else rby[Q] += mfma4_qsum(acc0 + acc1); /* b - x+ + [B A] v, the same in the four lanes of a quad */
                if (PIPE1)
                    W16_UNROLL for (int I = 0; I < NT; I++) cur[I] = nxt[I];
            }
        }
        W16R_TICK(3);
        /* [B A]' pi+ by variable tile row: tiles of [B A]' as they lie, operand by x */
        {
            double cur[PIPE1 ? NXT : 1], nxt[PIPE1 ? NXT : 1];
            if (PIPE1)
                W16_UNROLL for (int Q = 0; Q < NXT; Q++) cur[Q] = bt_tile(0, Q);
            W16_UNROLL for (int I = 0; I < NT; I++)
            {
                if (PIPE1 && I + 1 < NT)
                    W16_UNROLL for (int Q = 0; Q < NXT; Q++) nxt[Q] = bt_tile(I + 1, Q);
                if (!ROOMY) W16R_FENCE();
                double acc0 = 0.0, acc1 = 0.0;
                W16_UNROLL for (int Q = 0; Q < NXT; Q++)
                {
                    const double bv = PIPE1 ? cur[Q] : bt_tile(I, Q);
                    if (Q & 1) acc1 += bv * pix[Q]; else acc0 += bv * pix[Q];
                }
                if (ROOMY) bacc[I] = acc0 + acc1;
                else
                {
                    const double t = mfma4_qsum(acc0 + acc1);
                    if (x == 0) VXB[4 * I + y] = t;
                }
                if (PIPE1)
                    W16_UNROLL for (int Q = 0; Q < NXT; Q++) cur[Q] = nxt[Q];
            }
        }
        if (ROOMY)
        {
            W16_UNROLL for (int J = 0; J < NT; J++) { hacc[J] = mfma4_qsum(hacc[J]); bacc[J] = mfma4_qsum(bacc[J]); }
            W16_UNROLL for (int Q = 0; Q < NXT; Q++) rby[Q] += mfma4_qsum(racc[Q]);
            if (x == 0)
                W16_UNROLL for (int J = 0; J < NT; J++) { VXA[4 * J + y] = hacc[J]; VXB[4 * J + y] = bacc[J]; }
        }
        W16R_TICK(4);
        /* next stage: its [B A]' block by DMA; vectors, box rows and the descriptor after it into registers.  Where the register
         * file has room (the C3 shape) right here, as in ky_factor: the rest of the stage is ~6 k cycles there, less than a
         * loaded HBM round trip (requested behind the per-variable work the factor launch took 1.21 instead of 1.09 ms); at
         * nx = 24 the values in flight would be spilled, and the W product, the update of M and the Cholesky (> 12 k cycles)
         * lie between the later request and its use */
        constexpr bool PF_EARLY = BA_REG;
        auto next_stage_vectors = [&]()
        {
            c_bm = x_bm; c_em = x_em; c_nb = x_nb; c_oct = x_oct; c_ng = x_ng; c_og = x_og; c_ns = x_ns; c_os = x_os;
            prefetch_v(k - 1);
            load_desc(k > 1 ? k - 2 : 0);
        };
        if (BA_REG && k > 0) dma_b(k - 1);
        if (PF_EARLY && k > 0) next_stage_vectors();
        W16R_TICK(5);
        /* rb: norm and store by y (lanes x == 0) */
        W16_UNROLL for (int Q = 0; Q < NXT; Q++)
            if (x == 0)
            {
                nacc(nrm_b, rby[Q]);
                if (alive) WAT(D.rb, k * NX + 4 * Q + y) = rby[Q];
            }
        /* ---- exchange 2 complete: the per-variable work in the compact form ---- */
        GQP_ROWSYNC();
        double gt[R], gadd[R], gam[R], m[R];
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            const int lc = mine[s] ? row[s] : 0;
            const double hv = VXA[PAD + lc], bp = VXB[PAD + lc];
            gt[s] = 0.0; gadd[s] = 0.0; gam[s] = 0.0;
            if (mine[s])
            {
                obj += (0.5 * hv + g[s]) * v[s];
                gt[s] = bp + hv + g[s] - pik[s];
            }
            const bool has = !GEN && mine[s] && ((imask >> row[s]) & 1);
            if (GEN)
            {
                gt[s] -= gtr[s];
                gadd[s] = gar[s]; gam[s] = gmr[s];
            }
            if (has)
            {
                const int ib = popc64(bmask & (((uint64_t) 1 << row[s]) - 1));
                const bool al = (am >> ib) & 1, au = (am >> (nbg + ib)) & 1;
                const int el = o_ct + ib, eu = el + nbg;
                const double ll = al ? q_ll[s] : 0.0, lu = au ? q_lu[s] : 0.0;
                const double ttl = al ? q_tl[s] : 1.0, ttu = au ? q_tu[s] : 1.0;
                const double lbv = al ? q_dl[s] : 0.0, ubv = au ? q_du[s] : 0.0;
                const double rdl = al ? v[s] - lbv - ttl : 0.0, rdu = au ? ubv - v[s] - ttu : 0.0;
                const double rml = al ? ll * ttl - O.tau_min : 0.0, rmu = au ? lu * ttu - O.tau_min : 0.0;
                nacc(nrm_d, rdl); nacc(nrm_d, rdu); nacc(nrm_m, rml); nacc(nrm_m, rmu);
                musum += ll * ttl + lu * ttu;
                nact += (double) ((int) al + (int) au);
                gt[s] -= ll - lu;
                const double itl = frcp(ttl), itu = frcp(ttu);
                gam[s] = ll * itl + lu * itu;
                gadd[s] = (rml + ll * rdl) * itl - (rmu + lu * rdu) * itu;
                if (alive)
                {
                    WAT(D.rd, el) = rdl;
                    WAT(D.rd, eu) = rdu;
                }
            }
            if (fixed[s]) gt[s] = 0.0;
            if (mine[s]) { nacc(nrm_g, gt[s]); if (alive) WAT(D.rg, k * n + row[s]) = gt[s]; }
            m[s] = (fixed[s] || !mine[s]) ? 0.0 : gt[s] + gadd[s];
        }
        if (!PF_EARLY && k > 0) next_stage_vectors();
        W16R_TICK(6);
        /* ---- exchange 3: m and the diagonal terms by y ---- */
        GQP_ROWSYNC();
        if (PAD > 0 && l < PAD) { VXA[l] = 0.0; VXB[l] = 0.0; }
        W16_UNROLL for (int s = 0; s < R; s++)
            if (mine[s]) { VXA[PAD + row[s]] = m[s]; VXB[PAD + row[s]] = O.reg_prim + gam[s]; }
        GQP_ROWSYNC();
        W16_UNROLL for (int J = 0; J < NT; J++)
        {
            MTr[J] = x == 0 ? VXA[4 * J + y] : 0.0;
            MT[J][J] += x == y ? VXB[4 * J + y] : 0.0;
        }
        W16R_TICK(7);
        /* ---- W' tile row by tile row and M += W W': WT(C,I) = sum_{Q >= C} Lx(Q,C)' BA(Q,I); the rhs column of [B A] is rb,
         * that of W' gets lx+ added: w0 = lx+ + Lx+' rb ---- */
        double rbt[NXT]; /* the rhs column of [B A | rb] */
        W16_UNROLL for (int Q = 0; Q < NXT; Q++) rbt[Q] = x == 0 ? rby[Q] : 0.0;
        /* (!BA_REG: the tiles of [B A] come from LDS one tile row ahead of the products that use them -- the row of the next
         * (C, Q) pair is requested before the products of this one are issued, across the update of M as well; with all reads of a
         * pair in front of its own products every pair exposed an LDS round trip: 21 of them per stage at nx = 24) */
        constexpr bool BA_PIPE = !BA_REG && !W16TShape<NX, NU, NG>::TWO_WAVES; /* (at two waves per SIMD the second row in flight is the spill; the other wave covers the round trip there) */
        double bcur[BA_PIPE ? NT : 1], bnxt[BA_PIPE ? NT : 1];
        if (BA_PIPE)
            W16_UNROLL for (int I = 0; I < NT; I++) bcur[I] = ba_tile(0, I);
        W16_UNROLL for (int C = 0; C < NXT; C++)
        {
            double WT[NT + 1];
            W16_UNROLL for (int I = 0; I <= NT; I++) WT[I] = (I == NT && x == 0) ? lxy[C] : 0.0;
            W16_UNROLL for (int Q = C; Q < NXT; Q++)
            {
                if (BA_PIPE)
                {
                    const int Qn = Q + 1 < NXT ? Q + 1 : C + 1; /* the next pair: (C, Q + 1), or (C + 1, C + 1) behind the last row */
                    if (Qn < NXT)
                        W16_UNROLL for (int I = 0; I < NT; I++) bnxt[I] = ba_tile(Qn, I);
                    W16R_FENCE();
                }
                else if (!BA_REG) W16R_FENCE(); /* (the tiles of one row of [B A] in flight, not all of them) */
                W16_UNROLL for (int I = 0; I < NT; I++) WT[I] = gqp_mfma4(Lx[Q][C], BA_REG ? BA[Q][I] : (BA_PIPE ? bcur[I] : ba_tile(Q, I)), WT[I]);
                WT[NT] = gqp_mfma4(Lx[Q][C], rbt[Q], WT[NT]);
                if (BA_PIPE)
                    W16_UNROLL for (int I = 0; I < NT; I++) bcur[I] = bnxt[I];
            }
            W16_UNROLL for (int J = 0; J < NT; J++)
            {
                W16_UNROLL for (int I = J; I < NT; I++) MT[J][I] = gqp_mfma4(WT[J], WT[I], MT[J][I]);
                MTr[J] = gqp_mfma4(WT[J], WT[NT], MTr[J]);
            }
        }
        if (GEN)
        {
            /* general rows: M += A' diag(gamma) A as one more chain of tile products -- A(0,I)[y][x] = a_y[4I + x - PAD] from the
             * LDS copy of [D C], gamma_g still in the row vector */
            double At[NT], Ag[NT];
            const int rg_ = nbf + y < 16 ? nbf + y : 15;
            const double gmy = y < ng ? RW[16 + rg_] : 0.0;
            W16_UNROLL for (int I = 0; I < NT; I++)
            {
                const int c = 4 * I + x - PAD;
                double a = GTq[(y < NG ? y : 0) * n + (c > 0 ? c : 0)];
                if (y >= ng || c < 0) a = 0.0;
                At[I] = a; Ag[I] = gmy * a;
            }
            W16_UNROLL for (int J = 0; J < NT; J++)
                W16_UNROLL for (int I = J; I < NT; I++) MT[J][I] = gqp_mfma4(At[J], Ag[I], MT[J][I]);
        }
        W16R_TICK(8);
        if (!BA_REG && k > 0) dma_b(k - 1);
        if (emask) /* uniform: only a stage with fixed variables pays for the masking */
        {
            W16_UNROLL for (int J = 0; J < NT; J++)
            {
                const int r = 4 * J + y - PAD;
                const bool fr = r >= 0 && ((emask >> r) & 1);
                W16_UNROLL for (int I = J; I < NT; I++)
                {
                    const int c = 4 * I + x - PAD;
                    const bool fc = c >= 0 && ((emask >> c) & 1);
                    if (fr || fc) MT[J][I] = r == c ? 1.0 : 0.0;
                }
                if (fr) MTr[J] = 0.0;
            }
        }

        W16R_TICK(9);
        /* ---- blocked Cholesky M = U'U on the upper tiles, the rhs column riding along.  One wave per SIMD: nothing hides the
         * dependent chain of a diagonal block (four rsqrt refinements, ten quad broadcasts: ~1,000 cycles) but this wave's own
         * independent work -- so the loop is skewed by hand: step J issues the ONE panel product and the ONE trailing product
         * the next diagonal block needs, broadcasts its rows, and only then the rest of its panel / trailing products, the
         * stores of its finished row of U and the transposes for the next stage, all of which the chain of block J + 1 does
         * not depend on and the scheduler may run beside it. ---- */
        auto rows4 = [&](double dg, double &r0, double &r1, double &r2, double &r3)
        {
            /* rows of the diagonal block to every 16-lane row: r_j = row j of the block, indexed by x */
            r0 = gqp_mfma4(e4[0], dg, 0.0); r1 = gqp_mfma4(e4[1], dg, 0.0); r2 = gqp_mfma4(e4[2], dg, 0.0); r3 = gqp_mfma4(e4[3], dg, 0.0);
        };
        auto diag4 = [&](double r0, double r1, double r2, double r3, double &ut, double &G)
        {
            /* 4 x 4 Cholesky, every lane for its x: lxj = L[x][j]; a non-positive pivot zeroes its column (as ky_factor; the
             * NaN rsqrt returns for it is dropped by the select) */
            const double d0 = mfma4_qbc<0>(r0);
            double t0 = frsqrt(d0);
            W16R_OPAQUE(t0); /* (pinned: the select below must stay a select -- as a branch around the refinement it cuts the chain into basic blocks the products cannot be scheduled into) */
            const double i0 = d0 > 0.0 ? t0 : 0.0;
            const double lx0 = r0 * i0;
            const double l10 = mfma4_qbc<1>(lx0), l20 = mfma4_qbc<2>(lx0), l30 = mfma4_qbc<3>(lx0);
            const double s1 = r1 - l10 * lx0;
            const double d1 = mfma4_qbc<1>(s1);
            double t1 = frsqrt(d1);
            W16R_OPAQUE(t1);
            const double i1 = d1 > 0.0 ? t1 : 0.0;
            const double lx1 = s1 * i1;
            const double l21 = mfma4_qbc<2>(lx1), l31 = mfma4_qbc<3>(lx1);
            const double s2 = (r2 - l20 * lx0) - l21 * lx1;
            const double d2 = mfma4_qbc<2>(s2);
            double t2 = frsqrt(d2);
            W16R_OPAQUE(t2);
            const double i2 = d2 > 0.0 ? t2 : 0.0;
            const double lx2 = s2 * i2;
            const double l32 = mfma4_qbc<3>(lx2);
            const double s3 = ((r3 - l30 * lx0) - l31 * lx1) - l32 * lx2;
            const double d3 = mfma4_qbc<3>(s3);
            double t3 = frsqrt(d3);
            W16R_OPAQUE(t3);
            const double i3 = d3 > 0.0 ? t3 : 0.0;
            const double lx3 = s3 * i3;
            /* U_JJ = L_JJ': [y][x] = L[x][y], x >= y */
            const double uj = y == 0 ? lx0 : (y == 1 ? lx1 : (y == 2 ? lx2 : lx3));
            ut = x >= y ? uj : 0.0;
            /* G = L_JJ^-T: [y][x] = Linv[x][y], x >= y (every lane holds all of L_JJ; a zeroed column has a zero row / column here) */
            const double n10 = -l10 * i0 * i1, n21 = -l21 * i1 * i2, n32 = -l32 * i2 * i3;
            const double n20 = -(l20 * i0 + l21 * n10) * i2, n31 = -(l31 * i1 + l32 * n21) * i3;
            const double n30 = -(l30 * i0 + l31 * n10 + l32 * n20) * i3;
            const double gc0 = x == 0 ? i0 : (x == 1 ? n10 : (x == 2 ? n20 : n30));
            const double gc1 = x == 1 ? i1 : (x == 2 ? n21 : n31);
            const double gc2 = x == 2 ? i2 : n32;
            const double gsel = y == 0 ? gc0 : (y == 1 ? gc1 : (y == 2 ? gc2 : i3));
            G = x >= y ? gsel : 0.0;
        };
        double G, ut;
        {
            double r0, r1, r2, r3;
            rows4(MT[0][0], r0, r1, r2, r3);
            diag4(r0, r1, r2, r3, ut, G);
        }
        W16_UNROLL for (int J = 0; J < NT; J++)
        {
            MT[J][J] = ut;
            double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0, nu1 = 0.0;
            if (J + 1 < NT)
            {
                /* what the next diagonal block waits for: U(J,J+1) = L_JJ^-1 MT(J,J+1), MT(J+1,J+1) -= U(J,J+1)' U(J,J+1) */
                MT[J][J + 1] = gqp_mfma4(G, MT[J][J + 1], 0.0);
                nu1 = -MT[J][J + 1];
                MT[J + 1][J + 1] = gqp_mfma4(nu1, MT[J][J + 1], MT[J + 1][J + 1]);
                rows4(MT[J + 1][J + 1], r0, r1, r2, r3);
            }
            /* the rest of the panel: U(J,I) = L_JJ^-1 MT(J,I) */
            W16_UNROLL for (int I = J + 2; I < NT; I++) MT[J][I] = gqp_mfma4(G, MT[J][I], 0.0);
            MTr[J] = gqp_mfma4(G, MTr[J], 0.0);
            /* the rest of the trailing blocks: MT(K,I) -= U(J,K)' U(J,I) */
            W16_UNROLL for (int K = J + 1; K < NT; K++)
            {
                const double nu_ = K == J + 1 ? nu1 : -MT[J][K];
                W16_UNROLL for (int I = (K == J + 1 ? K + 1 : K); I < NT; I++) MT[K][I] = gqp_mfma4(nu_, MT[J][I], MT[K][I]);
                MTr[K] = gqp_mfma4(nu_, MTr[J], MTr[K]);
            }
            /* state block for the next (earlier) stage: natural tiles Lx(Q,C) = U(C,Q)' (one product with the identity), lx by y */
            if (J >= XT0)
            {
                W16_UNROLL for (int Q = J - XT0; Q < NXT; Q++) Lx[Q][J - XT0] = gqp_mfma4(MT[J][XT0 + Q], i4, 0.0);
                lxy[J - XT0] = MTr[J];
            }
            if (J + 1 < NT) diag4(r0, r1, r2, r3, ut, G);
            /* (behind the chain: the predicated stores are basic blocks of their own, the products above and the chain are one)
             * outputs of the finished row: U(J,I)[y][x] = L[4I + x - PAD][4J + y - PAD] into the packed factor; l by y */
            {
                const int c = 4 * J + y - PAD;
                W16_UNROLL for (int I = J; I < NT; I++)
                {
                    const int r = 4 * I + x - PAD;
                    if (alive && c >= 0 && r >= c) WAT(D.Lf, k * NP + PK(r, c)) = MT[J][I];
                }
                if (alive && x == 0 && c >= 0) WAT(D.lf, k * n + c) = MTr[J];
            }
        }
        W16R_TICK(10);
        W16R_TICK(12);
    }

    nrm_g = w16t_bmax(nrm_g, VXA, l); nrm_b = w16t_bmax(nrm_b, VXA, l); nrm_d = w16t_bmax(nrm_d, VXA, l); nrm_m = w16t_bmax(nrm_m, VXA, l);
    musum = w16t_bsum(musum, VXA, l); obj = w16t_bsum(obj, VXA, l);
    const double nact_d = w16t_bsum(nact, VXA, l);
    if (l == 0 && alive)
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
