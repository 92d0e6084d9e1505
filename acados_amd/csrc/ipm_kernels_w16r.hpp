/*
 * ipm_kernels_w16r.hpp -- SIXTEEN LANES PER INSTANCE, SEVERAL ROWS PER LANE: box-constrained stage blocks with
 * 17 <= nu + nx <= 32 (the nx = 24 classes of C5, the condensed shape of C3: nx = 8, nu = 15).
 *
 * The wave-per-instance kernels (ipm_kernels_wpi.hpp) are bound by ONE wave's dependent instruction stream
 * (profiles/r02_wpi_phase_cycles.txt: ~9,400 instructions and ~98,000 cycles per stage at n = 30), most of it LDS
 * round trips, barriers and address arithmetic on run-time dims.  This family keeps the register-row scheme of
 * ipm_kernels_w16.hpp -- four instances per wavefront, one per 16-lane DPP row, `row_newbcast` as the broadcast --
 * and gives every lane R = ceil(n / 16) rows of the stage matrices: lane l of a row owns the variables l, l + 16
 * (slot 0, slot 1).  Variable j lives in lane j & 15, slot j >> 4, so a broadcast of "the value of variable j" is
 * still ONE DPP move with an immediate lane, and one broadcast feeds R multiply-adds.
 *
 * Differences from the one-row family, all of them about registers (n = 30 leaves no room for everything):
 *   - the x-block of the factor of the stage handled before lives in the per-instance LDS tile TA and is read from
 *     there (uniform address per instance for the W product, column reads for Lx+' rb) instead of two register copies;
 *   - the column of the factor the forward sweep needs is read from the LDS tile TF where it is used;
 *   - only the lower triangle is carried: slot 0 (rows < 16) never touches columns >= 16.
 * Same algorithm, HBM arrays and slot conventions as ipm_kernels_w16.hpp / ipm_kernels_wpi.hpp (whose init / finalize
 * kernels serve this family too); box constraints without slacks (soft rows and general rows stay with the
 * wave-per-instance kernels).  Shapes are compile-time.
 * The CPU test tier runs these kernels under tests/hostsim like the one-row family.
 */
#ifndef IPM_KERNELS_W16R_HPP_
#define IPM_KERNELS_W16R_HPP_

#include "ipm_kernels_w16.hpp"

namespace gqp
{

/* per-instance LDS tile (doubles).
 * Factor sweep: exchange buffer (host simulation), pi+ vector, the x-block of the previous factor in CHUNKED rows (row q
 * holds columns 0 .. w(q/8) - 1, w(j) = min(8 (j + 1), NX): the rolled W loop reads a row up to its chunk bound, zero
 * above the diagonal), and a staging region that first carries the packed H block of the stage and then its [B A]'
 * block (leading dimension NX + 1).
 * rhs-only sweep: exchange buffer + the square x-block tile; forward sweep: exchange buffer + the full factor tile. */
template <int NX, int NU>
struct W16RLds
{
    static constexpr int n = NX + NU, R = (n + 15) / 16, LDX = NX + 1, LDF = n + 1, LDB = NX + 1, NP = n * (n + 1) / 2;
    static constexpr int XB = 0, TA = 16, TF = 16;
    static constexpr int SZ_A = 16 + NX * LDX, SZ_F = 16 + n * LDF;
    /* chunked x-block tile */
    static constexpr int NCH = (NX + 7) / 8;
    static constexpr int cw(int j) { return 8 * (j + 1) < NX ? 8 * (j + 1) : NX; }
    static constexpr int crow(int q) { return 32 * (q >> 3) * ((q >> 3) + 1) + (q & 7) * cw(q >> 3); }
    static constexpr int TAC_SZ = 32 * (NCH - 1) * NCH + (NX - 8 * (NCH - 1)) * NX; /* rows of the last chunk are NX wide */
    static constexpr int PV = 16, TAC = PV + 32, STG = TAC + TAC_SZ;
    static constexpr int STG_SZ = NP > n * LDB ? NP : n * LDB;
    static constexpr int SZ_K = STG + STG_SZ;
    static constexpr int SZ0 = SZ_A > SZ_F ? SZ_A : SZ_F;
    static constexpr int SZ = SZ0 > SZ_K ? SZ0 : SZ_K;
};

/* value of variable j: lane j & 15 of the row, slot j >> 4 (j is a compile-time constant after unrolling) */
#define W16R_BC(arr, j) w16_bcast((arr)[(j) >> 4], (j) & 15, xb)
/* does slot s hold a row >= c ?  (compile-time after unrolling: the lower triangle only) */
#define W16R_LOW(s, c) (16 * (s) + 15 >= (c))
/* the fully unrolled stage body is one basic block of several thousand instructions; left alone, the scheduler hoists
 * hundreds of loads and LDS reads to its top and spills.  A fence between the phases keeps each phase's loads inside it */
#if defined(__HIP_DEVICE_COMPILE__)
#define W16R_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define W16R_FENCE() do { } while (0)
#endif

/* ------------------------------------------------------------------------------------------------ factor */

template <int NX, int NU>
__global__ void __launch_bounds__(64) ky_factor(GqpDev D, GqpOpts O, int redo)
{
    GQP_DYN_SHARED(smem);
    typedef W16RLds<NX, NU> LY;
    constexpr int n = NX + NU, R = LY::R, NP = n * (n + 1) / 2, LDB = LY::LDB;
    constexpr int NH = (NP + 15) / 16, NBF = (n * NX + 15) / 16; /* flat 16-lane passes over the two stage blocks */
    const int l = threadIdx.x & 15, inst = blockIdx.x * 4 + (threadIdx.x >> 4);
    if (inst >= D.B) return;
    if (D.status[inst] != GQP_RUNNING) return;
    double *T = smem + (threadIdx.x >> 4) * LY::SZ, *xb = T + LY::XB;
    double *PV = T + LY::PV;   /* pi+ */
    double *TA = T + LY::TAC;  /* x-block of the factor of stage k+1, chunked rows, zero above the diagonal */
    double *SG = T + LY::STG;  /* staging: packed H block, then [B A]' with leading dimension LDB */
    int row[R], cx[R], tar[R];
    bool mine[R], isx[R];
    W16_UNROLL for (int s = 0; s < R; s++)
    {
        row[s] = l + 16 * s;
        mine[s] = row[s] < n;
        isx[s] = row[s] >= NU && row[s] < n;
        cx[s] = isx[s] ? row[s] - NU : 0;
        const int j = cx[s] >> 3, w = 8 * (j + 1) < NX ? 8 * (j + 1) : NX;
        tar[s] = 32 * j * (j + 1) + (cx[s] & 7) * w; /* LY::crow(cx) */
    }
    /* stage N has no successor: an all-zero tile */
    for (int e = l; e < LY::TAC_SZ; e += 16) TA[e] = 0.0;

    double lxn[R]; /* lx+ of this lane's states */
    W16_UNROLL for (int s = 0; s < R; s++) lxn[s] = 0.0;
    double nrm_g = 0.0, nrm_b = 0.0, nrm_d = 0.0, nrm_m = 0.0, musum = 0.0, obj = 0.0, nact = 0.0;

    for (int k = D.N; k >= 0; k--)
    {
        const GqpStage &S = D.st[k];
        const uint64_t imask = S.bmask & ~S.emask;
        const uint64_t am = WAT(D.amask, k * D.AW);
        const int nbg = S.nb;

        /* ---- loads: the packed H block and the [B A]' block of the stage, flat and coalesced over the 16 lanes;
         * rows / columns are then read from the LDS staging region ---- */
        double hb[NH], bb[NBF];
        W16_UNROLL for (int i = 0; i < NH; i++)
        {
            const int e = l + 16 * i;
            hb[i] = WAT(D.RSQ, k * NP + (e < NP ? e : 0));
        }
        W16_UNROLL for (int i = 0; i < NBF; i++)
        {
            const int e = l + 16 * i;
            bb[i] = WAT(D.BAt, k * n * NX + (e < n * NX ? e : 0));
        }
        double M[R][n], v[R], g[R], rb[R], pin[R], pik[R];
        bool fixed[R];
        int lc_[R], xc_[R];
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            lc_[s] = mine[s] ? row[s] : 0;
            xc_[s] = isx[s] ? cx[s] : 0;
            fixed[s] = mine[s] && ((S.emask >> row[s]) & 1);
            const double zm = mine[s] ? 1.0 : 0.0, zx = isx[s] ? 1.0 : 0.0;
            v[s] = zm * WAT(D.ux, k * n + lc_[s]);
            g[s] = zm * WAT(D.rq, k * n + lc_[s]);
            rb[s] = zx * (WAT(D.bvec, k * NX + xc_[s]) - WAT(D.ux, (k + 1) * n + NU + xc_[s]));
            pin[s] = zx * WAT(D.pi, (k + 1) * NX + xc_[s]);
            pik[s] = zx * WAT(D.pi, k * NX + xc_[s]);
        }
        GQP_ROWSYNC(); /* the previous stage is done with the staging region */
        W16_UNROLL for (int i = 0; i < NH; i++)
        {
            const int e = l + 16 * i;
            if (e < NP) SG[e] = hb[i];
        }
        W16_UNROLL for (int s = 0; s < R; s++)
            if (isx[s]) PV[cx[s]] = pin[s];
        GQP_ROWSYNC();
        /* symmetric row of H */
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            const double zm = mine[s] ? 1.0 : 0.0;
            W16_UNROLL for (int c = 0; c < n; c++) M[s][c] = zm * SG[c <= lc_[s] ? PK(lc_[s], c) : PK(c, lc_[s])];
        }
        GQP_ROWSYNC();
        W16_UNROLL for (int i = 0; i < NBF; i++)
        {
            const int e = l + 16 * i;
            if (e < n * NX) SG[(e / NX) * LDB + e % NX] = bb[i];
        }
        GQP_ROWSYNC();

        /* ---- rb += [B A] v (column cx of [B A]'), H v: one broadcast of v per variable serves both ---- */
        double hv[R];
        W16_UNROLL for (int s = 0; s < R; s++) hv[s] = 0.0;
        W16_UNROLL for (int r = 0; r < n; r++)
        {
            const double vr = W16R_BC(v, r);
            W16_UNROLL for (int s = 0; s < R; s++)
            {
                rb[s] += SG[r * LDB + xc_[s]] * vr; /* idle slots: clamped column, value unused */
                hv[s] += M[s][r] * vr;
            }
        }
        /* ---- W rows: W[c] = sum_{q >= c} Br[q] Lx+[q][c], and [B A]' pi+ with the same pass over the row of [B A]'.
         * A ROLLED loop over q (everything indexed by q lives in LDS; no broadcast inside), in chunks of eight rows
         * whose width is the chunk bound: straight-line code here let the scheduler hoist every LDS read of the
         * phase to the top of the stage and spill hundreds of registers ---- */
        double W[R][NX], gt[R], gadd[R], gam[R];
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            gt[s] = 0.0; gadd[s] = 0.0; gam[s] = 0.0;
            W16_UNROLL for (int c = 0; c < NX; c++) W[s][c] = 0.0;
        }
        W16_UNROLL for (int j = 0; j < LY::NCH; j++)
        {
            constexpr int dummy = 0; (void) dummy;
            const int q0 = 8 * j, q1 = 8 * (j + 1) < NX ? 8 * (j + 1) : NX, w = q1;
            const double *TAj = TA + 32 * j * (j + 1);
            _Pragma("unroll 2")
            for (int q = q0; q < q1; q++)
            {
                const double pc = PV[q];
                double Brq[R];
                W16_UNROLL for (int s = 0; s < R; s++)
                {
                    Brq[s] = (mine[s] ? 1.0 : 0.0) * SG[lc_[s] * LDB + q];
                    gt[s] += Brq[s] * pc;
                }
                const double *Tq = TAj + (q - q0) * w;
                W16_UNROLL for (int c = 0; c < NX; c++)
                    if (c < w)
                    {
                        const double Lqc = Tq[c];
                        W16_UNROLL for (int s = 0; s < R; s++) W[s][c] += Brq[s] * Lqc;
                    }
            }
        }
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            if (mine[s])
            {
                obj += (0.5 * hv[s] + g[s]) * v[s];
                gt[s] += hv[s] + g[s] - pik[s];
            }
            else gt[s] = 0.0;
            if (isx[s]) { nacc(nrm_b, rb[s]); WAT(D.rb, k * NX + cx[s]) = rb[s]; }
            const bool has = mine[s] && ((imask >> row[s]) & 1);
            if (has)
            {
                const int ib = popc64(S.bmask & (((uint64_t) 1 << row[s]) - 1));
                const bool al = (am >> ib) & 1, au = (am >> (nbg + ib)) & 1;
                const int el = S.o_ct + ib, eu = el + nbg;
                const double ll = al ? WAT(D.lam, el) : 0.0, lu = au ? WAT(D.lam, eu) : 0.0;
                const double ttl = al ? WAT(D.t, el) : 1.0, ttu = au ? WAT(D.t, eu) : 1.0;
                const double lbv = al ? WAT(D.dvec, el) : 0.0, ubv = au ? WAT(D.dvec, eu) : 0.0;
                const double rdl = al ? v[s] - lbv - ttl : 0.0, rdu = au ? ubv - v[s] - ttu : 0.0;
                const double rml = al ? ll * ttl - O.tau_min : 0.0, rmu = au ? lu * ttu - O.tau_min : 0.0;
                nacc(nrm_d, rdl); nacc(nrm_d, rdu); nacc(nrm_m, rml); nacc(nrm_m, rmu);
                musum += ll * ttl + lu * ttu;
                nact += (double) ((int) al + (int) au);
                gt[s] -= ll - lu;
                const double itl = frcp(ttl), itu = frcp(ttu);
                gam[s] = ll * itl + lu * itu;
                gadd[s] = (rml + ll * rdl) * itl - (rmu + lu * rdu) * itu;
                WAT(D.rd, el) = rdl;
                WAT(D.rd, eu) = rdu;
            }
            if (fixed[s]) gt[s] = 0.0;
            if (mine[s]) { nacc(nrm_g, gt[s]); WAT(D.rg, k * n + row[s]) = gt[s]; }
        }
        /* w0[c] (state slots) = lx+[c] + sum_{q >= c} Lx+[q][c] rb[q]: column c of the chunked tile */
        double w0[R];
        W16_UNROLL for (int s = 0; s < R; s++) w0[s] = lxn[s];
        W16_UNROLL for (int q = 0; q < NX; q++)
        {
            const double rbq = W16R_BC(rb, NU + q);
            W16_UNROLL for (int s = 0; s < R; s++) w0[s] += (q >= cx[s] ? TA[LY::crow(q) + cx[s]] : 0.0) * rbq;
        }
        W16_UNROLL for (int s = 0; s < R; s++)
            if (!isx[s]) w0[s] = 0.0;
        /* m = gt + gadd + W w0 */
        double m[R];
        W16_UNROLL for (int s = 0; s < R; s++) m[s] = gt[s] + gadd[s];
        W16_UNROLL for (int c = 0; c < NX; c++)
        {
            const double wc = W16R_BC(w0, NU + c);
            W16_UNROLL for (int s = 0; s < R; s++) m[s] += W[s][c] * wc;
        }
        W16_UNROLL for (int s = 0; s < R; s++)
            if (fixed[s] || !mine[s]) m[s] = 0.0;
        /* ---- M += W W' + reg + Gamma (lower triangle: slot s needs column c only if it holds a row >= c) ---- */
        W16_UNROLL for (int q = 0; q < NX; q++)
            W16_UNROLL for (int c = 0; c < n; c++)
            {
                const double wcq = w16_bcast(W[c >> 4][q], c & 15, xb);
                W16_UNROLL for (int s = 0; s < R; s++)
                    if (W16R_LOW(s, c)) M[s][c] += W[s][q] * wcq;
            }
        W16_UNROLL for (int s = 0; s < R; s++)
            W16_UNROLL for (int c = 0; c < n; c++)
                if (W16R_LOW(s, c)) M[s][c] += (c == row[s]) ? O.reg_prim + gam[s] : 0.0;
        if (S.emask) /* uniform: only a stage with fixed variables pays for the masking */
        {
            W16_UNROLL for (int s = 0; s < R; s++)
                W16_UNROLL for (int c = 0; c < n; c++)
                {
                    const bool fc = (S.emask >> c) & 1;
                    if (fixed[s] || fc) M[s][c] = (c == row[s]) ? 1.0 : 0.0;
                }
        }

        /* ---- Cholesky on register rows; the rhs entry m rides along (l = L^{-1} m) ---- */
        W16_UNROLL for (int j = 0; j < n; j++)
        {
            const double d = w16_bcast(M[j >> 4][j], j & 15, xb);
            const bool pos = d > 0.0;
            const double inv0 = frsqrt(pos ? d : 1.0);
            const double inv = pos ? inv0 : 0.0;
            const double lj = W16R_BC(m, j) * inv;
            double lo[R];
            W16_UNROLL for (int s = 0; s < R; s++)
            {
                lo[s] = 0.0;
                if (W16R_LOW(s, j))
                {
                    /* L[row][j] = M[row][j] inv for row >= j (the pivot holds d: d inv = sqrt(d), 0 for a non-positive
                     * pivot); rows above keep their entry */
                    const double Llj = row[s] >= j ? M[s][j] * inv : M[s][j];
                    M[s][j] = Llj;
                    lo[s] = row[s] > j ? Llj : 0.0; /* finished rows take no part in the trailing update */
                    m[s] = row[s] == j ? lj : m[s] - lo[s] * lj;
                }
            }
            W16_UNROLL for (int c = j + 1; c < n; c++)
            {
                const double lc = W16R_BC(lo, c);
                W16_UNROLL for (int s = 0; s < R; s++)
                    if (W16R_LOW(s, c)) M[s][c] -= lo[s] * lc;
            }
        }

        /* ---- outputs ---- */
        W16_UNROLL for (int s = 0; s < R; s++)
            if (mine[s])
            {
                W16_UNROLL for (int c = 0; c < n; c++)
                    if (W16R_LOW(s, c) && c <= row[s]) WAT(D.Lf, k * NP + PK(row[s], c)) = M[s][c];
                WAT(D.lf, k * n + row[s]) = m[s];
            }
        /* x-block for the next (earlier) stage: chunked rows, zero above the diagonal up to the chunk bound */
        GQP_ROWSYNC();
        W16_UNROLL for (int s = 0; s < R; s++)
            if (isx[s])
            {
                const int w = 8 * ((cx[s] >> 3) + 1) < NX ? 8 * ((cx[s] >> 3) + 1) : NX;
                W16_UNROLL for (int c = 0; c < NX; c++)
                    if (c < w) TA[tar[s] + c] = (W16R_LOW(s, NU + c) && c <= cx[s]) ? M[s][NU + c] : 0.0;
            }
        GQP_ROWSYNC();
        W16_UNROLL for (int s = 0; s < R; s++) lxn[s] = isx[s] ? m[s] : 0.0;
    }

    nrm_g = w16_rmax(nrm_g, xb); nrm_b = w16_rmax(nrm_b, xb); nrm_d = w16_rmax(nrm_d, xb); nrm_m = w16_rmax(nrm_m, xb);
    musum = w16_rsum(musum, xb); obj = w16_rsum(obj, xb);
    const double nact_d = w16_rsum(nact, xb);
    if (l == 0)
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

/* ------------------------------------------------------------------------------- rhs-only backward (p-form) */

template <int NX, int NU>
__global__ void __launch_bounds__(64) ky_backrhs(GqpDev D, GqpOpts O, int redo)
{
    GQP_DYN_SHARED(smem);
    typedef W16RLds<NX, NU> LY;
    constexpr int n = NX + NU, R = LY::R, NP = n * (n + 1) / 2, LDX = LY::LDX;
    const int l = threadIdx.x & 15, inst = blockIdx.x * 4 + (threadIdx.x >> 4);
    if (inst >= D.B) return;
    if (D.status[inst] != GQP_RUNNING) return;
    if (redo == 1 && !(D.alpha[inst] < 0.0)) return; /* redo = 2: sensitivity pass (direction only, every instance) */
    double *T = smem + (threadIdx.x >> 4) * LY::SZ, *xb = T + LY::XB, *TA = T + LY::TA;
    int row[R], cx[R];
    bool mine[R], isx[R];
    W16_UNROLL for (int s = 0; s < R; s++)
    {
        row[s] = l + 16 * s;
        mine[s] = row[s] < n;
        isx[s] = row[s] >= NU && row[s] < n;
        cx[s] = isx[s] ? row[s] - NU : 0;
    }
    const double smu = D.smu[inst];
    const double pscale = redo == 1 ? 0.0 : 1.0;
    for (int e = l; e < NX * LDX; e += 16) TA[e] = 0.0;
    GQP_ROWSYNC();
    double pn[R]; /* p of the stage handled before (state slots) */
    W16_UNROLL for (int s = 0; s < R; s++) pn[s] = 0.0;

    for (int k = D.N; k >= 0; k--)
    {
        const GqpStage &S = D.st[k];
        const uint64_t imask = S.bmask & ~S.emask;
        const uint64_t am = WAT(D.amask, k * D.AW);
        const int nbg = S.nb;
        /* row of the factor: the first NU columns and, for state slots, the x-block; row of [B A]' */
        double Lu[R][NU > 0 ? NU : 1], Lx[R][NX], Br[R][NX], rb[R], m[R];
        bool fixed[R];
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            const int lc_ = mine[s] ? row[s] : 0, xl_ = isx[s] ? row[s] : NU, xc_ = isx[s] ? cx[s] : 0;
            const double zm = mine[s] ? 1.0 : 0.0, zx = isx[s] ? 1.0 : 0.0;
            fixed[s] = mine[s] && ((S.emask >> row[s]) & 1);
            W16_UNROLL for (int c = 0; c < NU; c++) Lu[s][c] = (c <= lc_ ? zm : 0.0) * WAT(D.Lf, k * NP + PK(lc_, (c <= lc_ ? c : 0)));
            W16_UNROLL for (int c = 0; c < NX; c++) Lx[s][c] = (c <= xc_ ? zx : 0.0) * WAT(D.Lf, k * NP + PK(xl_, NU + (c <= xc_ ? c : 0)));
            W16_UNROLL for (int c = 0; c < NX; c++) Br[s][c] = zm * WAT(D.BAt, (k * n + lc_) * NX + c);
            rb[s] = zx * WAT(D.rb, k * NX + xc_);
            m[s] = zm * WAT(D.rg, k * n + lc_);
            const bool has = mine[s] && ((imask >> row[s]) & 1);
            if (has)
            {
                const int ib = popc64(S.bmask & (((uint64_t) 1 << row[s]) - 1));
                const bool al = (am >> ib) & 1, au = (am >> (nbg + ib)) & 1;
                const int el = S.o_ct + ib, eu = el + nbg;
                const double ll = al ? WAT(D.lam, el) : 0.0, lu = au ? WAT(D.lam, eu) : 0.0;
                const double ttl = al ? WAT(D.t, el) : 1.0, ttu = au ? WAT(D.t, eu) : 1.0;
                const double rdl = al ? WAT(D.rd, el) : 0.0, rdu = au ? WAT(D.rd, eu) : 0.0;
                const double rml = al ? ll * ttl - O.tau_min + pscale * WAT(D.pcorr, el) - smu : 0.0;
                const double rmu = au ? lu * ttu - O.tau_min + pscale * WAT(D.pcorr, eu) - smu : 0.0;
                m[s] += (rml + ll * rdl) * frcp(ttl) - (rmu + lu * rdu) * frcp(ttu);
            }
        }
        /* y = Lx+ (Lx+' rb) + p+ */
        double w0[R], y[R], a[R];
        W16_UNROLL for (int s = 0; s < R; s++) { w0[s] = 0.0; y[s] = pn[s]; a[s] = 0.0; }
        W16_UNROLL for (int q = 0; q < NX; q++)
        {
            const double rbq = W16R_BC(rb, NU + q);
            W16_UNROLL for (int s = 0; s < R; s++) w0[s] += TA[q * LDX + cx[s]] * rbq; /* zero above the diagonal */
        }
        W16_UNROLL for (int s = 0; s < R; s++)
            if (!isx[s]) w0[s] = 0.0;
        W16_UNROLL for (int c = 0; c < NX; c++)
        {
            const double wc = W16R_BC(w0, NU + c);
            W16_UNROLL for (int s = 0; s < R; s++) y[s] += TA[cx[s] * LDX + c] * wc; /* zero above the diagonal */
        }
        W16_UNROLL for (int s = 0; s < R; s++)
            if (!isx[s]) y[s] = 0.0;
        W16_UNROLL for (int c = 0; c < NX; c++)
        {
            const double yc = W16R_BC(y, NU + c);
            W16_UNROLL for (int s = 0; s < R; s++) a[s] += Br[s][c] * yc;
        }
        W16_UNROLL for (int s = 0; s < R; s++) m[s] = (fixed[s] || !mine[s]) ? 0.0 : m[s] + a[s];
        /* l_u = Lr^{-1} m_u; the rows below keep m_r -= L[r][j] l_j, which leaves p in the state slots */
        W16_UNROLL for (int j = 0; j < NU; j++)
        {
            const double d = w16_bcast(Lu[j >> 4][j], j & 15, xb);
            const double lj = d != 0.0 ? W16R_BC(m, j) * frcp(d) : 0.0;
            W16_UNROLL for (int s = 0; s < R; s++) m[s] = row[s] == j ? lj : (row[s] > j ? m[s] - Lu[s][j] * lj : m[s]);
        }
        W16_UNROLL for (int s = 0; s < R; s++)
            if (mine[s]) WAT(D.lf, k * n + row[s]) = m[s];
        /* this stage's x-block becomes "the stage handled before" */
        GQP_ROWSYNC();
        W16_UNROLL for (int s = 0; s < R; s++)
            if (isx[s])
            {
                W16_UNROLL for (int c = 0; c < NX; c++) TA[cx[s] * LDX + c] = Lx[s][c];
            }
        GQP_ROWSYNC();
        W16_UNROLL for (int s = 0; s < R; s++) pn[s] = isx[s] ? m[s] : 0.0;
    }
}

/* --------------------------------------------------------------------------------------------------- forward */

/* PFORM (= CORR): lf holds [l_u; p] (written by ky_backrhs), otherwise the plain l of the factor sweep */
template <int NX, int NU, bool CORR>
__global__ void __launch_bounds__(64) ky_fwd(GqpDev D, GqpOpts O, int redo)
{
    GQP_DYN_SHARED(smem);
    typedef W16RLds<NX, NU> LY;
    constexpr bool PFORM = CORR;
    constexpr int n = NX + NU, R = LY::R, NP = n * (n + 1) / 2, LDF = LY::LDF;
    const int l = threadIdx.x & 15, inst = blockIdx.x * 4 + (threadIdx.x >> 4);
    if (inst >= D.B) return;
    if (D.status[inst] != GQP_RUNNING) return;
    if (redo == 1 && !(D.alpha[inst] < 0.0)) return; /* redo = 2: sensitivity pass (direction only, every instance) */
    double *T = smem + (threadIdx.x >> 4) * LY::SZ, *xb = T + LY::XB, *TF = T + LY::TF;
    int row[R], cx[R];
    bool mine[R], isx[R];
    W16_UNROLL for (int s = 0; s < R; s++)
    {
        row[s] = l + 16 * s;
        mine[s] = row[s] < n;
        isx[s] = row[s] >= NU && row[s] < n;
        cx[s] = isx[s] ? row[s] - NU : 0;
    }
    const double smu = CORR ? D.smu[inst] : 0.0;
    const double pscale = (CORR && redo != 1) ? 1.0 : 0.0;
    double alpha = 1.0, S0 = 0.0, S1 = 0.0, S2 = 0.0, nact = 0.0;
    double dx[R]; /* dx of this lane's states for the stage being entered */
    W16_UNROLL for (int s = 0; s < R; s++) dx[s] = 0.0;

    for (int k = 0; k <= D.N; k++)
    {
        const GqpStage &S = D.st[k];
        const uint64_t imask = S.bmask & ~S.emask;
        const uint64_t am = WAT(D.amask, k * D.AW);
        const int nbg = S.nb;
        /* row of the factor (registers); columns are read from the LDS tile where they are used */
        double Lr[R][n], lv[R], rbv[R];
        int lc_[R], xc_[R];
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            lc_[s] = mine[s] ? row[s] : 0;
            xc_[s] = isx[s] ? cx[s] : 0;
            const double zm = mine[s] ? 1.0 : 0.0, zx = isx[s] ? 1.0 : 0.0;
            W16_UNROLL for (int c = 0; c < n; c++)
                Lr[s][c] = W16R_LOW(s, c) ? (c <= lc_[s] ? zm : 0.0) * WAT(D.Lf, k * NP + PK(lc_[s], (c <= lc_[s] ? c : 0))) : 0.0;
            lv[s] = zm * WAT(D.lf, k * n + lc_[s]);
            rbv[s] = zx * WAT(D.rb, k * NX + xc_[s]);
        }
        GQP_ROWSYNC();
        W16_UNROLL for (int s = 0; s < R; s++)
            if (mine[s])
            {
                W16_UNROLL for (int c = 0; c < n; c++) TF[row[s] * LDF + c] = Lr[s][c];
            }
        GQP_ROWSYNC();
        /* L[r][row] for this slot's row: column of the factor, zero above the diagonal and for idle slots */
#define W16R_LC(s, r) ((mine[s] && (r) >= row[s]) ? TF[(r) * LDF + lc_[s]] : 0.0)

        if (PFORM && k == 0)
        {
            /* the states of stage 0 are free: recover l_x = Lx^{-1} p */
            W16_UNROLL for (int j = NU; j < n; j++)
            {
                const double d = w16_bcast(Lr[j >> 4][j], j & 15, xb);
                const double lj = d != 0.0 ? W16R_BC(lv, j) * frcp(d) : 0.0;
                W16_UNROLL for (int s = 0; s < R; s++) lv[s] = row[s] == j ? lj : (row[s] > j ? lv[s] - Lr[s][j] * lj : lv[s]);
            }
        }
        /* dpi_k = Lx (Lx' dx) + p  (CORR only; k > 0) */
        if (CORR && k > 0)
        {
            double w0[R], a[R]; /* (Lx' dx)[cx] = sum_{q >= cx} L[NU+q][NU+cx] dx[q]: column of L */
            W16_UNROLL for (int s = 0; s < R; s++) { w0[s] = 0.0; a[s] = lv[s]; }
            W16_UNROLL for (int q = 0; q < NX; q++)
            {
                const double dq = W16R_BC(dx, NU + q);
                W16_UNROLL for (int s = 0; s < R; s++) w0[s] += W16R_LC(s, NU + q) * dq;
            }
            W16_UNROLL for (int s = 0; s < R; s++)
                if (!isx[s]) w0[s] = 0.0;
            W16_UNROLL for (int c = 0; c < NX; c++)
            {
                const double wc = W16R_BC(w0, NU + c);
                W16_UNROLL for (int s = 0; s < R; s++) a[s] += Lr[s][NU + c] * wc;
            }
            W16_UNROLL for (int s = 0; s < R; s++)
                if (isx[s]) WAT(D.dpi, k * NX + cx[s]) = a[s];
        }
        /* L' dv = -l for the free block: everything at k = 0, the inputs otherwise; dv of the states = dx for k > 0 */
        double dv[R], acc[R];
        W16_UNROLL for (int s = 0; s < R; s++) { dv[s] = (k > 0 && isx[s]) ? dx[s] : 0.0; acc[s] = -lv[s]; }
        if (k > 0)
        {
            W16_UNROLL for (int p = NU; p < n; p++)
            {
                const double dp = W16R_BC(dv, p);
                W16_UNROLL for (int s = 0; s < R; s++) acc[s] -= W16R_LC(s, p) * dp;
            }
        }
        W16_UNROLL for (int r = n - 1; r >= 0; r--)
        {
            if (k > 0 && r >= NU) continue; /* uniform: states are given */
            const double d = w16_bcast(Lr[r >> 4][r], r & 15, xb);
            const double dvr = d != 0.0 ? W16R_BC(acc, r) * frcp(d) : 0.0;
            W16_UNROLL for (int s = 0; s < R; s++)
            {
                if (row[s] == r) dv[s] = dvr;
                else if (row[s] < r) acc[s] -= W16R_LC(s, r) * dvr;
            }
        }
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            if (!mine[s]) dv[s] = 0.0;
            if (CORR && mine[s]) WAT(D.dux, k * n + row[s]) = dv[s];
        }
        /* dx of the next stage (column cx of [B A]' straight from memory) */
        double dxn[R];
        W16_UNROLL for (int s = 0; s < R; s++) dxn[s] = rbv[s];
        W16_UNROLL for (int r = 0; r < n; r++)
        {
            const double dr = W16R_BC(dv, r);
            W16_UNROLL for (int s = 0; s < R; s++) dxn[s] += WAT(D.BAt, (k * n + r) * NX + xc_[s]) * dr; /* idle slots: clamped column, value unused */
        }
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            const bool has = mine[s] && ((imask >> row[s]) & 1);
            if (has)
            {
                const int ib = popc64(S.bmask & (((uint64_t) 1 << row[s]) - 1));
                const bool al = (am >> ib) & 1, au = (am >> (nbg + ib)) & 1;
                const int el = S.o_ct + ib, eu = el + nbg;
                const double ll = al ? WAT(D.lam, el) : 0.0, lu = au ? WAT(D.lam, eu) : 0.0;
                const double ttl = al ? WAT(D.t, el) : 1.0, ttu = au ? WAT(D.t, eu) : 1.0;
                const double rdl = al ? WAT(D.rd, el) : 0.0, rdu = au ? WAT(D.rd, eu) : 0.0;
                const double pl = (CORR && al) ? WAT(D.pcorr, el) : 0.0, pu = (CORR && au) ? WAT(D.pcorr, eu) : 0.0;
                const double rml = al ? ll * ttl - O.tau_min + pscale * pl - smu : 0.0;
                const double rmu = au ? lu * ttu - O.tau_min + pscale * pu - smu : 0.0;
                const double dtl = al ? dv[s] + rdl : 0.0, dtu = au ? -dv[s] + rdu : 0.0;
                const double dll = al ? -(rml + ll * dtl) * frcp(ttl) : 0.0;
                const double dlu = au ? -(rmu + lu * dtu) * frcp(ttu) : 0.0;
                const double c1 = -ll * frcp(dll), c2 = -lu * frcp(dlu), c3 = -ttl * frcp(dtl), c4 = -ttu * frcp(dtu);
                alpha = (dll < 0.0 && c1 < alpha) ? c1 : alpha;
                alpha = (dlu < 0.0 && c2 < alpha) ? c2 : alpha;
                alpha = (dtl < 0.0 && c3 < alpha) ? c3 : alpha;
                alpha = (dtu < 0.0 && c4 < alpha) ? c4 : alpha;
                if (!CORR)
                {
                    S0 += ll * ttl + lu * ttu;
                    S1 += ll * dtl + ttl * dll + lu * dtu + ttu * dlu;
                    S2 += dll * dtl + dlu * dtu;
                    nact += (double) ((int) al + (int) au);
                    WAT(D.pcorr, el) = dll * dtl;
                    WAT(D.pcorr, eu) = dlu * dtu;
                }
                else
                {
                    WAT(D.dlam, el) = dll; WAT(D.dlam, eu) = dlu;
                    WAT(D.dt, el) = dtl; WAT(D.dt, eu) = dtu;
                }
            }
        }
        W16_UNROLL for (int s = 0; s < R; s++) dx[s] = isx[s] ? dxn[s] : 0.0;
#undef W16R_LC
    }

    if (redo == 2) return; /* sensitivity pass: dux, dpi, dlam, dt are the result */
    alpha = w16_rmin(alpha, xb);
    const int it = D.iter[inst];
    double *st = (inst < D.stat_inst && it + 1 < D.stat_rows) ? D.stat + (size_t) (it + 1) * GQP_STAT_COLS * D.stat_inst + inst : nullptr;
    if (!CORR)
    {
        S0 = w16_rsum(S0, xb); S1 = w16_rsum(S1, xb); S2 = w16_rsum(S2, xb);
        const double nact_d = w16_rsum(nact, xb);
        if (l == 0)
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
    GQP_ROWSYNC(); /* everybody has read alpha[inst] */
    if (O.cond_pred_corr && !redo && alpha < 0.1 * alpha_aff)
    {
        if (l == 0) D.alpha[inst] = -alpha_aff;
        return;
    }
    const double a = D.mu[inst] > 0.0 ? alpha * 0.995 : 1.0;
    /* update: one slot per variable / state / box row of the stage (dux, dpi, dlam, dt were written by these very
     * slots); separate loops with independent iterations, several stages in flight */
    W16_UNROLL for (int s = 0; s < R; s++)
    {
        const int lc_ = mine[s] ? row[s] : 0, xc_ = isx[s] ? cx[s] : 0;
        _Pragma("unroll 4")
        for (int k = 0; k <= D.N; k++)
        {
            const double u0 = WAT(D.ux, k * n + lc_), du = WAT(D.dux, k * n + lc_);
            if (mine[s]) WAT(D.ux, k * n + row[s]) = u0 + a * du;
        }
        _Pragma("unroll 4")
        for (int k = 1; k <= D.N; k++)
        {
            const double p0 = WAT(D.pi, k * NX + xc_), dp = WAT(D.dpi, k * NX + xc_);
            if (isx[s]) WAT(D.pi, k * NX + cx[s]) = p0 + a * dp;
        }
        /* distinct arrays: tell the compiler, so that the loads of several stages can be in flight */
        const GqpStage *__restrict__ st_ = D.st;
        const uint64_t *__restrict__ am_ = D.amask.p + (size_t) inst * D.amask.E;
        double *__restrict__ lam_ = D.lam.p + (size_t) inst * D.lam.E;
        double *__restrict__ t_ = D.t.p + (size_t) inst * D.t.E;
        const double *__restrict__ dlam_ = D.dlam.p + (size_t) inst * D.dlam.E;
        const double *__restrict__ dt_ = D.dt.p + (size_t) inst * D.dt.E;
        _Pragma("unroll 4")
        for (int k = 0; k <= D.N; k++)
        {
            const uint64_t bm = st_[k].bmask, imask = bm & ~st_[k].emask;
            const uint64_t am = am_[k * D.AW];
            const int nbg = st_[k].nb;
            const bool has = mine[s] && ((imask >> row[s]) & 1);
            const int ib = has ? popc64(bm & (((uint64_t) 1 << row[s]) - 1)) : 0;
            const int el = st_[k].o_ct + ib, eu = el + nbg;
            const bool al = has && ((am >> ib) & 1), au = has && ((am >> (nbg + ib)) & 1);
            const double laml = lam_[el] + a * dlam_[el], lamu = lam_[eu] + a * dlam_[eu];
            const double tl = t_[el] + a * dt_[el], tu = t_[eu] + a * dt_[eu];
            if (al) { lam_[el] = laml < O.lam_min ? O.lam_min : laml; t_[el] = tl < O.t_min ? O.t_min : tl; }
            if (au) { lam_[eu] = lamu < O.lam_min ? O.lam_min : lamu; t_[eu] = tu < O.t_min ? O.t_min : tu; }
        }
    }
    if (l == 0)
    {
        D.alpha[inst] = alpha;
        D.iter[inst] = it + 1;
        if (st) { st[4 * D.stat_inst] = alpha; st[5 * D.stat_inst] = alpha; }
    }
}

} // namespace gqp

#endif
