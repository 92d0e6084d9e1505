/*
 * pcond_kernels_w16.hpp -- partial condensing of the box-only class on register rows: SIXTEEN LANES PER BLOCK.
 *
 * kw_pcond (pcond_kernels.hpp) walks the blocks of one instance with a whole wavefront and run-time dims; on the C2
 * shape (nx = 8, nu = 3, blocks of 5) it issues some 3,000 wave instructions per stage for 8 x 23 matrices and takes
 * 17 ms per 65,536 instances.  Here one 16-lane DPP row condenses ONE BLOCK of one instance -- a wavefront carries the
 * same block of four neighbouring instances, the grid is (instances / 4) x (N2 + 1) -- with compile-time shapes
 * (NX, NU, BS) and everything that is O(n^3) on register rows, as in ipm_kernels_w16r.hpp:
 *     lane l, slot s  <->  column p = l + 16 s of X = [Bbar Abar] (NX entries), row p of Hbar (lower triangle), gbar[p]
 * with nc = BS * NU + NX <= 32 columns.  Per stage of the block (algorithm and notation of pcond_kernels.hpp):
 *     y = H [0; c] + g                    lane r computes y[r]
 *     gbar += Z' y                        broadcasts of y
 *     Hbar += Z' H Z                      R on the (ii, ii) input block, S X on the input rows (the lane of column col
 *                                         owns (col, ra) for col > ra; row ra collects the others by broadcast), X' (Q X)
 *                                         with T = Q X columns broadcast to the rows
 *     X <- A X (+ B on the stage's own input columns), c <- A c + b
 * H, [B A]', g, b of the stage are staged through a small LDS tile per block (flat 16-lane loads issued one stage
 * ahead) and read from there at compile-time offsets.  Columns of X that are still zero (inputs of later stages) are
 * skipped at compile time.
 * Class: every child row is a box row (input bounds anywhere, state bounds at block starts), no slacks -- what
 * gpu_batch.hip calls box_class; everything else stays with kw_pcond.  Same outputs, bit for bit up to the order of the
 * floating-point sums.
 */
#ifndef PCOND_KERNELS_W16_HPP_
#define PCOND_KERNELS_W16_HPP_

#include "pcond_kernels.hpp"
#include "ipm_kernels_w16r.hpp"

namespace gqp
{

template <int NX, int NU, int BS>
struct PcondzLds
{
    static constexpr int n = NX + NU, NP = n * (n + 1) / 2, NB = n * NX;
    static constexpr int XB = 0, HS = 16, BSO = HS + NP, GS = BSO + NB, BV = GS + n, SZ = (BV + NX + 1) & ~1;
    static constexpr int NSTG = NP + NB + n + NX; /* doubles of one stage: H, [B A]', g, b */
};

/* bit l of the result = predicate of lane l of this lane's 16-lane row */
__device__ static inline unsigned w16_rowbits(bool b, double *xb)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (unsigned) (__ballot(b) >> (threadIdx.x & 48)) & 0xFFFFu;
#else
    const int l = threadIdx.x & 15;
    xb[l] = b ? 1.0 : 0.0;
    GQP_ROWSYNC();
    unsigned m = 0;
    for (int j = 0; j < 16; j++) m |= xb[j] != 0.0 ? 1u << j : 0u;
    GQP_ROWSYNC();
    return m;
#endif
}

template <int NX, int NU, int BS>
__global__ void __launch_bounds__(64) kz_pcond(GqpDev P, GqpDev Cd, PcondMap Mp)
{
    GQP_DYN_SHARED(smem);
    typedef PcondzLds<NX, NU, BS> LY;
    constexpr int n = NX + NU, NP = LY::NP, NB = LY::NB, NUC = BS * NU, nc = NUC + NX, NPC = nc * (nc + 1) / 2, R = (nc + 15) / 16;
    const int l = threadIdx.x & 15, rq = threadIdx.x >> 4, jb = blockIdx.y;
    /* XCD-aware block -> instance-group map.  A wave-tiled parent keeps element e of 16 neighbouring instances in one
     * 128-byte line; this workgroup uses the 32 bytes of its four instances, the other three groups of the line are other
     * workgroups.  Workgroups go round-robin over the 8 XCDs (each with its own L2): with the identity map those four
     * groups sit on four different XCDs and every one of them pulls the line from HBM (PMC: 22 GB per launch for 65,536
     * C2 instances, 2.7x what the kernel needs).  Within each chunk of 32 workgroups, XCD c gets the groups 4c .. 4c + 3. */
    int gx = blockIdx.x;
    if ((gx | 31) < (int) gridDim.x) gx = (gx & ~31) + ((gx & 7) << 2) + ((gx >> 3) & 3);
    const int inst_raw = gx * 4 + rq;
    const bool alive = inst_raw < P.B;
    const int inst = alive ? inst_raw : P.B - 1; /* a row beyond the batch condenses the last instance and writes nothing */
    double *T = smem + rq * LY::SZ, *xb = T + LY::XB;
    const double *Hs = T + LY::HS, *Bs = T + LY::BSO, *gs = T + LY::GS, *bv = T + LY::BV;
    double *stg = T + LY::HS; /* H, [B A]', g, b lie one after the other */

    const int k0 = Mp.blk_start[jb < Mp.N2 ? jb : Mp.N2];
    const int bs = jb < Mp.N2 ? Mp.blk_start[jb + 1] - k0 : 1; /* terminal: stage N alone */
    const int mode = Mp.mode;

    int p[R];
    W16_UNROLL for (int s = 0; s < R; s++) p[s] = l + 16 * s;
    double Xc[R][NX], Hb[R][nc], gb[R], c[NX];
    W16_UNROLL for (int s = 0; s < R; s++)
    {
        gb[s] = 0.0;
        W16_UNROLL for (int r = 0; r < NX; r++) Xc[s][r] = p[s] == NUC + r ? 1.0 : 0.0;
        W16_UNROLL for (int cb = 0; cb < nc; cb++) Hb[s][cb] = 0.0;
    }
    W16_UNROLL for (int r = 0; r < NX; r++) c[r] = 0.0;

    /* the stage's data, flat over the 16 lanes, one stage ahead.  Layout of the parent decided once (wave-tiled or
     * instance-major): base pointer + element stride, no branch per access; the last pass over an array re-reads its last
     * element instead of branching */
    constexpr int PH = (NP + 15) / 16, PB = (NB + 15) / 16, PG = (n + 15) / 16, PX = (NX + 15) / 16;
    const size_t pes = P.RSQ.aos ? 1 : 64;
    auto pbase = [&](const GArr &a) { return a.p + (a.aos ? (size_t) inst * (size_t) a.E : (size_t) (inst >> 6) * (size_t) a.E * 64 + (size_t) (inst & 63)); };
    const double *pH = pbase(P.RSQ), *pB = pbase(P.BAt), *pG = pbase(P.rq), *pX = pbase(P.bvec);
    double fH[PH], fB[PB], fG[PG], fX[PX];
    auto prefetch = [&](int k)
    {
        W16_UNROLL for (int i = 0; i < PH; i++) { const int e = l + 16 * i; fH[i] = pH[(size_t) (k * NP + (e < NP ? e : NP - 1)) * pes]; }
        W16_UNROLL for (int i = 0; i < PB; i++) { const int e = l + 16 * i; fB[i] = pB[(size_t) (k * NB + (e < NB ? e : NB - 1)) * pes]; }
        W16_UNROLL for (int i = 0; i < PG; i++) { const int e = l + 16 * i; fG[i] = pG[(size_t) (k * n + (e < n ? e : n - 1)) * pes]; }
        W16_UNROLL for (int i = 0; i < PX; i++) { const int e = l + 16 * i; fX[i] = pX[(size_t) (k * NX + (e < NX ? e : NX - 1)) * pes]; }
    };
    prefetch(k0);

    W16_UNROLL for (int ii = 0; ii < BS; ii++)
    {
        if (ii < bs) /* uniform: every row of the wave condenses the same block index */
        {
            const int k = k0 + ii;
            GQP_ROWSYNC();
            W16_UNROLL for (int i = 0; i < PH; i++) { const int e = l + 16 * i; if (e < NP) stg[e] = fH[i]; }
            W16_UNROLL for (int i = 0; i < PB; i++) { const int e = l + 16 * i; if (e < NB) stg[NP + e] = fB[i]; }
            W16_UNROLL for (int i = 0; i < PG; i++) { const int e = l + 16 * i; if (e < n) stg[NP + NB + e] = fG[i]; }
            W16_UNROLL for (int i = 0; i < PX; i++) { const int e = l + 16 * i; if (e < NX) stg[NP + NB + n + e] = fX[i]; }
            GQP_ROWSYNC();
            if (ii + 1 < bs) prefetch(k + 1);
            W16_UNROLL for (int s = 0; s < R; s++) W16R_OPAQUE(p[s]);

            /* y[r] = g[r] + sum_q H[r][NU + q] c[q], lane r */
            const int yr = l < n ? l : 0;
            double yv = gs[yr];
            W16_UNROLL for (int q = 0; q < NX; q++)
            {
                const int cc = NU + q;
                yv += (yr >= cc ? Hs[PK(yr, 0) + cc] : Hs[PK(cc, 0) + yr]) * c[q];
            }
            if (l >= n) yv = 0.0;
            /* gbar += Z' y */
            W16_UNROLL for (int a = 0; a < NU; a++)
            {
                const double ya = w16_bcast(yv, a, xb);
                const int ra = ii * NU + a;
                gb[ra >> 4] += p[ra >> 4] == ra ? ya : 0.0;
            }
            W16_UNROLL for (int r = 0; r < NX; r++)
            {
                const double yx = w16_bcast(yv, NU + r, xb);
                W16_UNROLL for (int s = 0; s < R; s++) gb[s] += Xc[s][r] * yx;
            }
            if (mode & 1)
            {
                /* R on the (ii, ii) input block */
                W16_UNROLL for (int a = 0; a < NU; a++)
                    W16_UNROLL for (int b = 0; b <= a; b++)
                    {
                        const int ra = ii * NU + a;
                        Hb[ra >> 4][ii * NU + b] += p[ra >> 4] == ra ? Hs[PK(a, b)] : 0.0;
                    }
                /* S X on the input rows: s(col) = sum_r H[NU + r][a] X[r][col] lives in the lane of column col, which owns
                 * (col, ra) for col > ra; row ra collects the columns before it (the non-zero ones: inputs of earlier stages) */
                W16_UNROLL for (int a = 0; a < NU; a++)
                {
                    const int ra = ii * NU + a;
                    double sx[R];
                    W16_UNROLL for (int s = 0; s < R; s++)
                    {
                        sx[s] = 0.0;
                        W16_UNROLL for (int r = 0; r < NX; r++) sx[s] += Hs[PK(NU + r, a)] * Xc[s][r];
                        if (W16R_LOW(s, ra)) Hb[s][ra] += p[s] > ra ? sx[s] : (p[s] == ra ? 2.0 * sx[s] : 0.0);
                    }
                    W16_UNROLL for (int col = 0; col < NUC; col++) /* constant trip count: the bound depends on the outer index */
                        if (col < ii * NU)
                        {
                            const double sc = w16_bcast(sx[col >> 4], col & 15, xb);
                            Hb[ra >> 4][col] += p[ra >> 4] == ra ? sc : 0.0;
                        }
                }
                /* T = Q X (column per lane), Hbar += X' T: column cb of T is broadcast to the rows at or below cb.  Columns
                 * ii*NU .. NUC-1 of X are still zero */
                double Tc[R][NX];
                W16_UNROLL for (int s = 0; s < R; s++)
                    W16_UNROLL for (int r = 0; r < NX; r++)
                    {
                        double t = 0.0;
                        W16_UNROLL for (int q = 0; q < NX; q++) t += Hs[r >= q ? PK(NU + r, NU + q) : PK(NU + q, NU + r)] * Xc[s][q];
                        Tc[s][r] = t;
                    }
                W16_UNROLL for (int cb = 0; cb < nc; cb++)
                {
                    if (cb >= ii * NU && cb < NUC) continue;
                    W16_UNROLL for (int r = 0; r < NX; r++)
                    {
                        const double t = w16_bcast(Tc[cb >> 4][r], cb & 15, xb);
                        W16_UNROLL for (int s = 0; s < R; s++)
                            if (W16R_LOW(s, cb)) Hb[s][cb] += Xc[s][r] * t;
                    }
                }
            }
            /* (accumulators pinned at the end of their phase: see W16R_OPAQUE in ipm_kernels_w16r.hpp) */
            W16_UNROLL for (int s = 0; s < R; s++)
            {
                W16R_OPAQUE(gb[s]);
                W16_UNROLL for (int cb = 0; cb < nc; cb++)
                    if (W16R_LOW(s, cb)) W16R_OPAQUE(Hb[s][cb]);
            }
            /* propagate x_{k+1} = A x_k + B u_k + b (slot N of BAt / bvec is zero) */
            if (jb < Mp.N2)
            {
                double cn[NX], Xn[R][NX];
                W16_UNROLL for (int r = 0; r < NX; r++)
                {
                    double a = bv[r];
                    W16_UNROLL for (int q = 0; q < NX; q++) a += Bs[(NU + q) * NX + r] * c[q];
                    cn[r] = a;
                    W16_UNROLL for (int s = 0; s < R; s++)
                    {
                        double t = 0.0;
                        W16_UNROLL for (int q = 0; q < NX; q++) t += Bs[(NU + q) * NX + r] * Xc[s][q];
                        Xn[s][r] = t;
                    }
                }
                /* the stage's own input columns: column ii*NU + a of X gets B[:, a] */
                W16_UNROLL for (int s = 0; s < R; s++)
                {
                    const int a = p[s] - ii * NU;
                    const bool own = a >= 0 && a < NU;
                    const int ac = own ? a : 0;
                    W16_UNROLL for (int r = 0; r < NX; r++) Xn[s][r] += own ? Bs[ac * NX + r] : 0.0;
                }
                W16_UNROLL for (int r = 0; r < NX; r++)
                {
                    c[r] = cn[r];
                    W16_UNROLL for (int s = 0; s < R; s++) Xc[s][r] = Xn[s][r];
                }
            }
            W16_UNROLL for (int r = 0; r < NX; r++)
            {
                W16R_OPAQUE(c[r]);
                W16_UNROLL for (int s = 0; s < R; s++) W16R_OPAQUE(Xc[s][r]);
            }
        }
    }
    /* unused input slots of a short block (and all of them at the terminal stage): unit diagonal */
    const int used = jb < Mp.N2 ? bs * NU : 0;
    W16_UNROLL for (int s = 0; s < R; s++)
        W16_UNROLL for (int cb = 0; cb < NUC; cb++)
            if (W16R_LOW(s, cb)) Hb[s][cb] = (p[s] == cb && cb >= used) ? 1.0 : Hb[s][cb];

    if (!alive) return;
    /* ---- write child stage jb ---- */
    if (mode & 1)
    {
        W16_UNROLL for (int s = 0; s < R; s++)
            if (p[s] < nc)
            {
                W16_UNROLL for (int cb = 0; cb < nc; cb++)
                    if (W16R_LOW(s, cb) && cb <= p[s]) PLAT(Cd.RSQ, jb * NPC + PK(p[s], cb)) = Hb[s][cb];
                if (jb < Mp.N2)
                    W16_UNROLL for (int r = 0; r < NX; r++) PLAT(Cd.BAt, (jb * nc + p[s]) * NX + r) = Xc[s][r];
            }
    }
    if (mode & 2)
    {
        W16_UNROLL for (int s = 0; s < R; s++)
            if (p[s] < nc) PLAT(Cd.rq, jb * nc + p[s]) = gb[s];
        if (jb < Mp.N2)
        {
            double cl = 0.0;
            W16_UNROLL for (int r = 0; r < NX; r++) cl = l == r ? c[r] : cl;
            if (l < NX) PLAT(Cd.bvec, jb * NX + l) = cl;
        }
        /* box rows keep their bounds; activity bits; value of fixed variables */
        GQP_STAGE_REF Sc = Cd.st[jb];
        const int r0 = Mp.row_off[jb], nbc = Sc.nb;
        uint64_t amc = 0;
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            const int rc = p[s];
            const bool has = rc < nbc;
            const int kp = Mp.row_kp[r0 + (has ? rc : 0)], rp = Mp.row_rp[r0 + (has ? rc : 0)];
            GQP_STAGE_REF Sp = P.st[kp];
            const int su = Sp.nb + Sp.ng + rp;
            bool al = false, au = false;
            if (has)
            {
                PLAT(Cd.dvec, Sc.o_ct + rc) = PLAT(P.dvec, Sp.o_ct + rp);
                PLAT(Cd.dvec, Sc.o_ct + nbc + rc) = PLAT(P.dvec, Sp.o_ct + su);
                al = (PLAT(P.amask, kp * P.AW + (rp >> 6)) >> (rp & 63)) & 1;
                au = (PLAT(P.amask, kp * P.AW + (su >> 6)) >> (su & 63)) & 1;
            }
            const uint64_t bl = w16_rowbits(al, xb), bu = w16_rowbits(au, xb);
            amc |= (bl << (16 * s)) | (bu << (nbc + 16 * s));
        }
        if (l == 0) PLAT(Cd.amask, jb * Cd.AW) = amc;
        if (l < NX && ((Sc.emask >> (NUC + l)) & 1)) PLAT(Cd.ux, jb * nc + NUC + l) = PLAT(P.ux, k0 * n + NU + l);
    }
}

} // namespace gqp

#endif
