/*
 * pcond_kernels_mfma.hpp -- partial condensing of the box-only class on the FP64 MATRIX pipe: v_mfma_f64_4x4x4_4b_f64.
 *
 * The contraction of `ocp_qp_partial_condensing` (acados/ocp_qp/ocp_qp_partial_condensing.c:523-556 -> HPIPM's
 * d_part_cond_qp_cond; algorithm and notation: pcond_kernels.hpp) is, per stage i of a block,
 *     Hbar += Z_i' H_i Z_i,   gbar += Z_i' (H_i [0; c_i] + g_i),   X_{i+1} = [B A]_i Z_i,   c_{i+1} = A_i c_i + b_i
 * with Z_i = [E_i; X_i] (n x nc).  kz_pcond (pcond_kernels_w16.hpp) runs it on register rows with one DPP broadcast per
 * two multiply-adds: 388 registers, one wave per SIMD, 4.3 ms per 65,536 C2 instances at 0.23 of its HBM bound.  The 16x16x4
 * FP64 MFMA is no help at nx = 8 (half-empty tiles, and that shape of the instruction peaks below the vector pipe on
 * gfx950: profiles/r02_mfma_f64_probe.txt) -- but v_mfma_f64_4x4x4_4b_f64 is FOUR INDEPENDENT 4 x 4 x 4 products per
 * instruction, measured at 17-19 cycles per issue = 67-73 TFLOP/s (profiles/r04_mfma4x4x4_probe.txt; the DPP broadcast +
 * two v_fma_f64 it replaces: 14.3 cycles for half the flops), and nx = 8 fills its tiles exactly.  So here
 *     one wavefront = the same block of FOUR neighbouring instances, one instance per MFMA block
 * (kz_pcond's mapping; a workgroup is four such waves, see the instance map in the kernel), and every product above is a
 * chain of 4 x 4 tile products.
 *
 * Operand layout of the instruction (found by experiment, tools/mfma_f64_probe/probe3.hip,
 * profiles/r04_mfma4x4x4_layout.txt): block b = (lane >> 2) & 3; with x = lane & 3, y = lane >> 4
 *     A[i][k] sits in lane (x = i, y = k),  B[k][j] in lane (x = j, y = k),  D[i][j] in lane (x = j, y = i)
 * -- the 16x16x4 layout restricted to its block diagonal; a block is a quad of lanes in each of the four 16-lane rows, NOT a
 * 16-lane row.  Every tile here is held in the D layout ("lane (x, y) holds element [y][x]", one double per lane); a
 * D-layout tile P passed as the A operand is read as P', passed as the B operand as itself:
 *     gqp_mfma4(P, Q, C) = C + P' Q.
 * All products of the contraction have that shape already (H symmetric, [B A] stored transposed), so no tile is ever
 * transposed or moved between lanes:
 *     T  = H Z          T(I,J)  = sum_K H(K,I)' Z(K,J)                       (H(I,K) = H(K,I)')
 *     Hbar += Z' T      Hb(I,J) += sum_K Z(K,I)' T(K,J)        lower tiles only
 *     X+ = [B A] Z      Xn(I,J) = sum_K BAt(K,I)' Z(K,J)                     (BAt = [B A]' is what HBM holds)
 * The vectors ride along in ONE SPARE COLUMN (the padding of nc to a multiple of four: nc = 23 -> column 23): Z gets the
 * column [c_i; 0], T's copy of it gets g added, so that column of Z'T -- computed as the last ROW of tiles with the operands
 * swapped, T(K,last)' Z(K,J) -- is gbar, and that column of [B A] Z is A c (+ b).
 * Variables of a parent stage are taken in the order [x; u] inside the kernel (nx is a multiple of four: the propagated X
 * then fills whole tile rows and the selector rows E_i sit in tile rows of their own); the permutation is address
 * arithmetic in the loads, nothing else knows about it.  Tile columns of Z that are still zero (inputs of later stages) are
 * skipped at compile time.  No LDS: every lane loads exactly the tile elements it owns, one stage ahead.
 *
 * Class and outputs: those of kz_pcond (every child row a box row, no slacks; same child arrays up to the order of the
 * floating-point sums); ACADOS_AMD_PCOND_MFMA=0 keeps kz_pcond (the cross-check both test tiers run).
 */
#ifndef PCOND_KERNELS_MFMA_HPP_
#define PCOND_KERNELS_MFMA_HPP_

#include "pcond_kernels_w16.hpp"
#include "mfma4.hpp"

namespace gqp
{

#ifndef KM_PCOND_THREADS
#define KM_PCOND_THREADS 256 /* four waves = the sixteen instances of one 128-byte line of a wave-tiled parent */
#endif
template <int NX, int NU, int BS>
__global__ void __launch_bounds__(KM_PCOND_THREADS) km_pcond(GqpDev P, GqpDev Cd, PcondMap Mp)
{
    static_assert(NX % 4 == 0, "km_pcond: the state block must fill whole 4 x 4 tile rows");
    constexpr int n = NX + NU, NP = n * (n + 1) / 2, NB = n * NX, NUC = BS * NU, nc = NUC + NX, NPC = nc * (nc + 1) / 2;
    /* tile rows of x, of [x; u]; tile columns of [ubar xbar | c]; tile column and in-tile column of the vector column */
    constexpr int NXT = NX / 4, NRT = (n + 3) / 4, NUT = NRT - NXT, NCT = (nc + 4) / 4, JC = nc / 4, XC = nc % 4;
    static_assert(NUT * 4 >= NU && JC == NCT - 1, "km_pcond: tile bookkeeping");
    const int lane = threadIdx.x & 63, x = lane & 3, y = lane >> 4, bq = (lane >> 2) & 3, jb = blockIdx.y;
    const int l = x + 4 * y; /* 0..15 inside the instance's block: the tail (bounds, activity bits) runs one child row per lane */
    /* A wave-tiled parent keeps element e of 16 neighbouring instances in one 128-byte line: the FOUR WAVES of a workgroup
     * take the four instance groups of that line (no data is shared between them -- no LDS, no barrier -- but they run on one
     * CU and meet in its vector cache; with one wave per workgroup those four groups sat on four CUs, and before the
     * XCD-aware map of kz_pcond on four XCDs) */
    const int inst_raw = (blockIdx.x * (KM_PCOND_THREADS / 64) + (threadIdx.x >> 6)) * 4 + bq;
    const bool alive = inst_raw < P.B;
    const int inst = alive ? inst_raw : P.B - 1; /* a block beyond the batch condenses the last instance and writes nothing */

    const int k0 = Mp.blk_start[jb < Mp.N2 ? jb : Mp.N2];
    const int bs = jb < Mp.N2 ? Mp.blk_start[jb + 1] - k0 : 1; /* terminal: stage N alone */
    const int mode = Mp.mode;

    /* what this lane owns of a parent stage: natural index of the variable in row 4K + y / column 4I + x of the [x; u] order */
    int vy[NRT], vx[NRT];
    bool oky[NRT], okx[NRT];
    W16_UNROLL for (int K = 0; K < NRT; K++)
    {
        const int ry = 4 * K + y, rx = 4 * K + x;
        oky[K] = ry < n; okx[K] = rx < n;
        vy[K] = ry < NX ? NU + ry : (ry < n ? ry - NX : 0);
        vx[K] = rx < NX ? NU + rx : (rx < n ? rx - NX : 0);
    }
    const size_t pes = P.RSQ.aos ? 1 : 64;
    auto pbase = [&](const GArr &a) { return a.p + (a.aos ? (size_t) inst * (size_t) a.E : (size_t) (inst >> 6) * (size_t) a.E * 64 + (size_t) (inst & 63)); };
    const double *pH = pbase(P.RSQ), *pB = pbase(P.BAt), *pG = pbase(P.rq), *pX = pbase(P.bvec);
    /* the stage's tiles, one stage ahead: H (all NRT x NRT tiles of the symmetric block, [y][x] = H[vy][vx]), [B A]' (rows =
     * variables, columns = next state), g (by row), b (by row of the next state).  Padding rows / columns read element 0 and
     * are zeroed after the load (no branch around a load) */
    double fH[NRT][NRT], fB[NRT][NXT], fG[NRT], fX[NXT];
    auto prefetch = [&](int k)
    {
        W16_UNROLL for (int K = 0; K < NRT; K++)
        {
            W16_UNROLL for (int I = 0; I < NRT; I++)
            {
                const int r = vy[K] > vx[I] ? vy[K] : vx[I], c = vy[K] > vx[I] ? vx[I] : vy[K];
                fH[K][I] = pH[(size_t) (k * NP + PK(r, c)) * pes];
            }
            W16_UNROLL for (int I = 0; I < NXT; I++) fB[K][I] = pB[(size_t) (k * NB + vy[K] * NX + 4 * I + x) * pes];
            fG[K] = pG[(size_t) (k * n + vy[K]) * pes];
        }
        W16_UNROLL for (int I = 0; I < NXT; I++) fX[I] = pX[(size_t) (k * NX + 4 * I + y) * pes];
    };
    prefetch(k0);

    /* Z = [X; E | c; 0]: X_0 = [0 I], c_0 = 0 */
    double Z[NRT][NCT], Hb[NCT][NCT];
    W16_UNROLL for (int I = 0; I < NRT; I++)
        W16_UNROLL for (int J = 0; J < NCT; J++)
        {
            const int r = 4 * I + y, col = 4 * J + x;
            Z[I][J] = (I < NXT && col == NUC + r) ? 1.0 : 0.0;
        }
    W16_UNROLL for (int I = 0; I < NCT; I++)
        W16_UNROLL for (int J = 0; J < NCT; J++) Hb[I][J] = 0.0;

    W16_UNROLL for (int ii = 0; ii < BS; ii++)
    {
        if (ii < bs) /* uniform: every block of the wave condenses the same block index */
        {
            const int k = k0 + ii;
            /* zero tiles of Z at this stage, known at compile time: the X rows (tile rows < NXT) have no entries yet in the
             * columns of this and later stages' inputs, the selector rows E_ii only in this stage's input columns, and a tile
             * column of Z -- hence of T = H Z -- is zero where both are */
#define KM_ZX0(J) (4 * (J) >= ii * NU && 4 * (J) + 3 < NUC)
#define KM_ZU0(Iu, J) (4 * (J) + 3 < ii * NU + 4 * (Iu) || 4 * (J) > ii * NU + (4 * (Iu) + 3 < NU - 1 ? 4 * (Iu) + 3 : NU - 1))
#define KM_Z0(K, J) ((K) < NXT ? KM_ZX0(J) : KM_ZU0((K) - NXT, J))
#define KM_ZC(J) (4 * (J) >= (ii + 1) * NU && 4 * (J) + 3 < NUC)
            double cH[NRT][NRT], cB[NRT][NXT], cG[NRT], cX[NXT];
            W16_UNROLL for (int K = 0; K < NRT; K++)
            {
                W16_UNROLL for (int I = 0; I < NRT; I++) cH[K][I] = (oky[K] && okx[I]) ? fH[K][I] : 0.0;
                W16_UNROLL for (int I = 0; I < NXT; I++) cB[K][I] = oky[K] ? fB[K][I] : 0.0;
                cG[K] = oky[K] ? fG[K] : 0.0;
            }
            W16_UNROLL for (int I = 0; I < NXT; I++) cX[I] = fX[I];
            if (ii + 1 < bs) prefetch(k + 1);
            /* selector rows E_ii: input a of this stage is column ii * NU + a of the block */
            W16_UNROLL for (int I = 0; I < NUT; I++)
                W16_UNROLL for (int J = 0; J < NCT; J++)
                {
                    const int a = 4 * I + y, col = 4 * J + x;
                    Z[NXT + I][J] = (a < NU && col == ii * NU + a) ? 1.0 : 0.0;
                }
            /* T = H Z; the vector column gets g: T[:, nc] = H [c; 0] + g */
            double T[NRT][NCT];
            W16_UNROLL for (int I = 0; I < NRT; I++)
                W16_UNROLL for (int J = 0; J < NCT; J++) T[I][J] = (J == JC && x == XC) ? cG[I] : 0.0;
            W16_UNROLL for (int K = 0; K < NRT; K++)
                W16_UNROLL for (int I = 0; I < NRT; I++)
                    W16_UNROLL for (int J = 0; J < NCT; J++)
                        if (!KM_Z0(K, J)) T[I][J] = gqp_mfma4(cH[K][I], Z[K][J], T[I][J]);
            /* Hbar += Z' T, lower tiles; the last tile row with the operands swapped: rows nc - XC .. nc - 1 are the same by
             * symmetry, row nc is gbar' = (H [c; 0] + g)' Z */
            W16_UNROLL for (int K = 0; K < NRT; K++)
                W16_UNROLL for (int I = 0; I < NCT; I++)
                    W16_UNROLL for (int J = 0; J <= I; J++)
                        if (I == NCT - 1 ? !KM_Z0(K, J) : (!KM_Z0(K, I) && !KM_ZC(J)))
                            Hb[I][J] = I == NCT - 1 ? gqp_mfma4(T[K][I], Z[K][J], Hb[I][J]) : gqp_mfma4(Z[K][I], T[K][J], Hb[I][J]);
            /* X+ = [B A] Z, c+ = A c + b (slot N of BAt / bvec is zero) */
            if (jb < Mp.N2)
            {
                double Xn[NXT][NCT];
                W16_UNROLL for (int I = 0; I < NXT; I++)
                    W16_UNROLL for (int J = 0; J < NCT; J++) Xn[I][J] = (J == JC && x == XC) ? cX[I] : 0.0;
                W16_UNROLL for (int K = 0; K < NRT; K++)
                    W16_UNROLL for (int I = 0; I < NXT; I++)
                        W16_UNROLL for (int J = 0; J < NCT; J++)
                            if (!KM_Z0(K, J)) Xn[I][J] = gqp_mfma4(cB[K][I], Z[K][J], Xn[I][J]);
                W16_UNROLL for (int I = 0; I < NXT; I++)
                    W16_UNROLL for (int J = 0; J < NCT; J++) Z[I][J] = Xn[I][J];
            }
#undef KM_ZC
#undef KM_Z0
#undef KM_ZU0
#undef KM_ZX0
        }
    }

    if (!alive) return;
    /* ---- write child stage jb: lane (x, y) owns element [4 I + y][4 J + x] of every tile ---- */
    const int used = jb < Mp.N2 ? bs * NU : 0; /* unused input slots of a short block (all of them at the terminal stage): unit diagonal */
    if (mode & 1)
    {
        W16_UNROLL for (int I = 0; I < NCT; I++)
            W16_UNROLL for (int J = 0; J <= I; J++)
            {
                const int r = 4 * I + y, c = 4 * J + x;
                if (r < nc && c <= r) PLAT(Cd.RSQ, jb * NPC + PK(r, c)) = (r == c && c >= used && c < NUC) ? 1.0 : Hb[I][J];
            }
        if (jb < Mp.N2)
            W16_UNROLL for (int I = 0; I < NXT; I++)
                W16_UNROLL for (int J = 0; J < NCT; J++)
                {
                    const int c = 4 * J + x;
                    if (c < nc) PLAT(Cd.BAt, (jb * nc + c) * NX + 4 * I + y) = Z[I][J];
                }
    }
    if (mode & 2)
    {
        if (y == XC)
            W16_UNROLL for (int J = 0; J < NCT; J++)
            {
                const int c = 4 * J + x;
                if (c < nc) PLAT(Cd.rq, jb * nc + c) = Hb[NCT - 1][J];
            }
        if (jb < Mp.N2 && x == XC)
            W16_UNROLL for (int I = 0; I < NXT; I++) PLAT(Cd.bvec, jb * NX + 4 * I + y) = Z[I][JC];
        /* box rows keep their bounds; activity bits; value of fixed variables (one child row per lane, as kz_pcond) */
        constexpr int R = (nc + 15) / 16;
        GQP_STAGE_REF Sc = Cd.st[jb];
        const int r0 = Mp.row_off[jb], nbc = Sc.nb;
        uint64_t amc = 0;
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            const int rc = l + 16 * s;
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
            const uint64_t bl = mfma4_blockbits(al), bu = mfma4_blockbits(au);
            amc |= (bl << (16 * s)) | (bu << (nbc + 16 * s));
        }
        if (l == 0) PLAT(Cd.amask, jb * Cd.AW) = amc;
        if (l < NX && ((Sc.emask >> (NUC + l)) & 1)) PLAT(Cd.ux, jb * nc + NUC + l) = PLAT(P.ux, k0 * n + NU + l);
    }
}

} // namespace gqp

#endif
