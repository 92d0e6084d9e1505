/*
 * pcond_kernels.hpp -- partial condensing N -> N2 and expansion on the device.
 *
 * Stands in for `ocp_qp_partial_condensing` / `ocp_qp_partial_expansion`
 * (acados/ocp_qp/ocp_qp_partial_condensing.c:523-556, :664-689 in /root/reference, i.e. HPIPM's
 * d_part_cond_qp_cond / d_part_cond_qp_expand_sol, whose sources are absent there).  Block j of the
 * condensed QP covers stages k0..k1-1 of the original one; its variables are
 *     xbar = x_{k0},   ubar = [u_{k0}; u_{k0+1}; ...; u_{k1-1}]        (slot i*NU + a)
 * and the eliminated states are x_{k0+i} = X_i [ubar; xbar] + c_i.  With Z_i = [E_i; X_i]
 * (E_i selects u_{k0+i}) the block data are
 *     Hbar = sum_i Z_i' H_i Z_i,   gbar = sum_i Z_i'(H_i [0; c_i] + g_i),
 *     [Bbar Abar] = X_bs,  bbar = c_bs.
 * Input bounds stay box rows of ubar, bounds on x_{k0} stay box rows of xbar (this is where the
 * equality-flagged x0 rows live).  Every other inequality row a' v_{k0+i} -- a state bound of a stage inside the
 * block, a general row of any stage -- becomes a GENERAL row of the condensed stage with coefficients
 * a_u' E_i + a_x' X_i and bounds shifted by a_x' c_i; slacks (idxs_rev, also shared ones) travel with their rows,
 * one-sided rows keep their activity bits (wave-per-instance kernels kw_pcond / kw_pexpand; the compiled
 * one-instance-per-lane pair k_pcond / k_pexpand covers the box-only class and is the cross-check there).  The
 * expansion recovers pi of the eliminated dynamics from stationarity including the inequality terms.  Limits: a
 * condensed stage carries at most 64 rows / 128 inequality sides and nx + bs*nu <= 64; beyond that the full-space QP
 * is solved (the reference's default N2 = N) with a one-line notice.
 *
 * Mapping: one instance per lane like the IPM kernels; the block matrices (X: NX x nc,
 * Hbar: nc(nc+1)/2, nc = BS*NU + NX) are per-lane arrays addressed in rolled loops.  This is a
 * once-per-solve pre/post-processing step; the FP64-MFMA formulation of the Z'HZ contraction
 * named by the north star only pays for nx >= 16 and is left to a later round.
 */
#ifndef PCOND_KERNELS_HPP_
#define PCOND_KERNELS_HPP_

#include "ipm_kernels.hpp"

namespace gqp
{

struct PcondMap
{
    const int *blk_start; /* N2+1 entries: first original stage of each block; blk_start[N2] = N */
    const int *row_kp;    /* per child row (all child stages concatenated): parent stage */
    const int *row_rp;    /* ... parent sorted row index */
    const int *row_off;   /* N2+2 entries: first child row of each child stage */
    /* wave-per-instance kernels only: rows that become GENERAL rows of the condensed stage (state bounds of a stage
     * inside a block, every general row), listed by parent stage; their position in the list minus gk_off[k0] is the
     * general-row index in the child stage.  row_kp / row_rp cover box AND general child rows (sorted child order). */
    const int *gk_off;    /* N+2 entries: first list entry of parent stage k */
    const int *g_rp;      /* parent sorted row index (< nb: box row, else general row nb + g) */
    const int *g_var;     /* box rows: state index of the bounded variable */
    const int *slk_off;   /* N2+2 entries: first child slack of each child stage (all child slacks concatenated) */
    const int *slk_kp;    /* per child slack: parent stage */
    const int *slk_sp;    /* ... parent slack index */
    int N2;
    int mode;             /* 3: everything; 1: matrix part only (Hbar, Abar, Bbar -- condense_lhs);
                             2: vector part only (gbar, bbar, bounds -- condense_rhs), the split RTI
                             uses (ocp_qp_partial_condensing.c:575-598, :602-630) */
};

/* parent (NX, NU) -> child (NX, NUC = BSMAX*NU) */
template <int NX, int NU, int BSMAX>
__global__ void __launch_bounds__(64) k_pcond(GqpDev P, GqpDev Cd, PcondMap Mp)
{
    constexpr int n = NX + NU, NP = n * (n + 1) / 2;
    constexpr int NUC = BSMAX * NU, nc = NUC + NX, NPC = nc * (nc + 1) / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.B) return;
    const int Bp = P.Bp; /* parent and child share B, Bp */

    double X[NX * nc], Xn[NX * nc], T[NX * nc], c[NX], cn[NX], Hb[NPC], gb[nc], H[NP], y[n];

    for (int jb = 0; jb <= Mp.N2; jb++)
    {
        const int k0 = Mp.blk_start[jb < Mp.N2 ? jb : Mp.N2];
        const int k1 = jb < Mp.N2 ? Mp.blk_start[jb + 1] : P.N + 1; /* terminal: stage N alone */
        const int bs = jb < Mp.N2 ? k1 - k0 : 1;
        for (int e = 0; e < NX * nc; e++) X[e] = 0.0;
        for (int r = 0; r < NX; r++) { X[r * nc + NUC + r] = 1.0; c[r] = 0.0; }
        for (int e = 0; e < NPC; e++) Hb[e] = 0.0;
        for (int e = 0; e < nc; e++) gb[e] = 0.0;

        for (int ii = 0; ii < bs; ii++)
        {
            const int k = k0 + ii;
            for (int e = 0; e < NP; e++) H[e] = GATL(P.RSQ, k * NP + e);
            /* y = H [0; c] + g */
            for (int r = 0; r < n; r++)
            {
                double a = GATL(P.rq, k * n + r);
                for (int q = 0; q < NX; q++)
                {
                    const int cc = NU + q;
                    a += (r >= cc ? H[PK(r, cc)] : H[PK(cc, r)]) * c[q];
                }
                y[r] = a;
            }
            /* gbar += Z' y */
            for (int a = 0; a < NU; a++) gb[ii * NU + a] += y[a];
            for (int col = 0; col < nc; col++)
            {
                double s = 0.0;
                for (int r = 0; r < NX; r++) s += X[r * nc + col] * y[NU + r];
                gb[col] += s;
            }
            /* Hbar += Z' H Z  with Z = [E_ii; X]:  R on the (ii,ii) input block, S X on the
             * input rows, X' Q X everywhere */
            if (Mp.mode & 1)
            {
            for (int a = 0; a < NU; a++)
                for (int b = 0; b <= a; b++) Hb[PK(ii * NU + a, ii * NU + b)] += H[PK(a, b)];
            for (int a = 0; a < NU; a++)
            {
                const int ra = ii * NU + a;
                for (int col = 0; col < nc; col++)
                {
                    double s = 0.0; /* (S X)[a][col], S[a][r] = H[NU+r][a] */
                    for (int r = 0; r < NX; r++) s += H[PK(NU + r, a)] * X[r * nc + col];
                    if (col == ra) Hb[PK(ra, ra)] += 2.0 * s;
                    else if (col < ra) Hb[PK(ra, col)] += s;
                    else Hb[PK(col, ra)] += s;
                }
            }
            for (int r = 0; r < NX; r++)
                for (int col = 0; col < nc; col++)
                {
                    double s = 0.0; /* T = Q X */
                    for (int q = 0; q < NX; q++)
                        s += (r >= q ? H[PK(NU + r, NU + q)] : H[PK(NU + q, NU + r)]) * X[q * nc + col];
                    T[r * nc + col] = s;
                }
            for (int ra = 0; ra < nc; ra++)
                for (int cb = 0; cb <= ra; cb++)
                {
                    double s = 0.0;
                    for (int r = 0; r < NX; r++) s += X[r * nc + ra] * T[r * nc + cb];
                    Hb[PK(ra, cb)] += s;
                }
            }
            /* propagate x_{k+1} = A x_k + B u_k + b  (slot N of BAt/bvec is zero) */
            if (jb < Mp.N2)
            {
                for (int r = 0; r < NX; r++)
                {
                    double a = GATL(P.bvec, k * NX + r);
                    for (int q = 0; q < NX; q++) a += GATL(P.BAt, (k * n + NU + q) * NX + r) * c[q];
                    cn[r] = a;
                    for (int col = 0; col < nc; col++)
                    {
                        double s = 0.0;
                        for (int q = 0; q < NX; q++) s += GATL(P.BAt, (k * n + NU + q) * NX + r) * X[q * nc + col];
                        Xn[r * nc + col] = s;
                    }
                    for (int a2 = 0; a2 < NU; a2++) Xn[r * nc + ii * NU + a2] += GATL(P.BAt, (k * n + a2) * NX + r);
                }
                for (int e = 0; e < NX * nc; e++) X[e] = Xn[e];
                for (int r = 0; r < NX; r++) c[r] = cn[r];
            }
        }
        /* unused input slots of a short block (and all of them at the terminal stage): unit diagonal */
        const int used = jb < Mp.N2 ? bs * NU : 0;
        for (int s2 = used; s2 < NUC; s2++) Hb[PK(s2, s2)] = 1.0;

        /* ---- write child stage jb ---- */
        if (Mp.mode & 1)
        {
            for (int e = 0; e < NPC; e++) GATL(Cd.RSQ, jb * NPC + e) = Hb[e];
            if (jb < Mp.N2)
                for (int col = 0; col < nc; col++)
                    for (int r = 0; r < NX; r++) GATL(Cd.BAt, (jb * nc + col) * NX + r) = X[r * nc + col];
        }
        if (!(Mp.mode & 2)) continue;
        for (int e = 0; e < nc; e++) GATL(Cd.rq, jb * nc + e) = gb[e];
        if (jb < Mp.N2)
            for (int r = 0; r < NX; r++) GATL(Cd.bvec, jb * NX + r) = c[r];
        /* box rows, activity bits, value of fixed variables */
        GQP_STAGE_REF Sc = Cd.st[jb];
        const int r0 = Mp.row_off[jb], nbc = Sc.nb;
        uint64_t amc = 0;
        for (int rc = 0; rc < nbc; rc++)
        {
            const int kp = Mp.row_kp[r0 + rc], rp = Mp.row_rp[r0 + rc];
            GQP_STAGE_REF Sp = P.st[kp];
            const uint64_t amp = GATL(P.amask, kp);
            GATL(Cd.dvec, Sc.o_ct + rc) = GATL(P.dvec, Sp.o_ct + rp);
            GATL(Cd.dvec, Sc.o_ct + nbc + rc) = GATL(P.dvec, Sp.o_ct + Sp.nb + rp);
            if ((amp >> rp) & 1) amc |= (uint64_t) 1 << rc;
            if ((amp >> (Sp.nb + rp)) & 1) amc |= (uint64_t) 1 << (nbc + rc);
        }
        GATL(Cd.amask, jb) = amc;
        for (int r = 0; r < NX; r++)
            if ((Sc.emask >> (NUC + r)) & 1) GATL(Cd.ux, jb * nc + NUC + r) = GATL(P.ux, k0 * n + NU + r);
    }
}

/* expansion: original (ux, pi, lam, t) from the condensed solution */
template <int NX, int NU, int BSMAX>
__global__ void __launch_bounds__(64) k_pexpand(GqpDev P, GqpDev Cd, PcondMap Mp)
{
    constexpr int n = NX + NU, NP = n * (n + 1) / 2;
    constexpr int NUC = BSMAX * NU, nc = NUC + NX;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.B) return;
    const int Bp = P.Bp;
    double x[NX], xn[NX], u[NU], pk[NX], pn[NX];

    for (int jb = 0; jb <= Mp.N2; jb++)
    {
        const int k0 = Mp.blk_start[jb < Mp.N2 ? jb : Mp.N2];
        const int bs = jb < Mp.N2 ? Mp.blk_start[jb + 1] - k0 : 1;
        /* forward simulation of the eliminated states */
        for (int r = 0; r < NX; r++) x[r] = GATL(Cd.ux, jb * nc + NUC + r);
        for (int ii = 0; ii < bs; ii++)
        {
            const int k = k0 + ii;
            for (int a = 0; a < NU; a++) u[a] = jb < Mp.N2 ? GATL(Cd.ux, jb * nc + ii * NU + a) : 0.0;
            for (int a = 0; a < NU; a++) GATL(P.ux, k * n + a) = u[a];
            for (int r = 0; r < NX; r++) GATL(P.ux, k * n + NU + r) = x[r];
            if (jb < Mp.N2 && ii + 1 < bs)
            {
                for (int r = 0; r < NX; r++)
                {
                    double a = GATL(P.bvec, k * NX + r);
                    for (int q = 0; q < NX; q++) a += GATL(P.BAt, (k * n + NU + q) * NX + r) * x[q];
                    for (int q = 0; q < NU; q++) a += GATL(P.BAt, (k * n + q) * NX + r) * u[q];
                    xn[r] = a;
                }
                for (int r = 0; r < NX; r++) x[r] = xn[r];
            }
        }
        /* multipliers of the eliminated dynamics: pi_k = Q x_k + S' u_k + q_k + A_k' pi_{k+1}
         * (no inequality touches an eliminated state), backwards from the block boundary */
        if (jb < Mp.N2)
        {
            const int k1 = k0 + bs;
            for (int r = 0; r < NX; r++) { pn[r] = GATL(Cd.pi, (jb + 1) * NX + r); GATL(P.pi, k1 * NX + r) = pn[r]; }
            for (int k = k1 - 1; k > k0; k--)
            {
                for (int r = 0; r < NX; r++)
                {
                    double a = GATL(P.rq, k * n + NU + r);
                    for (int q = 0; q < n; q++)
                    {
                        const int rr = NU + r;
                        a += (rr >= q ? GATL(P.RSQ, k * NP + PK(rr, q)) : GATL(P.RSQ, k * NP + PK(q, rr))) * GATL(P.ux, k * n + q);
                    }
                    for (int q = 0; q < NX; q++) a += GATL(P.BAt, (k * n + NU + r) * NX + q) * pn[q];
                    pk[r] = a;
                }
                for (int r = 0; r < NX; r++) { pn[r] = pk[r]; GATL(P.pi, k * NX + r) = pk[r]; }
            }
        }
        /* inequality rows */
        GQP_STAGE_REF Sc = Cd.st[jb];
        const int r0 = Mp.row_off[jb], nbc = Sc.nb;
        for (int rc = 0; rc < nbc; rc++)
        {
            const int kp = Mp.row_kp[r0 + rc], rp = Mp.row_rp[r0 + rc];
            GQP_STAGE_REF Sp = P.st[kp];
            GATL(P.lam, Sp.o_ct + rp) = GATL(Cd.lam, Sc.o_ct + rc);
            GATL(P.lam, Sp.o_ct + Sp.nb + rp) = GATL(Cd.lam, Sc.o_ct + nbc + rc);
            GATL(P.t, Sp.o_ct + rp) = GATL(Cd.t, Sc.o_ct + rc);
            GATL(P.t, Sp.o_ct + Sp.nb + rp) = GATL(Cd.t, Sc.o_ct + nbc + rc);
        }
    }
    P.iter[i] = Cd.iter[i];
    P.status[i] = Cd.status[i];
    P.mu[i] = Cd.mu[i];
    P.obj[i] = Cd.obj[i];
    for (int q = 0; q < 4; q++) P.res[q * Bp + i] = Cd.res[q * Bp + i];
}

/* ---------------------------------------------------------------------------------------------------
 * Wave-per-instance versions (run-time dims, any NX + BS*NU <= 64): one 64-lane workgroup condenses /
 * expands one instance, the block matrices live in LDS and the lanes split the output entries.  Parent and
 * child arrays are addressed through a layout-agnostic accessor (either may be wave-tiled or instance-major).
 * Same arithmetic as k_pcond / k_pexpand above, which stay as the one-instance-per-lane reference
 * (ACADOS_AMD_PCOND_1TPI=1).
 * --------------------------------------------------------------------------------------------------- */
#ifndef GQP_DYN_SHARED
#define GQP_DYN_SHARED(name) extern __shared__ double name[]
#endif
#define PLAT(arr, e) (arr).p[(arr).aos ? (size_t) inst * (size_t) (arr).E + (size_t) (e) \
                                       : ((size_t) (inst >> 6) * (size_t) (arr).E + (size_t) (e)) * 64 + (size_t) (inst & 63)]

/* workgroup -> instance.  With a wave-tiled parent, element e of the instances 8j..8j+7 shares one 64-byte line;
 * workgroup b runs on XCD b % 8 (each XCD has its own L2), so the eight workgroups b = 8q..8q+7 -- dispatched together
 * -- would pull the same lines into eight L2s.  The map below gives every XCD its own eight-instance group of each
 * 64-instance tile (XCD x: instances 8x..8x+7 of the tile), consecutive workgroups of an XCD walk through it: each
 * line is fetched by one L2 only.  A bijection on every tile; the grid covers whole tiles. */
__device__ static inline int pcond_inst(int b) { return ((b >> 6) << 6) + ((b & 7) << 3) + ((b >> 3) & 7); }

__host__ __device__ static inline size_t pcondw_lds_doubles(int NX, int NU, int NUC)
{
    const int n = NX + NU, NP = n * (n + 1) / 2, nc = NUC + NX, ncp = nc | 1, NPC = nc * (nc + 1) / 2;
    return (size_t) NP + (size_t) n * NX + 3 * (size_t) NX * ncp + NPC + 2 * (size_t) nc + 6 * 64 + 8;
}

static __global__ void __launch_bounds__(64) kw_pcond(GqpDev P, GqpDev Cd, PcondMap Mp)
{
    GQP_DYN_SHARED(smem);
    const int NX = P.NX, NU = P.NU, n = NX + NU, NP = n * (n + 1) / 2;
    const int NUC = Cd.NU, nc = NUC + NX, ncp = nc | 1, NPC = nc * (nc + 1) / 2;
    const int inst = pcond_inst(blockIdx.x), lane = threadIdx.x;
    if (inst >= P.B) return;
    double *H = smem, *Bl = H + NP, *X = Bl + n * NX, *Xn = X + NX * ncp, *T = Xn + NX * ncp, *Hb = T + NX * ncp;
    double *gb = Hb + NPC, *y = gb + nc + nc, *g = y + 64, *c = g + 64, *cn = c + 64, *bl = cn + 64;
    auto Hs = [&](int r, int q) { return r >= q ? H[PK(r, q)] : H[PK(q, r)]; };

    for (int jb = 0; jb <= Mp.N2; jb++)
    {
        const int k0 = Mp.blk_start[jb < Mp.N2 ? jb : Mp.N2];
        const int k1 = jb < Mp.N2 ? Mp.blk_start[jb + 1] : P.N + 1; /* terminal: stage N alone */
        const int bs = jb < Mp.N2 ? k1 - k0 : 1;
        for (int e = lane; e < NX * ncp; e += 64) X[e] = 0.0;
        for (int e = lane; e < NPC; e += 64) Hb[e] = 0.0;
        for (int e = lane; e < nc; e += 64) gb[e] = 0.0;
        c[lane] = 0.0;
        __syncthreads();
        if (lane < NX) X[lane * ncp + NUC + lane] = 1.0;
        __syncthreads();

        for (int ii = 0; ii < bs; ii++)
        {
            const int k = k0 + ii, u0 = ii * NU;
            for (int e = lane; e < NP; e += 64) H[e] = PLAT(P.RSQ, k * NP + e);
            for (int e = lane; e < n * NX; e += 64) Bl[e] = PLAT(P.BAt, k * n * NX + e);
            if (lane < n) g[lane] = PLAT(P.rq, k * n + lane);
            if (lane < NX) bl[lane] = PLAT(P.bvec, k * NX + lane);
            __syncthreads();
            /* rows of stage k that are general rows of the condensed stage: a_row' v_k = a_row' (Z_ii vbar + [0; c]),
             * i.e. coefficients a_u E_ii + a_x X, bounds shifted by a_x' c */
            {
                const int g0 = Mp.gk_off[k], ngk = Mp.gk_off[k + 1] - g0, gbase = g0 - Mp.gk_off[k0];
                GQP_STAGE_REF Sp = P.st[k];
                GQP_STAGE_REF Sc = Cd.st[jb];
                if (Mp.mode & 1)
                    for (int e = lane; e < ngk * nc; e += 64)
                    {
                        const int gi = e / nc, col = e - gi * nc, rp = Mp.g_rp[g0 + gi];
                        double s;
                        if (rp < Sp.nb) s = X[Mp.g_var[g0 + gi] * ncp + col];
                        else
                        {
                            const int rowp = (Sp.o_g + rp - Sp.nb) * n;
                            s = (col >= u0 && col < u0 + NU) ? PLAT(P.DCt, rowp + col - u0) : 0.0;
                            for (int q = 0; q < NX; q++) s += PLAT(P.DCt, rowp + NU + q) * X[q * ncp + col];
                        }
                        PLAT(Cd.DCt, (Sc.o_g + gbase + gi) * nc + col) = s;
                    }
                if (Mp.mode & 2)
                    for (int gi = lane; gi < ngk; gi += 64)
                    {
                        const int rp = Mp.g_rp[g0 + gi], rc = Sc.nb + gbase + gi;
                        double sh = 0.0;
                        if (rp < Sp.nb) sh = c[Mp.g_var[g0 + gi]];
                        else
                            for (int q = 0; q < NX; q++) sh += PLAT(P.DCt, (Sp.o_g + rp - Sp.nb) * n + NU + q) * c[q];
                        PLAT(Cd.dvec, Sc.o_ct + rc) = PLAT(P.dvec, Sp.o_ct + rp) - sh;
                        PLAT(Cd.dvec, Sc.o_ct + Sc.nb + Sc.ng + rc) = PLAT(P.dvec, Sp.o_ct + Sp.nb + Sp.ng + rp) - sh;
                    }
            }
            /* y = H [0; c] + g */
            if (lane < n)
            {
                double a = g[lane];
                for (int q = 0; q < NX; q++) a += Hs(lane, NU + q) * c[q];
                y[lane] = a;
            }
            /* T = Q X */
            if (Mp.mode & 1)
                for (int e = lane; e < NX * nc; e += 64)
                {
                    /* x_k does not depend on the inputs of stage k and later: those columns of X (and of T) are zero */
                    const int r = e / nc, col = e - r * nc;
                    double s = 0.0;
                    if (!(col >= u0 && col < NUC))
                        for (int q = 0; q < NX; q++) s += Hs(NU + r, NU + q) * X[q * ncp + col];
                    T[r * ncp + col] = s;
                }
            __syncthreads();
            /* gbar += Z' y */
            for (int col = lane; col < nc; col += 64)
            {
                double s = (col >= u0 && col < u0 + NU) ? y[col - u0] : 0.0;
                for (int r = 0; r < NX; r++) s += X[r * ncp + col] * y[NU + r];
                gb[col] += s;
            }
            /* Hbar += Z' H Z: every packed entry (p, q), q <= p, by ONE lane:
             *   X' Q X  +  [p in U] (S X)[a_p][q]  +  [q in U] (S X)[a_q][p]  +  [p, q in U] R[a_p][a_q] */
            if (Mp.mode & 1)
            {
                int pr = 0, pc2 = 0; /* (row, col) of packed entry `lane`, advanced by 64 entries per step */
                {
                    int e = lane;
                    while (e > pr) { e -= pr + 1; pr++; }
                    pc2 = e;
                }
                for (int e = lane; e < NPC; e += 64)
                {
                    double s = 0.0;
                    const bool zp = pr >= u0 && pr < NUC, zq = pc2 >= u0 && pc2 < NUC; /* zero columns of X */
                    if (!zp && !zq) for (int r = 0; r < NX; r++) s += X[r * ncp + pr] * T[r * ncp + pc2];
                    const bool pu = pr >= u0 && pr < u0 + NU, qu = pc2 >= u0 && pc2 < u0 + NU;
                    if (pu && !zq) for (int r = 0; r < NX; r++) s += H[PK(NU + r, pr - u0)] * X[r * ncp + pc2];
                    if (qu && !zp) for (int r = 0; r < NX; r++) s += H[PK(NU + r, pc2 - u0)] * X[r * ncp + pr];
                    if (pu && qu) s += H[PK(pr - u0, pc2 - u0)];
                    Hb[e] += s;
                    /* advance (pr, pc2) by 64 packed entries */
                    int adv = 64;
                    while (adv > 0)
                    {
                        const int room = pr - pc2; /* entries left in this row after pc2 */
                        if (adv <= room) { pc2 += adv; adv = 0; }
                        else { adv -= room + 1; pr++; pc2 = 0; }
                    }
                }
            }
            /* propagate x_{k+1} = A x_k + B u_k + b */
            if (jb < Mp.N2)
            {
                if (lane < NX)
                {
                    double a = bl[lane];
                    for (int q = 0; q < NX; q++) a += Bl[(NU + q) * NX + lane] * c[q];
                    cn[lane] = a;
                }
                for (int e = lane; e < NX * nc; e += 64)
                {
                    const int r = e / nc, col = e - r * nc;
                    double s = (col >= u0 && col < u0 + NU) ? Bl[(col - u0) * NX + r] : 0.0;
                    if (!(col >= u0 && col < NUC))
                        for (int q = 0; q < NX; q++) s += Bl[(NU + q) * NX + r] * X[q * ncp + col];
                    Xn[r * ncp + col] = s;
                }
            }
            __syncthreads();
            if (jb < Mp.N2)
            {
                for (int e = lane; e < NX * ncp; e += 64) X[e] = Xn[e];
                if (lane < NX) c[lane] = cn[lane];
            }
            __syncthreads();
        }
        /* unused input slots of a short block (and all of them at the terminal stage): unit diagonal */
        const int used = jb < Mp.N2 ? bs * NU : 0;
        for (int s2 = used + lane; s2 < NUC; s2 += 64) Hb[PK(s2, s2)] = 1.0;
        __syncthreads();

        /* ---- write child stage jb ---- */
        if (Mp.mode & 1)
        {
            for (int e = lane; e < NPC; e += 64) PLAT(Cd.RSQ, jb * NPC + e) = Hb[e];
            if (jb < Mp.N2)
                for (int e = lane; e < nc * NX; e += 64)
                {
                    const int col = e / NX, r = e - col * NX;
                    PLAT(Cd.BAt, jb * nc * NX + e) = X[r * ncp + col];
                }
        }
        if (Mp.mode & 2)
        {
            for (int e = lane; e < nc; e += 64) PLAT(Cd.rq, jb * nc + e) = gb[e];
            if (jb < Mp.N2 && lane < NX) PLAT(Cd.bvec, jb * NX + lane) = c[lane];
            /* box rows, activity bits, value of fixed variables */
            GQP_STAGE_REF Sc = Cd.st[jb];
            const int r0 = Mp.row_off[jb], nbc = Sc.nb, nbgc = Sc.nb + Sc.ng, q0 = Mp.slk_off[jb];
            for (int rc = lane; rc < nbc; rc += 64) /* box rows keep their bounds (general rows: written above) */
            {
                const int kp = Mp.row_kp[r0 + rc], rp = Mp.row_rp[r0 + rc];
                GQP_STAGE_REF Sp = P.st[kp];
                PLAT(Cd.dvec, Sc.o_ct + rc) = PLAT(P.dvec, Sp.o_ct + rp);
                PLAT(Cd.dvec, Sc.o_ct + nbgc + rc) = PLAT(P.dvec, Sp.o_ct + Sp.nb + Sp.ng + rp);
            }
            for (int sc = lane; sc < Sc.ns; sc += 64) /* slacks travel unchanged: cost, bounds */
            {
                const int kp = Mp.slk_kp[q0 + sc], sp = Mp.slk_sp[q0 + sc];
                GQP_STAGE_REF Sp = P.st[kp];
                const int cp = Sp.o_ct + 2 * (Sp.nb + Sp.ng), cc = Sc.o_ct + 2 * nbgc;
                PLAT(Cd.dvec, cc + sc) = PLAT(P.dvec, cp + sp);
                PLAT(Cd.dvec, cc + Sc.ns + sc) = PLAT(P.dvec, cp + Sp.ns + sp);
                for (int w = 0; w < 2; w++)
                {
                    PLAT(Cd.Zz, (Sc.o_s + sc) * 2 + w) = PLAT(P.Zz, (Sp.o_s + sp) * 2 + w);
                    PLAT(Cd.Zz, (Sc.o_s + Sc.ns + sc) * 2 + w) = PLAT(P.Zz, (Sp.o_s + Sp.ns + sp) * 2 + w);
                }
            }
            if (lane == 0)
            {
                uint64_t amc[2] = {0, 0};
                auto take = [&](int kp, int side_p, int side_c) {
                    if ((PLAT(P.amask, kp * P.AW + (side_p >> 6)) >> (side_p & 63)) & 1) amc[side_c >> 6] |= (uint64_t) 1 << (side_c & 63);
                };
                for (int rc = 0; rc < nbgc; rc++)
                {
                    const int kp = Mp.row_kp[r0 + rc], rp = Mp.row_rp[r0 + rc];
                    take(kp, rp, rc);
                    take(kp, P.st[kp].nb + P.st[kp].ng + rp, nbgc + rc);
                }
                for (int sc = 0; sc < Sc.ns; sc++)
                {
                    const int kp = Mp.slk_kp[q0 + sc], sp = Mp.slk_sp[q0 + sc];
                    const int bp = 2 * (P.st[kp].nb + P.st[kp].ng);
                    take(kp, bp + sp, 2 * nbgc + sc);
                    take(kp, bp + P.st[kp].ns + sp, 2 * nbgc + Sc.ns + sc);
                }
                for (int w = 0; w < Cd.AW; w++) PLAT(Cd.amask, jb * Cd.AW + w) = amc[w];
            }
            if (lane < NX && ((Sc.emask >> (NUC + lane)) & 1)) PLAT(Cd.ux, jb * nc + NUC + lane) = PLAT(P.ux, k0 * n + NU + lane);
        }
        __syncthreads();
    }
}

/* expansion: original (ux, pi, lam, t) from the condensed solution */
static __global__ void __launch_bounds__(64) kw_pexpand(GqpDev P, GqpDev Cd, PcondMap Mp)
{
    GQP_DYN_SHARED(smem);
    const int NX = P.NX, NU = P.NU, n = NX + NU, NP = n * (n + 1) / 2;
    const int NUC = Cd.NU, nc = NUC + NX;
    const int inst = pcond_inst(blockIdx.x), lane = threadIdx.x;
    if (inst >= P.B) return;
    double *x = smem, *u = x + 64, *pn = u + 64, *ux = pn + 64; /* ux: [u; x] of the stage being visited */

    /* The blocks of an instance are independent of each other here -- every block starts from the condensed solution's own state and
     * multiplier -- so the grid is (instances, blocks): one wave walking all N2 + 1 blocks of its instance was a chain of N dependent
     * mat-vecs with one round trip to memory each (0.91 ms for 4,096 C3-shaped instances, a fifth of the device time of such a batch) */
    {
        const int jb = blockIdx.y;
        const int k0 = Mp.blk_start[jb < Mp.N2 ? jb : Mp.N2];
        const int bs = jb < Mp.N2 ? Mp.blk_start[jb + 1] - k0 : 1;
        /* forward simulation of the eliminated states */
        if (lane < NX) x[lane] = PLAT(Cd.ux, jb * nc + NUC + lane);
        __syncthreads();
        for (int ii = 0; ii < bs; ii++)
        {
            const int k = k0 + ii;
            if (lane < NU) u[lane] = jb < Mp.N2 ? PLAT(Cd.ux, jb * nc + ii * NU + lane) : 0.0;
            __syncthreads();
            if (lane < NU) PLAT(P.ux, k * n + lane) = u[lane];
            if (lane < NX) PLAT(P.ux, k * n + NU + lane) = x[lane];
            double xn = 0.0;
            const bool step = jb < Mp.N2 && ii + 1 < bs;
            if (step && lane < NX)
            {
                xn = PLAT(P.bvec, k * NX + lane);
                for (int q = 0; q < NX; q++) xn += PLAT(P.BAt, (k * n + NU + q) * NX + lane) * x[q];
                for (int q = 0; q < NU; q++) xn += PLAT(P.BAt, (k * n + q) * NX + lane) * u[q];
            }
            __syncthreads();
            if (step && lane < NX) x[lane] = xn;
            __syncthreads();
        }
        /* inequality rows and slacks: same constraints, same multipliers */
        GQP_STAGE_REF Sc = Cd.st[jb];
        const int r0 = Mp.row_off[jb], nbgc = Sc.nb + Sc.ng, q0 = Mp.slk_off[jb];
        for (int rc = lane; rc < nbgc; rc += 64)
        {
            const int kp = Mp.row_kp[r0 + rc], rp = Mp.row_rp[r0 + rc];
            GQP_STAGE_REF Sp = P.st[kp];
            const int up = Sp.o_ct + Sp.nb + Sp.ng + rp, uc = Sc.o_ct + nbgc + rc;
            PLAT(P.lam, Sp.o_ct + rp) = PLAT(Cd.lam, Sc.o_ct + rc);
            PLAT(P.lam, up) = PLAT(Cd.lam, uc);
            PLAT(P.t, Sp.o_ct + rp) = PLAT(Cd.t, Sc.o_ct + rc);
            PLAT(P.t, up) = PLAT(Cd.t, uc);
        }
        for (int sc = lane; sc < Sc.ns; sc += 64)
        {
            const int kp = Mp.slk_kp[q0 + sc], sp = Mp.slk_sp[q0 + sc];
            GQP_STAGE_REF Sp = P.st[kp];
            const int cp = Sp.o_ct + 2 * (Sp.nb + Sp.ng), cc = Sc.o_ct + 2 * nbgc;
            PLAT(P.sv, Sp.o_s + sp) = PLAT(Cd.sv, Sc.o_s + sc);
            PLAT(P.sv, Sp.o_s + Sp.ns + sp) = PLAT(Cd.sv, Sc.o_s + Sc.ns + sc);
            PLAT(P.lam, cp + sp) = PLAT(Cd.lam, cc + sc);
            PLAT(P.lam, cp + Sp.ns + sp) = PLAT(Cd.lam, cc + Sc.ns + sc);
            PLAT(P.t, cp + sp) = PLAT(Cd.t, cc + sc);
            PLAT(P.t, cp + Sp.ns + sp) = PLAT(Cd.t, cc + Sc.ns + sc);
        }
        __syncthreads(); /* the multipliers of the inner stages are read back below */
        /* multipliers of the eliminated dynamics, backwards:
         * pi_k = Q x_k + S' u_k + q_k + A_k' pi_{k+1} - J_x'(lam_lower - lam_upper) over the active rows of stage k */
        if (jb < Mp.N2)
        {
            const int k1 = k0 + bs;
            if (lane < NX) { pn[lane] = PLAT(Cd.pi, (jb + 1) * NX + lane); PLAT(P.pi, k1 * NX + lane) = pn[lane]; }
            __syncthreads();
            for (int k = k1 - 1; k > k0; k--)
            {
                if (lane < n) ux[lane] = PLAT(P.ux, k * n + lane);
                __syncthreads();
                double pk = 0.0;
                if (lane < NX)
                {
                    const int rr = NU + lane;
                    pk = PLAT(P.rq, k * n + rr);
                    for (int q = 0; q < n; q++) pk += PLAT(P.RSQ, k * NP + (rr >= q ? PK(rr, q) : PK(q, rr))) * ux[q];
                    for (int q = 0; q < NX; q++) pk += PLAT(P.BAt, (k * n + rr) * NX + q) * pn[q];
                    GQP_STAGE_REF Sp = P.st[k];
                    const int nbgp = Sp.nb + Sp.ng;
                    auto dlam = [&](int rp) { /* lam_lower - lam_upper of sorted row rp, active sides only */
                        const int su_ = nbgp + rp;
                        const bool al = (PLAT(P.amask, k * P.AW + (rp >> 6)) >> (rp & 63)) & 1;
                        const bool au = (PLAT(P.amask, k * P.AW + (su_ >> 6)) >> (su_ & 63)) & 1;
                        return (al ? PLAT(P.lam, Sp.o_ct + rp) : 0.0) - (au ? PLAT(P.lam, Sp.o_ct + su_) : 0.0);
                    };
                    if ((Sp.bmask >> rr) & 1) pk -= dlam(popc64_g(Sp.bmask & (((uint64_t) 1 << rr) - 1)));
                    for (int gq = 0; gq < Sp.ng; gq++) pk -= PLAT(P.DCt, (Sp.o_g + gq) * n + rr) * dlam(Sp.nb + gq);
                }
                __syncthreads();
                if (lane < NX) { pn[lane] = pk; PLAT(P.pi, k * NX + lane) = pk; }
                __syncthreads();
            }
        }
        __syncthreads();
    }
    if (lane == 0 && blockIdx.y == 0)
    {
        P.iter[inst] = Cd.iter[inst];
        P.status[inst] = Cd.status[inst];
        P.mu[inst] = Cd.mu[inst];
        P.obj[inst] = Cd.obj[inst];
        for (int q = 0; q < 4; q++) P.res[(size_t) q * P.Bp + inst] = Cd.res[(size_t) q * P.Bp + inst];
    }
}

/* ---------------------------------------------------------------------------------------------------
 * A solution (or initial guess) of the ORIGINAL QP restated in the variables of the condensed one: what
 * `condense_qp_out` does (ocp_qp_partial_condensing_condense_qp_out, ocp_qp_partial_condensing.c:559-571 ->
 * d_part_cond_qp_cond_sol; called by ocp_qp_xcond_solve when initialize_next_xcond_qp_from_qp_out is set,
 * ocp_qp_xcond_solver.c:554-565).  Pure index work, the inverse of the copies in kw_pexpand: block inputs stacked,
 * the block's first state, pi at the block boundaries, every inequality row / slack with its multipliers.
 * One instance per lane, layout-agnostic.
 * --------------------------------------------------------------------------------------------------- */
static __global__ void __launch_bounds__(64) k_pcond_sol(GqpDev P, GqpDev Cd, PcondMap Mp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.B) return;
    const int NX = P.NX, NU = P.NU, n = NX + NU, NUC = Cd.NU, nc = NUC + NX;
    for (int jb = 0; jb <= Mp.N2; jb++)
    {
        const int k0 = Mp.blk_start[jb < Mp.N2 ? jb : Mp.N2];
        const int bs = jb < Mp.N2 ? Mp.blk_start[jb + 1] - k0 : 0;
        for (int e = 0; e < NUC; e++)
            GATL(Cd.ux, jb * nc + e) = e < bs * NU ? GATL(P.ux, (k0 + e / NU) * n + e % NU) : 0.0;
        for (int r = 0; r < NX; r++) GATL(Cd.ux, jb * nc + NUC + r) = GATL(P.ux, k0 * n + NU + r);
        /* pi slot s = multiplier of the dynamics producing x_s: child slot jb+1 <-> parent slot blk_start[jb+1] */
        if (jb < Mp.N2)
            for (int r = 0; r < NX; r++) GATL(Cd.pi, (jb + 1) * NX + r) = GATL(P.pi, (k0 + bs) * NX + r);
        GQP_STAGE_REF Sc = Cd.st[jb];
        const int r0 = Mp.row_off[jb], nbgc = Sc.nb + Sc.ng, q0 = Mp.slk_off[jb];
        for (int rc = 0; rc < nbgc; rc++)
        {
            const int kp = Mp.row_kp[r0 + rc], rp = Mp.row_rp[r0 + rc];
            GQP_STAGE_REF Sp = P.st[kp];
            const int up = Sp.o_ct + Sp.nb + Sp.ng + rp, uc = Sc.o_ct + nbgc + rc;
            GATL(Cd.lam, Sc.o_ct + rc) = GATL(P.lam, Sp.o_ct + rp);
            GATL(Cd.lam, uc) = GATL(P.lam, up);
            GATL(Cd.t, Sc.o_ct + rc) = GATL(P.t, Sp.o_ct + rp);
            GATL(Cd.t, uc) = GATL(P.t, up);
        }
        for (int sc = 0; sc < Sc.ns; sc++)
        {
            const int kp = Mp.slk_kp[q0 + sc], sp = Mp.slk_sp[q0 + sc];
            GQP_STAGE_REF Sp = P.st[kp];
            const int cp = Sp.o_ct + 2 * (Sp.nb + Sp.ng), cc = Sc.o_ct + 2 * nbgc;
            GATL(Cd.sv, Sc.o_s + sc) = GATL(P.sv, Sp.o_s + sp);
            GATL(Cd.sv, Sc.o_s + Sc.ns + sc) = GATL(P.sv, Sp.o_s + Sp.ns + sp);
            GATL(Cd.lam, cc + sc) = GATL(P.lam, cp + sp);
            GATL(Cd.lam, cc + Sc.ns + sc) = GATL(P.lam, cp + Sp.ns + sp);
            GATL(Cd.t, cc + sc) = GATL(P.t, cp + sp);
            GATL(Cd.t, cc + Sc.ns + sc) = GATL(P.t, cp + Sp.ns + sp);
        }
    }
}

} // namespace gqp


#endif
