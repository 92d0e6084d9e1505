/*
 * ipm_kernels_w16.hpp -- SIXTEEN LANES PER INSTANCE: small stage blocks (nu + nx <= 16, box constraints).
 *
 * The wave-per-instance kernels (ipm_kernels_wpi.hpp) spend a whole wavefront on an 11 x 11 stage block and are
 * bound by LDS round trips.  Here one wavefront carries FOUR instances (one per 16-lane DPP row); lane l of a row
 * owns variable l AND row l of every stage matrix, in REGISTERS:
 *   - the O(n^3) parts (W = [B A]' Lx+, M = H~ + W W', Cholesky with the rhs riding along) run on register rows
 *     with `v_mov_b32_dpp row_newbcast:j` as the broadcast of lane j's value to its row -- no LDS, no barrier;
 *   - the few products that need a COLUMN of a factor (Lx+' rb, the back substitution) exchange through a small
 *     per-instance LDS tile (write rows, read columns);
 *   - arrays are instance-major as for the wave-per-instance family, whose init / finalize kernels are reused.
 * Same algorithm and HBM contract (arrays, slot conventions, p-form of the corrector sweep) as ipm_kernels_wpi.hpp.
 * Shapes are compile-time (NX, NU): the broadcast lane is an immediate of the DPP instruction.
 *
 * Rows are independent: a row whose instance has converged leaves the kernel, the other rows of the wave go on
 * (DPP row operations never cross rows).  Inside a live row all 16 lanes execute every broadcast.
 * The CPU test tier runs these kernels under tests/hostsim: lanes are host threads, the row broadcast goes through
 * a shared buffer and a per-row barrier (GQP_ROWSYNC).
 */
#ifndef IPM_KERNELS_W16_HPP_
#define IPM_KERNELS_W16_HPP_

#include "ipm_kernels_wpi.hpp"

namespace gqp
{

#if defined(__HIP_DEVICE_COMPILE__)
/* all lanes of a row see each other's LDS writes in program order (one wavefront); only the compiler must be kept
 * from reordering across the exchange */
#define GQP_ROWSYNC()                                                   \
    do {                                                                \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          \
        __builtin_amdgcn_wave_barrier();                                \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");          \
    } while (0)
template <int J>
__device__ static inline double w16_bc(double v)
{
    /* one v_mov_b64_dpp (gfx90a+: 64-bit DPP exists for row_newbcast); every lane of the row is written
     * (bound_ctrl, full masks), so there is no `old` value to initialise */
    return __builtin_amdgcn_update_dpp(v, v, 0x150 + J, 0xF, 0xF, true);
}
/* value of lane j (0..15) of this lane's row; j must be a compile-time constant after unrolling */
__device__ static inline double w16_bcast(double v, int j, double *)
{
    switch (j)
    {
        case 0: return w16_bc<0>(v); case 1: return w16_bc<1>(v); case 2: return w16_bc<2>(v); case 3: return w16_bc<3>(v);
        case 4: return w16_bc<4>(v); case 5: return w16_bc<5>(v); case 6: return w16_bc<6>(v); case 7: return w16_bc<7>(v);
        case 8: return w16_bc<8>(v); case 9: return w16_bc<9>(v); case 10: return w16_bc<10>(v); case 11: return w16_bc<11>(v);
        case 12: return w16_bc<12>(v); case 13: return w16_bc<13>(v); case 14: return w16_bc<14>(v); default: return w16_bc<15>(v);
    }
}
#else
/* host simulation (tests/hostsim defines GQP_ROWSYNC as a per-row thread barrier) and the host pass of hipcc
 * (kernels are only parsed there): the broadcast goes through the exchange buffer */
#ifndef GQP_ROWSYNC
#define GQP_ROWSYNC() do { } while (0)
#endif
__device__ static inline double w16_bcast(double v, int j, double *xbuf)
{
    const int l = threadIdx.x & 15;
    xbuf[l] = v;
    GQP_ROWSYNC();
    const double r = xbuf[j];
    GQP_ROWSYNC();
    return r;
}
#endif

/* per-instance LDS tile (doubles): exchange buffer, x-block tiles of two factors, one full factor tile, vectors */
template <int NX, int NU>
struct W16Lds
{
    static constexpr int n = NX + NU, LDX = NX + 1, LDF = n + 1;
    static constexpr int XB = 0, TA = 16, TB = TA + NX * LDX, TF = TB + NX * LDX, VEC = TF + n * LDF, SZ = VEC + 16;
};

#define W16_UNROLL _Pragma("unroll")

/* instance of row slot `slot` (GqpDev::perm), -1 beyond the live ones */
__device__ static inline int w16_slot_inst(const GqpDev &D, int slot)
{
    if (!D.perm) return slot < D.B ? slot : -1;
    return slot < D.n_perm ? D.perm[slot] : -1;
}
/* the still-iterating instances, densely: one thread per instance, order of arrival (the placement of an instance does not
 * enter its arithmetic) */
static __global__ void k_active_perm(GqpDev D, int *perm, int *count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < D.B && D.status[i] == GQP_RUNNING) perm[atomicAdd(count, 1)] = i;
}

/* reductions over the 16 lanes of a row (result in every lane) */
__device__ static inline double w16_rmax(double v, double *xb)
{
    double r = v;
    W16_UNROLL for (int j = 0; j < 16; j++)
    {
        const double o = w16_bcast(v, j, xb);
        r = (o > r || o != o) ? o : r;
    }
    return r;
}
__device__ static inline double w16_rsum(double v, double *xb)
{
    double r = 0.0;
    W16_UNROLL for (int j = 0; j < 16; j++) r += w16_bcast(v, j, xb);
    return r;
}
__device__ static inline double w16_rmin(double v, double *xb)
{
    double r = v;
    W16_UNROLL for (int j = 0; j < 16; j++)
    {
        const double o = w16_bcast(v, j, xb);
        r = o < r ? o : r;
    }
    return r;
}

/* slack index of sorted box row ib (< 16 here) without a lane-indexed memory access: the first 16 entries of the
 * stage's row -> slack map are two uniform 64-bit words, the lane picks its byte */
__device__ static inline int w16_srev(GQP_STAGE_REF S, int ib)
{
    const GQP_CONST_AS uint64_t *sp = (const GQP_CONST_AS uint64_t *) S.srev;
    const uint64_t w = ib < 8 ? sp[0] : sp[1];
    return (int) (int8_t) (w >> ((ib & 7) * 8));
}

/* ------------------------------------------------------------------------------------------------ factor */

/* SOFT: box rows may be soft, every slack belongs to exactly ONE box row (no general rows, no shared slacks -- the
 * host checks it, gpu_batch.hip): the slack block of a row is eliminated by the lane that owns the row, with the
 * cancellation-free formulas of ipm_kernels_wpi.hpp specialised to one row per slack (E = Z + Gamma_s, X = slack
 * stationarity + rho_s); nothing crosses lanes. */
/* Stages requested ahead of the one being computed (register ring, see kx_factor_body); a record is 3 n + NX + 12 doubles
 * per lane.  Measured at 7,281 instances (MI355X, factor launch, depth 0 -> 1): nx=4 N=100 396 -> 376 us, nx=8 nu=3 N=50
 * 364 -> 331 us, nx=12 nu=3 N=100 897 -> 806 us; depth 2 spills (nx=12: 1,344 us) or costs the second wave per SIMD. */
template <int NX, int NU, bool SOFT>
struct W16Prefetch
{
    static constexpr int FACTOR = SOFT ? 0 : 1;
};

template <int NX, int NU, bool SOFT>
__device__ static inline void kx_factor_body(const GqpDev &D, const GqpOpts &O, int redo)
{
    GQP_DYN_SHARED(smem);
    typedef W16Lds<NX, NU> LY;
    constexpr int n = NX + NU, NP = n * (n + 1) / 2, LDX = LY::LDX;
    const int l = threadIdx.x & 15, inst = w16_slot_inst(D, blockIdx.x * 4 + (threadIdx.x >> 4));
    if (inst < 0) return;
    if (D.status[inst] != GQP_RUNNING) return;
    double *T = smem + (threadIdx.x >> 4) * LY::SZ, *xb = T + LY::XB;
    double *TA = T + LY::TA; /* x-block of the factor of stage k+1, [q][c] */
    const bool mine = l < n, isx = l >= NU && l < n;
    const int cx = isx ? l - NU : 0; /* state index of a state lane */

    double Lp[NX];   /* row (NU + cx) of the x-block of the factor of stage k+1 (state lanes) */
    double LpT[NX];  /* column cx of the same block */
    W16_UNROLL for (int c = 0; c < NX; c++) { Lp[c] = 0.0; LpT[c] = 0.0; }
    double lxn = 0.0; /* lx+ of this lane's state */
    double nrm_g = 0.0, nrm_b = 0.0, nrm_d = 0.0, nrm_m = 0.0, musum = 0.0, obj = 0.0, nact = 0.0;

    /* Everything a stage reads from HBM (symmetric row l of H, row l and column cx of [B A]', vectors, the lane's box row:
     * n + NX + n + 12 doubles), as one record.  Small shapes keep a ring of PD records in registers: the record of stage
     * k - PD is requested while stage k computes.  A stage is a dependent chain load -> compute -> store, the per-instance
     * data of a batch do not fit any cache, and a batch of a few thousand small instances puts fewer than two waves on a
     * SIMD, so nothing else hides the HBM round trips: they, not the arithmetic, were the launch time of these shapes.
     * Every address is independent of loaded data (the box row of a lane follows from the stage's masks; a lane without a
     * row reads row 0 of the stage: always readable), the activity bits select afterwards. */
    struct StageLoads
    {
        double M[n], Br[NX], Bc[n], v, g, bv, xn, pin, pik, ll, lu, tl, tu, dl, du;
        uint64_t am;
    };
#if defined(W16_PF_DEPTH)
    constexpr int PD = W16_PF_DEPTH;
#else
    constexpr int PD = W16Prefetch<NX, NU, SOFT>::FACTOR;
#endif
    constexpr int PR = PD > 0 ? PD : 1;
    const int lc_ = mine ? l : 0, xc_ = isx ? cx : 0;
    auto load_stage = [&](int k, StageLoads &F)
    {
        GQP_STAGE_REF S = D.st[k];
        F.am = WAT(D.amask, k * D.AW);
        W16_UNROLL for (int c = 0; c < n; c++) F.M[c] = WAT(D.RSQ, k * NP + (c <= lc_ ? PK(lc_, c) : PK(c, lc_)));
        W16_UNROLL for (int c = 0; c < NX; c++) F.Br[c] = WAT(D.BAt, (k * n + lc_) * NX + c);
        W16_UNROLL for (int r = 0; r < n; r++) F.Bc[r] = WAT(D.BAt, (k * n + r) * NX + xc_);
        F.v = WAT(D.ux, k * n + lc_); F.g = WAT(D.rq, k * n + lc_);
        F.bv = WAT(D.bvec, k * NX + xc_); F.xn = WAT(D.ux, (k + 1) * n + NU + xc_);
        F.pin = WAT(D.pi, (k + 1) * NX + xc_); F.pik = WAT(D.pi, k * NX + xc_);
        const bool hs = mine && (((S.bmask & ~S.emask) >> l) & 1);
        const int ib = hs ? popc64(S.bmask & (((uint64_t) 1 << l) - 1)) : 0;
        const int el = S.o_ct + ib, eu = el + S.nb;
        F.ll = WAT(D.lam, el); F.lu = WAT(D.lam, eu);
        F.tl = WAT(D.t, el); F.tu = WAT(D.t, eu);
        F.dl = WAT(D.dvec, el); F.du = WAT(D.dvec, eu);
    };
    StageLoads R[PR];
    if (PD > 0)
    {
        W16_UNROLL for (int d = 0; d < PR; d++)
            if (D.N - d >= 0) load_stage(D.N - d, R[d]);
    }

    for (int k0 = D.N; k0 >= 0; k0 -= PR)
    W16_UNROLL for (int d = 0; d < PR; d++)
    {
        const int k = k0 - d;
        if (k < 0) break;
        GQP_STAGE_REF S = D.st[k];
        const uint64_t imask = S.bmask & ~S.emask;
        const int nbg = S.nb;
        const bool fixed = mine && ((S.emask >> l) & 1);
        StageLoads F;
        if (PD > 0)
        {
            F = R[d];
            if (k - PR >= 0) load_stage(k - PR, R[d]);
        }
        else load_stage(k, F);
        const uint64_t am = F.am;

        /* ---- idle lanes have read through clamped indices and zero the values now: no exec-masked branch per load ---- */
        const double zm = mine ? 1.0 : 0.0, zx = isx ? 1.0 : 0.0;
        double M[n], Br[NX], Bc[n];
        W16_UNROLL for (int c = 0; c < n; c++) M[c] = zm * F.M[c];
        W16_UNROLL for (int c = 0; c < NX; c++) Br[c] = zm * F.Br[c];
        W16_UNROLL for (int r = 0; r < n; r++) Bc[r] = zx * F.Bc[r];
        const double v = zm * F.v, g = zm * F.g;
        double rb = zx * (F.bv - F.xn);
        const double pin = zx * F.pin, pik = zx * F.pik;
        const bool has = mine && ((imask >> l) & 1);
        const int ib = has ? popc64(S.bmask & (((uint64_t) 1 << l) - 1)) : 0;
        const bool al = has && ((am >> ib) & 1), au = has && ((am >> (nbg + ib)) & 1);
        const int el = S.o_ct + ib, eu = el + nbg; /* row 0 of the stage when the lane has no row: always readable */
        const double ll = al ? F.ll : 0.0, lu = au ? F.lu : 0.0;
        const double ttl = al ? F.tl : 1.0, ttu = au ? F.tu : 1.0;
        const double lbv = al ? F.dl : 0.0, ubv = au ? F.du : 0.0;
        /* SOFT: the slack of this lane's row (values, cost, its two bound rows) */
        const int sj = (SOFT && has) ? w16_srev(S, ib) : -1;
        const int sq = sj >= 0 ? sj : 0, se0 = S.o_ct + 2 * nbg + sq, se1 = se0 + S.ns;
        const bool sal = sj >= 0 && ((am >> (2 * nbg + sq)) & 1), sau = sj >= 0 && ((am >> (2 * nbg + S.ns + sq)) & 1);
        double ssl = 0.0, ssu = 0.0;
        if (SOFT && sj >= 0) { ssl = WAT(D.sv, S.o_s + sq); ssu = WAT(D.sv, S.o_s + S.ns + sq); }

        /* ---- rb += [B A] v, H v (one broadcast of v per variable serves both) ---- */
        double hv = 0.0;
        W16_UNROLL for (int r = 0; r < n; r++)
        {
            const double vr = w16_bcast(v, r, xb);
            rb += Bc[r] * vr;
            hv += M[r] * vr;
        }
        /* [B A]' pi+ */
        double bp = 0.0;
        W16_UNROLL for (int c = 0; c < NX; c++) bp += Br[c] * w16_bcast(pin, NU + c, xb);
        double gt = 0.0, gadd = 0.0, gam = 0.0;
        if (mine)
        {
            obj += (0.5 * hv + g) * v;
            gt = bp + hv + g - pik;
        }
        if (isx) { nacc(nrm_b, rb); WAT(D.rb, k * NX + cx) = rb; }
        if (has)
        {
            const double rdl = al ? v + ssl - lbv - ttl : 0.0, rdu = au ? ubv - v + ssu - ttu : 0.0;
            const double rml = al ? ll * ttl - O.tau_min : 0.0, rmu = au ? lu * ttu - O.tau_min : 0.0;
            nacc(nrm_d, rdl); nacc(nrm_d, rdu); nacc(nrm_m, rml); nacc(nrm_m, rmu);
            musum += ll * ttl + lu * ttu;
            nact += (double) ((int) al + (int) au);
            gt -= ll - lu;
            const double itl = frcp(ttl), itu = frcp(ttu);
            const double bGl = ll * itl, bGu = lu * itu, bRl = (rml + ll * rdl) * itl, bRu = (rmu + lu * rdu) * itu;
            gam = bGl + bGu;
            gadd = bRl - bRu;
            WAT(D.rd, el) = rdl;
            WAT(D.rd, eu) = rdu;
            if (SOFT && sj >= 0)
            {
                /* loaded here, not with the stage's other loads: the kernel sits at the 256-VGPR line (two waves per
                 * SIMD) and these twelve values would stay live across the broadcast loops above */
                const double sll = sal ? WAT(D.lam, se0) : 0.0, slu = sau ? WAT(D.lam, se1) : 0.0;
                const double stl = sal ? WAT(D.t, se0) : 1.0, stu = sau ? WAT(D.t, se1) : 1.0;
                const double sdl = sal ? WAT(D.dvec, se0) : 0.0, sdu = sau ? WAT(D.dvec, se1) : 0.0;
                const double sZl = WAT(D.Zz, (S.o_s + sq) * 2), szl = WAT(D.Zz, (S.o_s + sq) * 2 + 1);
                const double sZu = WAT(D.Zz, (S.o_s + S.ns + sq) * 2), szu = WAT(D.Zz, (S.o_s + S.ns + sq) * 2 + 1);
                obj += (0.5 * sZl * ssl + szl) * ssl + (0.5 * sZu * ssu + szu) * ssu;
                const double srdl = sal ? ssl - sdl - stl : 0.0, srdu = sau ? ssu - sdu - stu : 0.0;
                const double srml = sal ? sll * stl - O.tau_min : 0.0, srmu = sau ? slu * stu - O.tau_min : 0.0;
                nacc(nrm_d, srdl); nacc(nrm_d, srdu); nacc(nrm_m, srml); nacc(nrm_m, srmu);
                musum += sll * stl + slu * stu;
                nact += (double) ((int) sal + (int) sau);
                const double sitl = frcp(stl), situ = frcp(stu);
                const double sGl = sll * sitl, sGu = slu * situ;
                const double sPl = (srml + sll * srdl) * sitl, sPu = (srmu + slu * srdu) * situ;
                WAT(D.rd, se0) = srdl;
                WAT(D.rd, se1) = srdu;
                const double Rl = sZl * ssl + szl - sll - ll, Ru = sZu * ssu + szu - slu - lu; /* slack stationarity */
                nacc(nrm_g, Rl); nacc(nrm_g, Ru);
                WAT(D.rgs, S.o_s + sq) = Rl; WAT(D.rgs, S.o_s + S.ns + sq) = Ru;
                const double El = sZl + sGl, Eu = sZu + sGu, Xl = Rl + sPl, Xu = Ru + sPu; /* D, r~ without the row */
                const double Dl = El + bGl, Du = Eu + bGu;
                WAT(D.sD, S.o_s + sq) = Dl; WAT(D.sD, S.o_s + S.ns + sq) = Du;
                WAT(D.sR, S.o_s + sq) = Xl + bRl; WAT(D.sR, S.o_s + S.ns + sq) = Xu + bRu;
                const double Il = Dl != 0.0 ? frcp(Dl) : 0.0, Iu = Du != 0.0 ? frcp(Du) : 0.0;
                gam = bGl * El * Il + bGu * Eu * Iu;
                gadd = (bRl * El - bGl * Xl) * Il - (bRu * Eu - bGu * Xu) * Iu;
            }
        }
        if (fixed) gt = 0.0;
        if (mine) { nacc(nrm_g, gt); WAT(D.rg, k * n + l) = gt; }

        /* ---- W row: W[c] = sum_{q >= c} Br[q] Lx+[q][c], Lx+[q][c] = entry c of the row held by lane NU + q ---- */
        double W[NX];
        W16_UNROLL for (int c = 0; c < NX; c++) W[c] = 0.0;
        W16_UNROLL for (int q = 0; q < NX; q++)
            W16_UNROLL for (int c = 0; c <= q; c++) W[c] += Br[q] * w16_bcast(Lp[c], NU + q, xb);
        /* w0[c] (state lanes) = lx+[c] + sum_{q >= c} Lx+[q][c] rb[q]: column c of Lx+ is LpT */
        double w0 = lxn;
        W16_UNROLL for (int q = 0; q < NX; q++)
        {
            const double rbq = w16_bcast(rb, NU + q, xb);
            w0 += (q >= cx ? LpT[q] : 0.0) * rbq;
        }
        if (!isx) w0 = 0.0;
        /* m = gt + gadd + W w0 */
        double m = gt + gadd;
        W16_UNROLL for (int c = 0; c < NX; c++) m += W[c] * w16_bcast(w0, NU + c, xb);
        if (fixed || !mine) m = 0.0;
        /* ---- M += W W' + reg + Gamma: M[c] += sum_q W_l[q] W_c[q] ---- */
        W16_UNROLL for (int q = 0; q < NX; q++)
            W16_UNROLL for (int c = 0; c < n; c++) M[c] += W[q] * w16_bcast(W[q], c, xb);
        W16_UNROLL for (int c = 0; c < n; c++) M[c] += (c == l) ? O.reg_prim + gam : 0.0;
        if (S.emask) /* uniform: only a stage with fixed variables pays for the masking */
        {
            W16_UNROLL for (int c = 0; c < n; c++)
            {
                const bool fc = (S.emask >> c) & 1;
                if (fixed || fc) M[c] = (c == l) ? 1.0 : 0.0;
            }
        }

        /* ---- Cholesky on register rows; the rhs entry m rides along (l = L^{-1} m) ---- */
        W16_UNROLL for (int j = 0; j < n; j++)
        {
            const double d = w16_bcast(M[j], j, xb);
            const bool pos = d > 0.0;
            const double inv0 = frsqrt(pos ? d : 1.0);
            const double inv = pos ? inv0 : 0.0;
            const double lj = w16_bcast(m, j, xb) * inv;
            /* L[l][j] = M[l][j] inv for l >= j (the pivot lane holds d: d inv = sqrt(d), 0 for a non-positive pivot);
             * rows above keep their entry */
            const double Llj = l >= j ? M[j] * inv : M[j];
            M[j] = Llj;
            const double lo = l > j ? Llj : 0.0; /* finished rows take no part in the trailing update */
            W16_UNROLL for (int c = j + 1; c < n; c++) M[c] -= lo * w16_bcast(lo, c, xb);
            m = l == j ? lj : m - lo * lj;
        }

        /* ---- outputs ---- */
        if (mine)
        {
            W16_UNROLL for (int c = 0; c < n; c++)
                if (c <= l) WAT(D.Lf, k * NP + PK(l, c)) = M[c];
            WAT(D.lf, k * n + l) = m;
        }
        /* x-block for the next (earlier) stage: rows stay in registers, columns go through the LDS tile */
        GQP_ROWSYNC();
        if (isx)
        {
            W16_UNROLL for (int c = 0; c < NX; c++)
            {
                Lp[c] = c <= cx ? M[NU + c] : 0.0;
                TA[cx * LDX + c] = Lp[c];
            }
        }
        GQP_ROWSYNC();
        if (isx)
        {
            W16_UNROLL for (int q = 0; q < NX; q++) LpT[q] = TA[q * LDX + cx];
        }
        lxn = isx ? m : 0.0;
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

template <int NX, int NU, bool SOFT>
__device__ static inline void kx_backrhs_body(const GqpDev &D, const GqpOpts &O, int redo)
{
    GQP_DYN_SHARED(smem);
    typedef W16Lds<NX, NU> LY;
    constexpr int n = NX + NU, NP = n * (n + 1) / 2, LDX = LY::LDX;
    const int l = threadIdx.x & 15, inst = w16_slot_inst(D, blockIdx.x * 4 + (threadIdx.x >> 4));
    if (inst < 0) return;
    if (D.status[inst] != GQP_RUNNING) return;
    if (redo == 1 && !(D.alpha[inst] < 0.0)) return; /* redo = 2: sensitivity pass (direction only, every instance) */
    double *T = smem + (threadIdx.x >> 4) * LY::SZ, *xb = T + LY::XB, *TA = T + LY::TA;
    const bool mine = l < n, isx = l >= NU && l < n;
    const int cx = isx ? l - NU : 0;
    const double smu = D.smu[inst];
    const double pscale = redo == 1 ? 0.0 : 1.0;
    double Lp[NX], LpT[NX];
    W16_UNROLL for (int c = 0; c < NX; c++) { Lp[c] = 0.0; LpT[c] = 0.0; }
    double pn = 0.0; /* p of the stage handled before (state lanes) */

    for (int k = D.N; k >= 0; k--)
    {
        GQP_STAGE_REF S = D.st[k];
        const uint64_t imask = S.bmask & ~S.emask;
        const uint64_t am = WAT(D.amask, k * D.AW);
        const int nbg = S.nb;
        const bool fixed = mine && ((S.emask >> l) & 1);
        /* row l of the factor: the first NU columns (Lr, Ls) and, for state lanes, the x-block */
        /* branch-free loads: clamped indices, idle lanes zero the value */
        const int lc_ = mine ? l : 0, xl_ = isx ? l : NU, xc_ = isx ? cx : 0;
        const double zm = mine ? 1.0 : 0.0, zx = isx ? 1.0 : 0.0;
        double Lu[NU > 0 ? NU : 1], Lx[NX], Br[NX];
        W16_UNROLL for (int c = 0; c < NU; c++) Lu[c] = (c <= lc_ ? zm : 0.0) * WAT(D.Lf, k * NP + PK(lc_, (c <= lc_ ? c : 0)));
        W16_UNROLL for (int c = 0; c < NX; c++) Lx[c] = (c <= xc_ ? zx : 0.0) * WAT(D.Lf, k * NP + PK(xl_, NU + (c <= xc_ ? c : 0)));
        W16_UNROLL for (int c = 0; c < NX; c++) Br[c] = zm * WAT(D.BAt, (k * n + lc_) * NX + c);
        const double rb = zx * WAT(D.rb, k * NX + xc_);
        double m = zm * WAT(D.rg, k * n + lc_);
        const bool has = mine && ((imask >> l) & 1);
        /* SOFT: the loads of the row's slack, issued with the stage's other loads */
        const int ibs = (SOFT && has) ? popc64(S.bmask & (((uint64_t) 1 << l) - 1)) : 0;
        const int sj = (SOFT && has) ? w16_srev(S, ibs) : -1;
        const int sq = sj >= 0 ? sj : 0, e0 = S.o_ct + 2 * nbg + sq, e1 = e0 + S.ns;
        const bool sal = sj >= 0 && ((am >> (2 * nbg + sq)) & 1), sau = sj >= 0 && ((am >> (2 * nbg + S.ns + sq)) & 1);
        double sll = 0.0, slu = 0.0, stl = 1.0, stu = 1.0, srdl = 0.0, srdu = 0.0, spl = 0.0, spu = 0.0, sgl = 0.0, sgu = 0.0,
               sZl = 0.0, sZu = 0.0, sDl = 0.0, sDu = 0.0;
        if (SOFT && sj >= 0)
        {
            sll = sal ? WAT(D.lam, e0) : 0.0; slu = sau ? WAT(D.lam, e1) : 0.0;
            stl = sal ? WAT(D.t, e0) : 1.0; stu = sau ? WAT(D.t, e1) : 1.0;
            srdl = sal ? WAT(D.rd, e0) : 0.0; srdu = sau ? WAT(D.rd, e1) : 0.0;
            spl = sal ? WAT(D.pcorr, e0) : 0.0; spu = sau ? WAT(D.pcorr, e1) : 0.0;
            sgl = WAT(D.rgs, S.o_s + sq); sgu = WAT(D.rgs, S.o_s + S.ns + sq);
            sZl = WAT(D.Zz, (S.o_s + sq) * 2); sZu = WAT(D.Zz, (S.o_s + S.ns + sq) * 2);
            sDl = WAT(D.sD, S.o_s + sq); sDu = WAT(D.sD, S.o_s + S.ns + sq);
        }
        if (has)
        {
            const int ib = popc64(S.bmask & (((uint64_t) 1 << l) - 1));
            const bool al = (am >> ib) & 1, au = (am >> (nbg + ib)) & 1;
            const int el = S.o_ct + ib, eu = el + nbg;
            const double ll = al ? WAT(D.lam, el) : 0.0, lu = au ? WAT(D.lam, eu) : 0.0;
            const double ttl = al ? WAT(D.t, el) : 1.0, ttu = au ? WAT(D.t, eu) : 1.0;
            const double rdl = al ? WAT(D.rd, el) : 0.0, rdu = au ? WAT(D.rd, eu) : 0.0;
            const double rml = al ? ll * ttl - O.tau_min + pscale * WAT(D.pcorr, el) - smu : 0.0;
            const double rmu = au ? lu * ttu - O.tau_min + pscale * WAT(D.pcorr, eu) - smu : 0.0;
            const double itl = frcp(ttl), itu = frcp(ttu);
            const double bRl = (rml + ll * rdl) * itl, bRu = (rmu + lu * rdu) * itu;
            if (!SOFT || sj < 0) m += bRl - bRu;
            else
            {
                const double bGl = ll * itl, bGu = lu * itu;
                const double srml = sal ? sll * stl - O.tau_min + pscale * spl - smu : 0.0;
                const double srmu = sau ? slu * stu - O.tau_min + pscale * spu - smu : 0.0;
                const double sitl = frcp(stl), situ = frcp(stu);
                const double Xl = sgl + (srml + sll * srdl) * sitl, Xu = sgu + (srmu + slu * srdu) * situ; /* r~ without the row */
                const double El = sZl + sll * sitl, Eu = sZu + slu * situ;
                WAT(D.sR, S.o_s + sj) = Xl + bRl; WAT(D.sR, S.o_s + S.ns + sj) = Xu + bRu;
                const double Il = sDl != 0.0 ? frcp(sDl) : 0.0, Iu = sDu != 0.0 ? frcp(sDu) : 0.0;
                m += (bRl * El - bGl * Xl) * Il - (bRu * Eu - bGu * Xu) * Iu;
            }
        }
        /* y = Lx+ (Lx+' rb) + p+ */
        double w0 = 0.0;
        W16_UNROLL for (int q = 0; q < NX; q++) w0 += (q >= cx ? LpT[q] : 0.0) * w16_bcast(rb, NU + q, xb);
        if (!isx) w0 = 0.0;
        double y = pn;
        W16_UNROLL for (int c = 0; c < NX; c++) y += Lp[c] * w16_bcast(w0, NU + c, xb); /* Lp[c] = 0 above the diagonal */
        if (!isx) y = 0.0;
        double a = 0.0;
        W16_UNROLL for (int c = 0; c < NX; c++) a += Br[c] * w16_bcast(y, NU + c, xb);
        m = (fixed || !mine) ? 0.0 : m + a;
        /* l_u = Lr^{-1} m_u; the rows below keep m_r -= L[r][j] l_j, which leaves p in the state lanes */
        W16_UNROLL for (int j = 0; j < NU; j++)
        {
            const double d = w16_bcast(Lu[j], j, xb);
            const double lj = d != 0.0 ? w16_bcast(m, j, xb) * frcp(d) : 0.0;
            m = l == j ? lj : (l > j ? m - Lu[j] * lj : m);
        }
        if (mine) WAT(D.lf, k * n + l) = m;
        /* this stage's x-block becomes "the stage handled before" */
        GQP_ROWSYNC();
        if (isx)
        {
            W16_UNROLL for (int c = 0; c < NX; c++) { Lp[c] = Lx[c]; TA[cx * LDX + c] = Lx[c]; }
        }
        GQP_ROWSYNC();
        if (isx)
        {
            W16_UNROLL for (int q = 0; q < NX; q++) LpT[q] = TA[q * LDX + cx];
        }
        pn = isx ? m : 0.0;
    }
}

/* --------------------------------------------------------------------------------------------------- forward */

/* PFORM (= CORR): lf holds [l_u; p] (written by kx_backrhs), otherwise the plain l of the factor sweep */
template <int NX, int NU, bool CORR, bool SOFT>
__device__ static inline void kx_fwd_body(const GqpDev &D, const GqpOpts &O, int redo)
{
    GQP_DYN_SHARED(smem);
    typedef W16Lds<NX, NU> LY;
    constexpr bool PFORM = CORR;
    constexpr int n = NX + NU, NP = n * (n + 1) / 2, LDF = LY::LDF;
    const int l = threadIdx.x & 15, inst = w16_slot_inst(D, blockIdx.x * 4 + (threadIdx.x >> 4));
    if (inst < 0) return;
    if (D.status[inst] != GQP_RUNNING) return;
    if (redo == 1 && !(D.alpha[inst] < 0.0)) return; /* redo = 2: sensitivity pass (direction only, every instance) */
    double *T = smem + (threadIdx.x >> 4) * LY::SZ, *xb = T + LY::XB, *TF = T + LY::TF;
    const bool mine = l < n, isx = l >= NU && l < n;
    const int cx = isx ? l - NU : 0;
    const double smu = CORR ? D.smu[inst] : 0.0;
    const double pscale = (CORR && redo != 1) ? 1.0 : 0.0;
    double alpha = 1.0, S0 = 0.0, S1 = 0.0, S2 = 0.0, nact = 0.0;
    double dx = 0.0; /* dx of this lane's state for the stage being entered */

    for (int k = 0; k <= D.N; k++)
    {
        GQP_STAGE_REF S = D.st[k];
        const uint64_t imask = S.bmask & ~S.emask;
        const uint64_t am = WAT(D.amask, k * D.AW);
        const int nbg = S.nb;
        /* row l of the factor (registers) and, through the LDS tile, column l */
        const int lc_ = mine ? l : 0, xc_ = isx ? cx : 0;
        const double zm = mine ? 1.0 : 0.0, zx = isx ? 1.0 : 0.0;
        double Lr[n], Lc[n], Bc[n];
        W16_UNROLL for (int c = 0; c < n; c++) Lr[c] = (c <= lc_ ? zm : 0.0) * WAT(D.Lf, k * NP + PK(lc_, (c <= lc_ ? c : 0)));
        W16_UNROLL for (int r = 0; r < n; r++) Bc[r] = zx * WAT(D.BAt, (k * n + r) * NX + xc_);
        double lv = zm * WAT(D.lf, k * n + lc_);
        const double rbv = zx * WAT(D.rb, k * NX + xc_);
        const bool has = mine && ((imask >> l) & 1);
        const int ib = has ? popc64(S.bmask & (((uint64_t) 1 << l) - 1)) : 0;
        const bool al = has && ((am >> ib) & 1), au = has && ((am >> (nbg + ib)) & 1);
        const int el = S.o_ct + ib, eu = el + nbg;
        const double ll = al ? WAT(D.lam, el) : 0.0, lu = au ? WAT(D.lam, eu) : 0.0;
        const double ttl = al ? WAT(D.t, el) : 1.0, ttu = au ? WAT(D.t, eu) : 1.0;
        const double rdl = al ? WAT(D.rd, el) : 0.0, rdu = au ? WAT(D.rd, eu) : 0.0;
        const double pl = (CORR && al) ? WAT(D.pcorr, el) : 0.0, pu = (CORR && au) ? WAT(D.pcorr, eu) : 0.0;
        /* SOFT: everything the slack of this lane's row needs, issued with the other loads of the stage (the
         * substitution below hides their latency) */
        const int sj = (SOFT && has) ? w16_srev(S, ib) : -1;
        const int sq = sj >= 0 ? sj : 0, e0 = S.o_ct + 2 * nbg + sq, e1 = e0 + S.ns;
        const bool sal = sj >= 0 && ((am >> (2 * nbg + sq)) & 1), sau = sj >= 0 && ((am >> (2 * nbg + S.ns + sq)) & 1);
        double sll = 0.0, slu = 0.0, stl = 1.0, stu = 1.0, sDl = 0.0, sDu = 0.0, rsl = 0.0, rsu = 0.0, sZl = 0.0, sZu = 0.0,
               spl = 0.0, spu = 0.0, srdl = 0.0, srdu = 0.0;
        if (SOFT && sj >= 0)
        {
            sll = sal ? WAT(D.lam, e0) : 0.0; slu = sau ? WAT(D.lam, e1) : 0.0;
            stl = sal ? WAT(D.t, e0) : 1.0; stu = sau ? WAT(D.t, e1) : 1.0;
            sDl = WAT(D.sD, S.o_s + sq); sDu = WAT(D.sD, S.o_s + S.ns + sq);
            rsl = WAT(D.sR, S.o_s + sq); rsu = WAT(D.sR, S.o_s + S.ns + sq);
            sZl = WAT(D.Zz, (S.o_s + sq) * 2); sZu = WAT(D.Zz, (S.o_s + S.ns + sq) * 2);
            spl = (CORR && sal) ? WAT(D.pcorr, e0) : 0.0; spu = (CORR && sau) ? WAT(D.pcorr, e1) : 0.0;
            srdl = sal ? WAT(D.rd, e0) : 0.0; srdu = sau ? WAT(D.rd, e1) : 0.0;
        }
        GQP_ROWSYNC();
        if (mine)
        {
            W16_UNROLL for (int c = 0; c < n; c++) TF[l * LDF + c] = Lr[c];
        }
        GQP_ROWSYNC();
        W16_UNROLL for (int r = 0; r < n; r++) Lc[r] = (mine && r >= l) ? TF[r * LDF + l] : 0.0; /* L[r][l] */

        if (PFORM && k == 0)
        {
            /* the states of stage 0 are free: recover l_x = Lx^{-1} p */
            W16_UNROLL for (int j = NU; j < n; j++)
            {
                const double d = w16_bcast(Lr[j], j, xb);
                const double lj = d != 0.0 ? w16_bcast(lv, j, xb) * frcp(d) : 0.0;
                lv = l == j ? lj : (l > j ? lv - Lr[j] * lj : lv);
            }
        }
        /* dpi_k = Lx (Lx' dx) + p  (CORR only; k > 0) */
        if (CORR && k > 0)
        {
            double w0 = 0.0; /* (Lx' dx)[cx] = sum_{q >= cx} L[NU+q][NU+cx] dx[q]: column l of L */
            W16_UNROLL for (int q = 0; q < NX; q++) w0 += Lc[NU + q] * w16_bcast(dx, NU + q, xb);
            if (!isx) w0 = 0.0;
            double a = lv;
            W16_UNROLL for (int c = 0; c < NX; c++) a += Lr[NU + c] * w16_bcast(w0, NU + c, xb);
            if (isx) WAT(D.dpi, k * NX + cx) = a;
        }
        /* L' dv = -l for the free block: everything at k = 0, the inputs otherwise; dv of the states = dx for k > 0 */
        double dv = (k > 0 && isx) ? dx : 0.0;
        double acc = -lv;
        if (k > 0)
        {
            W16_UNROLL for (int p = NU; p < n; p++) acc -= Lc[p] * w16_bcast(dv, p, xb);
        }
        W16_UNROLL for (int r = n - 1; r >= 0; r--)
        {
            if (k > 0 && r >= NU) continue; /* uniform: states are given */
            const double d = w16_bcast(Lr[r], r, xb);
            const double dvr = d != 0.0 ? w16_bcast(acc, r, xb) * frcp(d) : 0.0;
            if (l == r) dv = dvr;
            else if (l < r) acc -= Lc[r] * dvr;
        }
        if (!mine) dv = 0.0;
        if (CORR && mine) WAT(D.dux, k * n + l) = dv;
        /* dx of the next stage */
        double dxn = rbv;
        W16_UNROLL for (int r = 0; r < n; r++) dxn += Bc[r] * w16_bcast(dv, r, xb);
        if (has)
        {
            const double rml = al ? ll * ttl - O.tau_min + pscale * pl - smu : 0.0;
            const double rmu = au ? lu * ttu - O.tau_min + pscale * pu - smu : 0.0;
            double dcl = dv, dcu = -dv; /* dc + ds resp. -dc + ds: the row's own slack step included (SOFT) */
            if (SOFT && sj >= 0)
            {
                const double il = sDl != 0.0 ? frcp(sDl) : 0.0, iu = sDu != 0.0 ? frcp(sDu) : 0.0;
                const double gl_ = ll * frcp(ttl), gu_ = lu * frcp(ttu);
                const double dsl = (-rsl - gl_ * dv) * il, dsu = (-rsu + gu_ * dv) * iu;
                if (CORR) { WAT(D.dsv, S.o_s + sj) = dsl; WAT(D.dsv, S.o_s + S.ns + sj) = dsu; }
                const double sitl = frcp(stl), situ = frcp(stu);
                const double El = sZl + sll * sitl, Eu = sZu + slu * situ;
                dcl = (El * dv - rsl) * il;
                dcu = (-Eu * dv - rsu) * iu;
                /* the two bound rows of the slack */
                const double srml = sal ? sll * stl - O.tau_min + pscale * spl - smu : 0.0;
                const double srmu = sau ? slu * stu - O.tau_min + pscale * spu - smu : 0.0;
                const double sdtl = sal ? dsl + srdl : 0.0, sdtu = sau ? dsu + srdu : 0.0;
                const double sdll = sal ? -(srml + sll * sdtl) * sitl : 0.0, sdlu = sau ? -(srmu + slu * sdtu) * situ : 0.0;
                const double q1 = -sll * frcp(sdll), q2 = -slu * frcp(sdlu), q3 = -stl * frcp(sdtl), q4 = -stu * frcp(sdtu);
                alpha = (sdll < 0.0 && q1 < alpha) ? q1 : alpha;
                alpha = (sdlu < 0.0 && q2 < alpha) ? q2 : alpha;
                alpha = (sdtl < 0.0 && q3 < alpha) ? q3 : alpha;
                alpha = (sdtu < 0.0 && q4 < alpha) ? q4 : alpha;
                /* (the sums in the corrector sweep too: the conditional corrector asks for the duality measure its step ends at) */
                S0 += sll * stl + slu * stu;
                S1 += sll * sdtl + stl * sdll + slu * sdtu + stu * sdlu;
                S2 += sdll * sdtl + sdlu * sdtu;
                nact += (double) ((int) sal + (int) sau);
                if (!CORR)
                {
                    WAT(D.pcorr, e0) = sdll * sdtl;
                    WAT(D.pcorr, e1) = sdlu * sdtu;
                }
                else
                {
                    WAT(D.dlam, e0) = sdll; WAT(D.dlam, e1) = sdlu;
                    WAT(D.dt, e0) = sdtl; WAT(D.dt, e1) = sdtu;
                }
            }
            const double dtl = al ? dcl + rdl : 0.0, dtu = au ? dcu + rdu : 0.0;
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
            nact += (double) ((int) al + (int) au);
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
        dx = isx ? dxn : 0.0;
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
    if (O.cond_pred_corr && !redo)
    {
        /* conditional corrector: a step that would more than double the duality measure is taken again with the centering term
         * alone (redo pair of the host loop) */
        S0 = w16_rsum(S0, xb); S1 = w16_rsum(S1, xb); S2 = w16_rsum(S2, xb);
        const double nact_d = w16_rsum(nact, xb);
        if (nact_d > 0.0 && (S0 + alpha * S1 + alpha * alpha * S2) / nact_d > 2.0 * D.mu[inst])
        {
            if (l == 0) D.alpha[inst] = -alpha_aff;
            return;
        }
    }
    const double a = D.mu[inst] > 0.0 ? gqp_step_scale(alpha) : 1.0;
    if (O.ext_update)
    {
        /* launch-per-sweep loop: the step is applied by k_step_update (ipm_kernels.hpp), every element its own work item
         * (kx_solve, the whole solve in one launch, keeps the pass below) */
        if (l == 0)
        {
            D.apend[inst] = a;
            D.alpha[inst] = alpha;
            D.iter[inst] = it + 1;
            if (st) { st[4 * D.stat_inst] = alpha; st[5 * D.stat_inst] = alpha; }
        }
        return;
    }
    /* update: one lane per variable / state / box row of the stage (dux, dpi, dlam, dt were written by these very
     * lanes).  Everything the lane updates in CH consecutive stages -- variable, state multiplier, the two sides of its box
     * row, with SOFT the row's slack values and slack sides -- is loaded through clamped addresses before the first store
     * of the chunk: one memory round trip per chunk (three loops with four stages in flight each made 3 CH / 4 of them;
     * ipm_kernels_w16r.hpp has the measurement). */
    {
        constexpr int CH = SOFT ? 4 : 6;
        const int lc_ = mine ? l : 0, xc_ = isx ? cx : 0;
        const GqpStagePtr st_ = D.st;
        const uint64_t *__restrict__ am_ = D.amask.p + (size_t) inst * D.amask.E;
        for (int k0 = 0; k0 <= D.N; k0 += CH)
        {
            double u0[CH], du[CH], p0[CH], dp[CH], lm[CH][4], dl[CH][4], tt[CH][4], dt[CH][4], sv0[CH], sd0[CH], sv1[CH], sd1[CH];
            int e[CH][4], q0[CH], q1[CH];
            bool act[CH][4], hs[CH];
            W16_UNROLL for (int c = 0; c < CH; c++)
            {
                const int k = k0 + c <= D.N ? k0 + c : D.N; /* beyond the horizon: the last stage once more, nothing stored */
                const uint64_t bm = st_[k].bmask, imask = bm & ~st_[k].emask;
                const uint64_t am = am_[k * D.AW];
                const int nbg = st_[k].nb, o_ct = st_[k].o_ct;
                u0[c] = WAT(D.ux, k * n + lc_); du[c] = WAT(D.dux, k * n + lc_);
                p0[c] = WAT(D.pi, k * NX + xc_); dp[c] = WAT(D.dpi, k * NX + xc_);
                const bool has = mine && ((imask >> l) & 1);
                const int ib = has ? popc64(bm & (((uint64_t) 1 << l) - 1)) : 0;
                e[c][0] = o_ct + ib; e[c][1] = e[c][0] + nbg;
                act[c][0] = has && ((am >> ib) & 1); act[c][1] = has && ((am >> (nbg + ib)) & 1);
                hs[c] = false;
                if (SOFT)
                {
                    const int sj = has ? w16_srev(st_[k], ib) : -1;
                    const int ns = st_[k].ns, o_s = st_[k].o_s, sq = sj >= 0 ? sj : 0;
                    hs[c] = sj >= 0;
                    e[c][2] = o_ct + 2 * nbg + sq; e[c][3] = e[c][2] + ns;
                    act[c][2] = hs[c] && ((am >> (2 * nbg + sq)) & 1); act[c][3] = hs[c] && ((am >> (2 * nbg + ns + sq)) & 1);
                    q0[c] = o_s + sq; q1[c] = o_s + ns + sq;
                    const int s0 = q0[c] < D.sv.E ? q0[c] : 0, s1 = q1[c] < D.sv.E ? q1[c] : 0;
                    sv0[c] = WAT(D.sv, s0); sd0[c] = WAT(D.dsv, s0); sv1[c] = WAT(D.sv, s1); sd1[c] = WAT(D.dsv, s1);
                }
                W16_UNROLL for (int w = 0; w < (SOFT ? 4 : 2); w++)
                {
                    const int ec = e[c][w] < D.lam.E ? e[c][w] : 0;
                    lm[c][w] = WAT(D.lam, ec); dl[c][w] = WAT(D.dlam, ec); tt[c][w] = WAT(D.t, ec); dt[c][w] = WAT(D.dt, ec);
                }
            }
            W16_UNROLL for (int c = 0; c < CH; c++)
            {
                const int k = k0 + c;
                if (k > D.N) break;
                if (mine) WAT(D.ux, k * n + l) = u0[c] + a * du[c];
                if (isx && k >= 1) WAT(D.pi, k * NX + cx) = p0[c] + a * dp[c];
                W16_UNROLL for (int w = 0; w < (SOFT ? 4 : 2); w++)
                {
                    const double l2 = lm[c][w] + a * dl[c][w], t2 = tt[c][w] + a * dt[c][w];
                    if (act[c][w])
                    {
                        WAT(D.lam, e[c][w]) = l2 < O.lam_min ? O.lam_min : l2;
                        WAT(D.t, e[c][w]) = t2 < O.t_min ? O.t_min : t2;
                    }
                }
                if (SOFT && hs[c])
                {
                    WAT(D.sv, q0[c]) = sv0[c] + a * sd0[c];
                    WAT(D.sv, q1[c]) = sv1[c] + a * sd1[c];
                }
            }
        }
    }
    if (l == 0)
    {
        D.alpha[inst] = alpha;
        D.iter[inst] = it + 1;
        if (st) { st[4 * D.stat_inst] = alpha; st[5 * D.stat_inst] = alpha; }
    }
}

/* ------------------------------------------------------------------------------------------------ kernels */

template <int NX, int NU, bool SOFT = false>
__global__ void __launch_bounds__(64) kx_factor(GqpDev D, GqpOpts O, int redo) { kx_factor_body<NX, NU, SOFT>(D, O, redo); }
template <int NX, int NU, bool SOFT = false>
__global__ void __launch_bounds__(64) kx_backrhs(GqpDev D, GqpOpts O, int redo) { kx_backrhs_body<NX, NU, SOFT>(D, O, redo); }
template <int NX, int NU, bool CORR, bool SOFT = false>
__global__ void __launch_bounds__(64) kx_fwd(GqpDev D, GqpOpts O, int redo) { kx_fwd_body<NX, NU, CORR, SOFT>(D, O, redo); }

/* Between two sweeps of the whole-solve kernel: what the row's lanes stored (iterate, factor, the per-instance scalars lane 0
 * writes) is read by the other lanes of the row in the next sweep.  Same wave, same CU: completion of the stores is all it
 * takes (workgroup scope). */
#if defined(__HIP_DEVICE_COMPILE__)
#define W16_SWEEP_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup")
#else
#define W16_SWEEP_FENCE() GQP_ROWSYNC()
#endif

/* The whole solve in ONE launch (small batches: a QP of an acados control loop, a handful of them).  The rows of this
 * family are independent of each other -- no sweep needs anything from another instance -- so every 16-lane row runs the
 * host loop of run_ipm by itself: factor sweep (which also decides whether the instance is done), affine sweep,
 * corrector right-hand side, corrector sweep, and the conditional redo pair, until its instance leaves the RUNNING state
 * (converged, iteration limit, minimum step, NaN: all decided in the factor sweep, as in the launch-per-sweep loop).
 * Per-instance arithmetic is that of the separate kernels, bit for bit; what disappears is 4-6 launches, a device-to-host
 * copy and a stream synchronisation per IPM iteration -- for one QP most of the time of a solve.  Rows of a wave that
 * finish early idle until the wave's last row is done, which is why large batches stay on the launch-per-sweep loop
 * (dense list of the live instances, gpu_batch.hip). */
template <int NX, int NU, bool SOFT = false>
__global__ void __launch_bounds__(64) kx_solve(GqpDev D, GqpOpts O, int)
{
    const int inst = w16_slot_inst(D, blockIdx.x * 4 + (threadIdx.x >> 4));
    if (inst < 0) return;
    for (;;)
    {
        kx_factor_body<NX, NU, SOFT>(D, O, 0);
        W16_SWEEP_FENCE();
        if (D.status[inst] != GQP_RUNNING) break;
        kx_fwd_body<NX, NU, false, SOFT>(D, O, 0);
        W16_SWEEP_FENCE();
        for (int redo = 0; redo <= (O.cond_pred_corr ? 1 : 0); redo++)
        {
            kx_backrhs_body<NX, NU, SOFT>(D, O, redo);
            W16_SWEEP_FENCE();
            kx_fwd_body<NX, NU, true, SOFT>(D, O, redo);
            W16_SWEEP_FENCE();
        }
    }
}

} // namespace gqp

#endif
