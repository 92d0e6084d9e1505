/*
 * ipm_kernels_box.hpp -- fast-path IPM kernels for QPs whose inequalities are box constraints
 * only (ng = ns = 0 on every stage): configurations C1, C2, C5 and the reference's golden QPs.
 *
 * Same algorithm, same HBM layout and the same one-instance-per-lane mapping as the general
 * kernels in ipm_kernels.hpp; what changes is the shape of the code, dictated by what rocprofv3
 * and the ISA showed for the general kernels (profiles/r01_v0_*): ~200 serialised
 * `s_waitcnt vmcnt(0)` round trips per stage because per-variable uniform branches cut the stage
 * body into >100 basic blocks.  Here the stage body is STRAIGHT-LINE:
 *   - no k<N / k>0 branches: zero-filled extra slots (gpu_ipm_internal.h "slot conventions");
 *   - no per-row branches around loads: a row that does not exist is read through a clamped
 *     dummy index and masked arithmetically, so every load of a stage can be issued before the
 *     first use (one or two memory round trips per stage instead of ~200);
 *   - W = [B A]'Lx+ is formed in place over the BAt registers (peak ~180 live doubles);
 *   - no IEEE division/sqrt sequences: v_rcp_f64 / v_rsq_f64 + two Newton steps;
 *   - fewer bytes: rm is recomputed (lam*t), the affine sweep stores only dlam*dt (what the
 *     Mehrotra corrector needs) and neither dux nor dpi; mu_aff comes from three running sums
 *     instead of a second pass over lam,t,dlam,dt.
 * Equality-flagged rows (idxe) are not IPM rows here; their multipliers are recovered from
 * stationarity in kb_finalize.
 */
#ifndef IPM_KERNELS_BOX_HPP_
#define IPM_KERNELS_BOX_HPP_

#include "ipm_kernels.hpp"

namespace gqp
{

__device__ static inline double frcp(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
#else
    return 1.0 / x;
#endif
}

__device__ static inline double frsqrt(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    double e = __builtin_fma(-h * y, y, 0.5);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-h * y, y, 0.5);
    return __builtin_fma(y, e, y);
#else
    return 1.0 / sqrt(x);
#endif
}

__device__ static inline int popc64(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}

/* compiler-level fence: memory operations are not moved across it */

/* make wave-uniform values PROVABLY uniform for the compiler (scalar registers): the stage
 * structure is fetched with vector loads, and an SRD / soffset that is not provably uniform gets
 * every buffer op wrapped in a waterfall loop (cdna_hip_programming.md T20) */
__device__ static inline int uni(int v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readfirstlane(v);
#else
    return v;
#endif
}
__device__ static inline uint64_t uni64(uint64_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned int lo = __builtin_amdgcn_readfirstlane((int) (unsigned int) v);
    const unsigned int hi = __builtin_amdgcn_readfirstlane((int) (unsigned int) (v >> 32));
    return ((uint64_t) hi << 32) | lo;
#else
    return v;
#endif
}

/* stage structure in scalar registers */
struct StageU
{
    int nb, o_ct;
    uint64_t bmask, emask;
};
__device__ static inline StageU stage_u(GqpStagePtr st, int k)
{
    StageU S;
    S.nb = uni(st[k].nb);
    S.o_ct = uni(st[k].o_ct);
    S.bmask = uni64(st[k].bmask);
    S.emask = uni64(st[k].emask);
    return S;
}

/* ---------------------------------------------------------------------------------------
 * Acc: one HBM array rebased to (stage) element offset `e0`; element e of this lane is
 * base[(e0 + e) * Bp + i].  On the device it is a buffer resource (128-bit SRD in SGPRs):
 *     buffer_load_dwordx2 v, v_laneoff, s[srd], s_elemoff offen
 * i.e. ONE 32-bit VGPR (lane byte offset) serves every access and the element offset is scalar.
 * With plain pointers hipcc keeps a 64-bit VGPR address per access (360 v_lshl_add_u64 and as
 * many VGPR pairs per stage in the factor kernel), which is what pushed it into scratch.
 * The SRD points at this wave's tile of the array; element offsets are e * 512 B.
 * --------------------------------------------------------------------------------------- */
struct Acc
{
#if defined(__HIP_DEVICE_COMPILE__)
    __amdgpu_buffer_rsrc_t rs;
    unsigned int bp8, voff;
    typedef decltype(__builtin_amdgcn_raw_buffer_load_b64(rs, 0u, 0u, 0)) raw_t;
    __device__ inline double ld(int e) const
    {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, (unsigned int) e * bp8, 0));
    }
    /* load whose lane offset carries the (always zero) ordering token, see kb_factor */
    __device__ inline double ldo(int e, int ord) const
    {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, voff + (unsigned int) ord, (unsigned int) e * bp8, 0));
    }
    __device__ inline void st(int e, double v) const
    {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(raw_t, v), rs, voff, (unsigned int) e * bp8, 0);
    }
    /* element e0 + j, j a compile-time constant after unrolling: the low three bits of j travel in the instruction's
     * 12-bit immediate offset (8 elements x 512 B), so eight accesses share one scalar offset register instead of
     * computing one each */
    __device__ inline double ldj(int e0, int j) const
    {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, voff + (unsigned int) (j & 7) * 512u,
                                                                                  (unsigned int) (e0 + (j & ~7)) * bp8, 0));
    }
    __device__ inline void stj(int e0, int j, double v) const
    {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(raw_t, v), rs, voff + (unsigned int) (j & 7) * 512u,
                                              (unsigned int) (e0 + (j & ~7)) * bp8, 0);
    }
#else
    double *p;
    size_t bp;
    double ld(int e) const { return p[(size_t) e * bp]; }
    double ldo(int e, int) const { return p[(size_t) e * bp]; }
    void st(int e, double v) const { p[(size_t) e * bp] = v; }
    double ldj(int e0, int j) const { return p[(size_t) (e0 + j) * bp]; }
    void stj(int e0, int j, double v) const { p[(size_t) (e0 + j) * bp] = v; }
#endif
};

__device__ static inline Acc acc_at(GArr arr, size_t e0, int i)
{
    Acc a;
    /* this wave's tile of the array: blocks are single waves of 64 lanes, so the tile index is
     * blockIdx.x -- wave-uniform and provably so (a value derived from threadIdx would not be) */
    double *base = arr.p + ((size_t) blockIdx.x * (size_t) arr.E + e0) * 64;
#if defined(__HIP_DEVICE_COMPILE__)
    a.rs = __builtin_amdgcn_make_buffer_rsrc((void *) base, 0, 0xFFFFFFFFu, 0x00020000);
    a.bp8 = 512u; /* 64 lanes x 8 bytes per element */
    a.voff = (unsigned int) (i & 63) * 8u;
#else
    a.p = base + (i & 63);
    a.bp = 64;
#endif
    return a;
}
#define ACC(arr, e0) acc_at((arr), (size_t) (e0), i)
#define GATL_LD(arr, e) GATL(arr, e)

/* Scheduling fence: hipcc's machine scheduler hoists every load of a straight-line stage body
 * to its top (it only watches the 512-register ceiling), and the allocator then spills the
 * loaded values straight to scratch.  A sched_barrier between the load phases bounds how many
 * loaded blocks are live at once.  (Unlike an asm-based dependency it keeps the SRDs provably
 * wave-uniform -- no waterfall loops, cdna_hip_programming.md T20.) */
#if defined(__HIP_DEVICE_COMPILE__)
#define GQP_PHASE() __builtin_amdgcn_sched_barrier(0)
#else
#define GQP_PHASE() do { } while (0)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define GQP_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define GQP_OPAQUE(x) do { } while (0)
#endif

/* x (an int living in a VGPR, always 0 or a lane index) becomes data-dependent on val */
#if defined(__HIP_DEVICE_COMPILE__)
#define GQP_AFTER(x, val) asm volatile("" : "+v"(x) : "v"(val))
#else
#define GQP_AFTER(x, val) do { } while (0)
#endif

/* Lanes whose instance has finished ride along ("ghost" lanes) as long as one lane of their wave is still iterating: they
 * run the sweep on their unchanged iterate and store what is already there -- the factor sweep reproduces its own last
 * results bit for bit, the work arrays of the other sweeps are scratch -- and skip every per-instance scalar.  A wave with
 * holes writes partial 128-byte lines, and those cost far more than the bytes they carry (measured on C2: the factor
 * sweep with 82 % of the lanes live took 2.60 ms against 1.73 ms with every lane live; the sweeps that store little were
 * 2-3 % slower).  What keeps the factor of a finished instance intact: statuses change in the factor sweep only and every
 * lane of a wave takes the ride-along decision on the same statuses at the start of a launch, so a wave that ran the
 * rhs-only sweep (which overwrites lf) also runs the next factor sweep (which restores it), and the loop always ends behind a
 * factor sweep.  The host build runs one lane at a time -- a per-tile decision there would see statuses change between the
 * lanes of one launch and break exactly that -- and treats every wave as live: the ghost path is what it tests. */
#if defined(__HIP_DEVICE_COMPILE__)
#define GQP_WAVE_ANY(x) (__builtin_amdgcn_ballot_w64(x) != 0)
#else
#define GQP_WAVE_ANY(x) true
#endif

/* 1: the corrector sweep leaves its step to the next factor sweep (see kb_factor); 0: it applies it in a pass of its own (the
 * cross-check: make variant TAG=nofold DEFS=-DGQP_KB_FOLD=0) */
#ifndef GQP_KB_FOLD
#define GQP_KB_FOLD 1
#endif

#define GQP_ROW_CHUNK 4  /* rows of [B A]' fetched per load phase in kb_factor */
#define GQP_HROW_CHUNK 3 /* Hessian rows fetched per load phase in kb_factor */

/* row bookkeeping of variable j: exists?, compact row index (clamped to 0 when absent) */
#define GQP_ROW(j, has, ib)                                                                    \
    const bool has = (imask >> (j)) & 1;                                                       \
    const int ib = has ? popc64(S.bmask & (((uint64_t) 1 << (j)) - 1)) : 0

/* --------------------------------------------------------------- factor */

template <int NX, int NU, bool XBOX>
__global__ void __launch_bounds__(64) kb_factor(GqpDev D, GqpOpts O, int redo)
{
    constexpr int n = NX + NU, NP = n * (n + 1) / 2, NPX = NX * (NX + 1) / 2, NB = XBOX ? n : NU;
    const int Bp = D.Bp;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.B) return;
    const bool run = D.status[i] == GQP_RUNNING;
    if (!GQP_WAVE_ANY(run)) return;

    /* State rows of W = [B A]'Lx+ are parked in LDS ([element][lane], conflict-free 8-byte
     * accesses); only the NU input rows stay in registers.  W is the one block that must
     * coexist with the Hessian, and keeping both in VGPRs pushed the kernel into scratch.
     * The read index goes through an opaque move so that hipcc does not forward the stored
     * values straight back into registers. */
    constexpr bool W_IN_LDS = NX * NX * 64 * 8 <= 40 * 1024; /* 4 single-wave blocks per CU must fit */
    __shared__ double Wl[W_IN_LDS ? NX * NX * 64 : 1];
    double Wx[W_IN_LDS ? 1 : NX * NX]; /* larger shapes: per-lane block (scratch, see kernel_sets.h) */
    const int lane_w = threadIdx.x;
    int lane_r = threadIdx.x;
    GQP_OPAQUE(lane_r);
    double Lx[NPX], lx[NX];
    UNROLL for (int e = 0; e < NPX; e++) Lx[e] = 0.0;
    UNROLL for (int c = 0; c < NX; c++) lx[c] = 0.0;
    double nrm_g = 0.0, nrm_b = 0.0, nrm_d = 0.0, nrm_m = 0.0, musum = 0.0, obj = 0.0;
    int nact = 0;
    /* x_{k+1} and pi_{k+1} (slot k + 1 of ux / pi) are what stage k + 1 read as its own state and multiplier: carried in
     * registers instead of being fetched again (16 of the 227 loads of a C2 stage); slot N + 1 is zero by convention */
    double xn[NX], pin[NX];
    UNROLL for (int c = 0; c < NX; c++) { xn[c] = 0.0; pin[c] = 0.0; }
#if GQP_KB_FOLD
    /* FOLDED UPDATE.  The corrector sweep (kb_forward<CORR>) leaves its step (dux, dpi, dlam, dt) and the step length
     * (D.apend) behind instead of applying them in a pass of its own: this sweep reads the iterate anyway -- it adds the step
     * as it loads a stage and stores the new iterate.  Saves one read of (ux, pi, lam, t) per iteration, 31 of the ~ 950 doubles
     * an iteration moves per stage.  apd = 0: nothing pending (first iteration, a finished lane riding along: the iterate is
     * stored back as it is -- selects, not arithmetic, so that it stays bit for bit). */
    const double apd = run ? D.apend[i] : 0.0;
    const bool pend = apd != 0.0;
#endif

    for (int k = D.N; k >= 0; k--)
    {
        const StageU S = stage_u(D.st, k);
        const uint64_t imask = S.bmask & ~S.emask;
        const uint64_t am = GAT(D.amask, k);
        const int nbg = S.nb;

        /* The stage body is a chain of load phases.  `ord` is an always-zero lane offset that is
         * "laundered" (GQP_AFTER) through a value computed at the end of the previous phase and
         * added to the addresses of the next phase's loads: the compiler therefore cannot hoist
         * those loads above that computation.  This bounds the number of loaded blocks that are
         * live at once -- without it hipcc hoists all ~230 loads of the stage to the top, runs
         * out of registers and then serialises the loads one by one (profiles/r01_*).  Unlike an
         * instruction fence it leaves the scheduler free inside a phase. */
        int ord = 0;

        /* ---------------- phase 0 loads: what the dynamics rows need ---------------- */
        double rb[NX], v[n], gt[n];
        UNROLL for (int c = 0; c < NX; c++) rb[c] = ACC(D.bvec, 0).ld(k * NX + c) - xn[c];
        UNROLL for (int j = 0; j < n; j++) v[j] = ACC(D.ux, 0).ld(k * n + j);
#if GQP_KB_FOLD
        {
            double dv[n];
            UNROLL for (int j = 0; j < n; j++) dv[j] = ACC(D.dux, 0).ld(k * n + j);
            UNROLL for (int j = 0; j < n; j++) v[j] = pend ? v[j] + apd * dv[j] : v[j];
            UNROLL for (int j = 0; j < n; j++) ACC(D.ux, 0).st(k * n + j, v[j]);
        }
#endif

        /* ---------------- dynamics, GQP_ROW_CHUNK rows of [B A]' per load phase ----------------
         * rb += row*v_r, gt_r = row.pi+, W_r = row * Lx+ (state rows of W go to LDS) */
        double Wu[NU * NX];
        const Acc aBAt = ACC(D.BAt, k * n * NX);
        UNROLL for (int r = 0; r < n; r++)
        {
            if (r > 0 && r % GQP_ROW_CHUNK == 0) GQP_AFTER(ord, rb[0]);
            double row[NX];
            UNROLL for (int c = 0; c < NX; c++) row[c] = aBAt.ldo(r * NX + c, ord);
            double a = 0.0;
            UNROLL for (int c = 0; c < NX; c++)
            {
                a += row[c] * pin[c];
                rb[c] += row[c] * v[r];
            }
            gt[r] = a; /* BAt pi+ ; gradient and H v are added below */
            UNROLL for (int c = 0; c < NX; c++)
            {
                double w = 0.0;
                UNROLL for (int q = c; q < NX; q++) w += row[q] * Lx[PK(q, c)];
                if (r < NU) Wu[(r < NU ? r : 0) * NX + c] = w;
                else if (W_IN_LDS) Wl[W_IN_LDS ? (((r < NU ? NU : r) - NU) * NX + c) * 64 + lane_w : 0] = w;
                else Wx[W_IN_LDS ? 0 : ((r < NU ? NU : r) - NU) * NX + c] = w;
            }
        }
        /* w0 = Lx+' rb + lx+ (needs the final rb) */
        double w0[NX];
        UNROLL for (int c = 0; c < NX; c++)
        {
            double a = lx[c];
            UNROLL for (int q = c; q < NX; q++) a += Lx[PK(q, c)] * rb[q];
            w0[c] = a;
        }

        /* ---------------- box rows (need v only) ---------------- */
        GQP_AFTER(ord, w0[0]);
        double g[n], pik[NX];
        UNROLL for (int j = 0; j < n; j++) g[j] = ACC(D.rq, 0).ldo(k * n + j, ord);
        UNROLL for (int c = 0; c < NX; c++) pik[c] = ACC(D.pi, 0).ldo(k * NX + c, ord);
#if GQP_KB_FOLD
        if (k > 0) /* (slot 0 of pi is zero by convention and takes no step; slot k is the multiplier of the dynamics producing x_k) */
        {
            double dp[NX];
            UNROLL for (int c = 0; c < NX; c++) dp[c] = ACC(D.dpi, 0).ldo(k * NX + c, ord);
            UNROLL for (int c = 0; c < NX; c++) pik[c] = pend ? pik[c] + apd * dp[c] : pik[c];
            UNROLL for (int c = 0; c < NX; c++) ACC(D.pi, 0).st(k * NX + c, pik[c]);
        }
#endif
        double rdl[NB], rdu[NB], gadd[NB], gam[NB];
        UNROLL for (int j = 0; j < NB; j++)
        {
            GQP_ROW(j, has, ib);
            const int el = S.o_ct + ib, eu = el + nbg;
            const double lbv = ACC(D.dvec, 0).ldo(el, ord), ubv = ACC(D.dvec, 0).ldo(eu, ord);
#if GQP_KB_FOLD
            double laml = ACC(D.lam, 0).ldo(el, ord), lamu = ACC(D.lam, 0).ldo(eu, ord);
            double tl = ACC(D.t, 0).ldo(el, ord), tu = ACC(D.t, 0).ldo(eu, ord);
            const bool al = has && ((am >> ib) & 1), au = has && ((am >> (nbg + ib)) & 1);
            {
                /* the step of the row's multipliers and slacks, floored as the update pass floors them; a side that does not take
                 * part keeps its value; every existing row is written back (whole lines) */
                const double dll = ACC(D.dlam, 0).ldo(el, ord), dlu = ACC(D.dlam, 0).ldo(eu, ord);
                const double dtl = ACC(D.dt, 0).ldo(el, ord), dtu = ACC(D.dt, 0).ldo(eu, ord);
                const double nll = laml + apd * dll, nlu = lamu + apd * dlu, ntl = tl + apd * dtl, ntu = tu + apd * dtu;
                laml = (pend && al) ? (nll < O.lam_min ? O.lam_min : nll) : laml;
                lamu = (pend && au) ? (nlu < O.lam_min ? O.lam_min : nlu) : lamu;
                tl = (pend && al) ? (ntl < O.t_min ? O.t_min : ntl) : tl;
                tu = (pend && au) ? (ntu < O.t_min ? O.t_min : ntu) : tu;
                if (has)
                {
                    ACC(D.lam, 0).st(el, laml); ACC(D.lam, 0).st(eu, lamu);
                    ACC(D.t, 0).st(el, tl); ACC(D.t, 0).st(eu, tu);
                }
            }
#else
            const double laml = ACC(D.lam, 0).ldo(el, ord), lamu = ACC(D.lam, 0).ldo(eu, ord);
            const double tl = ACC(D.t, 0).ldo(el, ord), tu = ACC(D.t, 0).ldo(eu, ord);
            const bool al = has && ((am >> ib) & 1), au = has && ((am >> (nbg + ib)) & 1);
#endif
            const double ll = al ? laml : 0.0, lu = au ? lamu : 0.0;
            const double ttl = al ? tl : 1.0, ttu = au ? tu : 1.0;
            rdl[j] = al ? v[j] - lbv - ttl : 0.0;
            rdu[j] = au ? ubv - v[j] - ttu : 0.0;
            const double rml = al ? ll * ttl - O.tau_min : 0.0, rmu = au ? lu * ttu - O.tau_min : 0.0;
            nacc(nrm_d, rdl[j]); nacc(nrm_d, rdu[j]); nacc(nrm_m, rml); nacc(nrm_m, rmu);
            musum += ll * ttl + lu * ttu;
            nact += (int) al + (int) au;
            gt[j] -= ll - lu;
            const double itl = frcp(ttl), itu = frcp(ttu);
            gam[j] = ll * itl + lu * itu;
            gadd[j] = (rml + ll * rdl[j]) * itl - (rmu + lu * rdu[j]) * itu;
        }

        /* ---------------- Hessian rows in load phases: H v, M = H~ + W W', W w0 ---------------- */
        double M[NP], hv[n], mm[n];
        UNROLL for (int r = 0; r < n; r++) hv[r] = 0.0;
        const Acc aRSQ = ACC(D.RSQ, k * NP);
        UNROLL for (int r = 0; r < n; r++)
        {
            if (r % GQP_HROW_CHUNK == 0)
            {
                /* next chunk of Hessian rows; ordered after the previous chunk's last update */
                if (r > 0) GQP_AFTER(ord, M[PK(r - 1, r - 1)]);
                UNROLL for (int r2 = r; r2 < n && r2 < r + GQP_HROW_CHUNK; r2++)
                    UNROLL for (int c = 0; c <= r2; c++) M[PK(r2, c)] = aRSQ.ldo(PK(r2, c), ord);
            }
            UNROLL for (int c = 0; c < r; c++)
            {
                hv[r] += M[PK(r, c)] * v[c];
                hv[c] += M[PK(r, c)] * v[r];
            }
            hv[r] += M[PK(r, r)] * v[r];
            M[PK(r, r)] += O.reg_prim + (r < NB ? gam[r < NB ? r : 0] : 0.0);
            /* row r of W, then the rows c <= r two at a time (bounded LDS reads in flight) */
            int lr = lane_r;
            GQP_AFTER(lr, hv[r]);
            double wr[NX];
            UNROLL for (int q = 0; q < NX; q++)
                wr[q] = r < NU ? Wu[(r < NU ? r : 0) * NX + q]
                      : W_IN_LDS ? Wl[W_IN_LDS ? (((r < NU ? NU : r) - NU) * NX + q) * 64 + lr : 0]
                                 : Wx[W_IN_LDS ? 0 : ((r < NU ? NU : r) - NU) * NX + q];
            double a = 0.0;
            UNROLL for (int c = 0; c < NX; c++) a += wr[c] * w0[c];
            mm[r] = a + (r < NB ? gadd[r < NB ? r : 0] : 0.0);
            UNROLL for (int c = 0; c <= r; c++)
            {
                if (c >= NU && ((c - NU) & 1) == 0 && c > NU) GQP_AFTER(lr, M[PK(r, c - 1)]);
                double sacc = 0.0;
                UNROLL for (int q = 0; q < NX; q++)
                {
                    const double wc = c < NU ? Wu[(c < NU ? c : 0) * NX + q]
                                    : W_IN_LDS ? Wl[W_IN_LDS ? (((c < NU ? NU : c) - NU) * NX + q) * 64 + lr : 0]
                                               : Wx[W_IN_LDS ? 0 : ((c < NU ? NU : c) - NU) * NX + q];
                    sacc += wr[q] * wc;
                }
                M[PK(r, c)] += sacc;
            }
        }
        UNROLL for (int r = 0; r < n; r++)
        {
            obj += (0.5 * hv[r] + g[r]) * v[r];
            gt[r] += hv[r] + g[r];
        }
        UNROLL for (int c = 0; c < NX; c++) gt[NU + c] -= pik[c];
        /* residual norms; fixed variables carry no residual */
        UNROLL for (int j = 0; j < n; j++)
        {
            if ((S.emask >> j) & 1) gt[j] = 0.0;
            nacc(nrm_g, gt[j]);
        }
        UNROLL for (int c = 0; c < NX; c++) nacc(nrm_b, rb[c]);
        UNROLL for (int j = 0; j < n; j++) ACC(D.rg, 0).st(k * n + j, gt[j]);
        UNROLL for (int c = 0; c < NX; c++) ACC(D.rb, 0).st(k * NX + c, rb[c]);
        /* m = stationarity residual + condensed inequality terms + W w0 */
        UNROLL for (int j = 0; j < n; j++) gt[j] += mm[j];
        /* fixed variables: unit row/column, zero rhs (select, no branch) */
        UNROLL for (int r = 0; r < n; r++)
        {
            const bool fr = (S.emask >> r) & 1;
            if (fr) gt[r] = 0.0;
            UNROLL for (int c = 0; c <= r; c++)
            {
                const bool fc = (S.emask >> c) & 1;
                M[PK(r, c)] = (fr || fc) ? (r == c ? 1.0 : 0.0) : M[PK(r, c)];
            }
        }
        /* ---------------- Cholesky (inverse diagonal kept in registers) ---------------- */
        double invd[n];
        UNROLL for (int jc = 0; jc < n; jc++)
        {
            const double d = M[PK(jc, jc)];
            const bool pos = d > 0.0;
            const double r0 = frsqrt(pos ? d : 1.0); /* unconditional: no branch around v_rsq */
            const double inv = pos ? r0 : 0.0;
            invd[jc] = inv;
            M[PK(jc, jc)] = pos ? d * inv : 0.0;
            UNROLL for (int r = jc + 1; r < n; r++) M[PK(r, jc)] *= inv;
            UNROLL for (int c = jc + 1; c < n; c++)
                UNROLL for (int r = c; r < n; r++) M[PK(r, c)] -= M[PK(r, jc)] * M[PK(c, jc)];
        }
        UNROLL for (int e = 0; e < NP; e++) ACC(D.Lf, 0).st(k * NP + e, M[e]);
        UNROLL for (int r = 0; r < n; r++)
        {
            double a = gt[r];
            UNROLL for (int c = 0; c < r; c++) a -= M[PK(r, c)] * gt[c];
            gt[r] = a * invd[r];
            ACC(D.lf, 0).st(k * n + r, gt[r]);
        }
        UNROLL for (int r = 0; r < NX; r++)
        {
            lx[r] = gt[NU + r];
            UNROLL for (int c = 0; c <= r; c++) Lx[PK(r, c)] = M[PK(NU + r, NU + c)];
        }
        UNROLL for (int c = 0; c < NX; c++) { xn[c] = v[NU + c]; pin[c] = pik[c]; }
        /* rd of the existing rows (rm = lam*t - tau is recomputed by the consumers) */
        UNROLL for (int j = 0; j < NB; j++)
        {
            GQP_ROW(j, has, ib);
            if (has)
            {
                ACC(D.rd, 0).st(S.o_ct + ib, rdl[j]);
                ACC(D.rd, 0).st(S.o_ct + nbg + ib, rdu[j]);
            }
        }
    }

    if (!run) return;
#if GQP_KB_FOLD
    D.apend[i] = 0.0; /* applied */
#endif
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
    if (bad) status = 1;
    else if (nrm_g <= O.tol_stat && nrm_b <= O.tol_eq && nrm_d <= O.tol_ineq && nrm_m <= O.tol_comp) status = 0;
    else if (it >= O.iter_max) status = 2;
    else if (dabs(D.alpha[i]) <= O.alpha_min) status = 3;
    if (status != GQP_RUNNING)
    {
        D.status[i] = status;
        atomicSub(D.n_active, 1);
    }
}

/* ------------------------------------------------------- rhs-only backward */

template <int NX, int NU, bool XBOX>
__global__ void __launch_bounds__(64) kb_backrhs(GqpDev D, GqpOpts O, int redo)
{
    constexpr int n = NX + NU, NP = n * (n + 1) / 2, NPX = NX * (NX + 1) / 2, NB = XBOX ? n : NU;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.B) return;
    const bool run = D.status[i] == GQP_RUNNING;
    if (redo ? !(run && D.alpha[i] < 0.0) : !GQP_WAVE_ANY(run)) return;
    const double smu = D.smu[i];
    const double pscale = redo ? 0.0 : 1.0; /* redo = centering only: drop dlam_aff*dt_aff */

    double Lx[NPX], lx[NX];
    UNROLL for (int e = 0; e < NPX; e++) Lx[e] = 0.0;
    UNROLL for (int c = 0; c < NX; c++) lx[c] = 0.0;

    for (int k = D.N; k >= 0; k--)
    {
        const StageU S = stage_u(D.st, k);
        const uint64_t imask = S.bmask & ~S.emask;
        const uint64_t am = GAT(D.amask, k);
        const int nbg = S.nb;

        /* With box rows on the states (XBOX) the stage needs 8 x (NU+NX) row values on top of L and
         * [B A]': the loads are then issued in ordered phases (see kb_factor) -- rows, dynamics rows,
         * factor -- so that no phase holds more than one big block.  Without state rows everything
         * fits and is fetched in one go (ord stays a compile-time 0). */
        int ord = 0;
        double rb[NX], gt[n];
        UNROLL for (int c = 0; c < NX; c++) rb[c] = ACC(D.rb, 0).ld(k * NX + c);
        UNROLL for (int j = 0; j < n; j++) gt[j] = ACC(D.rg, 0).ld(k * n + j);
        UNROLL for (int j = 0; j < NB; j++)
        {
            GQP_ROW(j, has, ib);
            const int el = S.o_ct + ib, eu = el + nbg;
            const double laml = ACC(D.lam, 0).ld(el), lamu = ACC(D.lam, 0).ld(eu);
            const double tl = ACC(D.t, 0).ld(el), tu = ACC(D.t, 0).ld(eu);
            const double rdl = ACC(D.rd, 0).ld(el), rdu = ACC(D.rd, 0).ld(eu);
            const double pl = ACC(D.pcorr, 0).ld(el), pu = ACC(D.pcorr, 0).ld(eu);
            const bool al = has && ((am >> ib) & 1), au = has && ((am >> (nbg + ib)) & 1);
            const double ll = al ? laml : 0.0, lu = au ? lamu : 0.0;
            const double ttl = al ? tl : 1.0, ttu = au ? tu : 1.0;
            const double rml = al ? ll * ttl - O.tau_min + pscale * pl - smu : 0.0;
            const double rmu = au ? lu * ttu - O.tau_min + pscale * pu - smu : 0.0;
            const double dl = al ? rdl : 0.0, du = au ? rdu : 0.0;
            gt[j] += (rml + ll * dl) * frcp(ttl) - (rmu + lu * du) * frcp(ttu);
        }
        /* y = Lx+ (Lx+' rb + lx+) ; m = gt + BAt y */
        double w0[NX], y[NX];
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
        if (XBOX) GQP_AFTER(ord, gt[NB - 1]);
        const Acc aBAt = ACC(D.BAt, k * n * NX);
        UNROLL for (int r = 0; r < n; r++)
        {
            double a = 0.0;
            UNROLL for (int c = 0; c < NX; c++) a += aBAt.ldo(r * NX + c, ord) * y[c];
            gt[r] += a;
        }
        UNROLL for (int r = 0; r < n; r++) if ((S.emask >> r) & 1) gt[r] = 0.0;
        if (XBOX) GQP_AFTER(ord, gt[n - 1]);
        double L[NP];
        UNROLL for (int e = 0; e < NP; e++) L[e] = ACC(D.Lf, 0).ldo(k * NP + e, ord);
        UNROLL for (int r = 0; r < n; r++)
        {
            double a = gt[r];
            UNROLL for (int c = 0; c < r; c++) a -= L[PK(r, c)] * gt[c];
            const double d = L[PK(r, r)];
            gt[r] = d != 0.0 ? a * frcp(d) : 0.0;
            ACC(D.lf, 0).st(k * n + r, gt[r]);
        }
        UNROLL for (int r = 0; r < NX; r++)
        {
            lx[r] = gt[NU + r];
            UNROLL for (int c = 0; c <= r; c++) Lx[PK(r, c)] = L[PK(NU + r, NU + c)];
        }
    }
}

/* ----------------------------------------------------------------- forward */

template <int NX, int NU, bool XBOX, bool CORR>
__global__ void __launch_bounds__(64) kb_forward(GqpDev D, GqpOpts O, int redo)
{
    constexpr int n = NX + NU, NP = n * (n + 1) / 2, NB = XBOX ? n : NU;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.B) return;
    const bool run = D.status[i] == GQP_RUNNING;
    if (redo ? !(run && D.alpha[i] < 0.0) : !GQP_WAVE_ANY(run)) return;
    const double smu = CORR ? D.smu[i] : 0.0;
    const double pscale = (CORR && !redo) ? 1.0 : 0.0;

    double alpha = 1.0, S0 = 0.0, S1 = 0.0, S2 = 0.0;
    int nact = 0;
    double dx[NX];
    UNROLL for (int c = 0; c < NX; c++) dx[c] = 0.0;

    for (int k = 0; k <= D.N; k++)
    {
        const StageU S = stage_u(D.st, k);
        const uint64_t imask = S.bmask & ~S.emask;
        const uint64_t am = GAT(D.amask, k);
        const int nbg = S.nb;

        double L[NP], l[n], bat[n * NX], rbn[NX];
        UNROLL for (int e = 0; e < NP; e++) L[e] = ACC(D.Lf, 0).ld(k * NP + e);
        UNROLL for (int j = 0; j < n; j++) l[j] = ACC(D.lf, 0).ld(k * n + j);
        UNROLL for (int e = 0; e < n * NX; e++) bat[e] = ACC(D.BAt, 0).ld(k * n * NX + e);
        UNROLL for (int c = 0; c < NX; c++) rbn[c] = ACC(D.rb, 0).ld(k * NX + c);
        double laml[NB], lamu[NB], tl[NB], tu[NB], rdl[NB], rdu[NB], pl[NB], pu[NB];
        UNROLL for (int j = 0; j < NB; j++)
        {
            GQP_ROW(j, has, ib);
            const int el = S.o_ct + ib, eu = el + nbg;
            laml[j] = ACC(D.lam, 0).ld(el); lamu[j] = ACC(D.lam, 0).ld(eu);
            tl[j] = ACC(D.t, 0).ld(el); tu[j] = ACC(D.t, 0).ld(eu);
            rdl[j] = ACC(D.rd, 0).ld(el); rdu[j] = ACC(D.rd, 0).ld(eu);
            if (CORR) { pl[j] = ACC(D.pcorr, 0).ld(el); pu[j] = ACC(D.pcorr, 0).ld(eu); }
            else { pl[j] = 0.0; pu[j] = 0.0; }
        }

        double invd[n];
        UNROLL for (int r = 0; r < n; r++)
        {
            const double d = L[PK(r, r)];
            invd[r] = d != 0.0 ? frcp(d) : 0.0;
        }
        double dv[n];
        /* dpi_k = Lx (Lx' dx + lx): dx = 0 and the result is discarded at k = 0 (slot 0 stays 0
         * because the store is skipped there) */
        if (CORR)
        {
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
                if (k > 0) ACC(D.dpi, 0).st(k * NX + r, a);
            }
        }
        /* x part: given by the dynamics for k > 0, solved for k = 0 (select, same code) */
        const bool first = k == 0;
        UNROLL for (int r = n - 1; r >= NU; r--)
        {
            double a = -l[r];
            UNROLL for (int p = r + 1; p < n; p++) a -= L[PK(p, r)] * dv[p];
            dv[r] = first ? a * invd[r] : dx[r - NU];
        }
        UNROLL for (int r = NU - 1; r >= 0; r--)
        {
            double a = -l[r];
            UNROLL for (int p = r + 1; p < n; p++) a -= L[PK(p, r)] * dv[p];
            dv[r] = a * invd[r];
        }
        if (CORR) { UNROLL for (int j = 0; j < n; j++) ACC(D.dux, 0).st(k * n + j, dv[j]); }
        UNROLL for (int c = 0; c < NX; c++) dx[c] = rbn[c];
        UNROLL for (int r = 0; r < n; r++)
            UNROLL for (int c = 0; c < NX; c++) dx[c] += bat[r * NX + c] * dv[r];

        double o_dll[NB], o_dlu[NB], o_dtl[NB], o_dtu[NB];
        UNROLL for (int j = 0; j < NB; j++)
        {
            GQP_ROW(j, has, ib);
            const bool al = has && ((am >> ib) & 1), au = has && ((am >> (nbg + ib)) & 1);
            const double ll = al ? laml[j] : 0.0, lu = au ? lamu[j] : 0.0;
            const double ttl = al ? tl[j] : 1.0, ttu = au ? tu[j] : 1.0;
            const double rml = al ? ll * ttl - O.tau_min + pscale * pl[j] - smu : 0.0;
            const double rmu = au ? lu * ttu - O.tau_min + pscale * pu[j] - smu : 0.0;
            const double dtl = al ? dv[j] + rdl[j] : 0.0, dtu = au ? -dv[j] + rdu[j] : 0.0;
            const double dll = al ? -(rml + ll * dtl) * frcp(ttl) : 0.0;
            const double dlu = au ? -(rmu + lu * dtu) * frcp(ttu) : 0.0;
            /* ratio test, branch-free: candidate = -value/step where the step is negative */
            {
                const double c1 = -ll * frcp(dll), c2 = -lu * frcp(dlu), c3 = -ttl * frcp(dtl), c4 = -ttu * frcp(dtu);
                alpha = (dll < 0.0 && c1 < alpha) ? c1 : alpha;
                alpha = (dlu < 0.0 && c2 < alpha) ? c2 : alpha;
                alpha = (dtl < 0.0 && c3 < alpha) ? c3 : alpha;
                alpha = (dtu < 0.0 && c4 < alpha) ? c4 : alpha;
            }
            /* (corrector sweep too: the conditional corrector asks for the duality measure its step ends at) */
            S0 += ll * ttl + lu * ttu;
            S1 += ll * dtl + ttl * dll + lu * dtu + ttu * dlu;
            S2 += dll * dtl + dlu * dtu;
            nact += (int) al + (int) au;
            o_dll[j] = dll; o_dlu[j] = dlu; o_dtl[j] = dtl; o_dtu[j] = dtu;
        }
        UNROLL for (int j = 0; j < NB; j++)
        {
            GQP_ROW(j, has, ib);
            if (has)
            {
                const int el = S.o_ct + ib, eu = el + nbg;
                if (CORR)
                {
                    ACC(D.dlam, 0).st(el, o_dll[j]); ACC(D.dlam, 0).st(eu, o_dlu[j]);
                    ACC(D.dt, 0).st(el, o_dtl[j]); ACC(D.dt, 0).st(eu, o_dtu[j]);
                }
                else
                {
                    ACC(D.pcorr, 0).st(el, o_dll[j] * o_dtl[j]);
                    ACC(D.pcorr, 0).st(eu, o_dlu[j] * o_dtu[j]);
                }
            }
        }
    }

    const int it = D.iter[i];
    double *st = (i < D.stat_inst && it + 1 < D.stat_rows) ? D.stat + (size_t) (it + 1) * GQP_STAT_COLS * D.stat_inst + i : nullptr;
    if (!CORR)
    {
        if (!run) return;
        /* mu_aff = sum (lam + a dlam)(t + a dt) / nact, expanded in the three running sums */
        const double mu = D.mu[i];
        const double mu_aff = nact > 0 ? (S0 + alpha * S1 + alpha * alpha * S2) / nact : 0.0;
        double sigma = mu > 0.0 ? mu_aff / mu : 0.0;
        sigma = sigma * sigma * sigma;
        D.smu[i] = sigma * mu;
        D.alpha[i] = alpha;
        if (st) { st[0] = alpha; st[1 * D.stat_inst] = alpha; st[2 * D.stat_inst] = mu_aff; st[3 * D.stat_inst] = sigma; }
        return;
    }
    const double alpha_aff = dabs(D.alpha[i]);
    if (run && O.cond_pred_corr && !redo && nact > 0 && (S0 + alpha * S1 + alpha * alpha * S2) / nact > 2.0 * D.mu[i])
    {
        D.alpha[i] = -alpha_aff; /* the step would more than double the duality measure: flag for the redo pair (centering only) */
        return;
    }
    /* no inequality rows (mu == 0 exactly): the Newton step solves the QP, take it fully */
    const double a = !run ? 0.0 : D.mu[i] > 0.0 ? gqp_step_scale(alpha) : 1.0;
#if GQP_KB_FOLD
    /* folded update: the step stays in (dux, dpi, dlam, dt), the next factor sweep applies it (kb_factor) */
    if (!run) return;
    D.apend[i] = a;
    D.alpha[i] = alpha;
    D.iter[i] = it + 1;
    if (st) { st[4 * D.stat_inst] = alpha; st[5 * D.stat_inst] = alpha; }
    return;
#endif
    /* update pass.  Straight-line per stage like the sweep above: every load of the stage (62 for C2) is issued before
     * the first store, rows that do not exist are read through the clamped index, a side that does not take part gets
     * its own value written back; the stage structure is fetched one stage ahead.  (With a branch per row and side the
     * pass made one memory round trip per row -- 4 loads in flight per wave -- and ran at 3.1 TB/s.) */
    StageU Sn = stage_u(D.st, 0);
    uint64_t amn = GAT(D.amask, 0);
    for (int k = 0; k <= D.N; k++)
    {
        const StageU S = Sn;
        const uint64_t am = amn;
        {
            const int kn = k < D.N ? k + 1 : k;
            Sn = stage_u(D.st, kn);
            amn = GAT(D.amask, kn);
        }
        const uint64_t imask = S.bmask & ~S.emask;
        const int nbg = S.nb;
        const Acc aux_ = ACC(D.ux, k * n), adux_ = ACC(D.dux, k * n);
        const Acc api_ = ACC(D.pi, (k + 1) * NX), adpi_ = ACC(D.dpi, (k + 1) * NX);
        double vx[n], vdx[n], vp[NX], vdp[NX];
        UNROLL for (int j = 0; j < n; j++) { vx[j] = aux_.ld(j); vdx[j] = adux_.ld(j); }
        UNROLL for (int c = 0; c < NX; c++) { vp[c] = api_.ld(c); vdp[c] = adpi_.ld(c); }
        double vl[2 * NB], vdl[2 * NB], vt[2 * NB], vdt[2 * NB];
        UNROLL for (int j = 0; j < NB; j++)
        {
            GQP_ROW(j, has, ib);
            UNROLL for (int side = 0; side < 2; side++)
            {
                const int e = S.o_ct + side * nbg + ib;
                vl[2 * j + side] = ACC(D.lam, 0).ld(e); vdl[2 * j + side] = ACC(D.dlam, 0).ld(e);
                vt[2 * j + side] = ACC(D.t, 0).ld(e); vdt[2 * j + side] = ACC(D.dt, 0).ld(e);
            }
        }
        UNROLL for (int j = 0; j < n; j++) aux_.st(j, run ? vx[j] + a * vdx[j] : vx[j]);
        UNROLL for (int c = 0; c < NX; c++) api_.st(c, run ? vp[c] + a * vdp[c] : vp[c]);
        UNROLL for (int j = 0; j < NB; j++)
        {
            GQP_ROW(j, has, ib);
            if (!has) continue;
            UNROLL for (int side = 0; side < 2; side++)
            {
                const int e = S.o_ct + side * nbg + ib;
                const bool act = run && ((am >> (side * nbg + ib)) & 1);
                const double lam = vl[2 * j + side] + a * vdl[2 * j + side];
                const double t = vt[2 * j + side] + a * vdt[2 * j + side];
                ACC(D.lam, 0).st(e, act ? (lam < O.lam_min ? O.lam_min : lam) : vl[2 * j + side]);
                ACC(D.t, 0).st(e, act ? (t < O.t_min ? O.t_min : t) : vt[2 * j + side]);
            }
        }
    }
    if (!run) return;
    D.alpha[i] = alpha;
    D.iter[i] = it + 1;
    if (st) { st[4 * D.stat_inst] = alpha; st[5 * D.stat_inst] = alpha; }
}

/* ---------------------------------------------------------------- finalize */

/* multipliers of equality-flagged bounds from stationarity (what d_ocp_qp_restore_eq_dof does
 * after the reduced solve, ocp_qp_partial_condensing.c:683), t = 0 for them; natural slack and
 * zero multiplier for masked sides (ocp_qp_common.c:874-921 restated for those rows) */
template <int NX, int NU>
__global__ void __launch_bounds__(64) kb_finalize(GqpDev D)
{
    constexpr int n = NX + NU, NP = n * (n + 1) / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.B) return;
    for (int k = 0; k <= D.N; k++)
    {
        const StageU S = stage_u(D.st, k);
        if (S.nb == 0) continue;
        const int nbg = S.nb;
        const uint64_t am = GATL(D.amask, k);
        double v[n];
        UNROLL for (int j = 0; j < n; j++) v[j] = GATL_LD(D.ux, k * n + j);
        int ib = 0;
        UNROLL for (int j = 0; j < n; j++)
        {
            if (!((S.bmask >> j) & 1)) continue;
            const int el = S.o_ct + ib, eu = el + nbg;
            if ((S.emask >> j) & 1)
            {
                double a = GATL_LD(D.rq, k * n + j);
                UNROLL for (int c = 0; c < n; c++) a += GATL_LD(D.RSQ, k * NP + (c <= j ? PK(j, c) : PK(c, j))) * v[c];
                UNROLL for (int c = 0; c < NX; c++) a += GATL_LD(D.BAt, (k * n + j) * NX + c) * GATL_LD(D.pi, (k + 1) * NX + c);
                if (j >= NU) a -= GATL_LD(D.pi, k * NX + (j >= NU ? j - NU : 0));
                GATL(D.lam, el) = a > 0.0 ? a : 0.0;
                GATL(D.lam, eu) = a < 0.0 ? -a : 0.0;
                GATL(D.t, el) = 0.0;
                GATL(D.t, eu) = 0.0;
            }
            else
            {
                if (!((am >> ib) & 1)) { GATL(D.t, el) = v[j] - GATL(D.dvec, el); GATL(D.lam, el) = 0.0; }
                if (!((am >> (nbg + ib)) & 1)) { GATL(D.t, eu) = GATL(D.dvec, eu) - v[j]; GATL(D.lam, eu) = 0.0; }
            }
            ib++;
        }
    }
}

} // namespace gqp

#endif
