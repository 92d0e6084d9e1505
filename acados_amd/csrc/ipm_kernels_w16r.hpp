/*
 * ipm_kernels_w16r.hpp -- SIXTEEN LANES PER INSTANCE, SEVERAL ROWS PER LANE: stage blocks with 17 <= nu + nx <= 32 -- box
 * constrained (the nx = 24 classes of C5, the condensed shape of C3: nx = 8, nu = 15) or, in the GEN instantiations, with
 * general rows and slacks (C4).
 *
 * The wave-per-instance kernels (ipm_kernels_wpi.hpp) are bound by ONE wave's dependent instruction stream
 * (profiles/r02_wpi_phase_cycles.txt: ~9,400 instructions and ~98,000 cycles per stage at n = 30), most of it LDS
 * round trips, barriers and address arithmetic on run-time dims.  This family keeps the register-row scheme of
 * ipm_kernels_w16.hpp -- four instances per wavefront, one per 16-lane DPP row, `row_newbcast` as the broadcast --
 * and gives every lane R = ceil(n / 16) rows of the stage matrices: lane l of a row owns the variables l, l + 16
 * (slot 0, slot 1).  Variable j lives in lane j & 15, slot j >> 4, so a broadcast of "the value of variable j" is
 * still ONE DPP move with an immediate lane, and one broadcast feeds R multiply-adds.
 *
 * Differences from the one-row family, all of them about registers and latency (n = 30 needs ~420 registers: one wave per
 * SIMD, four workgroups per CU, 40 KB of LDS each):
 *   - what a stage reads from HBM arrives one stage ahead: the packed H / factor block and the [B A]' block by LDS-DMA
 *     (global_load_lds_dwordx4, no VGPR in between), vectors and box rows by plain loads into registers;
 *   - rows and columns of H, of the factor and of [B A]' are read from their LDS images where they are used (lane base +
 *     immediate offset); the previous factor's x-block stays in the register rows of the state slots and is broadcast from
 *     there (factor sweep) or sits in a small LDS tile (rhs-only sweep);
 *   - only the lower triangle is carried: slot 0 (rows < 16) never touches columns >= 16;
 *   - accumulators are pinned at the end of their phase and the lanes' row index is laundered once per stage: machine
 *     sinking and loop-invariant code motion otherwise spill hundreds of registers (W16R_OPAQUE).
 * Same algorithm, HBM arrays and slot conventions as ipm_kernels_w16.hpp / ipm_kernels_wpi.hpp (whose init / finalize
 * kernels serve this family too).  Shapes are compile-time; general rows and slacks: see "inequality rows with slacks"
 * below (one slack per row; shared slacks stay with the wave-per-instance kernels).
 * The CPU test tier runs these kernels under tests/hostsim like the one-row family.  One rule follows from it for the kernels
 * that use LDS-DMA: the host simulation copies a lane's 16 bytes when THAT lane reaches the call and keeps the lanes of a
 * block in step by counting their yields (every GQP_ROWSYNC, every broadcast), so the four rows of a workgroup must execute
 * the same number of broadcasts -- no broadcast inside a value-dependent conditional (`d != 0 ? bcast(x) / d : 0` let a row
 * with an empty LDS tile run a stage ahead of the others and overwrite the buffer they were still reading).  On the device
 * the wavefront executes in lockstep anyway.
 */
#ifndef IPM_KERNELS_W16R_HPP_
#define IPM_KERNELS_W16R_HPP_

#include "ipm_kernels_w16.hpp"

namespace gqp
{

/* per-instance LDS tile (doubles).
 * Factor sweep: exchange buffer (host simulation) + two regions filled by LDS-DMA one stage ahead: HR, the packed H block
 * of the stage as it lies in memory, and BR, its [B A]' block as it lies in memory (BR also carries the transposition tile
 * of the previous factor's x-block once the stage is done with [B A]').
 * rhs-only sweep: exchange buffer + the square x-block tile; forward sweep: exchange buffer + packed factor and [B A]' by
 * LDS-DMA (two buffers where 40 KB per workgroup allow).  GEN variants add the stage's general rows and the row vectors. */
template <int NX, int NU, int NG = 0>
struct W16RLds
{
    static constexpr int n = NX + NU, R = (n + 15) / 16, LDX = NX + 1, NP = n * (n + 1) / 2, NB = n * NX;
    /* GEN variants: general rows [D C] of the stage, [g][n], + four 16-entry row vectors (value in, gamma / gadd / dlam out) */
    static constexpr int GSZ = NG > 0 ? ((NG * n + 1) & ~1) + 80 : 0, RWO = (NG * n + 1) & ~1; /* (+ 16: row index of every lane) */
    static constexpr int XB = 0, TA = 16;
    static constexpr int GTA = 16 + NX * LDX, SZ_A = GTA + GSZ;
    static constexpr int HR = 16, HSZ = (NP + 1) & ~1, BR = HR + HSZ, BSZ = (NB + 1) & ~1;
    static constexpr int GT = BR + BSZ, SZ_K = GT + GSZ;
    /* forward sweep: the packed factor and the [B A]' block of the stage by LDS-DMA, in two buffers (the next stage
     * lands while this one computes) where four workgroups per CU still fit, in one otherwise */
    static constexpr int FBUF = HSZ + BSZ;
#if defined(W16R_NBUF1) /* development builds: one buffer in the forward sweep whatever fits */
    static constexpr int NBUF = 1;
#else
    static constexpr int NBUF = (16 + 2 * FBUF) * 4 * 8 <= 40960 ? 2 : 1;
#endif
    static constexpr int GTF = 16 + NBUF * FBUF, SZ_F2 = GTF + GSZ;
    static constexpr int SZ0 = SZ_A > SZ_F2 ? SZ_A : SZ_F2;
    static constexpr int SZ = ((SZ0 > SZ_K ? SZ0 : SZ_K) + 1) & ~1; /* even: every instance's tile starts on 16 bytes */
};

/* LDS-DMA (global_load_lds_dwordx4): the 64 lanes copy 16 bytes each from sbase + lane * 16 + IMM (global, wave-uniform
 * base) to ldsp + lane * 16 + IMM (LDS, wave-uniform base) without touching a VGPR.  Inline asm: the destination base
 * travels in M0, which the compiler reserves -- saved and restored inside the statement; the compiler neither counts
 * these requests nor orders LDS reads behind them, the kernel waits by hand (W16R_DMA_WAIT) and drains its own LDS reads
 * of a region before overwriting it (W16R_LDS_DRAIN).  tools/lds_dma_probe checks the addressing on the GPU. */
#if defined(__HIP_DEVICE_COMPILE__)
template <int IMM>
__device__ static inline void w16r_dma16(const double *sbase, const double *ldsp)
{
#if defined(W16R_NO_DMA) /* development builds: the same copy through registers, waits inserted by the compiler */
    const size_t o = (size_t) threadIdx.x * 16 + IMM;
    const double a = *(const double *) ((const char *) sbase + o), b = *(const double *) ((const char *) sbase + o + 8);
    *(double *) ((char *) ldsp + o) = a;
    *(double *) ((char *) ldsp + o + 8) = b;
    return;
#endif
    const unsigned voff = threadIdx.x * 16u;
    const unsigned lds = (unsigned) (uintptr_t) (__attribute__((address_space(3))) const double *) ldsp;
    unsigned keep;
#if defined(W16R_M0_DELAY) /* development builds: M0 restored long after the request was issued (is M0 sampled at issue?) */
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds), "n"(IMM) : "memory");
#else
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds), "n"(IMM) : "memory");
#endif
}
/* (s_barrier: a no-op for a one-wave workgroup as far as synchronisation goes; the programming guide orders the reads of
 * DMA'd data behind "vmcnt, then a barrier") */
#define W16R_DMA_WAIT() asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory")
/* The factor and the forward sweep are pinned to ONE wave per SIMD, whatever their register count.  History of the fault
 * that made this a rule (C3 shape <8,15>, development builds `make variant`, tools/stress_w16r.py / variant_rate.py):
 *   - rounds 2-3: the AFFINE forward sweep built for two waves per SIMD gave 30-160 wrong instances in 65,536 (whole
 *     workgroups, NaN, a different set every solve) as soon as two of its waves shared a SIMD; the corrector instantiation
 *     and the factor sweep built the same way were exact.  Not the LDS-DMA (a copy through registers fails the same way; the
 *     mechanism alone is exact at 2 and 4 waves per SIMD, tools/lds_dma_probe/probe2, probe3), not an LDS overrun, not the
 *     runtime's scratch handling, not the spill code as read;
 *   - round 3, narrowed to ONE construct: the stage descriptor fetched with VECTOR loads (what hipcc emitted for the
 *     plain-pointer stage table behind a kernel's first store: `global_load_dword v, v_zero, s[..]` x3) and consumed
 *     behind the compiler's PARTIAL waits (`s_waitcnt vmcnt(4)` / `vmcnt(3)` in front of `v_readfirstlane` / `v_mov`).
 *     Same two-waves build, descriptor loads followed by `s_waitcnt vmcnt(0)` (GQP_STAGE_VECTOR_FULLWAIT): exact.  Same
 *     build with the table in the constant address space (s_load; what ships now): exact at two waves per SIMD, and at
 *     three with 344 bytes of forced spills (W16R_WPE3_FWD_AFF).  Only reverting the table to the plain pointer
 *     (GQP_STAGE_VECTOR_LOADS) brings the failures back (33 / 58 / 83 per solve).  So: not occupancy as such, not spills --
 *     a partial vmcnt wait in front of the descriptor's consumers was not enough in that kernel under co-residency.  The
 *     load / wait pattern in isolation (tools/lds_dma_probe/probe4: uniform-address table loads, per-lane cold loads
 *     behind them, consumers behind vmcnt(4)...(2), 1 / 2 / 4 waves per SIMD) is exact, so what else in that kernel's
 *     VMEM stream the count missed is not identified; no shipped kernel contains the construct any more (the stage table
 *     is read with scalar loads in every family), and none of them has scratch;
 *   - two waves per SIMD do not pay here anyway: with the scalar table the affine sweep fits 254 registers without
 *     scratch and takes 0.513 ms per launch at two waves per SIMD against 0.506 at one (C3, same box).
 * One wave per SIMD is what every test and every measurement runs; the attribute keeps a future compiler from packing two. */
#define W16R_ONE_WAVE_PER_SIMD __attribute__((amdgpu_waves_per_eu(1, 1)))
/* development builds (make variant): two waves per SIMD for the factor and the forward sweeps (W16R_WPE2) or one of them (the rhs-only sweep has no LDS-DMA and no attribute) -- the open problem above */
#define W16R_TWO_WAVES_PER_SIMD __attribute__((amdgpu_waves_per_eu(2, 2)))
#if defined(W16R_WPE2) || defined(W16R_WPE2_FACT)
#define W16R_WPE_FACT W16R_TWO_WAVES_PER_SIMD
#else
#define W16R_WPE_FACT W16R_ONE_WAVE_PER_SIMD
#endif
#if defined(W16R_WPE2) || defined(W16R_WPE2_FWD)
#define W16R_WPE_FWD W16R_TWO_WAVES_PER_SIMD
#elif defined(W16R_WPE2_FWD_AFF) /* ... only the affine / only the corrector instantiation of the forward sweep */
#define W16R_WPE_FWD __attribute__((amdgpu_waves_per_eu(CORR ? 1 : 2, CORR ? 1 : 2)))
#elif defined(W16R_WPE3_FWD_AFF) /* ... three waves per SIMD (<= 170 registers: forces spills) for the affine instantiation */
#define W16R_WPE_FWD __attribute__((amdgpu_waves_per_eu(CORR ? 1 : 3, CORR ? 1 : 3)))
#elif defined(W16R_WPE2_FWD_CORR)
#define W16R_WPE_FWD __attribute__((amdgpu_waves_per_eu(CORR ? 2 : 1, CORR ? 2 : 1)))
#else
#define W16R_WPE_FWD W16R_ONE_WAVE_PER_SIMD
#endif
#define W16R_LDS_DRAIN() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
template <int IMM>
__device__ static inline void w16r_dma16(const double *sbase, const double *ldsp)
{
    const size_t o = (size_t) threadIdx.x * 16 + IMM;
    memcpy((char *) ldsp + o, (const char *) sbase + o, 16);
}
#define W16R_DMA_WAIT() GQP_ROWSYNC()
#define W16R_WPE_FACT
#define W16R_WPE_FWD
#define W16R_LDS_DRAIN() GQP_ROWSYNC()
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(W16R_NO_DMA) && !defined(W16R_DMA_UNGROUPED)
/* NF (1..4) full passes of 64 lanes x 16 bytes from sb / to lp with ONE write of M0: the passes differ in the instruction
 * offset only (it applies to both sides).  M0 is declared clobbered instead of saved and restored around every request:
 * written twice per request (set, restore) the scalar unit waited for the request in flight before each write -- the
 * issue of the 24 requests of a [B A]' block took ~5,000 cycles (profiles/r03_w16r_phase_cycles.txt) */
template <int NF>
__device__ static inline void w16r_dma_group(const double *sb, const double *lp)
{
    const unsigned voff = threadIdx.x * 16u;
    const unsigned lds = (unsigned) (uintptr_t) (__attribute__((address_space(3))) const double *) lp;
    if (NF == 1)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:0"
                     : : "v"(voff), "s"(sb), "s"(lds) : "memory", "m0");
    else if (NF == 2)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:0\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024"
                     : : "v"(voff), "s"(sb), "s"(lds) : "memory", "m0");
    else if (NF == 3)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:0\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:2048"
                     : : "v"(voff), "s"(sb), "s"(lds) : "memory", "m0");
    else
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:0\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072"
                     : : "v"(voff), "s"(sb), "s"(lds) : "memory", "m0");
}
template <int IMM>
__device__ static inline void w16r_dma_tail(const double *sb, const double *lp)
{
    const unsigned voff = threadIdx.x * 16u;
    const unsigned lds = (unsigned) (uintptr_t) (__attribute__((address_space(3))) const double *) lp;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3"
                 : : "v"(voff), "s"(sb), "s"(lds), "n"(IMM) : "memory", "m0");
}
/* G granules of 16 bytes, contiguous on both sides: groups of up to four full passes, then the lanes the last pass needs */
template <int G>
__device__ static inline void w16r_dma_region(const double *sbase, const double *ldsp)
{
    constexpr int NFULL = G / 64, REST = G % 64;
    W16_UNROLL for (int g4 = 0; g4 * 4 < NFULL; g4++)
    {
        const double *sb = sbase + g4 * 512, *lp = ldsp + g4 * 512;
        const int nf = NFULL - 4 * g4 < 4 ? NFULL - 4 * g4 : 4;
        if (nf == 1) w16r_dma_group<1>(sb, lp);
        else if (nf == 2) w16r_dma_group<2>(sb, lp);
        else if (nf == 3) w16r_dma_group<3>(sb, lp);
        else w16r_dma_group<4>(sb, lp);
    }
    if (REST > 0 && (int) threadIdx.x < REST)
    {
        constexpr int p = NFULL;
        const double *sb = sbase + (p >> 2) * 512, *lp = ldsp + (p >> 2) * 512;
        switch (p & 3)
        {
            case 0: w16r_dma_tail<0>(sb, lp); break;
            case 1: w16r_dma_tail<1024>(sb, lp); break;
            case 2: w16r_dma_tail<2048>(sb, lp); break;
            default: w16r_dma_tail<3072>(sb, lp); break;
        }
    }
}
#else
/* G granules of 16 bytes, contiguous on both sides: full passes of 64 lanes, then the lanes the last pass needs */
template <int G>
__device__ static inline void w16r_dma_region(const double *sbase, const double *ldsp)
{
    constexpr int NPASS = (G + 63) / 64;
    W16_UNROLL for (int p = 0; p < NPASS; p++)
    {
        const bool full = (p + 1) * 64 <= G;
        if (full || (int) threadIdx.x < G - 64 * p)
        {
            const double *sb = sbase + (p >> 2) * 512, *lp = ldsp + (p >> 2) * 512;
            switch (p & 3)
            {
                case 0: w16r_dma16<0>(sb, lp); break;
                case 1: w16r_dma16<1024>(sb, lp); break;
                case 2: w16r_dma16<2048>(sb, lp); break;
                default: w16r_dma16<3072>(sb, lp); break;
            }
        }
    }
}
#endif

#ifndef W16R_UPD_CH
#define W16R_UPD_CH 6 /* stages per chunk of the update pass of the corrector sweep */
#endif

/* value of variable j: lane j & 15 of the row, slot j >> 4 (j is a compile-time constant after unrolling) */
#define W16R_BC(arr, j) w16_bcast((arr)[(j) >> 4], (j) & 15, xb)
/* does slot s hold a row >= c ?  (compile-time after unrolling: the lower triangle only) */
#define W16R_LOW(s, c) (16 * (s) + 15 >= (c))
/* The lane-dependent predicates of the unrolled stage body (row == j, row > j, c <= cx, ... for every j) are loop
 * invariant: the compiler hoists hundreds of them out of the stage loop, runs out of scalar registers and spills them
 * lane by lane.  Laundering the lane's row index once per stage keeps each predicate next to its use. */
#if defined(__HIP_DEVICE_COMPILE__)
#define W16R_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define W16R_OPAQUE(x) do { } while (0)
#endif
/* development aid (`make timing`): cycles per phase of instance 0, slots of gqp_wpi_cycles */
#if defined(GQP_WPI_TIMING) && defined(__HIP_DEVICE_COMPILE__)
#define W16R_TICK(slot)                                                            \
    do {                                                                           \
        const unsigned long long now_ = clock64();                                 \
        if (inst == 0 && l == 0) gqp_wpi_cycles[slot] += now_ - tick_;             \
        tick_ = now_;                                                              \
    } while (0)
#else
#define W16R_TICK(slot) do { } while (0)
#endif
/* the fully unrolled stage body is one basic block of several thousand instructions; left alone, the scheduler hoists
 * hundreds of loads and LDS reads to its top and spills.  A fence keeps the broadcasts of the W product column by column
 * (what keeps the multiply-add chains of a phase in place is W16R_OPAQUE: the fence does not bind machine sinking) */
#if defined(__HIP_DEVICE_COMPILE__)
#define W16R_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define W16R_FENCE() do { } while (0)
#endif

/* ------------------------------------------------------------------- inequality rows with slacks (GEN variants)
 * The GEN instantiations (template parameter NG > 0: up to NG general rows per stage) also carry slacks, with the structure
 * of the sixteen-lanes SOFT kernels extended to general rows: every slack belongs to exactly ONE row, box or general, and a
 * stage has at most 16 inequality rows that take part (the host checks both; anything else stays with the
 * wave-per-instance kernels).  EVERY row -- the sorted box rows that are not equality-flagged, then the general rows -- is
 * processed by ONE lane (lane i <-> i-th such row) with the scalar formulas below; a row's value reaches its lane through a
 * 16-entry LDS vector (box row: v_j, written by the slot that owns variable j; general row: a'v, a 16-lane sum), its
 * results return the same way: (gamma, gadd, dlam) of a box row to the slot of its variable, a general row as the
 * rank-one terms gamma a a' / a gadd on the register rows.  Algebra: ipm_kernels_wpi.hpp (GEN kernels) specialised to one
 * row per slack: E = Z + Gamma_s, X = slack stationarity + rho_s, cancellation-free. */
struct W16RowF
{
    double gam, gadd, dlam; /* Hessian weight, gradient term, lam_lower - lam_upper */
};

/* sum over the 16 lanes of a row, result in every lane */
__device__ static inline double w16_rowsum(double v, double *xb)
{
#if defined(__HIP_DEVICE_COMPILE__)
    v += __builtin_amdgcn_update_dpp(0.0, v, 0x128, 0xF, 0xF, true); /* row_ror:8 */
    v += __builtin_amdgcn_update_dpp(0.0, v, 0x124, 0xF, 0xF, true); /* row_ror:4 */
    v += __builtin_amdgcn_update_dpp(0.0, v, 0x122, 0xF, 0xF, true); /* row_ror:2 */
    v += __builtin_amdgcn_update_dpp(0.0, v, 0x121, 0xF, 0xF, true); /* row_ror:1 */
    return v;
#else
    return w16_rsum(v, xb);
#endif
}

/* what the row functions need of the stage descriptor (held in registers, loaded one stage ahead where it matters) */
struct W16Dsc
{
    int nb, ng, ns, o_ct, o_s;
};
/* The row functions are BRANCH-FREE on their loads: every address is clamped into its array (a row that does not exist reads
 * element 0 of the stage / of the array), all loads are issued back to back, the values are selected afterwards.  (A
 * `cond ? load : 0` per element makes the compiler branch around each load and wait for it alone: some thirty exposed
 * memory latencies per row.)  Stores stay behind their predicate. */
#define W16R_CLAMP(arr, e) ((e) < (arr).E ? (e) : 0)

/* factor sweep: residuals, norms, duality measure and the row's contribution to the condensed stage system */
__device__ static inline W16RowF w16r_row_factor(const GqpDev &D, const GqpOpts &O, int inst, bool alive, const W16Dsc S, uint64_t am,
                                                 bool has, int row_, int sj_, double val, double &nrm_g, double &nrm_d, double &nrm_m,
                                                 double &musum, double &nact, double &obj)
{
    const int nbg = S.nb + S.ng, row = has ? row_ : 0, sj = has ? sj_ : -1;
    const bool al = has && ((am >> row) & 1), au = has && ((am >> (nbg + row)) & 1), soft = sj >= 0;
    const int el = W16R_CLAMP(D.lam, S.o_ct + row), eu = W16R_CLAMP(D.lam, S.o_ct + nbg + row);
    const int sq = soft ? sj : 0, se0 = W16R_CLAMP(D.lam, S.o_ct + 2 * nbg + sq), se1 = W16R_CLAMP(D.lam, S.o_ct + 2 * nbg + S.ns + sq);
    const int q0 = W16R_CLAMP(D.sv, S.o_s + sq), q1 = W16R_CLAMP(D.sv, S.o_s + S.ns + sq);
    const bool sal = soft && ((am >> (2 * nbg + sq)) & 1), sau = soft && ((am >> (2 * nbg + S.ns + sq)) & 1);
    const bool hs = D.sv.E > 0; /* uniform: the batch has slack arrays at all */
    /* loads */
    const double v_ll = WAT(D.lam, el), v_lu = WAT(D.lam, eu), v_tl = WAT(D.t, el), v_tu = WAT(D.t, eu);
    const double v_dl = WAT(D.dvec, el), v_du = WAT(D.dvec, eu);
    double v_sl = 0.0, v_su = 0.0, v_sll = 0.0, v_slu = 0.0, v_stl = 1.0, v_stu = 1.0, v_sdl = 0.0, v_sdu = 0.0, sZl = 0.0, szl = 0.0, sZu = 0.0, szu = 0.0;
    if (hs)
    {
        v_sl = WAT(D.sv, q0); v_su = WAT(D.sv, q1);
        v_sll = WAT(D.lam, se0); v_slu = WAT(D.lam, se1); v_stl = WAT(D.t, se0); v_stu = WAT(D.t, se1);
        v_sdl = WAT(D.dvec, se0); v_sdu = WAT(D.dvec, se1);
        sZl = WAT(D.Zz, q0 * 2); szl = WAT(D.Zz, q0 * 2 + 1); sZu = WAT(D.Zz, q1 * 2); szu = WAT(D.Zz, q1 * 2 + 1);
    }
    W16RowF r = {0.0, 0.0, 0.0};
    const double ll = al ? v_ll : 0.0, lu = au ? v_lu : 0.0, ttl = al ? v_tl : 1.0, ttu = au ? v_tu : 1.0;
    const double lbv = al ? v_dl : 0.0, ubv = au ? v_du : 0.0;
    const double ssl = soft ? v_sl : 0.0, ssu = soft ? v_su : 0.0;
    const double rdl = al ? val + ssl - lbv - ttl : 0.0, rdu = au ? ubv - val + ssu - ttu : 0.0;
    const double rml = al ? ll * ttl - O.tau_min : 0.0, rmu = au ? lu * ttu - O.tau_min : 0.0;
    nacc(nrm_d, rdl); nacc(nrm_d, rdu); nacc(nrm_m, rml); nacc(nrm_m, rmu);
    musum += ll * ttl + lu * ttu;
    nact += (double) ((int) al + (int) au);
    r.dlam = ll - lu;
    const double itl = frcp(ttl), itu = frcp(ttu);
    const double bGl = ll * itl, bGu = lu * itu, bRl = (rml + ll * rdl) * itl, bRu = (rmu + lu * rdu) * itu;
    r.gam = bGl + bGu;
    r.gadd = bRl - bRu;
    if (has && alive)
    {
        WAT(D.rd, el) = rdl;
        WAT(D.rd, eu) = rdu;
    }
    /* the slack of the row (everything below is neutral for a hard row: soft = false) */
    const double sll = sal ? v_sll : 0.0, slu = sau ? v_slu : 0.0, stl = sal ? v_stl : 1.0, stu = sau ? v_stu : 1.0;
    const double sdl = sal ? v_sdl : 0.0, sdu = sau ? v_sdu : 0.0;
    if (soft) obj += (0.5 * sZl * ssl + szl) * ssl + (0.5 * sZu * ssu + szu) * ssu;
    const double srdl = sal ? ssl - sdl - stl : 0.0, srdu = sau ? ssu - sdu - stu : 0.0;
    const double srml = sal ? sll * stl - O.tau_min : 0.0, srmu = sau ? slu * stu - O.tau_min : 0.0;
    nacc(nrm_d, srdl); nacc(nrm_d, srdu); nacc(nrm_m, srml); nacc(nrm_m, srmu);
    musum += sll * stl + slu * stu;
    nact += (double) ((int) sal + (int) sau);
    const double sitl = frcp(stl), situ = frcp(stu);
    const double sGl = sll * sitl, sGu = slu * situ;
    const double sPl = (srml + sll * srdl) * sitl, sPu = (srmu + slu * srdu) * situ;
    const double Rl = soft ? sZl * ssl + szl - sll - ll : 0.0, Ru = soft ? sZu * ssu + szu - slu - lu : 0.0; /* slack stationarity */
    nacc(nrm_g, Rl); nacc(nrm_g, Ru);
    const double El = sZl + sGl, Eu = sZu + sGu, Xl = Rl + sPl, Xu = Ru + sPu; /* D, r~ without the row */
    const double Dl = El + bGl, Du = Eu + bGu;
    if (soft && alive)
    {
        WAT(D.rd, se0) = srdl;
        WAT(D.rd, se1) = srdu;
        WAT(D.rgs, q0) = Rl; WAT(D.rgs, q1) = Ru;
        WAT(D.sD, q0) = Dl; WAT(D.sD, q1) = Du;
        WAT(D.sR, q0) = Xl + bRl; WAT(D.sR, q1) = Xu + bRu;
    }
    const double Il = Dl != 0.0 ? frcp(Dl) : 0.0, Iu = Du != 0.0 ? frcp(Du) : 0.0;
    if (soft)
    {
        r.gam = bGl * El * Il + bGu * Eu * Iu;
        r.gadd = (bRl * El - bGl * Xl) * Il - (bRu * Eu - bGu * Xu) * Iu;
    }
    return r;
}

/* rhs-only sweep: the row's gradient term with the corrector's complementarity rhs (rd as stored by the factor sweep) */
__device__ static inline double w16r_row_rhs(const GqpDev &D, const GqpOpts &O, int inst, bool alive, const W16Dsc S, uint64_t am,
                                             bool has, int row_, int sj_, double smu, double pscale)
{
    const int nbg = S.nb + S.ng, row = has ? row_ : 0, sj = has ? sj_ : -1;
    const bool al = has && ((am >> row) & 1), au = has && ((am >> (nbg + row)) & 1), soft = sj >= 0;
    const int el = W16R_CLAMP(D.lam, S.o_ct + row), eu = W16R_CLAMP(D.lam, S.o_ct + nbg + row);
    const int sq = soft ? sj : 0, e0 = W16R_CLAMP(D.lam, S.o_ct + 2 * nbg + sq), e1 = W16R_CLAMP(D.lam, S.o_ct + 2 * nbg + S.ns + sq);
    const int q0 = W16R_CLAMP(D.sv, S.o_s + sq), q1 = W16R_CLAMP(D.sv, S.o_s + S.ns + sq);
    const bool sal = soft && ((am >> (2 * nbg + sq)) & 1), sau = soft && ((am >> (2 * nbg + S.ns + sq)) & 1);
    const bool hs = D.sv.E > 0;
    const double v_ll = WAT(D.lam, el), v_lu = WAT(D.lam, eu), v_tl = WAT(D.t, el), v_tu = WAT(D.t, eu);
    const double v_dl = WAT(D.rd, el), v_du = WAT(D.rd, eu), v_pl = WAT(D.pcorr, el), v_pu = WAT(D.pcorr, eu);
    double v_sll = 0.0, v_slu = 0.0, v_stl = 1.0, v_stu = 1.0, v_srl = 0.0, v_sru = 0.0, v_spl = 0.0, v_spu = 0.0, sgl = 0.0, sgu = 0.0,
           sZl = 0.0, sZu = 0.0, sDl = 0.0, sDu = 0.0;
    if (hs)
    {
        v_sll = WAT(D.lam, e0); v_slu = WAT(D.lam, e1); v_stl = WAT(D.t, e0); v_stu = WAT(D.t, e1);
        v_srl = WAT(D.rd, e0); v_sru = WAT(D.rd, e1); v_spl = WAT(D.pcorr, e0); v_spu = WAT(D.pcorr, e1);
        sgl = WAT(D.rgs, q0); sgu = WAT(D.rgs, q1);
        sZl = WAT(D.Zz, q0 * 2); sZu = WAT(D.Zz, q1 * 2);
        sDl = WAT(D.sD, q0); sDu = WAT(D.sD, q1);
    }
    const double ll = al ? v_ll : 0.0, lu = au ? v_lu : 0.0, ttl = al ? v_tl : 1.0, ttu = au ? v_tu : 1.0;
    const double rdl = al ? v_dl : 0.0, rdu = au ? v_du : 0.0;
    const double rml = al ? ll * ttl - O.tau_min + pscale * v_pl - smu : 0.0;
    const double rmu = au ? lu * ttu - O.tau_min + pscale * v_pu - smu : 0.0;
    const double itl = frcp(ttl), itu = frcp(ttu);
    const double bRl = (rml + ll * rdl) * itl, bRu = (rmu + lu * rdu) * itu;
    const double sll = sal ? v_sll : 0.0, slu = sau ? v_slu : 0.0, stl = sal ? v_stl : 1.0, stu = sau ? v_stu : 1.0;
    const double srdl = sal ? v_srl : 0.0, srdu = sau ? v_sru : 0.0, spl = sal ? v_spl : 0.0, spu = sau ? v_spu : 0.0;
    const double bGl = ll * itl, bGu = lu * itu;
    const double srml = sal ? sll * stl - O.tau_min + pscale * spl - smu : 0.0;
    const double srmu = sau ? slu * stu - O.tau_min + pscale * spu - smu : 0.0;
    const double sitl = frcp(stl), situ = frcp(stu);
    const double Xl = sgl + (srml + sll * srdl) * sitl, Xu = sgu + (srmu + slu * srdu) * situ; /* r~ without the row */
    const double El = sZl + sll * sitl, Eu = sZu + slu * situ;
    if (soft && alive) { WAT(D.sR, q0) = Xl + bRl; WAT(D.sR, q1) = Xu + bRu; }
    const double Il = sDl != 0.0 ? frcp(sDl) : 0.0, Iu = sDu != 0.0 ? frcp(sDu) : 0.0;
    return soft ? (bRl * El - bGl * Xl) * Il - (bRu * Eu - bGu * Xu) * Iu : bRl - bRu;
}

/* forward sweep: steps of the row's multipliers / slacks for the step dc of the row's value, step length, sums of the
 * Mehrotra centring (affine sweep) or the steps written out (corrector sweep) */
template <bool CORR>
__device__ static inline void w16r_row_fwd(const GqpDev &D, const GqpOpts &O, int inst, bool alive, const W16Dsc S, uint64_t am,
                                           bool has, int row_, int sj_, double dc, double smu, double pscale, double &alpha, double &S0,
                                           double &S1, double &S2, double &nact)
{
    const int nbg = S.nb + S.ng, row = has ? row_ : 0, sj = has ? sj_ : -1;
    const bool al = has && ((am >> row) & 1), au = has && ((am >> (nbg + row)) & 1), soft = sj >= 0;
    const int el = W16R_CLAMP(D.lam, S.o_ct + row), eu = W16R_CLAMP(D.lam, S.o_ct + nbg + row);
    const int sq = soft ? sj : 0, e0 = W16R_CLAMP(D.lam, S.o_ct + 2 * nbg + sq), e1 = W16R_CLAMP(D.lam, S.o_ct + 2 * nbg + S.ns + sq);
    const int q0 = W16R_CLAMP(D.sv, S.o_s + sq), q1 = W16R_CLAMP(D.sv, S.o_s + S.ns + sq);
    const bool sal = soft && ((am >> (2 * nbg + sq)) & 1), sau = soft && ((am >> (2 * nbg + S.ns + sq)) & 1);
    const bool hs = D.sv.E > 0;
    const double v_ll = WAT(D.lam, el), v_lu = WAT(D.lam, eu), v_tl = WAT(D.t, el), v_tu = WAT(D.t, eu);
    const double v_dl = WAT(D.rd, el), v_du = WAT(D.rd, eu);
    const double v_pl = CORR ? WAT(D.pcorr, el) : 0.0, v_pu = CORR ? WAT(D.pcorr, eu) : 0.0;
    double v_sll = 0.0, v_slu = 0.0, v_stl = 1.0, v_stu = 1.0, v_srl = 0.0, v_sru = 0.0, v_spl = 0.0, v_spu = 0.0, rsl = 0.0, rsu = 0.0,
           sZl = 0.0, sZu = 0.0, sDl = 0.0, sDu = 0.0;
    if (hs)
    {
        v_sll = WAT(D.lam, e0); v_slu = WAT(D.lam, e1); v_stl = WAT(D.t, e0); v_stu = WAT(D.t, e1);
        v_srl = WAT(D.rd, e0); v_sru = WAT(D.rd, e1);
        if (CORR) { v_spl = WAT(D.pcorr, e0); v_spu = WAT(D.pcorr, e1); }
        rsl = WAT(D.sR, q0); rsu = WAT(D.sR, q1);
        sZl = WAT(D.Zz, q0 * 2); sZu = WAT(D.Zz, q1 * 2);
        sDl = WAT(D.sD, q0); sDu = WAT(D.sD, q1);
    }
    const double ll = al ? v_ll : 0.0, lu = au ? v_lu : 0.0, ttl = al ? v_tl : 1.0, ttu = au ? v_tu : 1.0;
    const double rdl = al ? v_dl : 0.0, rdu = au ? v_du : 0.0;
    const double pl = al ? v_pl : 0.0, pu = au ? v_pu : 0.0;
    const double rml = al ? ll * ttl - O.tau_min + pscale * pl - smu : 0.0;
    const double rmu = au ? lu * ttu - O.tau_min + pscale * pu - smu : 0.0;
    /* the row's own slack: steps of the slack and of its two bound rows */
    const double sll = sal ? v_sll : 0.0, slu = sau ? v_slu : 0.0, stl = sal ? v_stl : 1.0, stu = sau ? v_stu : 1.0;
    const double srdl = sal ? v_srl : 0.0, srdu = sau ? v_sru : 0.0, spl = sal ? v_spl : 0.0, spu = sau ? v_spu : 0.0;
    const double il = sDl != 0.0 ? frcp(sDl) : 0.0, iu = sDu != 0.0 ? frcp(sDu) : 0.0;
    const double gl_ = ll * frcp(ttl), gu_ = lu * frcp(ttu);
    const double dsl = (-rsl - gl_ * dc) * il, dsu = (-rsu + gu_ * dc) * iu;
    if (CORR && soft && alive) { WAT(D.dsv, q0) = dsl; WAT(D.dsv, q1) = dsu; }
    const double sitl = frcp(stl), situ = frcp(stu);
    const double El = sZl + sll * sitl, Eu = sZu + slu * situ;
    const double dcl = soft ? (El * dc - rsl) * il : dc;   /* dc + ds resp. -dc + ds: the row's own slack step included */
    const double dcu = soft ? (-Eu * dc - rsu) * iu : -dc;
    const double srml = sal ? sll * stl - O.tau_min + pscale * spl - smu : 0.0;
    const double srmu = sau ? slu * stu - O.tau_min + pscale * spu - smu : 0.0;
    const double sdtl = sal ? dsl + srdl : 0.0, sdtu = sau ? dsu + srdu : 0.0;
    const double sdll = sal ? -(srml + sll * sdtl) * sitl : 0.0, sdlu = sau ? -(srmu + slu * sdtu) * situ : 0.0;
    const double q1_ = -sll * frcp(sdll), q2_ = -slu * frcp(sdlu), q3_ = -stl * frcp(sdtl), q4_ = -stu * frcp(sdtu);
    alpha = (sdll < 0.0 && q1_ < alpha) ? q1_ : alpha;
    alpha = (sdlu < 0.0 && q2_ < alpha) ? q2_ : alpha;
    alpha = (sdtl < 0.0 && q3_ < alpha) ? q3_ : alpha;
    alpha = (sdtu < 0.0 && q4_ < alpha) ? q4_ : alpha;
    const double dtl = al ? dcl + rdl : 0.0, dtu = au ? dcu + rdu : 0.0;
    const double dll = al ? -(rml + ll * dtl) * frcp(ttl) : 0.0;
    const double dlu = au ? -(rmu + lu * dtu) * frcp(ttu) : 0.0;
    const double c1 = -ll * frcp(dll), c2 = -lu * frcp(dlu), c3 = -ttl * frcp(dtl), c4 = -ttu * frcp(dtu);
    alpha = (dll < 0.0 && c1 < alpha) ? c1 : alpha;
    alpha = (dlu < 0.0 && c2 < alpha) ? c2 : alpha;
    alpha = (dtl < 0.0 && c3 < alpha) ? c3 : alpha;
    alpha = (dtu < 0.0 && c4 < alpha) ? c4 : alpha;
    /* (the sums in the corrector sweep too: the conditional corrector asks for the duality measure its step ends at) */
    S0 += ll * ttl + lu * ttu + sll * stl + slu * stu;
    S1 += ll * dtl + ttl * dll + lu * dtu + ttu * dlu + sll * sdtl + stl * sdll + slu * sdtu + stu * sdlu;
    S2 += dll * dtl + dlu * dtu + sdll * sdtl + sdlu * sdtu;
    nact += (double) ((int) al + (int) au + (int) sal + (int) sau);
    if (!CORR)
    {
        if (has && alive) { WAT(D.pcorr, el) = dll * dtl; WAT(D.pcorr, eu) = dlu * dtu; }
        if (soft && alive) { WAT(D.pcorr, e0) = sdll * sdtl; WAT(D.pcorr, e1) = sdlu * sdtu; }
    }
    else
    {
        if (has && alive)
        {
            WAT(D.dlam, el) = dll; WAT(D.dlam, eu) = dlu;
            WAT(D.dt, el) = dtl; WAT(D.dt, eu) = dtu;
        }
        if (soft && alive)
        {
            WAT(D.dlam, e0) = sdll; WAT(D.dlam, e1) = sdlu;
            WAT(D.dt, e0) = sdtl; WAT(D.dt, e1) = sdtu;
        }
    }
}

/* update pass of the corrector sweep for one row (and its slack), in two halves: every load of the row -- its two sides,
 * the two slack values, the two slack sides: 20 -- through clamped addresses, and, once the loads of every row the lane
 * handles in this stage are in flight, the stores; a side that does not take part keeps its value (its stores stay behind
 * the predicate).  (With a branch per side the pass made five memory round trips per row and stage, one row after the
 * other: more than half of the C4 corrector sweep.) */
struct W16RowU
{
    int e[4], q0, q1;
    bool act[4], hs;
    double lm[4], dl[4], tt[4], dt[4], v0, d0, v1, d1;
};
__device__ static inline W16RowU w16r_row_update_load(const GqpDev &D, int inst, const W16Dsc S, uint64_t am, bool has, int row, int sj)
{
    W16RowU u;
    const int nbg = S.nb + S.ng, ns = S.ns, o_s = S.o_s;
    const int r = has ? row : 0, sq = (has && sj >= 0) ? sj : 0;
    u.hs = has && sj >= 0;
    u.e[0] = S.o_ct + r; u.e[1] = u.e[0] + nbg; u.e[2] = S.o_ct + 2 * nbg + sq; u.e[3] = u.e[2] + ns;
    u.act[0] = has && ((am >> r) & 1); u.act[1] = has && ((am >> (nbg + r)) & 1);
    u.act[2] = u.hs && ((am >> (2 * nbg + sq)) & 1); u.act[3] = u.hs && ((am >> (2 * nbg + ns + sq)) & 1);
    W16_UNROLL for (int w = 0; w < 4; w++)
    {
        const int ec = W16R_CLAMP(D.lam, u.e[w]);
        u.lm[w] = WAT(D.lam, ec); u.dl[w] = WAT(D.dlam, ec); u.tt[w] = WAT(D.t, ec); u.dt[w] = WAT(D.dt, ec);
    }
    u.q0 = o_s + sq; u.q1 = o_s + ns + sq;
    const int s0 = W16R_CLAMP(D.sv, u.q0), s1 = W16R_CLAMP(D.sv, u.q1);
    u.v0 = WAT(D.sv, s0); u.d0 = WAT(D.dsv, s0); u.v1 = WAT(D.sv, s1); u.d1 = WAT(D.dsv, s1);
    return u;
}
__device__ static inline void w16r_row_update_store(const GqpDev &D, const GqpOpts &O, int inst, const W16RowU &u, double a)
{
    W16_UNROLL for (int w = 0; w < 4; w++)
    {
        const double l2 = u.lm[w] + a * u.dl[w], t2 = u.tt[w] + a * u.dt[w];
        if (u.act[w])
        {
            WAT(D.lam, u.e[w]) = l2 < O.lam_min ? O.lam_min : l2;
            WAT(D.t, u.e[w]) = t2 < O.t_min ? O.t_min : t2;
        }
    }
    if (u.hs)
    {
        WAT(D.sv, u.q0) = u.v0 + a * u.d0;
        WAT(D.sv, u.q1) = u.v1 + a * u.d1;
    }
}

/* ------------------------------------------------------------------------------------------------ factor */

template <int NX, int NU, int NG = 0>
__global__ void __launch_bounds__(64) W16R_WPE_FACT ky_factor(GqpDev D, GqpOpts O, int redo)
{
    GQP_DYN_SHARED(smem);
    typedef W16RLds<NX, NU, NG> LY;
    constexpr bool GEN = NG > 0; /* general rows and slacks (one slack per row): rows through w16r_row_factor */
    constexpr int n = NX + NU, R = LY::R, NP = LY::NP, NB = LY::NB, NGP = (NG * n + 15) / 16;
    int l = threadIdx.x & 15;
    const int rq = threadIdx.x >> 4;
    /* Liveness per 16-lane row.  No row leaves the kernel while another one of the wave is alive: all 64 lanes take part
     * in the LDS-DMA of every live row.  A dead row (converged instance, or beyond the batch) computes on the data of a
     * valid instance and writes nothing. */
    const int inst0 = blockIdx.x * 4;
    bool aq[4], any = false, alive = false;
    int iq[4], inst = 0;
    W16_UNROLL for (int q = 0; q < 4; q++)
    {
        const int ir = w16_slot_inst(D, inst0 + q);
        iq[q] = ir >= 0 ? ir : D.B - 1;
        aq[q] = ir >= 0 && D.status[iq[q]] == GQP_RUNNING;
        any = any || aq[q];
        if (q == rq) { alive = aq[q]; inst = iq[q]; }
    }
    if (!any) return;
    double *T = smem + rq * LY::SZ, *xb = T + LY::XB;
    double *HRq = T + LY::HR; /* packed H block of the stage */
    double *BRq = T + LY::BR; /* [B A]' block of the stage, [row][NX]; later the x-block of the previous factor, [q][NX] */
    double *GTq = T + LY::GT; /* GEN: general rows of the stage, [g][n] */
    double *RW = GTq + LY::RWO; /* GEN: row vectors, 4 x 16 */
    int *RI = (int *) (RW + 64); /* GEN: inequality row handled by every lane */
    int row[R], cx[R];
    bool mine[R], isx[R];
    W16_UNROLL for (int s = 0; s < R; s++) row[s] = l + 16 * s;

    /* What a stage reads from HBM arrives ONE STAGE AHEAD while the O(n^3) phases of the stage before run (one wave per
     * SIMD: nobody else hides the latency).  The two big blocks go straight into LDS by DMA: H as soon as the rows of the
     * current one are in registers, [B A]' as soon as the stage is done with the current one.  The vectors, the box rows
     * of the slots (a slot without a row reads row 0 of the stage: always readable) and the descriptor of the stage after
     * the next land in registers; all of it is waited for once, at the top of the stage. */
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
    double p_v[R], p_g[R], p_b[R], p_xn[R], p_pin[R], p_pik[R], p_ll[R], p_lu[R], p_tl[R], p_tu[R], p_dl[R], p_du[R];
    double p_ht = 0.0, p_bt = 0.0; /* last element of an odd-sized block (the DMA moves pairs) */
    uint64_t p_am, c_bm, c_em, n_bm, n_em; /* activity bits of the next stage; box / equality masks of the current and next one */
    int c_nb, c_oct, n_nb, n_oct, c_ng = 0, c_og = 0, n_ng = 0, n_og = 0, c_ns = 0, c_os = 0, n_ns = 0, n_os = 0;
    double p_G[NGP > 0 ? NGP : 1];
    auto load_desc = [&](int kk)
    {
        GQP_STAGE_REF Sn = D.st[kk];
        n_bm = Sn.bmask; n_em = Sn.emask; n_nb = Sn.nb; n_oct = Sn.o_ct;
        if (GEN) { n_ng = Sn.ng; n_og = Sn.o_g; n_ns = Sn.ns; n_os = Sn.o_s; }
    };
    /* vectors and box rows of stage kk, whose descriptor is in c_* */
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
    if (!GEN) prefetch_v(D.N);
    load_desc(D.N > 0 ? D.N - 1 : 0);

    double Lp[R][NX]; /* rows of the x-block of the factor of stage k+1 held by the state slots, zero above the diagonal */
    double lxn[R];    /* lx+ of this lane's states */
    W16_UNROLL for (int s = 0; s < R; s++)
    {
        lxn[s] = 0.0;
        W16_UNROLL for (int c = 0; c < NX; c++) Lp[s][c] = 0.0;
    }
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
        /* GEN: the same for the instance and the lane index.  Everything derived from them alone -- the lane's base address
         * in each of the ~20 arrays the stage touches, the element indices of the general rows -- is loop invariant, gets
         * hoisted out of the stage loop, lives across its register peak and was spilled: 80 registers, reloaded from
         * scratch one by one in front of the loads they address.  Recomputing them per stage costs a multiply-add each. */
        if (GEN) { W16R_OPAQUE(inst); W16R_OPAQUE(l); }
        /* GEN: the vectors and the general rows of the stage are loaded HERE, in front of the wait for the DMA'd blocks (one
         * exposed latency), not one stage ahead: the register file of this variant has no room for values that live across
         * the stage, and a prefetched value that is spilled on arrival waits alone behind its own load -- thirty serialised
         * memory latencies, 27 k cycles per stage */
        if (GEN) prefetch_v(k);
        W16R_DMA_WAIT(); /* everything issued for this stage has landed */
        const uint64_t bmask = c_bm, emask = c_em, imask = bmask & ~emask, am = p_am;
        const int nbg = c_nb, o_ct = c_oct;
        double M[R][n], v[R], g[R], rb[R], pin[R], pik[R], q_ll[R], q_lu[R], q_tl[R], q_tu[R], q_dl[R], q_du[R];
        bool fixed[R];
        int lc_[R], xc_[R];
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            lc_[s] = mine[s] ? row[s] : 0;
            xc_[s] = isx[s] ? cx[s] : 0;
            fixed[s] = mine[s] && ((emask >> row[s]) & 1);
            const double zm = mine[s] ? 1.0 : 0.0, zx = isx[s] ? 1.0 : 0.0;
            v[s] = zm * p_v[s];
            g[s] = zm * p_g[s];
            rb[s] = zx * (p_b[s] - p_xn[s]);
            pin[s] = zx * p_pin[s];
            pik[s] = zx * p_pik[s];
            if (!GEN) { q_ll[s] = p_ll[s]; q_lu[s] = p_lu[s]; q_tl[s] = p_tl[s]; q_tu[s] = p_tu[s]; q_dl[s] = p_dl[s]; q_du[s] = p_du[s]; }
        }
        const int ng = GEN ? c_ng : 0;
        const W16Dsc dsc = {c_nb, ng, c_ns, c_oct, c_os};
        const int nbf = popc64(imask); /* box rows that take part (equality-flagged ones do not) */
        if (GEN)
        {
            W16_UNROLL for (int i = 0; i < NGP; i++)
            {
                const int e = l + 16 * i;
                if (e < NG * n) GTq[e] = e < ng * n ? p_G[i] : 0.0;
            }
            GQP_ROWSYNC();
        }
        if ((NP & 1) || (NB & 1))
        {
            if (l == 0)
            {
                if (NP & 1) HRq[NP - 1] = p_ht;
                if (NB & 1) BRq[NB - 1] = p_bt;
            }
            GQP_ROWSYNC();
        }
        /* the descriptor loaded a stage ago becomes the next stage's */
        const uint64_t x_bm = n_bm, x_em = n_em;
        const int x_nb = n_nb, x_oct = n_oct, x_ng = n_ng, x_og = n_og, x_ns = n_ns, x_os = n_os;
        /* ---- symmetric row of H from the packed block: H[row][c] = packed[PK(row, c)] for c <= row, packed[PK(c, row)]
         * above.  A slot knows at compile time which one it is except inside its own 16 x 16 diagonal block, where both
         * are read and the lane picks (every access: lane base + immediate offset) ---- */
        auto load_M = [&]()
        {
            W16_UNROLL for (int s = 0; s < R; s++)
            {
                const double zm = mine[s] ? 1.0 : 0.0;
                const double *lowp = HRq + PK(lc_[s], 0), *upp = HRq + lc_[s];
                W16_UNROLL for (int c = 0; c < n; c++)
                {
                    double h;
                    if (c < 16 * s) h = lowp[c];
                    else if (c > 16 * s + 15) h = upp[PK(c, 0)];
                    else if (GEN) h = *(c <= lc_[s] ? lowp + c : upp + PK(c, 0)); /* the lane picks the ADDRESS: one read */
                    else
                    {
                        const double lo = lowp[c < n - 1 ? c : n - 1], up = upp[PK(c, 0)];
                        h = c <= lc_[s] ? lo : up;
                    }
                    M[s][c] = zm * h;
                }
                if (GEN) W16R_FENCE(); /* slot by slot: all reads of both slots in flight at once was the register peak of the stage */
            }
            if (k > 0) dma_h(k - 1);
        };
        /* GEN: the rows of H come AFTER the inequality rows.  The row functions keep ~40 loaded values and as many addresses
         * alive per lane; with the 54 entries of M beside them (and the previous factor's x-block, which must stay) the
         * register file overflowed: 126 spilled registers, and their scratch traffic -- 16,384 instances x 256 B per
         * register -- is HBM traffic (the factor launch moved 2.26x its algorithmic bytes).  H v then takes its own pass of
         * broadcasts of v (n DPP moves), which is nothing beside a spill. */
        if (!GEN) load_M();
        W16R_TICK(0);

        /* ---- rb += [B A] v (column cx of [B A]'), H v: one broadcast of v per variable serves both ---- */
        double hv[R];
        W16_UNROLL for (int s = 0; s < R; s++) hv[s] = 0.0;
        W16_UNROLL for (int r = 0; r < n; r++)
        {
            const double vr = W16R_BC(v, r);
            W16_UNROLL for (int s = 0; s < R; s++)
            {
                rb[s] += BRq[r * NX + xc_[s]] * vr; /* idle slots: clamped column, value unused */
                if (!GEN) hv[s] += M[s][r] * vr;
            }
        }
        W16_UNROLL for (int s = 0; s < R; s++) { W16R_OPAQUE(rb[s]); if (!GEN) W16R_OPAQUE(hv[s]); }
        W16R_TICK(1);
        double gtr[R], gar[R], gmr[R]; /* GEN: what the inequality rows add to the stationarity residual / gradient / Hessian diagonal */
        W16_UNROLL for (int s = 0; s < R; s++) { gtr[s] = 0.0; gar[s] = 0.0; gmr[s] = 0.0; }
        if (GEN)
        {
            /* ONE row per lane: lane i < nb + ng processes inequality row i (sorted box rows, then general rows).  The row
             * values travel through LDS by row index (box rows: written above by the slot that owns the variable; general
             * rows: a'v, a 16-lane sum), the results come back the same way: the slot of a box row picks its row's
             * (gamma, gadd, dlam), a general row enters as rank-one terms M += gamma a a', gradient += a gadd,
             * stationarity residual -= a (lam_l - lam_u) */
            W16_UNROLL for (int s = 0; s < R; s++)
                if (mine[s] && ((imask >> row[s]) & 1)) /* value and row index of the slot's box row, to the lane that processes it */
                {
                    const int jb = popc64(imask & (((uint64_t) 1 << row[s]) - 1));
                    RW[jb] = v[s];
                    RI[jb] = popc64(bmask & (((uint64_t) 1 << row[s]) - 1));
                }
            W16_UNROLL for (int g = 0; g < NG; g++)
            {
                double t = 0.0;
                W16_UNROLL for (int s = 0; s < R; s++) t += (mine[s] ? GTq[g * n + lc_[s]] : 0.0) * v[s];
                t = w16_rowsum(t, xb);
                if (l == g && g < ng) { RW[nbf + g] = t; RI[nbf + g] = nbg + g; }
            }
            W16R_TICK(7);
            GQP_ROWSYNC();
            const bool hr = l < nbf + ng;
            const int rr = hr ? RI[l] : 0;
            const int sjr = hr ? (int) D.st[k].srev[rr] : -1;
            const W16RowF rf = w16r_row_factor(D, O, inst, alive, dsc, am, hr, rr, sjr, RW[hr ? l : 0], nrm_g, nrm_d, nrm_m, musum, nact, obj);
            W16R_TICK(8);
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
            W16_UNROLL for (int g = 0; g < NG; g++)
            {
                const int rg_ = nbf + g < 16 ? nbf + g : 15;
                const double Ag = g < ng ? RW[32 + rg_] : 0.0, Lg = g < ng ? RW[48 + rg_] : 0.0;
                W16_UNROLL for (int s = 0; s < R; s++)
                {
                    const double ap = mine[s] ? GTq[g * n + lc_[s]] : 0.0;
                    gtr[s] += ap * Lg;
                    gar[s] += ap * Ag;
                }
            }
            W16_UNROLL for (int s = 0; s < R; s++) { W16R_OPAQUE(gtr[s]); W16R_OPAQUE(gar[s]); W16R_OPAQUE(gmr[s]); }
            W16R_TICK(9);
            /* now the rows of H, H v, and the general rows' rank-one terms M += gamma a a' (gamma of row g still sits in the
             * row vector, a in the LDS copy of [D C]) */
            load_M();
            W16_UNROLL for (int r = 0; r < n; r++)
            {
                const double vr = W16R_BC(v, r);
                W16_UNROLL for (int s = 0; s < R; s++) hv[s] += M[s][r] * vr;
            }
            W16_UNROLL for (int s = 0; s < R; s++) W16R_OPAQUE(hv[s]);
            int zg = 0; /* always zero; laundered through the previous row's result: the reads of row g + 1 depend on it */
            W16_UNROLL for (int g = 0; g < NG; g++)
            {
                /* one general row at a time: hoisted together, the NG x n reads of [D C] were the register peak of the stage */
                const double *Gr = GTq + zg + g * n;
                const int rg_ = nbf + g < 16 ? nbf + g : 15;
                const double Gg = g < ng ? RW[16 + rg_] : 0.0;
                W16_UNROLL for (int s = 0; s < R; s++)
                {
                    const double ga = Gg * (mine[s] ? Gr[lc_[s]] : 0.0);
                    W16_UNROLL for (int c = 0; c < n; c++)
                        if (W16R_LOW(s, c)) M[s][c] += ga * Gr[c];
                }
                /* every multiply-add of this row in place before the next row's reads are issued */
                W16_UNROLL for (int s = 0; s < R; s++)
                    W16_UNROLL for (int c = 0; c < n; c++)
                        if (W16R_LOW(s, c)) W16R_OPAQUE(M[s][c]);
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" : "+v"(zg) : "v"(M[R - 1][0]));
#endif
            }
            W16_UNROLL for (int s = 0; s < R; s++)
                W16_UNROLL for (int c = 0; c < n; c++)
                    if (W16R_LOW(s, c)) W16R_OPAQUE(M[s][c]);
        }
        /* ---- W rows: W[c] = sum_{q >= c} Br[q] Lx+[q][c]: Lx+[q][c] is entry c of the register row of the slot that
         * holds state q, one broadcast feeds both slots; [B A]' pi+ rides along ---- */
        double W[R][NX], gt[R], gadd[R], gam[R];
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            gt[s] = 0.0; gadd[s] = 0.0; gam[s] = 0.0;
            W16_UNROLL for (int c = 0; c < NX; c++) W[s][c] = 0.0;
        }
        double brn[R]; /* row entries of [B A]' one column ahead of their use */
        W16_UNROLL for (int s = 0; s < R; s++) brn[s] = BRq[lc_[s] * NX];
        W16_UNROLL for (int q = 0; q < NX; q++)
        {
            /* the scheduler must not pull the broadcasts of later columns up here: they are all ready at the top of the
             * phase, and hoisted they spill */
            W16R_FENCE();
            double brq[R];
            W16_UNROLL for (int s = 0; s < R; s++)
            {
                brq[s] = mine[s] ? brn[s] : 0.0;
                brn[s] = BRq[lc_[s] * NX + (q + 1 < NX ? q + 1 : q)];
            }
            const double pc = W16R_BC(pin, NU + q);
            W16_UNROLL for (int s = 0; s < R; s++) gt[s] += brq[s] * pc;
            W16_UNROLL for (int c = 0; c <= q; c++)
            {
                const double Lqc = w16_bcast(Lp[(NU + q) >> 4][c], (NU + q) & 15, xb);
                W16_UNROLL for (int s = 0; s < R; s++) W[s][c] += brq[s] * Lqc;
            }
        }
        /* The accumulators of the phase are pinned at its end.  Their readers sit in later basic blocks (behind the box-row
         * branches): machine sinking would move every multiply-add chain down there, while the broadcasts feeding them
         * (convergent) stay up here -- three hundred values alive across the gap, spilled to scratch. */
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            W16R_OPAQUE(gt[s]);
            W16_UNROLL for (int c = 0; c < NX; c++) W16R_OPAQUE(W[s][c]);
        }
        W16R_TICK(2);
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            if (mine[s])
            {
                obj += (0.5 * hv[s] + g[s]) * v[s];
                gt[s] += hv[s] + g[s] - pik[s];
            }
            else gt[s] = 0.0;
            if (isx[s]) { nacc(nrm_b, rb[s]); if (alive) WAT(D.rb, k * NX + cx[s]) = rb[s]; }
            const bool has = mine[s] && ((imask >> row[s]) & 1);
            if (GEN) { }
            else if (has)
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
            if (!GEN)
            {
                if (fixed[s]) gt[s] = 0.0;
                if (mine[s]) { nacc(nrm_g, gt[s]); if (alive) WAT(D.rg, k * n + row[s]) = gt[s]; }
            }
        }
        if (GEN)
        {
            W16_UNROLL for (int s = 0; s < R; s++)
            {
                gt[s] -= gtr[s];
                gadd[s] = gar[s]; gam[s] = gmr[s];
                if (fixed[s]) gt[s] = 0.0;
                if (mine[s]) { nacc(nrm_g, gt[s]); if (alive) WAT(D.rg, k * n + row[s]) = gt[s]; }
            }
        }
        W16R_TICK(10);
        /* w0[c] (state slots) = lx+[c] + sum_{q >= c} Lx+[q][c] rb[q] needs COLUMN c of Lx+: the rows go through the
         * [B A]' region (the stage is done with it), the slots read their column back */
        GQP_ROWSYNC();
        W16_UNROLL for (int s = 0; s < R; s++)
            if (isx[s])
            {
                W16_UNROLL for (int c = 0; c < NX; c++) BRq[cx[s] * NX + c] = Lp[s][c];
            }
        GQP_ROWSYNC();
        double w0[R];
        W16_UNROLL for (int s = 0; s < R; s++) w0[s] = lxn[s];
        W16_UNROLL for (int q = 0; q < NX; q++)
        {
            const double rbq = W16R_BC(rb, NU + q);
            W16_UNROLL for (int s = 0; s < R; s++) w0[s] += BRq[q * NX + xc_[s]] * rbq; /* zero above the diagonal */
        }
        W16_UNROLL for (int s = 0; s < R; s++)
            if (!isx[s]) w0[s] = 0.0;
        W16R_TICK(11);
        if (k > 0)
        {
            /* next stage: its [B A]' block by DMA; vectors, box rows and the descriptor after it into registers */
            dma_b(k - 1);
            W16R_TICK(13);
            c_bm = x_bm; c_em = x_em; c_nb = x_nb; c_oct = x_oct; c_ng = x_ng; c_og = x_og; c_ns = x_ns; c_os = x_os;
            if (!GEN) prefetch_v(k - 1); /* GEN: at the top of the stage, see there */
            W16R_TICK(14);
            load_desc(k > 1 ? k - 2 : 0);
        }
        W16R_TICK(12);
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
        W16_UNROLL for (int s = 0; s < R; s++) W16R_OPAQUE(m[s]);
        W16R_TICK(3);
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
        W16_UNROLL for (int s = 0; s < R; s++)
            W16_UNROLL for (int c = 0; c < n; c++)
                if (W16R_LOW(s, c)) W16R_OPAQUE(M[s][c]);
        if (emask) /* uniform: only a stage with fixed variables pays for the masking */
        {
            W16_UNROLL for (int s = 0; s < R; s++)
                W16_UNROLL for (int c = 0; c < n; c++)
                {
                    const bool fc = (emask >> c) & 1;
                    if (fixed[s] || fc) M[s][c] = (c == row[s]) ? 1.0 : 0.0;
                }
        }
        W16R_TICK(4);

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
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            W16R_OPAQUE(m[s]);
            W16_UNROLL for (int c = 0; c < n; c++)
                if (W16R_LOW(s, c)) W16R_OPAQUE(M[s][c]);
        }
        W16R_TICK(5);

        /* ---- outputs: the rows of the factor from registers (both LDS regions are being filled for the next stage) ---- */
        if (GEN) W16R_OPAQUE(inst); /* the store addresses are formed here, not at the top of the stage (see there) */
        W16_UNROLL for (int s = 0; s < R; s++)
            if (mine[s] && alive)
            {
                W16_UNROLL for (int c = 0; c < n; c++)
                    if (W16R_LOW(s, c) && c <= row[s]) WAT(D.Lf, k * NP + PK(row[s], c)) = M[s][c];
                WAT(D.lf, k * n + row[s]) = m[s];
            }
        /* x-block for the next (earlier) stage: register rows of the state slots */
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            W16_UNROLL for (int c = 0; c < NX; c++)
                Lp[s][c] = (W16R_LOW(s, NU + c) && isx[s] && c <= cx[s]) ? M[s][NU + c] : 0.0;
            lxn[s] = isx[s] ? m[s] : 0.0;
        }
        W16R_TICK(6);
    }

    nrm_g = w16_rmax(nrm_g, xb); nrm_b = w16_rmax(nrm_b, xb); nrm_d = w16_rmax(nrm_d, xb); nrm_m = w16_rmax(nrm_m, xb);
    musum = w16_rsum(musum, xb); obj = w16_rsum(obj, xb);
    const double nact_d = w16_rsum(nact, xb);
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

/* ------------------------------------------------------------------------------- rhs-only backward (p-form) */

/* PF (shapes whose rows fit the register file twice, e.g. the condensed C3 shape): everything a stage reads is loaded one
 * stage ahead -- one wave per SIMD, nobody else hides the latency */
template <int NX, int NU, int NG = 0>
__global__ void __launch_bounds__(64) ky_backrhs(GqpDev D, GqpOpts O, int redo)
{
    GQP_DYN_SHARED(smem);
    typedef W16RLds<NX, NU, NG> LY;
    constexpr bool GEN = NG > 0;
    constexpr int n = NX + NU, R = LY::R, NP = n * (n + 1) / 2, LDX = LY::LDX, NGP = (NG * n + 15) / 16;
    constexpr bool PF = !GEN && R * (NU + 2 * NX + 12) <= 96;
    const int l = threadIdx.x & 15, inst = w16_slot_inst(D, blockIdx.x * 4 + (threadIdx.x >> 4));
    if (inst < 0) return;
    if (D.status[inst] != GQP_RUNNING) return;
    if (redo == 1 && !(D.alpha[inst] < 0.0)) return; /* redo = 2: sensitivity pass (direction only, every instance) */
    double *T = smem + (threadIdx.x >> 4) * LY::SZ, *xb = T + LY::XB, *TA = T + LY::TA, *GTq = T + LY::GTA, *RW = GTq + LY::RWO;
    int *RI = (int *) (RW + 64);
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

    /* loads of one stage: row of the factor (first NU columns; x-block for state slots), row of [B A]', vectors, the
     * slot's box row (a slot without a row reads row 0 of the stage: always readable) */
    struct StageRegs
    {
        double Lu[R][NU > 0 ? NU : 1], Lx[R][NX], Br[R][NX], rb[R], rg[R], ll[R], lu[R], tl[R], tu[R], dl[R], du[R], pl[R], pu[R];
        double gt[NGP > 0 ? NGP : 1]; /* GEN: the general rows of the stage, flat over the 16 lanes */
        uint64_t am, bm, em;
        int nb, oct;
    };
    /* GEN: the stage descriptor one stage ahead in registers, so that everything a stage reads can be issued at once */
    struct Desc
    {
        uint64_t bm, em;
        int nb, oct, ng, ns, os, og;
    };
    auto load_desc = [&](int k, Desc &d)
    {
        GQP_STAGE_REF S = D.st[k];
        d.bm = S.bmask; d.em = S.emask; d.nb = S.nb; d.oct = S.o_ct; d.ng = S.ng; d.ns = S.ns; d.os = S.o_s; d.og = S.o_g;
    };
    Desc cd = {}, nd = {};
    if (GEN) load_desc(D.N, cd);
    auto load = [&](int k, StageRegs &G)
    {
        GQP_STAGE_REF S = D.st[k];
        if (GEN)
        {
            G.bm = cd.bm; G.em = cd.em; G.nb = cd.nb; G.oct = cd.oct;
            W16_UNROLL for (int i = 0; i < NGP; i++)
            {
                const int e = l + 16 * i;
                G.gt[i] = WAT(D.DCt, cd.og * n + (e < cd.ng * n ? e : 0));
            }
        }
        else { G.bm = S.bmask; G.em = S.emask; G.nb = S.nb; G.oct = S.o_ct; }
        G.am = WAT(D.amask, k * D.AW);
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            const int lc_ = mine[s] ? row[s] : 0, xl_ = isx[s] ? row[s] : NU, xc_ = isx[s] ? cx[s] : 0;
            W16_UNROLL for (int c = 0; c < NU; c++) G.Lu[s][c] = WAT(D.Lf, k * NP + PK(lc_, (c <= lc_ ? c : 0)));
            W16_UNROLL for (int c = 0; c < NX; c++) G.Lx[s][c] = WAT(D.Lf, k * NP + PK(xl_, NU + (c <= xc_ ? c : 0)));
            W16_UNROLL for (int c = 0; c < NX; c++) G.Br[s][c] = WAT(D.BAt, (k * n + lc_) * NX + c);
            G.rb[s] = WAT(D.rb, k * NX + xc_);
            G.rg[s] = WAT(D.rg, k * n + lc_);
            if (!GEN)
            {
                const bool hs = mine[s] && (((S.bmask & ~S.emask) >> row[s]) & 1);
                const int ib = hs ? popc64(S.bmask & (((uint64_t) 1 << row[s]) - 1)) : 0;
                const int el = S.o_ct + ib, eu = el + S.nb;
                G.ll[s] = WAT(D.lam, el); G.lu[s] = WAT(D.lam, eu);
                G.tl[s] = WAT(D.t, el); G.tu[s] = WAT(D.t, eu);
                G.dl[s] = WAT(D.rd, el); G.du[s] = WAT(D.rd, eu);
                G.pl[s] = WAT(D.pcorr, el); G.pu[s] = WAT(D.pcorr, eu);
            }
        }
    };
    StageRegs Gn;
    if (PF) load(D.N, Gn);

    for (int k = D.N; k >= 0; k--)
    {
        StageRegs G;
        if (PF)
        {
            G = Gn;
            if (k > 0) load(k - 1, Gn);
        }
        else load(k, G);
        if (GEN && k > 0) load_desc(k - 1, nd);
        const uint64_t imask = G.bm & ~G.em, am = G.am;
        const int nbg = G.nb;
        double Lu[R][NU > 0 ? NU : 1], Lx[R][NX], Br[R][NX], rb[R], m[R];
        bool fixed[R];
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            const int lc_ = mine[s] ? row[s] : 0, xc_ = isx[s] ? cx[s] : 0;
            const double zm = mine[s] ? 1.0 : 0.0, zx = isx[s] ? 1.0 : 0.0;
            fixed[s] = mine[s] && ((G.em >> row[s]) & 1);
            W16_UNROLL for (int c = 0; c < NU; c++) Lu[s][c] = (c <= lc_ ? zm : 0.0) * G.Lu[s][c];
            W16_UNROLL for (int c = 0; c < NX; c++) Lx[s][c] = (c <= xc_ ? zx : 0.0) * G.Lx[s][c];
            W16_UNROLL for (int c = 0; c < NX; c++) Br[s][c] = zm * G.Br[s][c];
            rb[s] = zx * G.rb[s];
            m[s] = zm * G.rg[s];
            const bool has = mine[s] && ((imask >> row[s]) & 1);
            if (GEN) { }
            else if (has)
            {
                const int ib = popc64(G.bm & (((uint64_t) 1 << row[s]) - 1));
                const bool al = (am >> ib) & 1, au = (am >> (nbg + ib)) & 1;
                const double ll = al ? G.ll[s] : 0.0, lu = au ? G.lu[s] : 0.0;
                const double ttl = al ? G.tl[s] : 1.0, ttu = au ? G.tu[s] : 1.0;
                const double rdl = al ? G.dl[s] : 0.0, rdu = au ? G.du[s] : 0.0;
                const double rml = al ? ll * ttl - O.tau_min + pscale * G.pl[s] - smu : 0.0;
                const double rmu = au ? lu * ttu - O.tau_min + pscale * G.pu[s] - smu : 0.0;
                m[s] += (rml + ll * rdl) * frcp(ttl) - (rmu + lu * rdu) * frcp(ttu);
            }
        }
        if (GEN)
        {
            /* general rows: gradient term of row g from lane g, applied as a gadd through the rows of [D C] */
            GQP_STAGE_REF S = D.st[k];
            const int ng = cd.ng;
            GQP_ROWSYNC();
            W16_UNROLL for (int i = 0; i < NGP; i++)
            {
                const int e = l + 16 * i;
                if (e < NG * n) GTq[e] = e < ng * n ? G.gt[i] : 0.0;
            }
            GQP_ROWSYNC();
            /* one inequality row per lane (sorted box rows, then general rows); results back through LDS by row index */
            const W16Dsc dsc = {cd.nb, cd.ng, cd.ns, cd.oct, cd.os};
            const int nbf = popc64(imask);
            W16_UNROLL for (int s = 0; s < R; s++)
                if (mine[s] && ((imask >> row[s]) & 1))
                    RI[popc64(imask & (((uint64_t) 1 << row[s]) - 1))] = popc64(G.bm & (((uint64_t) 1 << row[s]) - 1));
            if (l < ng) RI[nbf + l] = cd.nb + l;
            GQP_ROWSYNC();
            const bool hr = l < nbf + ng;
            const int rr = hr ? RI[l] : 0;
            RW[32 + l] = w16r_row_rhs(D, O, inst, true, dsc, am, hr, rr, hr ? (int) S.srev[rr] : -1, smu, pscale);
            GQP_ROWSYNC();
            W16_UNROLL for (int s = 0; s < R; s++)
            {
                const bool has = mine[s] && ((imask >> row[s]) & 1);
                m[s] += has ? RW[32 + popc64(imask & (((uint64_t) 1 << row[s]) - 1))] : 0.0;
            }
            W16_UNROLL for (int g = 0; g < NG; g++)
            {
                const int rg_ = nbf + g < 16 ? nbf + g : 15;
                const double Ag = g < ng ? RW[32 + rg_] : 0.0;
                W16_UNROLL for (int s = 0; s < R; s++) m[s] += (mine[s] ? GTq[g * n + (mine[s] ? row[s] : 0)] : 0.0) * Ag;
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
            const double d = w16_bcast(Lu[j >> 4][j], j & 15, xb), mj = W16R_BC(m, j);
            const double lj = d != 0.0 ? mj * frcp(d) : 0.0;
            W16_UNROLL for (int s = 0; s < R; s++) m[s] = row[s] == j ? lj : (row[s] > j ? m[s] - Lu[s][j] * lj : m[s]);
        }
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            W16R_OPAQUE(m[s]);
            if (mine[s]) WAT(D.lf, k * n + row[s]) = m[s];
        }
        /* this stage's x-block becomes "the stage handled before" */
        GQP_ROWSYNC();
        W16_UNROLL for (int s = 0; s < R; s++)
            if (isx[s])
            {
                W16_UNROLL for (int c = 0; c < NX; c++) TA[cx[s] * LDX + c] = Lx[s][c];
            }
        GQP_ROWSYNC();
        W16_UNROLL for (int s = 0; s < R; s++) pn[s] = isx[s] ? m[s] : 0.0;
        if (GEN) cd = nd;
    }
}

/* --------------------------------------------------------------------------------------------------- forward */

/* PFORM (= CORR): lf holds [l_u; p] (written by ky_backrhs), otherwise the plain l of the factor sweep.
 * The packed factor and the [B A]' block of a stage arrive in LDS by DMA (two buffers where they fit: the next stage lands
 * while this one computes); rows AND columns of the factor are read from the packed block where they are used -- no
 * register copy, no transposition tile.  Vectors and box rows are loaded one stage ahead into registers.  As in the factor
 * sweep no row leaves while another row of the wave is alive (the DMA needs all 64 lanes); a dead row writes nothing. */
template <int NX, int NU, bool CORR, int NG = 0>
__global__ void __launch_bounds__(64) W16R_WPE_FWD ky_fwd(GqpDev D, GqpOpts O, int redo)
{
    GQP_DYN_SHARED(smem);
    typedef W16RLds<NX, NU, NG> LY;
    constexpr bool PFORM = CORR, GEN = NG > 0;
    constexpr int n = NX + NU, R = LY::R, NP = LY::NP, NB = LY::NB, NBUF = LY::NBUF, NGP = (NG * n + 15) / 16;
    const int l = threadIdx.x & 15, rq = threadIdx.x >> 4;
    const int inst0 = blockIdx.x * 4;
    bool aq[4], any = false, alive = false;
    int iq[4], inst = 0;
    W16_UNROLL for (int q = 0; q < 4; q++)
    {
        const int ir = w16_slot_inst(D, inst0 + q);
        iq[q] = ir >= 0 ? ir : D.B - 1;
        aq[q] = ir >= 0 && D.status[iq[q]] == GQP_RUNNING && !(redo == 1 && !(D.alpha[iq[q]] < 0.0));
        any = any || aq[q];
        if (q == rq) { alive = aq[q]; inst = iq[q]; }
    }
    if (!any) return; /* redo = 1: only the instances whose corrector step was rejected; redo = 2: sensitivity pass */
    double *T = smem + rq * LY::SZ, *xb = T + LY::XB, *GTq = T + LY::GTF, *RW = GTq + LY::RWO;
    int *RI = (int *) (RW + 64);
    int row[R], cx[R];
    bool mine[R], isx[R];
    W16_UNROLL for (int s = 0; s < R; s++) row[s] = l + 16 * s;
    const double smu = CORR ? D.smu[inst] : 0.0;
    const double pscale = (CORR && redo != 1) ? 1.0 : 0.0;
    double alpha = 1.0, S0 = 0.0, S1 = 0.0, S2 = 0.0, nact = 0.0;
    double dx[R]; /* dx of this lane's states for the stage being entered */
    W16_UNROLL for (int s = 0; s < R; s++) dx[s] = 0.0;

    auto dma = [&](int kk)
    {
        W16R_LDS_DRAIN();
        const int off = 16 + (NBUF == 2 ? (kk & 1) * LY::FBUF : 0);
        W16_UNROLL for (int q = 0; q < 4; q++)
            if (aq[q])
            {
                w16r_dma_region<NP / 2>(D.Lf.p + (size_t) iq[q] * (size_t) D.Lf.E + (size_t) kk * NP, smem + q * LY::SZ + off);
                w16r_dma_region<NB / 2>(D.BAt.p + (size_t) iq[q] * (size_t) D.BAt.E + (size_t) kk * NB, smem + q * LY::SZ + off + LY::HSZ);
            }
    };
    double p_lv[R], p_rb[R], p_ll[R], p_lu[R], p_tl[R], p_tu[R], p_dl[R], p_du[R], p_pl[R], p_pu[R];
    double p_ft = 0.0, p_bt = 0.0;
    uint64_t p_am, c_bm, c_em, n_bm, n_em;
    int c_nb, c_oct, n_nb, n_oct, c_ng = 0, c_ns = 0, c_os = 0, c_og = 0, n_ng = 0, n_ns = 0, n_os = 0, n_og = 0;
    auto load_desc = [&](int kk)
    {
        GQP_STAGE_REF Sn = D.st[kk];
        n_bm = Sn.bmask; n_em = Sn.emask; n_nb = Sn.nb; n_oct = Sn.o_ct;
        if (GEN) { n_ng = Sn.ng; n_ns = Sn.ns; n_os = Sn.o_s; n_og = Sn.o_g; }
#if defined(GQP_STAGE_VECTOR_FULLWAIT) && defined(__HIP_DEVICE_COMPILE__)
        /* development build (with GQP_STAGE_VECTOR_LOADS): the vector-loaded descriptor complete before anything else is issued */
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(n_nb), "+v"(n_oct), "+v"(n_bm), "+v"(n_em) : : "memory");
#endif
    };
    auto prefetch_v = [&](int kk) /* descriptor of stage kk in c_* */
    {
        p_am = WAT(D.amask, kk * D.AW);
        if (NP & 1) p_ft = WAT(D.Lf, kk * NP + NP - 1);
        if (NB & 1) p_bt = WAT(D.BAt, kk * NB + NB - 1);
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            const bool mn = row[s] < n, ix = row[s] >= NU && row[s] < n;
            const int lc = mn ? row[s] : 0, xc = ix ? row[s] - NU : 0;
            p_lv[s] = WAT(D.lf, kk * n + lc);
            p_rb[s] = WAT(D.rb, kk * NX + xc);
            if (!GEN)
            {
                const bool hs = mn && (((c_bm & ~c_em) >> row[s]) & 1);
                const int ib = hs ? popc64(c_bm & (((uint64_t) 1 << row[s]) - 1)) : 0;
                const int el = c_oct + ib, eu = el + c_nb;
                p_ll[s] = WAT(D.lam, el); p_lu[s] = WAT(D.lam, eu);
                p_tl[s] = WAT(D.t, el); p_tu[s] = WAT(D.t, eu);
                p_dl[s] = WAT(D.rd, el); p_du[s] = WAT(D.rd, eu);
                p_pl[s] = CORR ? WAT(D.pcorr, el) : 0.0; p_pu[s] = CORR ? WAT(D.pcorr, eu) : 0.0;
            }
        }
    };
    load_desc(0);
    c_bm = n_bm; c_em = n_em; c_nb = n_nb; c_oct = n_oct; c_ng = n_ng; c_ns = n_ns; c_os = n_os; c_og = n_og;
    dma(0);
    prefetch_v(0);
    load_desc(D.N > 0 ? 1 : 0);

    for (int k = 0; k <= D.N; k++)
    {
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            W16R_OPAQUE(row[s]);
            mine[s] = row[s] < n;
            isx[s] = row[s] >= NU && row[s] < n;
            cx[s] = isx[s] ? row[s] - NU : 0;
        }
        /* GEN: the general rows of the stage, issued in front of the wait for the DMA'd blocks (descriptor in registers) */
        double gtl[NGP > 0 ? NGP : 1];
        if (GEN)
        {
            W16_UNROLL for (int i = 0; i < NGP; i++)
            {
                const int e = l + 16 * i;
                gtl[i] = WAT(D.DCt, c_og * n + (e < c_ng * n ? e : 0));
            }
        }
        W16R_DMA_WAIT();
        double *LF = T + 16 + (NBUF == 2 ? (k & 1) * LY::FBUF : 0); /* packed factor of the stage */
        double *BRq = LF + LY::HSZ;                                 /* [B A]' of the stage, [row][NX] */
        const uint64_t bmask = c_bm, emask = c_em, imask = bmask & ~emask, am = p_am;
        const int nbg = c_nb, o_ct = c_oct;
        double lv[R], rbv[R], q_ll[R], q_lu[R], q_tl[R], q_tu[R], q_dl[R], q_du[R], q_pl[R], q_pu[R];
        int lc_[R], xc_[R];
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            lc_[s] = mine[s] ? row[s] : 0;
            xc_[s] = isx[s] ? cx[s] : 0;
            lv[s] = mine[s] ? p_lv[s] : 0.0;
            rbv[s] = isx[s] ? p_rb[s] : 0.0;
            if (!GEN)
            {
                q_ll[s] = p_ll[s]; q_lu[s] = p_lu[s]; q_tl[s] = p_tl[s]; q_tu[s] = p_tu[s];
                q_dl[s] = p_dl[s]; q_du[s] = p_du[s]; q_pl[s] = p_pl[s]; q_pu[s] = p_pu[s];
            }
        }
        W16Dsc gdsc = {c_nb, 0, 0, c_oct, 0};
        const int nbf = popc64(imask);
        if (GEN)
        {
            /* general rows of the stage (read again below: a'dv), descriptor fields and slack indices of the rows */
            gdsc.ng = c_ng; gdsc.ns = c_ns; gdsc.o_s = c_os;
            W16_UNROLL for (int i = 0; i < NGP; i++)
            {
                const int e = l + 16 * i;
                if (e < NG * n) GTq[e] = e < c_ng * n ? gtl[i] : 0.0;
            }
            GQP_ROWSYNC();
        }
        if ((NP & 1) || (NB & 1))
        {
            if (l == 0)
            {
                if (NP & 1) LF[NP - 1] = p_ft;
                if (NB & 1) BRq[NB - 1] = p_bt;
            }
            GQP_ROWSYNC();
        }
        if (k < D.N)
        {
            if (NBUF == 2) dma(k + 1);
            c_bm = n_bm; c_em = n_em; c_nb = n_nb; c_oct = n_oct; c_ng = n_ng; c_ns = n_ns; c_os = n_os; c_og = n_og;
            prefetch_v(k + 1);
            load_desc(k + 2 <= D.N ? k + 2 : D.N);
        }
        /* entries of the packed factor: row of this slot (zero above the diagonal and for idle slots) and column of
         * this slot's variable (zero above the diagonal) */
#define W16R_LR(s, c) ((mine[s] && (c) <= row[s]) ? LF[PK(lc_[s], 0) + (c)] : 0.0)
#define W16R_LC(s, r) ((mine[s] && (r) >= row[s]) ? LF[PK((r), 0) + lc_[s]] : 0.0)
        /* reciprocal of the diagonal entry of this slot's row (0 for a zero pivot): every slot computes its own once, the
         * substitutions below broadcast it -- no reciprocal in their dependent chains */
        double dg[R];
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            const double d = mine[s] ? W16R_LR(s, lc_[s]) : 0.0;
            dg[s] = d != 0.0 ? frcp(d) : 0.0;
        }

        if (PFORM && k == 0)
        {
            /* the states of stage 0 are free: recover l_x = Lx^{-1} p */
            W16_UNROLL for (int j = NU; j < n; j++)
            {
                const double lj = W16R_BC(lv, j) * W16R_BC(dg, j);
                W16_UNROLL for (int s = 0; s < R; s++)
                    if (W16R_LOW(s, j)) lv[s] = row[s] == j ? lj : (row[s] > j ? lv[s] - W16R_LR(s, j) * lj : lv[s]);
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
                W16_UNROLL for (int s = 0; s < R; s++)
                    if (W16R_LOW(s, NU + c)) a[s] += W16R_LR(s, NU + c) * wc;
            }
            W16_UNROLL for (int s = 0; s < R; s++)
                if (isx[s] && alive) WAT(D.dpi, k * NX + cx[s]) = a[s];
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
            const double dvr = W16R_BC(acc, r) * W16R_BC(dg, r);
            W16_UNROLL for (int s = 0; s < R; s++)
            {
                if (row[s] == r) dv[s] = dvr;
                else if (row[s] < r) acc[s] -= W16R_LC(s, r) * dvr;
            }
        }
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            if (!mine[s]) dv[s] = 0.0;
            if (CORR && mine[s] && alive) WAT(D.dux, k * n + row[s]) = dv[s];
        }
        /* dx of the next stage */
        double dxn[R];
        W16_UNROLL for (int s = 0; s < R; s++) dxn[s] = rbv[s];
        W16_UNROLL for (int r = 0; r < n; r++)
        {
            const double dr = W16R_BC(dv, r);
            W16_UNROLL for (int s = 0; s < R; s++) dxn[s] += BRq[r * NX + xc_[s]] * dr; /* idle slots: clamped column, value unused */
        }
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            const bool has = mine[s] && ((imask >> row[s]) & 1);
            if (GEN)
            {
                if (has) /* step and row index of the slot's box row, to the lane that processes it */
                {
                    const int jb = popc64(imask & (((uint64_t) 1 << row[s]) - 1));
                    RW[jb] = dv[s];
                    RI[jb] = popc64(bmask & (((uint64_t) 1 << row[s]) - 1));
                }
            }
            else if (has)
            {
                const int ib = popc64(bmask & (((uint64_t) 1 << row[s]) - 1));
                const bool al = (am >> ib) & 1, au = (am >> (nbg + ib)) & 1;
                const int el = o_ct + ib, eu = el + nbg;
                const double ll = al ? q_ll[s] : 0.0, lu = au ? q_lu[s] : 0.0;
                const double ttl = al ? q_tl[s] : 1.0, ttu = au ? q_tu[s] : 1.0;
                const double rdl = al ? q_dl[s] : 0.0, rdu = au ? q_du[s] : 0.0;
                const double pl = (CORR && al) ? q_pl[s] : 0.0, pu = (CORR && au) ? q_pu[s] : 0.0;
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
                S0 += ll * ttl + lu * ttu;
                S1 += ll * dtl + ttl * dll + lu * dtu + ttu * dlu;
                S2 += dll * dtl + dlu * dtu;
                nact += (double) ((int) al + (int) au);
                if (!CORR)
                {
                    if (alive)
                    {
                        WAT(D.pcorr, el) = dll * dtl;
                        WAT(D.pcorr, eu) = dlu * dtu;
                    }
                }
                else if (alive)
                {
                    WAT(D.dlam, el) = dll; WAT(D.dlam, eu) = dlu;
                    WAT(D.dt, el) = dtl; WAT(D.dt, eu) = dtu;
                }
            }
        }
        if (GEN)
        {
            /* one inequality row per lane: lane i processes row i (sorted box rows, then general rows) with the step of the
             * row's value from LDS (box rows: written above; general rows: a'dv, a 16-lane sum) */
            W16_UNROLL for (int g = 0; g < NG; g++)
            {
                double t = 0.0;
                W16_UNROLL for (int s = 0; s < R; s++) t += (mine[s] ? GTq[g * n + lc_[s]] : 0.0) * dv[s];
                t = w16_rowsum(t, xb);
                if (l == g && g < gdsc.ng) { RW[nbf + g] = t; RI[nbf + g] = gdsc.nb + g; }
            }
            GQP_ROWSYNC();
            const bool hr = l < nbf + gdsc.ng;
            const int rr = hr ? RI[l] : 0;
            const int sjr = hr ? (int) D.st[k].srev[rr] : -1;
            w16r_row_fwd<CORR>(D, O, inst, alive, gdsc, am, hr, rr, sjr, RW[hr ? l : 0], smu, pscale, alpha, S0, S1, S2, nact);
            GQP_ROWSYNC();
        }
        W16_UNROLL for (int s = 0; s < R; s++)
        {
            dx[s] = isx[s] ? dxn[s] : 0.0;
            W16R_OPAQUE(dx[s]);
        }
        W16R_OPAQUE(alpha);
        if (NBUF == 1 && k < D.N) dma(k + 1);
#undef W16R_LR
#undef W16R_LC
    }
    if (!alive) return; /* rows are independent from here on */
    W16_UNROLL for (int s = 0; s < R; s++)
    {
        mine[s] = row[s] < n;
        isx[s] = row[s] >= NU && row[s] < n;
        cx[s] = isx[s] ? row[s] - NU : 0;
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
#if defined(W16R_SKIP_UPDATE) /* development builds: what the update pass costs (the solve no longer converges) */
    if (l == 0) { D.alpha[inst] = alpha; D.iter[inst] = it + 1; }
    return;
#endif
    if (O.ext_update)
    {
        /* launch-per-sweep loop: the step is applied by k_step_update (ipm_kernels.hpp), every element its own work item */
        if (l == 0)
        {
            D.apend[inst] = a;
            D.alpha[inst] = alpha;
            D.iter[inst] = it + 1;
            if (st) { st[4 * D.stat_inst] = alpha; st[5 * D.stat_inst] = alpha; }
        }
        return;
    }
    /* update: one slot per variable / state / box row of the stage (dux, dpi, dlam, dt were written by these very
     * slots).  Everything the lane updates in W16R_UPD_CH consecutive stages -- both slots: variable, state multiplier, the
     * two sides of the box row -- is loaded through clamped addresses before the first store of the chunk: one memory
     * round trip per chunk.  (Three loops per slot with four stages in flight each made six times as many, and the pass
     * cost 0.68 ms of the C3 corrector sweep's 1.68: measured with the pass compiled out.) */
    {
        constexpr int CH = W16R_UPD_CH;
        const GqpStagePtr st_ = D.st;
        const uint64_t *__restrict__ am_ = D.amask.p + (size_t) inst * D.amask.E;
        for (int k0 = 0; k0 <= D.N; k0 += CH)
        {
            double u0[CH][R], du[CH][R], p0[CH][R], dp[CH][R], ll[CH][R], lu[CH][R], dll[CH][R], dlu[CH][R], tl[CH][R], tu[CH][R],
                   dtl[CH][R], dtu[CH][R];
            int el[CH][R], eu[CH][R];
            bool al[CH][R], au[CH][R];
            W16_UNROLL for (int c = 0; c < CH; c++)
            {
                const int k = k0 + c <= D.N ? k0 + c : D.N; /* beyond the horizon: the last stage once more, nothing stored */
                const uint64_t bm = st_[k].bmask, imask = bm & ~st_[k].emask;
                const uint64_t am = GEN ? 0 : am_[k * D.AW];
                const int nbg = st_[k].nb, o_ct = st_[k].o_ct;
                W16_UNROLL for (int s = 0; s < R; s++)
                {
                    const int lc_ = mine[s] ? row[s] : 0, xc_ = isx[s] ? cx[s] : 0;
                    u0[c][s] = WAT(D.ux, k * n + lc_); du[c][s] = WAT(D.dux, k * n + lc_);
                    p0[c][s] = WAT(D.pi, k * NX + xc_); dp[c][s] = WAT(D.dpi, k * NX + xc_);
                    if (!GEN)
                    {
                        const bool has = mine[s] && ((imask >> row[s]) & 1);
                        const int ib = has ? popc64(bm & (((uint64_t) 1 << row[s]) - 1)) : 0;
                        el[c][s] = o_ct + ib; eu[c][s] = el[c][s] + nbg;
                        al[c][s] = has && ((am >> ib) & 1); au[c][s] = has && ((am >> (nbg + ib)) & 1);
                        ll[c][s] = WAT(D.lam, el[c][s]); lu[c][s] = WAT(D.lam, eu[c][s]);
                        dll[c][s] = WAT(D.dlam, el[c][s]); dlu[c][s] = WAT(D.dlam, eu[c][s]);
                        tl[c][s] = WAT(D.t, el[c][s]); tu[c][s] = WAT(D.t, eu[c][s]);
                        dtl[c][s] = WAT(D.dt, el[c][s]); dtu[c][s] = WAT(D.dt, eu[c][s]);
                    }
                }
            }
            W16_UNROLL for (int c = 0; c < CH; c++)
            {
                const int k = k0 + c;
                if (k > D.N) break;
                W16_UNROLL for (int s = 0; s < R; s++)
                {
                    if (mine[s]) WAT(D.ux, k * n + row[s]) = u0[c][s] + a * du[c][s];
                    if (isx[s] && k >= 1) WAT(D.pi, k * NX + cx[s]) = p0[c][s] + a * dp[c][s];
                    if (!GEN)
                    {
                        const double laml = ll[c][s] + a * dll[c][s], lamu = lu[c][s] + a * dlu[c][s];
                        const double ttl = tl[c][s] + a * dtl[c][s], ttu = tu[c][s] + a * dtu[c][s];
                        if (al[c][s]) { WAT(D.lam, el[c][s]) = laml < O.lam_min ? O.lam_min : laml; WAT(D.t, el[c][s]) = ttl < O.t_min ? O.t_min : ttl; }
                        if (au[c][s]) { WAT(D.lam, eu[c][s]) = lamu < O.lam_min ? O.lam_min : lamu; WAT(D.t, eu[c][s]) = ttu < O.t_min ? O.t_min : ttu; }
                    }
                }
            }
        }
    }
    if (GEN)
    {
        /* box row of every slot (equality-flagged rows take no part) and general row l (slot 0): the loads of all of them,
         * then the stores -- the row -> slack map and the values: two memory round trips per stage.  (Issuing the loads of
         * stage k + 1 in front of the stores of stage k as well: measured, no gain.) */
        for (int k = 0; k <= D.N; k++)
        {
            GQP_STAGE_REF S = D.st[k];
            const W16Dsc dsc = {S.nb, S.ng, S.ns, S.o_ct, S.o_s};
            const uint64_t am = WAT(D.amask, k * D.AW);
            W16RowU ub[R];
            W16_UNROLL for (int s = 0; s < R; s++)
            {
                const bool hb = mine[s] && (((S.bmask & ~S.emask) >> row[s]) & 1);
                const int ib = hb ? popc64(S.bmask & (((uint64_t) 1 << row[s]) - 1)) : 0;
                ub[s] = w16r_row_update_load(D, inst, dsc, am, hb, ib, (int) S.srev[ib]);
            }
            const bool hg = l < S.ng;
            const W16RowU ug = w16r_row_update_load(D, inst, dsc, am, hg, S.nb + (hg ? l : 0), (int) S.srev[S.nb + (hg ? l : 0)]);
            W16_UNROLL for (int s = 0; s < R; s++) w16r_row_update_store(D, O, inst, ub[s], a);
            w16r_row_update_store(D, O, inst, ug, a);
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
