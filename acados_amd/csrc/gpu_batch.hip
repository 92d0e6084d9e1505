/*
 * gpu_batch.hip -- host side of the device-resident OCP-QP batch: HBM allocation, layout
 * conversion (pack / unpack kernels), kernel dispatch by shape, and the IPM launch loop.
 * C-ABI declared in include/acados_amd/ocp_qp_gpu_batch.h.
 */
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <climits>
#include <vector>

#include "acados_amd/ocp_qp_gpu_batch.h"
#include "gpu_ipm_internal.h"
#include "ipm_kernels.hpp"
#include "ipm_kernels_box.hpp"
#include "ipm_kernels_box_small.hpp"
#include "ipm_kernels_wpi.hpp"
#include "ipm_kernels_w16.hpp"
#include "ipm_kernels_w16r.hpp"
#include "ipm_kernels_w16t.hpp"
#include "ipm_kernels_wpi_mfma.hpp"
#include "res_kernels.hpp"
#include "pcond_kernels_w16.hpp"
#include "pcond_kernels_mfma.hpp"
#include "dense_kernels.hpp"
#include "kernel_sets.h"

/* a failing HIP call is not something a solve can recover from (lost device, out of HBM): message + exit(1), acados'
 * convention for errors that are not a solver status (`printf(...); exit(1);` throughout acados/ocp_qp/); what a solve
 * itself can report -- NaN, MAXITER, MINSTEP, infeasible -- comes back per instance as acados return codes */
/* ... with one exception (round-3 review): a HIP error is not the caller's mistake, and n host threads of an MPC fleet may
 * share this process (rendezvous mode).  HIPCHK reports and THROWS; every extern "C" entry that does device work catches at
 * the boundary (function-try-block) and returns -1 (NULL for the creators): the adapters turn that into ACADOS_QP_FAILURE
 * for the instances of the call, the process lives on.  No exception ever crosses the C-ABI. */
struct gqp_hip_failure { int code; };
#define HIPCHK(x)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess)                                                                  \
        {                                                                                      \
            fprintf(stderr, "\nerror: acados_amd: HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            throw gqp_hip_failure{(int) e_};                                                   \
        }                                                                                      \
    } while (0)

namespace
{

#define GQP_WPI_MIN_N 13       /* nu+nx from which the wave-per-instance kernels serve every batch */
#define GQP_WPI_BATCH_MAX 8192 /* batch size up to which they also serve the smaller stage blocks */
#define GQP_W16_BATCH_MAX 20480 /* ... where a 16-lanes-per-instance instantiation exists (crossover of the C2 shape: ~20k) */
#define GQP_WPI_GEN_SMALL_MAX 2048   /* ... for nu + nx <= 6 with general rows / shared slacks (compiled general set) */
#define GQP_W16_SMALL_MAX 4096       /* ... against the pipelined small-block kernels (nu + nx <= 6) */
#define GQP_W16_SMALL_XBOX_MAX 12288 /* ... the same with box rows on the states */

/* compiled shape classes; a batch is served by the cheapest one that covers it */
const KernelSet g_ksets[] = {
    GQP_KSET(4, 1, 0, 0),
    GQP_KSET(4, 1, 2, 3),
    GQP_KSET(4, 4, 0, 0), /* child shape of (4,1) condensed in blocks of <= 4 */
    GQP_KSET(8, 3, 0, 0),
    GQP_KSET(12, 3, 0, 0),
};

/* sixteen-lanes-per-instance sweeps (ipm_kernels_w16.hpp): compiled (NX, NU) with nu + nx <= 16 */
struct W16Set
{
    int NX, NU, NG; /* NG > 0: two-rows-per-lane GEN kernels (general rows + one slack per row) in the SOFT slots */
    kern_redo_t fact, rhs, faff, fcor;
    kern_redo_t sfact, srhs, sfaff, sfcor; /* SOFT variants: soft box rows, one slack per row */
    size_t shmem;
    kern_redo_t solve, ssolve; /* the whole solve in one launch (small batches); null: launch per sweep only */
    kern_redo_t tfact;         /* two-rows shapes, box rows: the factor sweep on 4 x 4 MFMA tiles (ipm_kernels_w16t.hpp), the default */
    size_t tshmem;
};
#define GQP_W16(NX, NU)                                                                                       \
    {NX, NU, 0, gqp::kx_factor<NX, NU>, gqp::kx_backrhs<NX, NU>, gqp::kx_fwd<NX, NU, false>, gqp::kx_fwd<NX, NU, true>, \
     gqp::kx_factor<NX, NU, true>, gqp::kx_backrhs<NX, NU, true>, gqp::kx_fwd<NX, NU, false, true>,            \
     gqp::kx_fwd<NX, NU, true, true>, 4 * gqp::W16Lds<NX, NU>::SZ * sizeof(double), gqp::kx_solve<NX, NU>, gqp::kx_solve<NX, NU, true>, \
     nullptr, 0}
/* ... and with 17 <= nu + nx <= 32 (ipm_kernels_w16r.hpp: two rows per lane; box rows without slacks) */
#define GQP_W16R(NX, NU)                                                                                      \
    {NX, NU, 0, gqp::ky_factor<NX, NU>, gqp::ky_backrhs<NX, NU>, gqp::ky_fwd<NX, NU, false>, gqp::ky_fwd<NX, NU, true>, \
     nullptr, nullptr, nullptr, nullptr, 4 * gqp::W16RLds<NX, NU>::SZ * sizeof(double), nullptr, nullptr,                \
     gqp::kt_factor<NX, NU>, 4 * gqp::W16TLds<NX, NU>::SZ * sizeof(double)}
/* ... with general rows and slacks (one slack per row): the C4 class; the same kernels at nu + nx <= 16 (R = 1 row per lane:
 * <12,4,4>, <8,3,4>) put general rows + slacks of the small shapes on the sixteen-lanes family too */
#define GQP_W16G(NX, NU, NG)                                                                                  \
    {NX, NU, NG, nullptr, nullptr, nullptr, nullptr, gqp::ky_factor<NX, NU, NG>, gqp::ky_backrhs<NX, NU, NG>,     \
     gqp::ky_fwd<NX, NU, false, NG>, gqp::ky_fwd<NX, NU, true, NG>, 4 * gqp::W16RLds<NX, NU, NG>::SZ * sizeof(double), nullptr, nullptr, \
     gqp::kt_factor<NX, NU, NG>, 4 * gqp::W16TLds<NX, NU, NG>::SZ * sizeof(double)}
/* ... the same on register rows only (ky_factor): the tile sweep of this instantiation does not fit the register file
 * (kt_factor<24,3,8>: 6 spilled VGPRs; spill traffic is HBM traffic there -- the ISA lint of `make` refuses it) */
#define GQP_W16G_ROWS(NX, NU, NG)                                                                             \
    {NX, NU, NG, nullptr, nullptr, nullptr, nullptr, gqp::ky_factor<NX, NU, NG>, gqp::ky_backrhs<NX, NU, NG>,     \
     gqp::ky_fwd<NX, NU, false, NG>, gqp::ky_fwd<NX, NU, true, NG>, 4 * gqp::W16RLds<NX, NU, NG>::SZ * sizeof(double), nullptr, nullptr, \
     nullptr, 0}
const W16Set g_w16_sets[] = {GQP_W16(4, 1), GQP_W16(8, 3), GQP_W16(12, 3), GQP_W16(4, 4), GQP_W16(12, 4),
                             GQP_W16R(8, 15), GQP_W16R(24, 6), GQP_W16G(24, 3, 4), GQP_W16G_ROWS(24, 3, 8), GQP_W16G(12, 4, 4), GQP_W16G(8, 3, 4)};

/* condensing of the box-only class on register rows, sixteen lanes per block (pcond_kernels_w16.hpp): compiled
 * (NX, NU, block size) with nx + bs * nu <= 32 */
struct PcondzSet
{
    int NX, NU, BS;
    kern_pcond_t cond;
    size_t shmem;
    kern_pcond_t cond_m; /* the same contraction on the FP64 matrix pipe, v_mfma_f64_4x4x4_4b_f64 (pcond_kernels_mfma.hpp): the default */
};
#define GQP_PCONDZ(NX, NU, BS) {NX, NU, BS, gqp::kz_pcond<NX, NU, BS>, 4 * gqp::PcondzLds<NX, NU, BS>::SZ * sizeof(double), gqp::km_pcond<NX, NU, BS>}
const PcondzSet g_pcondz_sets[] = {GQP_PCONDZ(8, 3, 5), GQP_PCONDZ(4, 1, 4)};

} // namespace

struct ocp_qp_gpu_batch
{
    int B = 0, Bp = 0, N = 0, device = 0;
    std::vector<int> nx, nu, nbx, nbu, nb, ng, ns, nbxe;
    std::vector<std::vector<int>> idxb, idxs_rev, idxe; /* as given (original row order) */
    std::vector<std::vector<int>> perm;                   /* original box row -> sorted row */
    bool finalized = false;
    bool use_box = false; /* box-only fast path */
    int aos = 0;          /* instance-major arrays: wave-per-instance kernel family */
    int AW = 1;           /* activity words per stage (64 inequality sides each); 2 only for wave-per-instance batches */
    int wpi = 0;          /* wave-per-instance kernels (ipm_kernels_wpi.hpp): one workgroup per instance */
    bool wpi_mfma = false; /* ... whose factor sweep is the blocked-Cholesky / MFMA kernel (17 <= n <= 32) */
    int w16 = 0;          /* ... whose four sweeps are the 16-lanes-per-instance kernels (ipm_kernels_w16.hpp): 4 instances per workgroup */
    size_t shmem = 0;     /* their dynamic LDS bytes (rhs sweep, init, finalize) */
    size_t shmem_fwd = 0; /* ... of the forward sweeps (one factor buffer instead of two) */
    size_t shmem_fact = 0; /* dynamic LDS bytes of the factor sweep */
    bool w16_soft = false; /* ... with soft box rows (one slack per row) */
    int w16_ng = 0;        /* ... of the GEN two-rows-per-lane set: general rows per stage it carries (slacks on them too) */
    KernelSet wpi_ks;      /* the wave-per-instance set of the same padded dims (fallback of w16_soft) */
    int w16_slots = 0;     /* row slots a sweep launch covers: B, or the live instances once GqpDev::perm lists them */
    size_t w16_shmem = 0;  /* dynamic LDS bytes of a 16-lanes-per-instance workgroup (4 instances) */
    size_t w16_shmem_fact = 0; /* ... of the factor sweep (the tile sweep of the small two-rows shapes runs two waves per SIMD: its own, smaller tile) */
    int w16_tiles = 0;     /* two-rows family: factor sweep on 4 x 4 MFMA tiles (kt_factor) */
    int *d_side_map = nullptr; /* k_step_update: side -> stage * 128 + activity bit (-1: equality-flagged row), built on first use */
    kern_redo_t w16_solve = nullptr; /* whole-solve kernel of the family (batches of at most solve_max instances) */
    int solve_max = 256;   /* largest batch that is solved in one launch (option "solve_max", 0 = off) */
    int n_single_launch = 0;
    KernelSet own_ks;     /* runtime-shaped kernel set of a wpi batch (ks points here) */
    int xbox = 0;
    const KernelSet *ks = nullptr;
    std::string kname;
    std::vector<GqpStage> st;
    GqpStage *d_st = nullptr;
    GqpDev D = {};
    GqpOpts O;
    bool has_slack = false;            /* some stage has slack variables (set with the dims) */
    /* full condensing of any size (option "full_dense": dense_kernels.hpp, dense_solve below) */
    int full_dense = 0;
    gqp::KdRow *d_kd_rows = nullptr;
    gqp::KdDims kd = {};
    int kd_unsupported = 0;
    double *d_kd_ws = nullptr;
    int kd_slice = 0;
    /* terminal polishing step (option "polish", opt-in: polish_pass below) */
    int polish = 0;
    double polish_ratio = 1e-3;
    double polish_min = 0.0; /* ... and min(lam, t) above this (absolute): a pair whose smaller member is already below the accuracy wanted cannot move the point by more */
    int n_polished = 0, n_polish_reverted = 0;
    int *d_pol_status = nullptr, *d_pol_iter = nullptr, *d_pol_flag = nullptr, *d_pol_cnt = nullptr;
    double *d_pol_sc = nullptr;
    GArr pol_ux = {nullptr, 0, 0}, pol_sv = {nullptr, 0, 0}, pol_pi = {nullptr, 0, 0}, pol_lam = {nullptr, 0, 0}, pol_t = {nullptr, 0, 0};
    double tol_comp_soft_scale = 1.0; /* effective_opts: opt-in tighter exit on complementarity of a soft-constrained class (1 = the tolerance as given, the reference's semantics) */
    int nct_tot = 0, ns2_tot = 0, ng_tot = 0;
    std::vector<void *> allocs;
    size_t bytes = 0;
    double *d_stage = nullptr; /* staging for host->device field blocks */
    double *d_chunks = nullptr; /* the input blob handed over in pieces (_set_bulk_chunk): its own buffer -- d_stage is reused by */
    size_t chunks_cap = 0;      /* every other transfer (a hot start's _set_bulk_out comes between the chunks and the scatter) */
    /* zero-copy gather (ocp_qp_gpu_batch_gather_tables / _gather_run): word tables of the full input blob [0] and of its vector part [1] */
    struct GatherTab { int P = 0, n_words = 0, full = 0; int *d_slot = nullptr, *d_off = nullptr, *d_pos = nullptr; unsigned char *d_neg = nullptr; } gtab[2];
    const double **d_gptrs = nullptr;
    size_t gptrs_cap = 0;
    long chunks_got = 0;        /* instances handed over since the last _set_bulk_staged (it refuses to scatter a partial blob) */
    hipEvent_t chunks_ev = nullptr; /* recorded in front of the first chunk of a round: time_pack covers copies + scatter as _set_bulk's does */
    size_t stage_cap = 0;
    int *d_map = nullptr;
    int map_cap = 0;
    int *h_nact = nullptr; /* pinned */
    int *d_perm = nullptr, *d_perm_cnt = nullptr; /* sixteen-lanes sweeps: dense list of the still-iterating instances */
    int *h_ints = nullptr; /* pinned, 2 * Bp ints: per-instance status / iteration read-backs of a solve (no heap traffic per call) */
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double time_tot = 0.0, time_pack = 0.0;
    int last_iters = 0, launches = 0;
    int print_level = 0;
    double t0_min = 1e-16, lam0_min = 1e-16; /* lower clips of t / lam at a hot start (HPIPM args of the same name) */
    int profile = 0;                 /* per-kernel-class HIP event timing on the launch stream */
    bool kb_plain = true;            /* one-instance-per-lane box batch: the phase-ordered kernels (false: the pipelined small-block ones) */
    int stream_priority = 0;         /* HIP stream priority of this batch and of the sub-batches it creates (0: default) */
    std::vector<hipEvent_t> prof_ev; /* pool, pairs (start, stop) */
    std::vector<int> prof_cls;       /* kernel class of each recorded pair */
    double prof_ms[6] = {0, 0, 0, 0, 0, 0};
    int prof_cnt[6] = {0, 0, 0, 0, 0, 0};
    int stat_inst = 0, stat_rows = 0;
    /* partial condensing (pcond_kernels.hpp) */
    int cond_N = 0;                 /* requested N2; 0 or N = full space */
    int cond_keep_iterate = 0;      /* hot start of a condensed solve from the child's own last iterate (not from the root's) */
    int force_NX = 0, force_NU = 0; /* child batches are pinned to the kernel shape the condense kernel writes */
    std::vector<int> user_blocks;   /* cond_block_size (N2 entries) or empty: N/N2 each, remainder to the first blocks */
    int pcond_state = 0;            /* 0 unchecked, 1 active, -1 not applicable (message printed once) */
    ocp_qp_gpu_batch *child = nullptr;
    const PcondSet *pc = nullptr;   /* compiled one-instance-per-lane condensing kernels, or ... */
    int pc_rt = 0;                  /* ... the run-time-shaped wave-per-instance ones (kw_pcond / kw_pexpand) */
    int pc_lane_expand = 0;         /* the expansion runs on the compiled one-instance-per-lane kernel although condensing does not */
    const PcondzSet *pcz = nullptr; /* condensing on register rows, sixteen lanes per block (box-only class, compiled shapes) */
    int pcz_mfma = 1;               /* ... on the FP64 matrix pipe (km_pcond); ACADOS_AMD_PCOND_MFMA=0: on register rows with DPP broadcasts (kz_pcond) */
    size_t pc_shmem = 0;
    std::vector<int> blk_start;
    gqp::PcondMap pmap;
    double time_xcond = 0.0;
    bool lhs_ready = false;         /* condense_lhs done: the next solve only condenses the vector part */
    /* compaction of the still-iterating instances into a dense sub-batch (see run_ipm) */
    ocp_qp_gpu_batch *compact = nullptr; /* next level, capacity <= B/2, created on first use */
    const KernelSet *force_ks = nullptr; /* sub-batches run the very same kernel set */
    int compact_min = 1 << 30;           /* levels smaller than this are not compacted; off by default: it only
                                            pays once every sweep kernel is bandwidth-bound */
    ocp_qp_gpu_batch *tail = nullptr;    /* wave-per-instance sub-batch for the last survivors of a one-instance-per-lane level */
    int tail_max = 12288;                /* switch to it when at most this many instances (and 1 / tail_div of the level) remain; 0 = off */
    int tail_div = 4;
    int n_tail_switches = 0;
    /* solution sensitivities / factor at the solution */
    bool factor_stale = false;           /* the last solve finished instances on a sub-level: Lf of the root is not theirs */
    bool sens_open = false;              /* seeds are being collected (rg, rb, rd hold seeds, not residuals) */
    GArr sfix = {nullptr, 0, 0};         /* derivative of the equality-flagged variables (e.g. x0), [N+2][n] */
    int *d_saved_status = nullptr;
    ocp_qp_gpu_batch *sens_child = nullptr; /* wave-per-instance sub-batch the sensitivities of a one-instance-per-lane batch run in */
    int *d_slist = nullptr;
    int sens_cap = 0;
    int tail_cap = 0;
    int *d_list = nullptr;               /* instance index of every slot of `compact` / `tail` */
    int list_cap = 0;
    int n_compactions = 0;
    /* KKT residuals of the current (data, iterate) on demand (res_kernels.hpp) */
    gqp::ResOut R = {{nullptr, 0, 0}, {nullptr, 0, 0}, {nullptr, 0, 0}, {nullptr, 0, 0}, {nullptr, 0, 0}, nullptr, 0};
    /* bulk pack / unpack (one H2D + one launch per direction) */
    struct BulkMap
    {
        bool built = false;
        int len = 0, nm = 0;
        std::vector<std::string> fields; /* per segment */
        std::vector<int> seg_stage, seg_off, seg_len;
        int *d_arr = nullptr, *d_elem = nullptr, *d_moff = nullptr, *d_mstage = nullptr, *d_mbit = nullptr;
        int *d_sgn = nullptr, *d_elem2 = nullptr; /* seed blob only: sign of the entry in the residual arrays, slot in sfix */
        int *d_arr_g = nullptr, *d_elem_g = nullptr; /* input blob, READ direction (_get_bulk_in): the strict upper triangles of
                                                        Q and R (not written: only the lower triangle of the caller's block is
                                                        valid) are read from the mirrored element */
        gqp::GArrTable T;
    } bulk_in, bulk_out, bulk_seed, bulk_vec;
};

namespace
{

template <class T>
T *dalloc(ocp_qp_gpu_batch *b, size_t cnt)
{
    void *p = nullptr;
    size_t bytes = sizeof(T) * (cnt ? cnt : 1);
    HIPCHK(hipMalloc(&p, bytes));
    HIPCHK(hipMemset(p, 0, bytes));
    b->allocs.push_back(p);
    b->bytes += bytes;
    return (T *) p;
}

/* wave-tiled per-instance array with E elements per instance (gpu_ipm_internal.h, GArrT) */
template <class T>
GArrT<T> garr(ocp_qp_gpu_batch *b, size_t E)
{
    GArrT<T> a;
    a.aos = b->aos;
    a.E = (int) (E ? E : 1);
    a.p = dalloc<T>(b, (size_t) a.E * (size_t) b->Bp);
    return a;
}

void opts_default(GqpOpts &o)
{
    /* acados defaults on top of mode BALANCE: ocp_qp_hpipm.c:101-113 */
    o.mu0 = 1e0;
    o.tol_stat = 1e-6; o.tol_eq = 1e-8; o.tol_ineq = 1e-8; o.tol_comp = 1e-8;
    o.alpha_min = 1e-8; o.tau_min = 0.0; o.lam_min = 1e-16; o.t_min = 1e-16; o.reg_prim = 1e-15;
    o.iter_max = 50; o.pred_corr = 1; o.cond_pred_corr = 1; o.warm_start = 0; o.ext_update = 0;
    o.t0_init = 2; /* acados_ocp_options.py:1128-1143: the default is the residual-based start */
}

/* options as the kernels see them.
 * (1) OPT-IN (option "tol_comp_soft_scale", default 1 = off: the solver stops at the tol_comp it is given, as HPIPM does,
 *     ocp_qp_hpipm.c:104-107).  With a value < 1, soft-constrained classes (any stage with slacks) use
 *     tol_comp * tol_comp_soft_scale in the exit test.  Why one may want it: a soft row with a small
 *     multiplier lam* sits at t = mu / lam* on the central path, and with slack penalties of 1e2 the primal solution is
 *     flat: at mu ~ 1e-8 the C4 iterate is a median 3e-7 / worst 1e-4 (relative) away from the exact solution although
 *     all four KKT residuals are <= 1e-8 (profiles/r04_c4_distance_to_solution.txt) -- two solvers stopping inside that
 *     ball cannot agree to 1e-6.  Three more orders of mu cost 1.4 iterations of 12.7 and bring it to 3e-10 / 3e-6 * sqrt.
 * (2) Barrier floor: the complementarity target never drops below 1e-3 of that tolerance (see oracle/ocp_qp_oracle.c
 *     tau_eff: below it only the rounding error of the Newton step grows). */
GqpOpts effective_opts(const GqpOpts &o, const ocp_qp_gpu_batch *b)
{
    GqpOpts e = o;
    if (b->has_slack && b->tol_comp_soft_scale > 0.0 && b->tol_comp_soft_scale < 1.0) e.tol_comp = o.tol_comp * b->tol_comp_soft_scale;
    const double f = 1e-3 * e.tol_comp;
    if (e.tau_min < f) e.tau_min = f;
    return e;
}

int padded_var(const ocp_qp_gpu_batch *b, int k, int iv)
{
    /* index in [u(nu_k); x(nx_k)] -> index in padded [u(NU); x(NX)] */
    return iv < b->nu[k] ? iv : b->ks->NU + (iv - b->nu[k]);
}

void finalize_structure(ocp_qp_gpu_batch *b)
{
    if (b->finalized) return;
    const int N = b->N, NX = b->ks->NX, NU = b->ks->NU;
    const int n = NX + NU, NP = n * (n + 1) / 2;
    b->st.assign(N + 1, GqpStage());
    b->perm.assign(N + 1, std::vector<int>());
    int o_ct = 0, o_s = 0, o_g = 0;
    for (int k = 0; k <= N; k++)
    {
        GqpStage &S = b->st[k];
        memset(&S, 0, sizeof(S));
        S.nb = b->nb[k]; S.ng = b->ng[k]; S.ns = b->ns[k];
        if (S.ns > 0) b->has_slack = true;
        S.o_ct = o_ct; S.o_s = o_s; S.o_g = o_g; S.has_dyn = k < N;
        const int nbg = S.nb + S.ng, nct = 2 * nbg + 2 * S.ns;
        if (nct > 64 * b->AW || nbg > GQP_MAX_ROWS)
        {
            fprintf(stderr, "acados_amd: stage %d has %d inequality sides (> %d): unsupported\n", k, nct, 64 * b->AW);
            exit(1); /* not reachable through ocp_qp_gpu_batch_create, which refuses such dims */
        }
        /* sort box rows by variable */
        std::vector<int> order(S.nb);
        for (int r = 0; r < S.nb; r++) order[r] = r;
        std::sort(order.begin(), order.end(), [&](int a, int c) { return b->idxb[k][a] < b->idxb[k][c]; });
        b->perm[k].assign(S.nb, 0);
        for (int sp = 0; sp < S.nb; sp++)
        {
            const int ob = order[sp];
            b->perm[k][ob] = sp;
            const int iv = b->idxb[k][ob];
            if (iv < 0 || iv >= b->nu[k] + b->nx[k]) { fprintf(stderr, "acados_amd: idxb out of range at stage %d\n", k); exit(1); }
            const int pv = padded_var(b, k, iv);
            if ((S.bmask >> pv) & 1) { fprintf(stderr, "acados_amd: duplicate idxb entry at stage %d: unsupported\n", k); exit(1); }
            S.bmask |= (uint64_t) 1 << pv;
        }
        for (int r = 0; r < GQP_MAX_ROWS; r++) S.srev[r] = -1;
        for (int r = 0; r < nbg; r++)
        {
            const int sj = b->idxs_rev[k][r];
            if (sj >= S.ns) { fprintf(stderr, "acados_amd: idxs_rev out of range at stage %d\n", k); exit(1); }
            const int row = r < S.nb ? b->perm[k][r] : r;
            S.srev[row] = (int8_t) sj;
        }
        for (size_t e = 0; e < b->idxe[k].size(); e++)
        {
            const int ob = b->idxe[k][e];
            if (ob < 0 || ob >= S.nb) { fprintf(stderr, "acados_amd: idxe out of range at stage %d\n", k); exit(1); }
            S.emask |= (uint64_t) 1 << padded_var(b, k, b->idxb[k][ob]);
        }
        o_ct += nct; o_s += 2 * S.ns; o_g += S.ng;
    }
    b->nct_tot = o_ct; b->ns2_tot = o_s; b->ng_tot = o_g;
    if (b->w16 && b->w16_soft)
    {
        /* the SOFT sixteen-lanes kernels eliminate a slack inside the lane that owns its row: every slack must belong
         * to exactly one box row that is not an equality-flagged one; anything else runs on the general
         * wave-per-instance kernels of the same padded dims */
        bool ok = true;
        for (int k = 0; k <= N && ok; k++)
        {
            const GqpStage &S = b->st[k];
            std::vector<int> refs(S.ns, 0);
            for (int r = 0; r < S.nb + (b->w16_ng ? S.ng : 0); r++)
                if (S.srev[r] >= 0) refs[S.srev[r]]++;
            for (int q = 0; q < S.ns; q++) ok = ok && refs[q] == 1;
            if (!b->w16_ng) ok = ok && S.ng == 0;
            else ok = ok && __builtin_popcountll(S.bmask & ~S.emask) + S.ng <= 16; /* GEN: one inequality row per lane */
            for (size_t e = 0; e < b->idxe[k].size(); e++) ok = ok && b->idxs_rev[k][b->idxe[k][e]] < 0;
        }
        if (!ok)
        {
            b->own_ks = b->wpi_ks;
            b->w16 = 0;
            b->w16_solve = nullptr;
            b->w16_soft = false;
            b->w16_tiles = 0;
            char nm[160];
            snprintf(nm, sizeof(nm), "wpi-gen(nx=%d,nu=%d,ng=%d,ns=%d,lds=%zuB)", b->ks->NX, b->ks->NU, b->ks->NG, b->ks->NS, b->shmem_fact);
            b->kname = nm;
        }
    }
    b->use_box = o_g == 0 && o_s == 0 && !getenv("ACADOS_AMD_GENERAL_KERNELS");
    b->xbox = 0;
    for (int k = 0; k <= N; k++)
        if (((b->st[k].bmask & ~b->st[k].emask) >> NU) != 0) b->xbox = 1;
    if (b->use_box && !b->wpi)
    {
        /* nu + nx <= 6: the pipelined kernels of ipm_kernels_box_small.hpp ("1tpi-pipe"); ACADOS_AMD_KB_SMALL=0 keeps the
         * phase-ordered ones of ipm_kernels_box.hpp (cross-check) */
        const char *small = getenv("ACADOS_AMD_KB_SMALL");
        b->kb_plain = (small && atoi(small) == 0) || NX + NU > 6;
        char nm[160];
        snprintf(nm, sizeof(nm), "1tpi-%s<NX=%d,NU=%d,XBOX=%d>", b->kb_plain ? "box" : "pipe", NX, NU, b->xbox);
        b->kname = nm;
    }
    if (b->wpi)
    {
        b->use_box = true;
        if (std::max(b->shmem, b->shmem_fact) > 64 * 1024)
        {
            /* more than the default dynamic LDS limit: raise it for the four sweep kernels */
            const void *fns[] = {(const void *) b->own_ks.back_fact, (const void *) b->own_ks.back_rhs,
                                 (const void *) b->own_ks.fwd_aff, (const void *) b->own_ks.fwd_corr};
            for (const void *f : fns)
                HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int) std::max(b->shmem, b->shmem_fact)));
        }
    }
    b->d_st = dalloc<GqpStage>(b, N + 1);
    HIPCHK(hipMemcpy(b->d_st, b->st.data(), sizeof(GqpStage) * (N + 1), hipMemcpyHostToDevice));

    GqpDev &D = b->D;
    const size_t Bp = b->Bp;
    D.B = b->B; D.Bp = b->Bp; D.N = N; D.NX = NX; D.NU = NU; D.NG = b->ks->NG; D.NS = b->ks->NS;
    D.st = (GqpStagePtr) (uintptr_t) b->d_st;
    const size_t RP = 16; /* spare row elements (clamped dummy row index) */
    D.BAt = garr<double>(b, (size_t) ((N + 1) * n * NX));
    D.bvec = garr<double>(b, (size_t) ((N + 1) * NX));
    D.RSQ = garr<double>(b, (size_t) ((N + 1) * NP));
    D.rq = garr<double>(b, (size_t) ((N + 1) * n));
    D.dvec = garr<double>(b, (size_t) ((o_ct + RP)));
    D.AW = b->AW;
    D.amask = garr<uint64_t>(b, (size_t) ((N + 1) * b->AW));
    /* (+ n spare elements: the sixteen-lanes GEN kernels prefetch the stage's rows of [D C] branch-free, a stage WITHOUT general rows
     * reads element 0 of its (empty) block -- for such stages at the end of the horizon that is the element behind the last row:
     * found by the structure fuzz of round 6, seed 9193, a fault only where the array ended on a page boundary) */
    D.DCt = garr<double>(b, (size_t) (o_g * n + n));
    /* (at least one (Z, z) pair: the branch-free row functions of the sixteen-lanes GEN kernels read the pair of slack 0 through
     * clamped addresses also in a batch without slacks -- with one element per instance the z of the LAST instance lay behind the
     * allocation: found by the structure fuzz of round 5, a fault only where the array ended on a page boundary) */
    D.Zz = garr<double>(b, (size_t) (o_s > 0 ? o_s * 2 : 2));
    D.ux = garr<double>(b, (size_t) ((N + 2) * n));
    D.sv = garr<double>(b, (size_t) (o_s));
    D.pi = garr<double>(b, (size_t) ((N + 2) * NX));
    D.lam = garr<double>(b, (size_t) ((o_ct + RP)));
    D.t = garr<double>(b, (size_t) ((o_ct + RP)));
    D.rg = garr<double>(b, (size_t) ((N + 1) * n));
    D.rgs = garr<double>(b, (size_t) (o_s));
    D.rb = garr<double>(b, (size_t) ((N + 1) * NX));
    D.rd = garr<double>(b, (size_t) ((o_ct + RP)));
    D.rm = garr<double>(b, (size_t) ((o_ct + RP)));
    D.dux = garr<double>(b, (size_t) ((N + 2) * n));
    D.dsv = garr<double>(b, (size_t) (o_s));
    D.dpi = garr<double>(b, (size_t) ((N + 2) * NX));
    D.dlam = garr<double>(b, (size_t) ((o_ct + RP)));
    D.dt = garr<double>(b, (size_t) ((o_ct + RP)));
    D.pcorr = garr<double>(b, (size_t) ((o_ct + RP)));
    D.sD = garr<double>(b, (size_t) (o_s));
    D.sR = garr<double>(b, (size_t) (o_s));
    D.Lf = garr<double>(b, (size_t) ((N + 1) * NP));
    D.lf = garr<double>(b, (size_t) ((N + 1) * n));
    D.res = dalloc<double>(b, 4 * Bp);
    D.mu = dalloc<double>(b, Bp); D.smu = dalloc<double>(b, Bp);
    D.alpha = dalloc<double>(b, Bp); D.obj = dalloc<double>(b, Bp);
    D.apend = dalloc<double>(b, Bp);
    D.iter = dalloc<int>(b, Bp); D.status = dalloc<int>(b, Bp);
    D.n_active = dalloc<int>(b, 1);
    b->stat_inst = b->B < 64 ? b->B : 64;
    b->stat_rows = 0;
    D.stat = nullptr; D.stat_inst = 0; D.stat_rows = 0;
    D.perm = nullptr; D.n_perm = 0;

    /* neutral padding: unit Hessian diagonal on padded variables */
    const int grid = (b->B + 63) / 64;
    for (int k = 0; k <= N; k++)
    {
        for (int j = 0; j < n; j++)
        {
            const bool real = j < NU ? j < b->nu[k] : (j - NU) < b->nx[k];
            if (!real)
                hipLaunchKernelGGL(gqp::k_fill_strided, dim3(grid), dim3(64), 0, b->stream, D.RSQ, 1.0, b->B,
                                   k * NP + PK(j, j));
        }
        /* activity: every existing row side, minus equality-flagged rows */
        const GqpStage &S = b->st[k];
        const int nbg = S.nb + S.ng, nct = 2 * nbg + 2 * S.ns;
        uint64_t m[2] = {0, 0};
        for (int e = 0; e < nct; e++) m[e >> 6] |= (uint64_t) 1 << (e & 63);
        for (size_t e = 0; e < b->idxe[k].size(); e++)
        {
            const int sp = b->perm[k][b->idxe[k][e]];
            m[sp >> 6] &= ~((uint64_t) 1 << (sp & 63));
            m[(nbg + sp) >> 6] &= ~((uint64_t) 1 << ((nbg + sp) & 63));
        }
        for (int w = 0; w < b->AW; w++)
            hipLaunchKernelGGL(gqp::k_fill_u64, dim3(grid), dim3(64), 0, b->stream, D.amask, m[w], b->Bp, k * b->AW + w);
    }
    HIPCHK(hipStreamSynchronize(b->stream));
    b->finalized = true;
}

void ensure_stat(ocp_qp_gpu_batch *b)
{
    const int rows = b->O.iter_max + 2;
    if (b->stat_rows >= rows) return;
    /* (re)allocate the statistics table for the first stat_inst instances */
    b->D.stat = dalloc<double>(b, (size_t) rows * GQP_STAT_COLS * b->stat_inst);
    b->stat_rows = rows;
    b->D.stat_rows = rows;
    b->D.stat_inst = b->stat_inst;
}

/* element map of a numeric field: for every source element e the target element index
 * in `*arr` (units of Bp doubles), or -1.  Returns the field length, -1 if unknown. */
int field_map(ocp_qp_gpu_batch *b, const char *f, int k, std::vector<int> &map, GArr *arr,
              std::vector<int> *map2 = nullptr, GArr *arr2 = nullptr)
{
    const int N = b->N, NX = b->ks->NX, NU = b->ks->NU, n = NX + NU, NP = n * (n + 1) / 2;
    const GqpDev &D = b->D;
    const GqpStage &S = b->st[k];
    const int nx = b->nx[k], nu = b->nu[k], nx1 = k < N ? b->nx[k + 1] : 0;
    const int nbu = b->nbu[k], nbx = b->nbx[k], ng = S.ng, ns = S.ns, nbg = S.nb + S.ng;
    map.clear();
    auto dyn = [&]() { return k < N; };
    if (!strcmp(f, "A"))
    {
        if (!dyn()) return -1;
        *arr = D.BAt;
        for (int c = 0; c < nx; c++) for (int r = 0; r < nx1; r++) map.push_back((k * n + NU + c) * NX + r);
    }
    else if (!strcmp(f, "B"))
    {
        if (!dyn()) return -1;
        *arr = D.BAt;
        for (int c = 0; c < nu; c++) for (int r = 0; r < nx1; r++) map.push_back((k * n + c) * NX + r);
    }
    else if (!strcmp(f, "b"))
    {
        if (!dyn()) return -1;
        *arr = D.bvec;
        for (int r = 0; r < nx1; r++) map.push_back(k * NX + r);
    }
    else if (!strcmp(f, "Q"))
    {
        *arr = D.RSQ;
        for (int c = 0; c < nx; c++) for (int r = 0; r < nx; r++) map.push_back(r >= c ? k * NP + PK(NU + r, NU + c) : -1);
    }
    else if (!strcmp(f, "R"))
    {
        *arr = D.RSQ;
        for (int c = 0; c < nu; c++) for (int r = 0; r < nu; r++) map.push_back(r >= c ? k * NP + PK(r, c) : -1);
    }
    else if (!strcmp(f, "S"))
    {
        *arr = D.RSQ; /* S is nu x nx: u'Sx */
        for (int jx = 0; jx < nx; jx++) for (int iu = 0; iu < nu; iu++) map.push_back(k * NP + PK(NU + jx, iu));
    }
    else if (!strcmp(f, "q")) { *arr = D.rq; for (int r = 0; r < nx; r++) map.push_back(k * n + NU + r); }
    else if (!strcmp(f, "r")) { *arr = D.rq; for (int r = 0; r < nu; r++) map.push_back(k * n + r); }
    else if (!strcmp(f, "lbu") || !strcmp(f, "ubu") || !strcmp(f, "lbx") || !strcmp(f, "ubx"))
    {
        const bool up = f[0] == 'u', isx = f[2] == 'x';
        *arr = D.dvec;
        const int cnt = isx ? nbx : nbu, off = isx ? nbu : 0;
        for (int e = 0; e < cnt; e++) map.push_back(S.o_ct + (up ? nbg : 0) + b->perm[k][off + e]);
        if (!up && isx && map2)
        {
            /* equality-flagged x bounds: the bound value is the value of the variable */
            *arr2 = D.ux;
            map2->assign(cnt, -1);
            for (size_t q = 0; q < b->idxe[k].size(); q++)
            {
                const int ob = b->idxe[k][q];
                if (ob >= off && ob < off + cnt) (*map2)[ob - off] = k * n + padded_var(b, k, b->idxb[k][ob]);
            }
        }
    }
    else if (!strcmp(f, "lg") || !strcmp(f, "ug"))
    {
        *arr = D.dvec;
        for (int g = 0; g < ng; g++) map.push_back(S.o_ct + (f[0] == 'u' ? nbg : 0) + S.nb + g);
    }
    else if (!strcmp(f, "C"))
    {
        *arr = D.DCt;
        for (int c = 0; c < nx; c++) for (int g = 0; g < ng; g++) map.push_back((S.o_g + g) * n + NU + c);
    }
    else if (!strcmp(f, "D"))
    {
        *arr = D.DCt;
        for (int c = 0; c < nu; c++) for (int g = 0; g < ng; g++) map.push_back((S.o_g + g) * n + c);
    }
    else if (!strcmp(f, "Zl")) { *arr = D.Zz; for (int j = 0; j < ns; j++) map.push_back((S.o_s + j) * 2); }
    else if (!strcmp(f, "zl")) { *arr = D.Zz; for (int j = 0; j < ns; j++) map.push_back((S.o_s + j) * 2 + 1); }
    else if (!strcmp(f, "Zu")) { *arr = D.Zz; for (int j = 0; j < ns; j++) map.push_back((S.o_s + ns + j) * 2); }
    else if (!strcmp(f, "zu")) { *arr = D.Zz; for (int j = 0; j < ns; j++) map.push_back((S.o_s + ns + j) * 2 + 1); }
    else if (!strcmp(f, "lls")) { *arr = D.dvec; for (int j = 0; j < ns; j++) map.push_back(S.o_ct + 2 * nbg + j); }
    else if (!strcmp(f, "lus")) { *arr = D.dvec; for (int j = 0; j < ns; j++) map.push_back(S.o_ct + 2 * nbg + ns + j); }
    /* iterate */
    else if (!strcmp(f, "x")) { *arr = D.ux; for (int r = 0; r < nx; r++) map.push_back(k * n + NU + r); }
    else if (!strcmp(f, "u")) { *arr = D.ux; for (int r = 0; r < nu; r++) map.push_back(k * n + r); }
    else if (!strcmp(f, "sl")) { *arr = D.sv; for (int j = 0; j < ns; j++) map.push_back(S.o_s + j); }
    else if (!strcmp(f, "su")) { *arr = D.sv; for (int j = 0; j < ns; j++) map.push_back(S.o_s + ns + j); }
    else if (!strcmp(f, "pi"))
    {
        if (!dyn()) return -1;
        *arr = D.pi; /* acados pi[k] = multiplier of the dynamics producing x_{k+1}: slot k+1 */
        for (int r = 0; r < nx1; r++) map.push_back((k + 1) * NX + r);
    }
    else if (!strcmp(f, "ric_L"))
    {
        /* Cholesky factor of the stage matrix of the last factorisation, variables [u;x] of this
         * stage (padding removed), nv x nv column-major, lower triangle (upper = 0) */
        *arr = D.Lf;
        const int nv = nu + nx;
        for (int c = 0; c < nv; c++)
            for (int r = 0; r < nv; r++)
                map.push_back(r >= c ? k * NP + PK(padded_var(b, k, r), padded_var(b, k, c)) : -1);
    }
    else if (!strcmp(f, "ric_l"))
    {
        *arr = D.lf;
        for (int r = 0; r < nu + nx; r++) map.push_back(k * n + padded_var(b, k, r));
    }
    else if (!strcmp(f, "lam") || !strcmp(f, "t"))
    {
        *arr = f[0] == 'l' ? D.lam : D.t;
        for (int side = 0; side < 2; side++)
        {
            for (int r = 0; r < S.nb; r++) map.push_back(S.o_ct + side * nbg + b->perm[k][r]);
            for (int g = 0; g < ng; g++) map.push_back(S.o_ct + side * nbg + S.nb + g);
        }
        for (int j = 0; j < 2 * ns; j++) map.push_back(S.o_ct + 2 * nbg + j);
    }
    else
        return -1;
    return (int) map.size();
}

/* bit positions of a mask field */
int mask_bits(ocp_qp_gpu_batch *b, const char *f, int k, std::vector<int> &bits)
{
    const GqpStage &S = b->st[k];
    const int nbu = b->nbu[k], nbx = b->nbx[k], nbg = S.nb + S.ng, ns = S.ns;
    bits.clear();
    std::vector<char> is_eq(S.nb, 0);
    for (size_t e = 0; e < b->idxe[k].size(); e++) is_eq[b->idxe[k][e]] = 1;
    auto box = [&](int off, int cnt, int side) {
        for (int e = 0; e < cnt; e++) bits.push_back(is_eq[off + e] ? -1 : side * nbg + b->perm[k][off + e]);
    };
    if (!strcmp(f, "lbu_mask")) box(0, nbu, 0);
    else if (!strcmp(f, "ubu_mask")) box(0, nbu, 1);
    else if (!strcmp(f, "lbx_mask")) box(nbu, nbx, 0);
    else if (!strcmp(f, "ubx_mask")) box(nbu, nbx, 1);
    else if (!strcmp(f, "lg_mask")) for (int g = 0; g < S.ng; g++) bits.push_back(S.nb + g);
    else if (!strcmp(f, "ug_mask")) for (int g = 0; g < S.ng; g++) bits.push_back(nbg + S.nb + g);
    else if (!strcmp(f, "lls_mask")) for (int j = 0; j < ns; j++) bits.push_back(2 * nbg + j);
    else if (!strcmp(f, "lus_mask")) for (int j = 0; j < ns; j++) bits.push_back(2 * nbg + ns + j);
    else return -1;
    return (int) bits.size();
}

int *upload_map(ocp_qp_gpu_batch *b, const std::vector<int> &map)
{
    if ((int) map.size() > b->map_cap)
    {
        b->map_cap = (int) map.size() * 2 + 64;
        b->d_map = dalloc<int>(b, b->map_cap);
    }
    /* the previous launch using d_map must be done before it is overwritten */
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipMemcpy(b->d_map, map.data(), sizeof(int) * map.size(), hipMemcpyHostToDevice));
    return b->d_map;
}

const double *stage_in(ocp_qp_gpu_batch *b, const double *data, size_t cnt, int is_device)
{
    if (is_device) return data;
    if (cnt > b->stage_cap)
    {
        b->stage_cap = cnt * 2;
        b->d_stage = dalloc<double>(b, b->stage_cap);
    }
    HIPCHK(hipMemcpyAsync(b->d_stage, data, sizeof(double) * cnt, hipMemcpyHostToDevice, b->stream));
    return b->d_stage;
}

} // namespace

/* one array of a level <-> the same array of its sub-level (dir 0: gather the listed instances, 1: scatter back);
 * levels of different layouts exchange through the LDS-transposing kernel */
template <class T>
static void copy_level(const GArrT<T> &big, const GArrT<T> &small, const int *d_list, int cnt, int dir, hipStream_t s)
{
    if (!(big.E > 0 && big.p && small.p) || cnt <= 0) return;
    const dim3 grid((cnt + 63) / 64, (big.E + 63) / 64), block(64);
    if (big.aos != small.aos) GQP_LAUNCH_COOP(gqp::k_compact_tile<T>, grid, block, 0, s, big, small, d_list, cnt, dir);
    else hipLaunchKernelGGL(gqp::k_compact_copy<T>, grid, block, 0, s, big, small, d_list, cnt, dir);
}

extern "C" {

/* force_ks: a compaction sub-batch runs the very kernel set of its parent; force_wpi: a tail / sensitivity sub-batch is
 * pinned to the wave-per-instance family at the parent's padded dims.  Plain arguments: the library keeps no state
 * outside the objects its caller owns (SURVEY 8b "Threading"). */
static ocp_qp_gpu_batch *batch_create_shape(int N, const int *nx, const int *nu, const int *nbx, const int *nbu,
                                            const int *ng, const int *ns, int n_batch, int device, int force_NX, int force_NU,
                                            const KernelSet *g_force_ks = nullptr, bool g_force_wpi = false);

ocp_qp_gpu_batch *ocp_qp_gpu_batch_create(int N, const int *nx, const int *nu, const int *nbx, const int *nbu,
                                          const int *ng, const int *ns, int n_batch, int device)
try
{
    return batch_create_shape(N, nx, nu, nbx, nbu, ng, ns, n_batch, device, 0, 0);
}
catch (const gqp_hip_failure &) { return nullptr; }

static ocp_qp_gpu_batch *batch_create_shape(int N, const int *nx, const int *nu, const int *nbx, const int *nbu,
                                            const int *ng, const int *ns, int n_batch, int device, int force_NX, int force_NU,
                                            const KernelSet *g_force_ks, bool g_force_wpi)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    {
        fprintf(stderr, "acados_amd: no HIP device available -- the GPU OCP-QP path cannot run\n");
        return nullptr;
    }
    if (device >= 0) HIPCHK(hipSetDevice(device));
    ocp_qp_gpu_batch *b = new ocp_qp_gpu_batch();
    HIPCHK(hipGetDevice(&b->device));
    b->force_NX = force_NX; b->force_NU = force_NU;
    b->B = n_batch;
    b->Bp = (n_batch + 63) / 64 * 64;
    b->N = N;
    int mx = 0, mu = 0, mg = 0, ms = 0;
    for (int k = 0; k <= N; k++)
    {
        b->nx.push_back(nx[k]); b->nu.push_back(nu[k]); b->nbx.push_back(nbx[k]); b->nbu.push_back(nbu[k]);
        b->nb.push_back(nbx[k] + nbu[k]); b->ng.push_back(ng[k]); b->ns.push_back(ns[k]); b->nbxe.push_back(0);
        mx = std::max(mx, nx[k]); mu = std::max(mu, nu[k]); mg = std::max(mg, ng[k]); ms = std::max(ms, ns[k]);
        std::vector<int> ib;
        for (int i = 0; i < nbu[k]; i++) ib.push_back(i);
        for (int i = 0; i < nbx[k]; i++) ib.push_back(nu[k] + i);
        b->idxb.push_back(ib);
        b->idxs_rev.push_back(std::vector<int>(nbx[k] + nbu[k] + ng[k], -1));
        b->idxe.push_back(std::vector<int>());
    }
    double best = 1e300;
    auto consider = [&](const KernelSet &ks) {
        if (ks.NX < mx || ks.NU < mu || ks.NG < mg || ks.NS < ms) return;
        if (b->force_NX && (ks.NX != b->force_NX || ks.NU != b->force_NU || ks.NG || ks.NS)) return;
        const double n = ks.NX + ks.NU;
        const double cost = n * n * n + 10.0 * (ks.NG + ks.NS) * n * n;
        if (cost < best) { best = cost; b->ks = &ks; }
    };
    for (const KernelSet &ks : g_ksets) consider(ks);
    {
        /* the rolled-loop / scratch-resident instantiations only serve as a cross-check of the wave-per-instance
         * family (ACADOS_AMD_WPI=0); by default a shape no register-resident instantiation covers goes there */
        const char *e0 = getenv("ACADOS_AMD_WPI");
        if (e0 && atoi(e0) == 0)
            for (int q = 0; q < g_n_ksets_large; q++) consider(g_ksets_large[q]);
    }
    if (g_force_ks) b->ks = g_force_ks;
    /* wave-per-instance family (ipm_kernels_wpi.hpp): box-constrained QPs whose stage block is too large for
     * the one-instance-per-lane register mapping.  Dimensions are runtime values there: no padding to a
     * compiled shape.  ACADOS_AMD_WPI=0/1 overrides the size rule (tests). */
    /* inequality sides per stage: 64 per activity word; the one-instance-per-lane kernels read one word, the
     * wave-per-instance ones up to two */
    int nct_max = 0;
    for (int k = 0; k <= N; k++) nct_max = std::max(nct_max, 2 * (nbx[k] + nbu[k] + ng[k]) + 2 * ns[k]);
    if (nct_max > 128)
    {
        fprintf(stderr, "acados_amd: a stage has %d inequality sides (2(nb+ng)+2ns > 128): unsupported\n", nct_max);
        delete b;
        return nullptr;
    }
    const bool need_wpi = nct_max > 64;
    if (need_wpi) b->ks = nullptr; /* no one-instance-per-lane kernel may serve it */
    if (!g_force_ks || g_force_wpi)
    {
        int wx = force_NX ? force_NX : mx, wu = force_NU ? force_NU : mu;
        const bool gen = mg > 0 || ms > 0;
        const char *env = getenv("ACADOS_AMD_WPI");
        const char *v1 = getenv("ACADOS_AMD_WPI_V1");
        const bool ref = v1 && atoi(v1) != 0 && !gen;
        /* size rule: large stage blocks always; small ones while the batch is too small to fill the chip with 64
         * instances per wave (measured crossover on the C2 shape between 4,096 and 16,384 instances,
         * tools/family_crossover.py).  ACADOS_AMD_WPI_BATCH_MAX overrides the batch threshold. */
        const char *bm = getenv("ACADOS_AMD_WPI_BATCH_MAX");
        /* sixteen lanes per instance: the smallest compiled shape that covers the dims (per-stage dims live inside
         * the padded shape); a sub-level created for a hand-over must match its parent's padded dims exactly */
        const bool soft_dims = mg == 0 && ms > 0; /* slacks on box rows only: the SOFT variants, if the structure allows */
        const W16Set *w16 = nullptr;
        {
            const char *e16 = getenv("ACADOS_AMD_W16"), *e16r = getenv("ACADOS_AMD_W16R");
            const char *e16g = getenv("ACADOS_AMD_W16G");
            if (!need_wpi && !ref && !(e16 && atoi(e16) == 0))
                for (const W16Set &ws : g_w16_sets)
                {
                    bool fits = force_NX ? (ws.NX == wx && ws.NU == wu) : (ws.NX >= wx && ws.NU >= wu);
                    if (!gen) fits = fits && ws.fact && ws.NG == 0;
                    else if (soft_dims) fits = fits && ws.sfact && ws.NG == 0;                       /* slacks on box rows only */
                    /* general rows (a small block with a compiled one-instance-per-lane general set -- the golden shared-slack
                     * structure, nx = 4, nu = 1 -- keeps the dispatch it was measured with: gen_small below) */
                    else fits = fits && ws.sfact && ws.NG >= mg && !(e16g && atoi(e16g) == 0) && !(b->ks && b->ks->NX + b->ks->NU <= 6);
                    if (ws.NX + ws.NU > 16 && (e16r && atoi(e16r) == 0)) fits = false;
                    /* two rows per lane: dims run PADDED inside the compiled shape, so its ~2.3x over the wave-per-instance
                     * kernels (which take dims at run time) is gone once the padded block has more than twice the work:
                     * the shape must cover at least 77 % of the compiled nu + nx */
                    if (ws.NX + ws.NU > 16 && !force_NX && 13 * (wx + wu) < 10 * (ws.NX + ws.NU)) fits = false;
                    if (fits && (!w16 || ws.NX + ws.NU < w16->NX + w16->NU)) w16 = &ws;
                }
        }
        const bool has_w16 = w16 != nullptr;
        /* state bounds after stage 0 (the mass-spring class): the one-instance-per-lane kernels of that class run out
         * of registers, the sixteen-lanes kernels win at every batch size measured (tools/xbox_rate.py) */
        bool xbox_dims = false;
        for (int k = 1; k <= N; k++) xbox_dims = xbox_dims || nbx[k] > 0;
        /* nu + nx <= 6: the pipelined one-instance-per-lane kernels (ipm_kernels_box_small.hpp) take over much earlier
         * -- measured crossover on nx = 4, nu = 1 between 2,048 and 4,096 instances (N = 100; 4,096 ... 7,281 at N = 20),
         * with bounds on every state between 7,281 and 16,384; at 65,536 they are 2.7-2.9x (1.4x) ahead
         * (tools/small_shape_crossover.py) */
        const bool kb_small = !gen && b->ks && b->ks->NX + b->ks->NU <= 6;
        /* general rows / shared slacks on a small block with a compiled one-instance-per-lane set (the reference's golden
         * structure pend_idxs_rev_min_qp0: nx = 4, nu = 1, two general rows sharing one slack): 1,024 instances 3.45 ms
         * there against 2.65 ms on the wave-per-instance kernels, 8,192: 4.4 against 9.2 ms, 65,536: 8.4 against 74 ms */
        const bool gen_small = gen && b->ks && b->ks->NX + b->ks->NU <= 6;
        const int batch_max = bm ? atoi(bm)
                            : !has_w16 ? (gen_small ? GQP_WPI_GEN_SMALL_MAX : GQP_WPI_BATCH_MAX)
                            : kb_small ? (xbox_dims ? GQP_W16_SMALL_XBOX_MAX : GQP_W16_SMALL_MAX)
                            /* general rows / slacks on the sixteen-lanes GEN kernels: the one-instance-per-lane alternative is the
                             * general set of a (much) larger padded shape with its blocks in scratch -- never */
                            : (xbox_dims || (gen && !soft_dims && !gen_small) ? INT_MAX : GQP_W16_BATCH_MAX);
        /* (a shape no compiled one-instance-per-lane set covers runs here whatever the override says) */
        const bool want = g_force_wpi || need_wpi || !b->ks || (env ? atoi(env) != 0 : (wx + wu >= GQP_WPI_MIN_N || n_batch <= batch_max));
        if (want && wx + wu <= 64 && wx >= 1 && mg <= 32 && ms <= 32)
        {
            if (w16) { wx = w16->NX; wu = w16->NU; }
            /* factor sweep: register-tile kernel for the tile count of this shape; rhs-only and forward sweeps on
             * the packed factor; GEN variants carry general constraints and slacks.  ACADOS_AMD_WPI_V1=1 selects
             * the plain LDS-resident reference kernels of the family (box-constrained QPs only), kept for
             * cross-checking. */
            static const kern_redo_t fact_box[8] = {gqp::kw_factor<1, false>, gqp::kw_factor<2, false>, gqp::kw_factor<3, false>,
                                                    gqp::kw_factor<4, false>, gqp::kw_factor<5, false>, gqp::kw_factor<6, false>,
                                                    gqp::kw_factor<7, false>, gqp::kw_factor<8, false>};
            static const kern_redo_t fact_gen[8] = {gqp::kw_factor<1, true>, gqp::kw_factor<2, true>, gqp::kw_factor<3, true>,
                                                    gqp::kw_factor<4, true>, gqp::kw_factor<5, true>, gqp::kw_factor<6, true>,
                                                    gqp::kw_factor<7, true>, gqp::kw_factor<8, true>};
            const int t8 = (wx + wu + 7) / 8 - 1;
            /* 17 <= n <= 32: blocked Cholesky + the O(n^3) parts on the FP64 matrix pipe (ipm_kernels_wpi_mfma.hpp).
             * Measured (tools/factor_variants.py, 4,096 instances): C4 class 3.97 vs 4.10 ms per factor launch, box class
             * nx=24 nu=6 2.40 vs 2.07 ms, condensed C3 shape 0.35 vs 0.33 ms -- the FP64 matrix pipe is slower than the
             * vector pipe on this chip (tools/mfma_f64_probe), so it serves the class it helps (general rows + slacks) by
             * default; ACADOS_AMD_WPI_MFMA=1 / 0 forces it on / off for every shape in range */
            const char *emf = getenv("ACADOS_AMD_WPI_MFMA");
            const bool mfma = !ref && !w16 && wx + wu > 16 && wx + wu <= 32 && (emf ? atoi(emf) != 0 : gen);
            kern_redo_t fact_ct = nullptr, rhs_ct = nullptr, faff_ct = nullptr, fcor_ct = nullptr;
            {
                /* compile-time dims where they pay: the condensed C3 shape (nx=8 nu=15, box): factor sweep 0.328 -> 0.287 ms
                 * per 4,096 launch.  Measured and NOT kept: nx=24 nu=3 general (4.09 -> 5.13 ms) and nx=24 nu=6 box (2.07 ->
                 * 2.15 ms) -- full unrolling costs those shapes more in registers than the folded arithmetic saves
                 * (tools/factor_variants.py).  ACADOS_AMD_WPI_CT=0 keeps the run-time-shaped instantiations */
                const char *ect = getenv("ACADOS_AMD_WPI_CT");
                if (!ref && !w16 && !(ect && atoi(ect) == 0))
                {
#define GQP_CT(GENV, X, U, T)                                                                                     \
    if (gen == GENV && wx == X && wu == U)                                                                        \
    {                                                                                                             \
        fact_ct = gqp::kw_factor<T, GENV, X, U>; rhs_ct = gqp::kw_backrhs<GENV, X, U>;                               \
        faff_ct = gqp::kw_fwd<false, GENV, X, U>; fcor_ct = gqp::kw_fwd<true, GENV, X, U>;                             \
    }
                    GQP_CT(false, 8, 15, 3)
#undef GQP_CT
                }
            }
            static const kern_redo_t fact_mf[2][3] = {{gqp::kw_factor_m<false, 0>, gqp::kw_factor_m<false, 1>, gqp::kw_factor_m<false, 2>},
                                                      {gqp::kw_factor_m<true, 0>, gqp::kw_factor_m<true, 1>, gqp::kw_factor_m<true, 2>}};
            const char *epf = getenv("ACADOS_AMD_WPI_MFMA_PF");
            const int pf = epf ? std::max(0, std::min(2, atoi(epf))) : 0; /* register prefetch did not pay (VGPR pressure) */
            const kern_redo_t fact = ref ? gqp::kw_backward<true> : mfma ? fact_mf[gen ? 1 : 0][pf] : fact_ct ? fact_ct : gen ? fact_gen[t8] : fact_box[t8];
            b->wpi_mfma = mfma;
            kern_redo_t rhs = ref ? gqp::kw_backward<false> : gen ? gqp::kw_backrhs<true> : gqp::kw_backrhs<false>;
            kern_redo_t faff = ref ? gqp::kw_forward<false> : gen ? gqp::kw_fwd<false, true> : gqp::kw_fwd<false, false>;
            kern_redo_t fcor = ref ? gqp::kw_forward<true> : gen ? gqp::kw_fwd<true, true> : gqp::kw_fwd<true, false>;
            if (rhs_ct) { rhs = rhs_ct; faff = faff_ct; fcor = fcor_ct; }

            const kern_opts_t init = gen ? gqp::kw_init<true> : gqp::kw_init<false>;
            const kern_plain_t fin = gen ? gqp::kw_finalize<true> : gqp::kw_finalize<false>;
            b->own_ks = KernelSet{wx, wu, mg, ms, init, fact, rhs, faff, fcor, fin,
                                  {fact, fact}, {rhs, rhs}, {faff, faff}, {fcor, fcor}, fin};
            b->ks = &b->own_ks;
            b->wpi = 1;
            b->aos = 1;
            b->AW = need_wpi ? 2 : 1;
            /* small box-constrained stage blocks: four instances per wave, register rows + DPP row broadcasts */
            if (w16)
            {
                const W16Set &ws = *w16;
                b->wpi_ks = b->own_ks; /* what the batch falls back to if the slack structure is not one-slack-per-box-row */
                const char *et = getenv("ACADOS_AMD_W16T"); /* 0: the factor sweep of the two-rows family on register rows (ky_factor) */
                const char *etg = getenv("ACADOS_AMD_W16T_GEN"); /* ... of the GEN instantiations alone */
                const bool tiles = ws.tfact && !(et && atoi(et) == 0) && !(gen && etg && atoi(etg) == 0);
                const kern_redo_t kf = tiles ? ws.tfact : (gen ? ws.sfact : ws.fact), kr = gen ? ws.srhs : ws.rhs;
                const kern_redo_t ka = gen ? ws.sfaff : ws.faff, kc = gen ? ws.sfcor : ws.fcor;
                b->own_ks.back_fact = kf; b->own_ks.back_rhs = kr; b->own_ks.fwd_aff = ka; b->own_ks.fwd_corr = kc;
                for (int q = 0; q < 2; q++)
                {
                    b->own_ks.box_fact[q] = kf; b->own_ks.box_rhs[q] = kr;
                    b->own_ks.box_fwd_aff[q] = ka; b->own_ks.box_fwd_corr[q] = kc;
                }
                b->w16 = 1;
                b->w16_slots = n_batch;
                b->w16_soft = gen;
                b->w16_ng = ws.NG;
                b->w16_shmem = ws.shmem;
                b->w16_shmem_fact = tiles ? ws.tshmem : ws.shmem;
                b->w16_tiles = tiles;
                if (const char *ea = getenv("ACADOS_AMD_W16_LDS_ALIGN")) /* development: allocation rounded up / padded */
                {
                    const size_t a = (size_t) atoi(ea);
                    if (a > 1) { b->w16_shmem = (b->w16_shmem + a - 1) / a * a; b->w16_shmem_fact = (b->w16_shmem_fact + a - 1) / a * a; }
                }
                b->w16_solve = gen ? ws.ssolve : ws.solve;
            }
            const size_t con = gqp::wpi_con_doubles(wx + wu, mg, ms);
            b->shmem = (ref ? gqp::wpi_lds_doubles(wx, wu) : gqp::wpi3_lds_doubles(wx, wu) + con) * sizeof(double);
            b->shmem_fwd = (ref ? gqp::wpi_lds_doubles(wx, wu) : gqp::wpi3_lds_doubles(wx, wu, 1) + con) * sizeof(double);
            b->shmem_fact = (ref ? gqp::wpi_lds_doubles(wx, wu) : (mfma ? gqp::wpim_lds_doubles(wx, wu) : gqp::wpi2_lds_doubles(wx, wu)) + con) * sizeof(double);
        }
    }
    if (!b->ks)
    {
        fprintf(stderr, "acados_amd: no kernel instantiation covers nx<=%d nu<=%d ng<=%d ns<=%d\n", mx, mu, mg, ms);
        delete b;
        return nullptr;
    }
    char nm[128];
    if (b->w16 && b->w16_ng) snprintf(nm, sizeof(nm), "w16r-gen<NX=%d,NU=%d,NG=%d>", b->ks->NX, b->ks->NU, b->w16_ng);
    else if (b->w16) snprintf(nm, sizeof(nm), b->w16_soft ? "w16-soft<NX=%d,NU=%d>" : b->ks->NX + b->ks->NU > 16 ? "w16r-box<NX=%d,NU=%d>" : "w16-box<NX=%d,NU=%d>", b->ks->NX, b->ks->NU);
    else if (b->wpi && (b->ks->NG || b->ks->NS))
        snprintf(nm, sizeof(nm), "wpi-gen(nx=%d,nu=%d,ng=%d,ns=%d,lds=%zuB%s)", b->ks->NX, b->ks->NU, b->ks->NG, b->ks->NS, b->shmem_fact, b->wpi_mfma ? ",mfma" : "");
    else if (b->wpi) snprintf(nm, sizeof(nm), "wpi-box(nx=%d,nu=%d,lds=%zuB%s)", b->ks->NX, b->ks->NU, b->wpi_mfma ? b->shmem_fact : b->shmem, b->wpi_mfma ? ",mfma" : "");
    else snprintf(nm, sizeof(nm), "1tpi<NX=%d,NU=%d,NG=%d,NS=%d>", b->ks->NX, b->ks->NU, b->ks->NG, b->ks->NS);
    b->kname = nm;
    opts_default(b->O);
    HIPCHK(hipStreamCreate(&b->stream));
    HIPCHK(hipEventCreate(&b->ev0));
    HIPCHK(hipEventCreate(&b->ev1));
    HIPCHK(hipHostMalloc((void **) &b->h_nact, sizeof(int)));
    HIPCHK(hipHostMalloc((void **) &b->h_ints, sizeof(int) * 2 * (size_t) b->Bp));
    return b;
}

void ocp_qp_gpu_batch_destroy(ocp_qp_gpu_batch *b)
try
{
    if (!b) return;
    (void) hipSetDevice(b->device);
    (void) hipStreamSynchronize(b->stream);
    for (void *p : b->allocs) (void) hipFree(p);
    (void) hipHostFree(b->h_nact);
    (void) hipHostFree(b->h_ints);
    (void) hipEventDestroy(b->ev0);
    (void) hipEventDestroy(b->ev1);
    if (b->chunks_ev) (void) hipEventDestroy(b->chunks_ev);
    for (hipEvent_t e : b->prof_ev) (void) hipEventDestroy(e);
    (void) hipStreamDestroy(b->stream);
    if (b->child) ocp_qp_gpu_batch_destroy(b->child);
    if (b->compact) ocp_qp_gpu_batch_destroy(b->compact);
    if (b->tail) ocp_qp_gpu_batch_destroy(b->tail);
    if (b->sens_child) ocp_qp_gpu_batch_destroy(b->sens_child);
    delete b;
}
catch (const gqp_hip_failure &) {}

int ocp_qp_gpu_batch_set_int(ocp_qp_gpu_batch *b, const char *f, int k, const int *v, int cnt)
try
{
    if (b->finalized)
    {
        fprintf(stderr, "acados_amd: structure field %s must be set before numeric data\n", f);
        return -1;
    }
    if (k < 0 || k > b->N) return -1;
    if (!strcmp(f, "idxb")) { b->idxb[k].assign(v, v + b->nb[k]); return 0; }
    if (!strcmp(f, "idxbu")) { for (int i = 0; i < b->nbu[k]; i++) b->idxb[k][i] = v[i]; return 0; }
    if (!strcmp(f, "idxbx")) { for (int i = 0; i < b->nbx[k]; i++) b->idxb[k][b->nbu[k] + i] = b->nu[k] + v[i]; return 0; }
    if (!strcmp(f, "idxs_rev")) { b->idxs_rev[k].assign(v, v + b->nb[k] + b->ng[k]); return 0; }
    if (!strcmp(f, "idxe")) { b->idxe[k].assign(v, v + cnt); b->nbxe[k] = cnt; return 0; }
    fprintf(stderr, "acados_amd: ocp_qp_gpu_batch_set_int: unknown field %s\n", f);
    return -1;
}
catch (const gqp_hip_failure &) { return -1; }

int ocp_qp_gpu_batch_set(ocp_qp_gpu_batch *b, const char *f, int stage, const double *data, int is_device)
try
{
    HIPCHK(hipSetDevice(b->device));
    finalize_structure(b);
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, b->stream));
    const int k0 = stage < 0 ? 0 : stage, k1 = stage < 0 ? b->N : stage;
    const int grid = (b->B + 63) / 64;
    int rc = 0;
    const double *src = nullptr;
    int src_len = -1;
    for (int k = k0; k <= k1; k++)
    {
        std::vector<int> map, map2;
        GArr arr = {nullptr, 0, 0}, arr2 = {nullptr, 0, 0};
        const size_t flen = strlen(f);
        if (flen > 5 && !strcmp(f + flen - 5, "_mask"))
        {
            const int len = mask_bits(b, f, k, map);
            if (len < 0) { rc = -1; break; }
            if (len == 0) continue;
            if (src_len != len) { src = stage_in(b, data, (size_t) b->B * len, is_device); src_len = len; }
            int *dm = upload_map(b, map);
            hipLaunchKernelGGL(gqp::k_setmask, dim3(grid), dim3(64), 0, b->stream, src, b->B, len, dm, b->D.amask, k, b->AW);
            continue;
        }
        const int len = field_map(b, f, k, map, &arr, &map2, &arr2);
        if (len < 0)
        {
            if (stage < 0) continue; /* field absent at this stage (e.g. A at N) */
            rc = -1;
            break;
        }
        if (len == 0) continue;
        if (src_len != len) { src = stage_in(b, data, (size_t) b->B * len, is_device); src_len = len; }
        int *dm = upload_map(b, map);
        hipLaunchKernelGGL(gqp::k_scatter, dim3(grid), dim3(64), 0, b->stream, src, b->B, len, dm, arr);
        if (arr2.p)
        {
            dm = upload_map(b, map2);
            hipLaunchKernelGGL(gqp::k_scatter, dim3(grid), dim3(64), 0, b->stream, src, b->B, len, dm, arr2);
        }
    }
    HIPCHK(hipEventRecord(e1, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    b->time_pack += ms * 1e-3;
    HIPCHK(hipEventDestroy(e0)); HIPCHK(hipEventDestroy(e1));
    if (rc) fprintf(stderr, "acados_amd: ocp_qp_gpu_batch_set: unknown field %s (stage %d)\n", f, stage);
    return rc;
}
catch (const gqp_hip_failure &) { return -1; }

/* (re)create the batch's stream with a priority; the batch must be idle */
static void set_stream_priority(ocp_qp_gpu_batch *b, int prio)
{
    int lo = 0, hi = 0; /* hip: "least" (numerically largest) and "greatest" (numerically smallest) priority */
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    const int p = prio < hi ? hi : (prio > lo ? lo : prio);
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipStreamDestroy(b->stream));
    HIPCHK(hipStreamCreateWithPriority(&b->stream, hipStreamDefault, p));
    b->stream_priority = prio;
    /* sub-batches that exist already and launch on streams of their own (the sensitivity slices, the condensed batch) follow;
     * the ones created later take the priority at creation (round-3 advice) */
    if (b->sens_child) set_stream_priority(b->sens_child, prio);
    if (b->child) set_stream_priority(b->child, prio);
}

int ocp_qp_gpu_batch_opts_set(ocp_qp_gpu_batch *b, const char *f, const void *v)
try
{
    GqpOpts &o = b->O;
    const double *d = (const double *) v;
    const int *i = (const int *) v;
    if (!strcmp(f, "iter_max")) o.iter_max = *i;
    else if (!strcmp(f, "tol_stat")) o.tol_stat = *d;
    else if (!strcmp(f, "tol_eq")) o.tol_eq = *d;
    else if (!strcmp(f, "tol_ineq")) o.tol_ineq = *d;
    else if (!strcmp(f, "tol_comp")) o.tol_comp = *d;
    else if (!strcmp(f, "warm_start")) o.warm_start = *i;
    else if (!strcmp(f, "mu0")) { if (*d > 0.0) o.mu0 = *d; }
    else if (!strcmp(f, "alpha_min")) o.alpha_min = *d;
    else if (!strcmp(f, "tau_min")) o.tau_min = *d;
    else if (!strcmp(f, "tol_comp_soft_scale")) b->tol_comp_soft_scale = *d;
    else if (!strcmp(f, "full_dense")) b->full_dense = *i != 0;
    else if (!strcmp(f, "polish")) b->polish = *i < 0 ? 0 : (*i > 8 ? 8 : *i);
    else if (!strcmp(f, "polish_ratio")) b->polish_ratio = *d;
    else if (!strcmp(f, "polish_min")) b->polish_min = *d;
    else if (!strcmp(f, "reg_prim")) o.reg_prim = *d;
    else if (!strcmp(f, "cond_pred_corr")) o.cond_pred_corr = *i;
    else if (!strcmp(f, "print_level")) b->print_level = *i;
    else if (!strcmp(f, "profile")) b->profile = *i;
    else if (!strcmp(f, "cond_keep_iterate")) b->cond_keep_iterate = *i;
    else if (!strcmp(f, "marker"))
    {
        /* one empty launch named k_marker<id> on the batch's stream (id 0..15): section mark for rocprofv3 summaries */
        typedef void (*marker_t)(int *);
        static const marker_t mk[16] = {gqp::k_marker<0>, gqp::k_marker<1>, gqp::k_marker<2>, gqp::k_marker<3>, gqp::k_marker<4>,
                                        gqp::k_marker<5>, gqp::k_marker<6>, gqp::k_marker<7>, gqp::k_marker<8>, gqp::k_marker<9>,
                                        gqp::k_marker<10>, gqp::k_marker<11>, gqp::k_marker<12>, gqp::k_marker<13>,
                                        gqp::k_marker<14>, gqp::k_marker<15>};
        HIPCHK(hipSetDevice(b->device));
        hipLaunchKernelGGL(mk[*i & 15], dim3(1), dim3(64), 0, b->stream, (int *) nullptr);
        HIPCHK(hipStreamSynchronize(b->stream));
    }
    else if (!strcmp(f, "stream_priority"))
    {
        /* several batches solved concurrently from host threads (the C5 classes): the long ones on a high-priority
         * stream (negative value, clamped to the device's range) are dispatched first, the short ones fill the gaps */
        set_stream_priority(b, *i); /* (the condensed batch and the sensitivity slices follow inside) */
        if (b->tail) set_stream_priority(b->tail, *i);
    }
    else if (!strcmp(f, "compact_min")) b->compact_min = *i;
    else if (!strcmp(f, "tail_max")) b->tail_max = *i;
    else if (!strcmp(f, "tail_div")) b->tail_div = *i < 1 ? 1 : *i;
    else if (!strcmp(f, "solve_max")) b->solve_max = *i;
    else if (!strcmp(f, "cond_N"))
    {
        if (*i != b->cond_N)
        {
            b->cond_N = *i;
            b->user_blocks.clear();
            b->pcond_state = 0;
            if (b->child) { ocp_qp_gpu_batch_destroy(b->child); b->child = nullptr; }
        }
    }
    else if (!strcmp(f, "cond_block_size"))
    {
        /* user block sizes, N2 + 1 entries as ocp_qp_partial_condensing.c:305-313; set cond_N first.  They must sum to
         * N (:346-356).  A non-zero LAST entry (the reference's own test: qp_solver_cond_block_size = [6, 5, 4, 2, 2, 1],
         * examples/acados_python/tests/pcond_getters_test.py:200) asks HPIPM to fold the last stages' inputs into the terminal
         * stage; here those stages form one more block in front of an input-free terminal stage (the condensed QP has
         * N2 + 1 stages with inputs instead of N2 -- the solution of the original QP is the same, the layout of the condensed
         * one differs: INTEGRATION.md 3). */
        const int N2 = b->cond_N;
        if (N2 <= 0 || N2 >= b->N) { fprintf(stderr, "acados_amd: cond_block_size needs cond_N in 1..N-1 first\n"); return -1; }
        int sum = 0;
        for (int j = 0; j <= N2; j++) sum += i[j];
        bool ok = sum == b->N && i[N2] >= 0;
        for (int j = 0; j < N2; j++) ok = ok && i[j] >= 1;
        if (!ok)
        {
            fprintf(stderr, "acados_amd: partial condensing: block sizes must be >= 1 (the last one >= 0) and sum to N = %d (got %d)\n", b->N, sum);
            return -1;
        }
        const std::vector<int> blocks(i, i + N2 + (i[N2] > 0 ? 1 : 0));
        if (blocks != b->user_blocks) /* the same sizes again (an adapter sends its options before every solve): nothing to redo */
        {
            b->user_blocks = blocks;
            b->pcond_state = 0;
            if (b->child) { ocp_qp_gpu_batch_destroy(b->child); b->child = nullptr; }
        }
    }
    else if (!strcmp(f, "t0_min")) b->t0_min = *d;
    else if (!strcmp(f, "lam0_min")) b->lam0_min = *d;
    else if (!strcmp(f, "update_fact_exit")) { /* the factor sweep factorises before it decides: the factor at the exit
                                                  point is always there (and re-done lazily after a hand-over) */ }
    else if (!strcmp(f, "t0_init"))
    {
        /* acados_ocp_options.py:1128-1143: 0 lam = t = sqrt(mu0); 1 lam = mu0, t = 1; 2 from the constraint residuals */
        if (*i < 0 || *i > 2) { fprintf(stderr, "acados_amd: t0_init must be 0, 1 or 2, got %d\n", *i); return -1; }
        o.t0_init = *i;
    }
    else if (!strcmp(f, "ric_alg"))
    {
        if (*i != 1) fprintf(stderr, "acados_amd: ric_alg=%d requested, only the square-root Riccati (1) is implemented\n", *i);
    }
    else if (!strcmp(f, "hpipm_mode"))
    {
        /* modes only reset defaults in the reference (ocp_qp_hpipm.c:142-165) */
        const char *mode = (const char *) v;
        if (strcmp(mode, "BALANCE") && strcmp(mode, "SPEED") && strcmp(mode, "SPEED_ABS") && strcmp(mode, "ROBUST"))
        {
            fprintf(stderr, "acados_amd: got non-supported mode %s\n", mode);
            return -1;
        }
        const double t1 = o.tol_stat, t2 = o.tol_eq, t3 = o.tol_ineq, t4 = o.tol_comp;
        const int im = o.iter_max, ws = o.warm_start;
        opts_default(o);
        o.tol_stat = t1; o.tol_eq = t2; o.tol_ineq = t3; o.tol_comp = t4; o.iter_max = im; o.warm_start = ws;
    }
    else
    {
        fprintf(stderr, "acados_amd: ocp_qp_gpu_batch_opts_set: unknown option %s\n", f);
        return -1;
    }
    return 0;
}
catch (const gqp_hip_failure &) { return -1; }


/* ---- partial condensing (ocp_qp_partial_condensing.c:523-556, :664-689) ---- */
static void pcond_setup(ocp_qp_gpu_batch *b)
{
    /* blocks with inputs: cond_N of them, one more when the user's last block size is not 0 (see "cond_block_size") */
    const int N = b->N, N2 = (int) b->user_blocks.size() == b->cond_N + 1 ? b->cond_N + 1 : b->cond_N;
    b->pcond_state = -1;
    auto decline = [&](const char *why) {
        if (b->full_dense) return; /* the caller knows: this batch only probes whether ONE block fits a condensed stage (FULL_CONDENSING) */
        fprintf(stderr, "acados_amd: cond_N=%d requested but %s; solving the full-space QP (N2 = N, the default of "
                        "ocp_qp_partial_condensing.c:243-265) -- the solution is identical\n", N2, why);
    };
    if (b->cond_N <= 0 || b->cond_N >= N) return;
    /* block sizes as d_part_cond_qp_compute_block_size: N/N2 each, remainder to the first blocks */
    b->blk_start.assign(N2 + 1, 0);
    int bsmax = 0;
    for (int j = 0; j < N2; j++)
    {
        const int bs = (int) b->user_blocks.size() == N2 ? b->user_blocks[j] : N / N2 + (j < N % N2 ? 1 : 0);
        b->blk_start[j + 1] = b->blk_start[j] + bs;
        bsmax = std::max(bsmax, bs);
    }
    std::vector<char> is_start(N + 1, 0);
    for (int j = 0; j <= N2; j++) is_start[b->blk_start[j]] = 1;
    /* box-only class: input bounds anywhere, state bounds at block starts only -- every child row is a box row;
     * anything else (state bounds inside a block, general rows, slacks) makes general rows in the condensed stages,
     * which only the wave-per-instance condensing kernels write */
    bool box_class = true;
    for (int k = 0; k <= N; k++)
        if (b->ng[k] || b->ns[k] || (b->nbx[k] && !is_start[k])) box_class = false;
    b->pc = nullptr;
    b->pc_rt = 0;
    for (int q = 0; q < g_n_pcond_sets; q++)
        if (g_pcond_sets[q].NX == b->ks->NX && g_pcond_sets[q].NU == b->ks->NU && g_pcond_sets[q].BSMAX >= bsmax &&
            (!b->pc || g_pcond_sets[q].BSMAX < b->pc->BSMAX))
            b->pc = &g_pcond_sets[q];
    /* the wave-per-instance condensing kernels take every shape with nx + bs*nu <= 64 at run time; the compiled
     * one-instance-per-lane ones remain as the cross-check (ACADOS_AMD_PCOND_1TPI=1) */
    {
        const char *e1 = getenv("ACADOS_AMD_PCOND_1TPI");
        const bool want_1tpi = e1 && atoi(e1) != 0 && box_class;
        if (!(want_1tpi && b->pc) && b->ks->NX + bsmax * b->ks->NU <= 64) b->pc_rt = 1;
    }
    /* Expansion of a WAVE-TILED parent: element e of eight neighbouring instances shares a 64-byte line, so a kernel that
     * walks one instance per wave moves eight lines for every one it uses -- kw_pexpand is bound by that traffic (13.3 ms
     * per 65,536 instances of the C2 shape).  The compiled one-instance-per-lane kernel reads the same data coalesced:
     * 1.6 ms.  (Condensing stays with kw_pcond: the per-lane block matrices of k_pcond do not fit the register file.) */
    {
        const char *e2 = getenv("ACADOS_AMD_PCOND_LANE_EXPAND");
        b->pc_lane_expand = b->pc && b->pc->BSMAX == bsmax && box_class && !b->aos && !(e2 && atoi(e2) == 0);
    }
    b->pcz = nullptr;
    {
        const char *e3 = getenv("ACADOS_AMD_PCOND_W16");
        if (box_class && !(e3 && atoi(e3) == 0))
            for (const PcondzSet &z : g_pcondz_sets)
                if (z.NX == b->ks->NX && z.NU == b->ks->NU && z.BS == bsmax) b->pcz = &z;
        const char *e4 = getenv("ACADOS_AMD_PCOND_MFMA");
        b->pcz_mfma = !(e4 && atoi(e4) == 0);
    }
    if (!box_class && !b->pc_rt) { decline("the condensed stage (nx + block size * nu > 64) is beyond the condensing kernels"); return; }
    if (!b->pc && !b->pc_rt) { decline("no condensing kernel covers this shape / block size"); return; }
    const int NU = b->ks->NU, BS = b->pc_rt ? bsmax : b->pc->BSMAX;
    b->pc_shmem = gqp::pcondw_lds_doubles(b->ks->NX, NU, BS * NU) * sizeof(double);

    /* child dims + structure.  Child rows in their original order: [input boxes of the block's stages][state boxes of
     * the block's first stage][general rows: per stage of the block, its state boxes (inner stages) then its general
     * rows]; child slacks: the block's stages one after the other */
    std::vector<int> cnx(N2 + 1), cnu(N2 + 1), cnbx(N2 + 1), cnbu(N2 + 1), cng(N2 + 1, 0), cns(N2 + 1, 0);
    std::vector<std::vector<int>> cidxb(N2 + 1), row_kp(N2 + 1), row_op(N2 + 1), crev(N2 + 1);
    std::vector<int> cidxe, h_gk(N + 2, 0), h_grp, h_gvar, h_soff(N2 + 2, 0), h_skp, h_ssp, slack_base(N + 1, 0);
    for (int j = 0; j <= N2; j++)
    {
        const int k0 = b->blk_start[j], k1 = j < N2 ? b->blk_start[j + 1] : N + 1;
        const int kend = j < N2 ? k1 : k0 + 1; /* stages of this block */
        cnx[j] = b->nx[k0];
        cnu[j] = j < N2 ? (k1 - k0) * NU : 0;
        for (int k = k0; k < (j < N2 ? k1 : k0); k++)
            for (int r = 0; r < b->nbu[k]; r++)
            {
                cidxb[j].push_back((k - k0) * NU + b->idxb[k][r]);
                row_kp[j].push_back(k); row_op[j].push_back(r);
            }
        cnbu[j] = (int) cidxb[j].size();
        for (int r = 0; r < b->nbx[k0]; r++)
        {
            cidxb[j].push_back(cnu[j] + (b->idxb[k0][b->nbu[k0] + r] - b->nu[k0]));
            row_kp[j].push_back(k0); row_op[j].push_back(b->nbu[k0] + r);
        }
        cnbx[j] = b->nbx[k0];
        if (j == 0)
            for (size_t e = 0; e < b->idxe[0].size(); e++) cidxe.push_back(cnbu[0] + (b->idxe[0][e] - b->nbu[0]));
        if (j > 0 && !b->idxe[k0].empty()) { decline("equality-flagged bounds after stage 0"); return; }
        if (j == N2 && b->nbu[N]) { decline("input bounds at the terminal stage"); return; }
        for (int k = k0; k < kend; k++)
        {
            if (k > k0 && !b->idxe[k].empty()) { decline("equality-flagged bounds after stage 0"); return; }
            h_gk[k] = (int) h_grp.size();
            if (k > k0)
                for (int r = b->nbu[k]; r < b->nb[k]; r++)
                {
                    h_grp.push_back(b->perm[k][r]); h_gvar.push_back(b->idxb[k][r] - b->nu[k]);
                    row_kp[j].push_back(k); row_op[j].push_back(r);
                }
            for (int g = 0; g < b->ng[k]; g++)
            {
                h_grp.push_back(b->nb[k] + g); h_gvar.push_back(0);
                row_kp[j].push_back(k); row_op[j].push_back(b->nb[k] + g);
            }
            slack_base[k] = cns[j];
            for (int q = 0; q < b->ns[k]; q++) { h_skp.push_back(k); h_ssp.push_back(q); }
            cns[j] += b->ns[k];
        }
        cng[j] = (int) row_kp[j].size() - (int) cidxb[j].size();
        h_soff[j + 1] = h_soff[j] + cns[j];
        const int nrow = (int) row_kp[j].size();
        if (nrow > GQP_MAX_ROWS || 2 * nrow + 2 * cns[j] > 128)
        {
            decline("a condensed stage would carry more than 64 inequality rows / 128 sides");
            return;
        }
        crev[j].assign(nrow, -1);
        for (int oc = 0; oc < nrow; oc++)
        {
            const int k = row_kp[j][oc], sj = b->idxs_rev[k].empty() ? -1 : b->idxs_rev[k][row_op[j][oc]];
            if (sj >= 0) crev[j][oc] = slack_base[k] + sj;
        }
    }
    h_gk[N + 1] = (int) h_grp.size();
    ocp_qp_gpu_batch *c = batch_create_shape(N2, cnx.data(), cnu.data(), cnbx.data(), cnbu.data(), cng.data(), cns.data(),
                                             b->B, b->device, b->ks->NX, BS * NU);
    if (c && b->stream_priority) set_stream_priority(c, b->stream_priority);
    if (!c) { decline("the condensed shape has no kernel instantiation"); return; }
    for (int j = 0; j <= N2; j++)
    {
        if (!cidxb[j].empty()) ocp_qp_gpu_batch_set_int(c, "idxb", j, cidxb[j].data(), (int) cidxb[j].size());
        if (cns[j]) ocp_qp_gpu_batch_set_int(c, "idxs_rev", j, crev[j].data(), (int) crev[j].size());
    }
    ocp_qp_gpu_batch_set_int(c, "idxe", 0, cidxe.data(), (int) cidxe.size());
    finalize_structure(c);
    /* row maps in the sorted (device) row order of both batches: box rows sorted by variable, general rows as listed */
    std::vector<int> h_off(N2 + 2, 0), h_kp, h_rp;
    for (int j = 0; j <= N2; j++)
    {
        const int nb = (int) cidxb[j].size(), nrow = (int) row_kp[j].size();
        std::vector<int> kp(nrow), rp(nrow);
        for (int oc = 0; oc < nrow; oc++)
        {
            const int rc = oc < nb ? c->perm[j][oc] : oc;
            const int k = row_kp[j][oc], op = row_op[j][oc];
            kp[rc] = k;
            rp[rc] = op < b->nb[k] ? b->perm[k][op] : op;
        }
        h_kp.insert(h_kp.end(), kp.begin(), kp.end());
        h_rp.insert(h_rp.end(), rp.begin(), rp.end());
        h_off[j + 1] = h_off[j] + nrow;
    }
    auto up = [&](const std::vector<int> &v) {
        int *d = dalloc<int>(b, std::max<size_t>(1, v.size()));
        if (!v.empty()) HIPCHK(hipMemcpy(d, v.data(), sizeof(int) * v.size(), hipMemcpyHostToDevice));
        return d;
    };
    int *d_start = up(b->blk_start), *d_kp = up(h_kp), *d_rp = up(h_rp), *d_off = up(h_off);
    b->pmap.gk_off = up(h_gk); b->pmap.g_rp = up(h_grp); b->pmap.g_var = up(h_gvar);
    b->pmap.slk_off = up(h_soff); b->pmap.slk_kp = up(h_skp); b->pmap.slk_sp = up(h_ssp);
    b->pmap.blk_start = d_start; b->pmap.row_kp = d_kp; b->pmap.row_rp = d_rp; b->pmap.row_off = d_off; b->pmap.N2 = N2;
    b->child = c;
    b->pcond_state = 1;
}

static void pcond_launch(ocp_qp_gpu_batch *b, bool expand)
{
    ocp_qp_gpu_batch *c = b->child;
    if (!expand && b->pcz && b->pc_rt && b->AW <= 1 && c->AW <= 1)
    {
        const dim3 grid((b->B + 3) / 4, b->pmap.N2 + 1);
        if (b->pcz_mfma)
            GQP_LAUNCH_COOP(b->pcz->cond_m, dim3((b->B + KM_PCOND_THREADS / 16 - 1) / (KM_PCOND_THREADS / 16), b->pmap.N2 + 1), dim3(KM_PCOND_THREADS), 0,
                            b->stream, b->D, c->D, b->pmap);
        else GQP_LAUNCH_COOP(b->pcz->cond, grid, dim3(64), b->pcz->shmem, b->stream, b->D, c->D, b->pmap);
        return;
    }
    if ((b->pc_rt || b->AW > 1 || c->AW > 1) && !(expand && b->pc_lane_expand && b->AW <= 1 && c->AW <= 1))
    {
        /* (the expansion keeps four 64-entry vectors in LDS, not the condensing's block matrices: with the small allocation a CU holds
         *  all the waves its SIMDs take) */
        if (expand) GQP_LAUNCH_COOP(gqp::kw_pexpand, dim3(b->Bp, b->pmap.N2 + 1), dim3(64), 4 * 64 * sizeof(double), b->stream, b->D, c->D, b->pmap);
        else GQP_LAUNCH_COOP(gqp::kw_pcond, dim3(b->Bp), dim3(64), b->pc_shmem, b->stream, b->D, c->D, b->pmap);
    }
    else
    {
        const kern_pcond_t kern = expand ? b->pc->expand : b->pc->cond;
        hipLaunchKernelGGL(kern, dim3((b->B + 63) / 64), dim3(64), 0, b->stream, b->D, c->D, b->pmap);
    }
}

static int pcond_solve(ocp_qp_gpu_batch *b, int mode = 3)
{
    ocp_qp_gpu_batch *c = b->child;
    b->pmap.mode = mode;
    const dim3 grid((b->B + 63) / 64), block(64);
    hipEvent_t e0, e1, e2, e3;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventCreate(&e2)); HIPCHK(hipEventCreate(&e3));
    c->O = b->O;
    c->tol_comp_soft_scale = b->tol_comp_soft_scale;
    c->polish = b->polish; c->polish_ratio = b->polish_ratio; c->polish_min = b->polish_min;
    c->print_level = b->print_level;
    HIPCHK(hipEventRecord(e0, b->stream));
    pcond_launch(b, false);
    if (b->O.warm_start >= 2 && !b->cond_keep_iterate)
        /* hot start: the root's iterate restated in the condensed variables (condense_qp_out,
         * ocp_qp_xcond_solver.c:554-565) is the child's starting point -- unless the caller keeps the child's own last
         * iterate (the reference's default: initialize_next_xcond_qp_from_qp_out clear) */
        hipLaunchKernelGGL(gqp::k_pcond_sol, grid, block, 0, b->stream, b->D, c->D, b->pmap);
    HIPCHK(hipEventRecord(e1, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    c->t0_min = b->t0_min; c->lam0_min = b->lam0_min;
    c->profile = b->profile; /* per-class event times of the condensed solve are reported on the root */
    const int bad = ocp_qp_gpu_batch_solve(c);
    for (int q = 0; q < 6; q++)
    {
        b->prof_ms[q] += c->prof_ms[q]; b->prof_cnt[q] += c->prof_cnt[q];
        c->prof_ms[q] = 0.0; c->prof_cnt[q] = 0;
    }
    HIPCHK(hipEventRecord(e2, b->stream));
    pcond_launch(b, true);
    HIPCHK(hipEventRecord(e3, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipGetLastError());
    float m0 = 0.f, m1 = 0.f;
    HIPCHK(hipEventElapsedTime(&m0, e0, e1));
    HIPCHK(hipEventElapsedTime(&m1, e2, e3));
    b->time_xcond = (m0 + m1) * 1e-3;
    b->time_tot = c->time_tot + b->time_xcond;
    b->last_iters = c->last_iters;
    b->launches = c->launches + 2;
    HIPCHK(hipEventDestroy(e0)); HIPCHK(hipEventDestroy(e1)); HIPCHK(hipEventDestroy(e2)); HIPCHK(hipEventDestroy(e3));
    return bad;
}

/* ---- the IPM launch loop of one level (root batch or compaction sub-batch) ---- */
struct IpmKernels
{
    kern_redo_t fact, rhs, faff, fcorr;
    kern_plain_t final_;
};

/* launch geometry of the IPM kernels of one level: 64 instances per single-wave block (one instance per
 * lane) or one single-wave block per instance with the stage matrices in dynamic LDS (wpi) */
#define GQP_IPM_LAUNCH_SHM(b, kern, shm, s, ...)                                                              \
    do {                                                                                                      \
        if ((b)->wpi) GQP_LAUNCH_COOP(kern, dim3((b)->B), dim3(64), shm, s, __VA_ARGS__);                     \
        else hipLaunchKernelGGL(kern, dim3(((b)->B + 63) / 64), dim3(64), 0, s, __VA_ARGS__);                 \
    } while (0)
#define GQP_IPM_LAUNCH(b, kern, s, ...) GQP_IPM_LAUNCH_SHM(b, kern, (b)->shmem, s, __VA_ARGS__)
/* the four sweeps: 16-lanes-per-instance batches pack 4 instances into one 64-lane workgroup */
#define GQP_SWEEP_LAUNCH(b, kern, shm, s, ...)                                                                \
    do {                                                                                                      \
        if ((b)->w16) GQP_LAUNCH_COOP(kern, dim3(((b)->w16_slots + 3) / 4), dim3(64), (b)->w16_shmem, s, __VA_ARGS__);  \
        else GQP_IPM_LAUNCH_SHM(b, kern, shm, s, __VA_ARGS__);                                                \
    } while (0)

/* the factor sweep: its own LDS tile in the sixteen-lanes families */
#define GQP_FACT_LAUNCH(b, kern, s, ...)                                                                      \
    do {                                                                                                      \
        if ((b)->w16) GQP_LAUNCH_COOP(kern, dim3(((b)->w16_slots + 3) / 4), dim3(64), (b)->w16_shmem_fact, s, __VA_ARGS__); \
        else GQP_IPM_LAUNCH_SHM(b, kern, (b)->shmem_fact, s, __VA_ARGS__);                                    \
    } while (0)

static IpmKernels pick_kernels(const ocp_qp_gpu_batch *b)
{
    const KernelSet *ks = b->ks;
    const int xb = b->xbox;
    IpmKernels k;
    k.fact = b->use_box ? ks->box_fact[xb] : ks->back_fact;
    k.rhs = b->use_box ? ks->box_rhs[xb] : ks->back_rhs;
    k.faff = b->use_box ? ks->box_fwd_aff[xb] : ks->fwd_aff;
    k.fcorr = b->use_box ? ks->box_fwd_corr[xb] : ks->fwd_corr;
    if (b->use_box && !b->wpi && b->kb_plain && ks->kb_fact[xb])
    {
        k.fact = ks->kb_fact[xb]; k.rhs = ks->kb_rhs[xb]; k.faff = ks->kb_fwd_aff[xb]; k.fcorr = ks->kb_fwd_corr[xb];
    }
    k.final_ = b->use_box ? ks->box_finalize : ks->finalize;
    return k;
}

/* per-kernel-class HIP event timing, accumulated on the ROOT batch */
struct Prof
{
    ocp_qp_gpu_batch *root;
    size_t used = 0;
    void begin(int cls, hipStream_t s)
    {
        if (!root->profile) return;
        if (used + 2 > root->prof_ev.size())
        {
            hipEvent_t e0, e1;
            HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
            root->prof_ev.push_back(e0); root->prof_ev.push_back(e1);
        }
        root->prof_cls.push_back(cls);
        HIPCHK(hipEventRecord(root->prof_ev[used], s));
    }
    void end(hipStream_t s)
    {
        if (!root->profile) return;
        HIPCHK(hipEventRecord(root->prof_ev[used + 1], s));
        used += 2;
    }
};

static void compact_into(ocp_qp_gpu_batch *b, ocp_qp_gpu_batch *root, int nact, hipStream_t s, bool tail);
static void compact_back(ocp_qp_gpu_batch *b, hipStream_t s, ocp_qp_gpu_batch *c, int it0);

/*
 * One level of the IPM loop.  After the factor kernel of every iteration the host reads the
 * number of still-iterating instances.  Converged instances are scattered over the waves, so a
 * wave keeps paying full HBM traffic as long as ONE of its 64 lanes is active; when at most half
 * of the level is still iterating, the survivors (QP data + iterate, ~98 KB each for C2) are
 * copied into a dense sub-batch, the loop continues there (recursively), and the results are
 * scattered back.  Per-instance arithmetic is unchanged, so results are bit-identical.
 */
/* side -> (stage, activity bit) of k_step_update, once per batch */
static const int *side_map(ocp_qp_gpu_batch *b)
{
    if (b->d_side_map) return b->d_side_map;
    std::vector<int> h((size_t) std::max(b->nct_tot, 1), -1);
    for (int k = 0; k <= b->N; k++)
    {
        const GqpStage &S = b->st[k];
        const int nbg = S.nb + S.ng;
        int sp = 0;
        for (int pv = 0; pv < 64; pv++)
            if ((S.bmask >> pv) & 1)
            {
                if (!((S.emask >> pv) & 1)) { h[S.o_ct + sp] = k * 128 + sp; h[S.o_ct + nbg + sp] = k * 128 + nbg + sp; }
                sp++;
            }
        for (int g = 0; g < S.ng; g++) { h[S.o_ct + S.nb + g] = k * 128 + S.nb + g; h[S.o_ct + nbg + S.nb + g] = k * 128 + nbg + S.nb + g; }
        for (int q = 0; q < 2 * S.ns; q++) h[S.o_ct + 2 * nbg + q] = k * 128 + 2 * nbg + q;
    }
    b->d_side_map = dalloc<int>(b, h.size());
    HIPCHK(hipMemcpy(b->d_side_map, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice));
    return b->d_side_map;
}

static void run_ipm(ocp_qp_gpu_batch *b, ocp_qp_gpu_batch *root, Prof &prof, hipStream_t s, int it)
{
    const IpmKernels K = pick_kernels(b);
    GqpDev D = b->D;
    GqpOpts O = effective_opts(root->O, root);
    /* sixteen-lanes families, launch per sweep: the step is applied by a launch of its own (k_step_update) instead of a pass at
     * the end of the corrector sweep, which walks the stages one after the other inside a latency-bound kernel.  Measured
     * (tools/ext_update_ab.py, profiles/r05_ext_update_ab.txt; identical iterates): general rows + slacks (two memory round trips
     * per stage in the pass) C4 124.0 -> 120.4 ms; box classes of 7,281 instances +0.2 ... +1.1 %; the condensed C3 batch
     * (65,536 instances, bandwidth-bound: the pass overlaps with other workgroups' sweeps, a launch of its own does not) 48.4 ->
     * 49.4 ms; short horizons (N = 20) lose what one more launch per iteration costs (r05_ext_update_ab2.txt: -2.6 % at nx = 12) -- so:
     * GEN always, box classes up to 16,384 instances from N = 50 on.  ACADOS_AMD_EXT_UPDATE=0 / 1 forces the pass / the launch */
    const char *eext = getenv("ACADOS_AMD_EXT_UPDATE");
    const bool ext_update = b->w16 && (eext ? atoi(eext) != 0 : (b->w16_ng > 0 || (b->B <= 16384 && b->N >= 50)));
    GqpOpts Oc = O; /* options of the corrector-sweep launches */
    Oc.ext_update = ext_update ? 1 : 0;
    const int *smap = ext_update ? side_map(b) : nullptr;
    b->w16_slots = b->B;
    /* sixteen lanes per instance: once a sixth of the slots has converged the sweeps run over a dense list of the live
     * instances (row slot -> instance, GqpDev::perm) -- no data moves, the grid shrinks, every wave carries four live rows */
    const char *eperm = getenv("ACADOS_AMD_W16_PERM");
    const bool use_perm = b->w16 && !(eperm && atoi(eperm) == 0);
    if (b == root && b->w16 && b->w16_solve && b->B <= root->solve_max && !D.perm)
    {
        /* small batch: every 16-lane row runs this loop by itself inside one launch (kx_solve) */
        prof.begin(1, s);
        GQP_SWEEP_LAUNCH(b, b->w16_solve, 0, s, D, O, 0);
        prof.end(s);
        root->launches++;
        root->n_single_launch++;
        HIPCHK(hipMemcpyAsync(b->h_nact, D.n_active, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (*b->h_nact <= 0) return;
        /* (never taken: a row leaves the kernel only with its instance out of the RUNNING state) */
    }
    for (;; it++)
    {
        if (b == root) prof.begin(1, s); /* per-class timing covers the root level only (full-batch launches) */
        GQP_FACT_LAUNCH(b, K.fact, s, D, O, 0);
        if (b == root) prof.end(s);
        root->launches++;
        HIPCHK(hipMemcpyAsync(b->h_nact, D.n_active, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        const int nact = *b->h_nact;
        if (root->print_level > 1) printf("acados_amd: ipm iter %d level size %d active %d\n", it, b->B, nact);
        if (nact <= 0 || it > O.iter_max) break;
        if (use_perm && nact >= 1 && 6 * nact <= 5 * b->w16_slots)
        {
            if (!b->d_perm) { b->d_perm = dalloc<int>(b, b->Bp); b->d_perm_cnt = dalloc<int>(b, 1); }
            HIPCHK(hipMemsetAsync(b->d_perm_cnt, 0, sizeof(int), s));
            hipLaunchKernelGGL(gqp::k_active_perm, dim3((b->B + 255) / 256), dim3(256), 0, s, D, b->d_perm, b->d_perm_cnt);
            D.perm = b->d_perm;
            D.n_perm = nact;
            b->w16_slots = nact;
        }
        if (!b->wpi && root->tail_max > 0 && nact <= root->tail_max && (long) root->tail_div * nact <= b->B)
        {
            /* the last survivors of a one-instance-per-lane level: a wave that still has ONE active lane pays
             * the full per-wave latency of every sweep, so the tail continues one wave per instance */
            compact_into(b, root, nact, s, true);
            root->n_tail_switches++;
            run_ipm(b->tail, root, prof, s, it);
            /* the family's own finalize first: the general one-instance-per-lane kernels refresh the multipliers
             * of the fixed variables in every factor sweep and their finalize relies on that, the wave-per-instance
             * kernels compute them once at the end */
            GQP_IPM_LAUNCH(b->tail, pick_kernels(b->tail).final_, s, b->tail->D);
            root->launches++;
            compact_back(b, s, b->tail, it);
            break;
        }
        if (b->B >= root->compact_min && 2 * nact <= b->B)
        {
            compact_into(b, root, nact, s, false);
            root->n_compactions++;
            run_ipm(b->compact, root, prof, s, it);
            compact_back(b, s, b->compact, it);
            break;
        }
        if (b == root) prof.begin(2, s);
        GQP_SWEEP_LAUNCH(b, K.faff, b->shmem_fwd, s, D, O, 0);
        if (b == root) prof.end(s);
        if (b == root) prof.begin(3, s);
        GQP_SWEEP_LAUNCH(b, K.rhs, b->shmem, s, D, O, 0);
        if (b == root) prof.end(s);
        if (b == root) prof.begin(4, s);
        GQP_SWEEP_LAUNCH(b, K.fcorr, b->shmem_fwd, s, D, Oc, 0);
        if (b == root) prof.end(s);
        root->launches += 3;
        if (O.cond_pred_corr)
        {
            GQP_SWEEP_LAUNCH(b, K.rhs, b->shmem, s, D, O, 1);
            GQP_SWEEP_LAUNCH(b, K.fcorr, b->shmem_fwd, s, D, Oc, 1);
            root->launches += 2;
        }
        if (ext_update)
        {
            /* (behind the redo pair: an instance whose corrector collapsed gets its step length there) */
            const int blocks = D.ux.aos ? b->B : (b->B + GQP_UPD_THREADS - 1) / GQP_UPD_THREADS;
            hipLaunchKernelGGL(gqp::k_step_update, dim3(blocks), dim3(GQP_UPD_THREADS), 0, s, D, O, smap, b->nct_tot, b->ns2_tot);
            root->launches++;
        }
    }
    b->w16_slots = b->B; /* launches outside this loop (sensitivity passes) cover every instance again */
}

/*
 * Terminal polishing step (option "polish" = 1, opt-in; status and iteration counts of the solve are unchanged).  Behind the loop,
 * converged instances that still hold a BALANCED complementarity pair -- min(lam, t) > polish_ratio * max(lam, t) on a row that takes
 * part: the weakly active rows an IPM leaves at t = mu / lam*, the whole of the distance between its exit point and the solution
 *  -- run ONE more iteration of the same four sweeps (k_polish_select: iteration counter 0 against iter_max 1, exit
 * tolerances that cannot be met, statistics table off).  Near the solution a Mehrotra iteration is the affine step but for
 * sigma = (mu_aff / mu)^3 ~ 1e-9: mu falls by three orders and more.  The polished point must pass the exit test the solve was
 * run with; an instance whose does not gets its iterate back (k_polish_restore / k_polish_revert: counted in "polish_reverted").
 * Cost: one iteration + one factor sweep over the instances selected -- in the one-instance-per-lane family over every 64-instance
 * tile that holds one.  Measured: profiles/NOTES.md round 6.
 */
static void run_ipm(ocp_qp_gpu_batch *b, ocp_qp_gpu_batch *root, Prof &prof, hipStream_t s, int it);
static void polish_pass(ocp_qp_gpu_batch *b, Prof &prof, hipStream_t s)
{
    b->n_polished = b->n_polish_reverted = 0;
    GqpDev &D = b->D;
    if (!b->d_pol_status)
    {
        b->d_pol_status = dalloc<int>(b, b->Bp); b->d_pol_iter = dalloc<int>(b, b->Bp); b->d_pol_flag = dalloc<int>(b, b->Bp);
        b->d_pol_cnt = dalloc<int>(b, 2);
        b->d_pol_sc = dalloc<double>(b, (size_t) 6 * b->Bp);
        b->pol_ux = garr<double>(b, D.ux.E); b->pol_sv = garr<double>(b, D.sv.E); b->pol_pi = garr<double>(b, D.pi.E);
        b->pol_lam = garr<double>(b, D.lam.E); b->pol_t = garr<double>(b, D.t.E);
    }
    const dim3 g64((b->B + 63) / 64), blk(64);
    HIPCHK(hipMemsetAsync(b->d_pol_cnt, 0, 2 * sizeof(int), s));
    hipLaunchKernelGGL(gqp::k_polish_select, g64, blk, 0, s, D, side_map(b), b->nct_tot, b->polish_ratio, b->polish_min, b->d_pol_status, b->d_pol_iter, b->d_pol_sc,
                       b->d_pol_cnt);
    HIPCHK(hipMemcpyAsync(b->h_nact, b->d_pol_cnt, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const int nsel = *b->h_nact;
    b->launches++;
    if (nsel <= 0)
    {
        hipLaunchKernelGGL(gqp::k_status_restore, g64, blk, 0, s, D, b->d_pol_status); /* (nothing changed; keeps the two paths alike) */
        return;
    }
    const struct { GArr src, dst; } cp[] = {{D.ux, b->pol_ux}, {D.sv, b->pol_sv}, {D.pi, b->pol_pi}, {D.lam, b->pol_lam}, {D.t, b->pol_t}};
    for (const auto &c : cp) HIPCHK(hipMemcpyAsync(c.dst.p, c.src.p, sizeof(double) * (size_t) c.src.E * b->Bp, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(D.n_active, b->h_nact, sizeof(int), hipMemcpyHostToDevice, s));
    const GqpOpts keep = b->O;
    const int keep_stat = D.stat_inst;
    b->O.tol_stat = b->O.tol_eq = b->O.tol_ineq = b->O.tol_comp = -1.0; /* the loop's exit test cannot pass: it ends at iter_max */
    b->O.iter_max = b->polish; /* option value = iterations of the pass (1 unless asked otherwise) */
    const GqpOpts Oeff = effective_opts(keep, b);
    b->O.tau_min = Oeff.tau_min; /* the barrier floor of the solve (derived from ITS tol_comp) */
    D.stat_inst = 0;             /* the statistics table keeps the solve's rows */
    run_ipm(b, b, prof, s, 0);
    b->O = keep;
    D.stat_inst = keep_stat;
    hipLaunchKernelGGL(gqp::k_polish_restore, g64, blk, 0, s, D, Oeff, b->d_pol_status, b->d_pol_iter, b->d_pol_sc, b->d_pol_flag, b->d_pol_cnt + 1);
    HIPCHK(hipMemcpyAsync(b->h_nact, b->d_pol_cnt + 1, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    b->launches++;
    b->n_polished = nsel;
    b->n_polish_reverted = *b->h_nact;
    if (b->n_polish_reverted > 0)
    {
        const int blocks = D.ux.aos ? b->B : (b->B + GQP_UPD_THREADS - 1) / GQP_UPD_THREADS;
        hipLaunchKernelGGL(gqp::k_polish_revert, dim3(blocks), dim3(GQP_UPD_THREADS), 0, s, D, b->d_pol_flag, b->pol_ux, b->pol_sv, b->pol_pi, b->pol_lam,
                           b->pol_t);
        b->launches++;
    }
}

/* arrays that define a QP instance and its iterate (everything else is recomputed) */
#define GQP_FOR_STATE_ARRAYS(X) X(BAt) X(bvec) X(RSQ) X(rq) X(dvec) X(DCt) X(Zz) X(ux) X(sv) X(pi) X(lam) X(t)
#define GQP_FOR_RESULT_ARRAYS(X) X(ux) X(sv) X(pi) X(lam) X(t)

static void compact_into(ocp_qp_gpu_batch *b, ocp_qp_gpu_batch *root, int nact, hipStream_t s, bool tail)
{
    /* sorted list of the still-iterating instances (sorted => the gather reads stay coalesced) */
    int *st = b->h_ints, *list = b->h_ints + b->Bp; /* pinned, owned by the level */
    HIPCHK(hipMemcpy(st, b->D.status, sizeof(int) * b->B, hipMemcpyDeviceToHost));
    int cnt = 0;
    for (int i = 0; i < b->B; i++) if (st[i] == GQP_RUNNING) list[cnt++] = i;
    /* capacities follow the level's CAPACITY (Bp), not its current count: a level is re-used by later solves with
     * more survivors (its B changes between solves) */
    if (!b->d_list || cnt > b->list_cap)
    {
        b->list_cap = std::max(cnt, (b->Bp + 1) / 2);
        b->d_list = dalloc<int>(b, b->list_cap);
    }
    ocp_qp_gpu_batch *&slot = tail ? b->tail : b->compact;
    if (slot && cnt > (tail ? b->tail_cap : slot->Bp))
    {
        /* more survivors than the sub-batch was sized for (tail_max raised, or a later solve of a re-used level) */
        ocp_qp_gpu_batch_destroy(slot);
        slot = nullptr;
    }
    if (!slot)
    {
        /* same kernel set (compaction) or the wave-per-instance family at the very same padded dims (tail) */
        const int cap = tail ? std::max(cnt, std::min(b->tail_max, b->list_cap)) : std::max(cnt, (b->Bp + 1) / 2);
        if (tail) b->tail_cap = cap;
        ocp_qp_gpu_batch *c = batch_create_shape(b->N, b->nx.data(), b->nu.data(), b->nbx.data(), b->nbu.data(),
                                                 b->ng.data(), b->ns.data(), cap, b->device, tail ? b->ks->NX : 0, tail ? b->ks->NU : 0,
                                                 b->ks, tail);
        if (c && b->stream_priority) set_stream_priority(c, b->stream_priority);
        if (!c) { fprintf(stderr, "acados_amd: cannot create the compaction sub-batch\n"); exit(1); }
        c->idxb = b->idxb; c->idxs_rev = b->idxs_rev; c->idxe = b->idxe; c->nbxe = b->nbxe;
        c->compact_min = b->compact_min;
        c->tail_max = b->tail_max;
        c->tail_div = b->tail_div;
        if (!tail) { c->aos = b->aos; c->wpi = b->wpi; c->shmem = b->shmem; c->shmem_fwd = b->shmem_fwd; c->shmem_fact = b->shmem_fact; c->w16 = b->w16; c->w16_soft = b->w16_soft; c->w16_ng = b->w16_ng; c->w16_shmem = b->w16_shmem; c->w16_shmem_fact = b->w16_shmem_fact; c->w16_tiles = b->w16_tiles; }
        finalize_structure(c);
        slot = c;
    }
    ocp_qp_gpu_batch *c = slot;
    HIPCHK(hipMemcpyAsync(b->d_list, list, sizeof(int) * cnt, hipMemcpyHostToDevice, s));
    c->B = cnt; /* the level works on `cnt` slots of its capacity */
    c->D.B = cnt;
    const dim3 block(64);
#define GQP_COPY_IN(A) copy_level(b->D.A, c->D.A, b->d_list, cnt, 0, s);
    GQP_FOR_STATE_ARRAYS(GQP_COPY_IN)
#undef GQP_COPY_IN
    copy_level(b->D.amask, c->D.amask, b->d_list, cnt, 0, s);
    hipLaunchKernelGGL(gqp::k_compact_scalars, dim3((cnt + 63) / 64), block, 0, s, b->D, c->D, b->d_list, cnt, 0);
    *c->h_nact = cnt;
    HIPCHK(hipMemcpyAsync(c->D.n_active, c->h_nact, sizeof(int), hipMemcpyHostToDevice, s));
    /* the level keeps its own statistics rows for its first slots; merged into the parent's table afterwards */
    c->O = root->O;
    c->tol_comp_soft_scale = root->tol_comp_soft_scale;
    ensure_stat(c);
    HIPCHK(hipMemsetAsync(c->D.stat, 0, sizeof(double) * (size_t) c->stat_rows * GQP_STAT_COLS * c->stat_inst, s));
    HIPCHK(hipStreamSynchronize(s)); /* `list` (host) must outlive the copy */
}

static void compact_back(ocp_qp_gpu_batch *b, hipStream_t s, ocp_qp_gpu_batch *c, int it0)
{
    const int cnt = c->B;
    if (b->D.stat && c->D.stat)
        hipLaunchKernelGGL(gqp::k_stat_merge, dim3(1, std::max(1, std::min(b->stat_rows, c->stat_rows) - it0)), dim3(64), 0, s, b->D,
                           c->D, b->d_list, cnt < 64 ? cnt : 64, it0);
    const dim3 block(64);
#define GQP_COPY_OUT(A) copy_level(b->D.A, c->D.A, b->d_list, cnt, 1, s);
    GQP_FOR_RESULT_ARRAYS(GQP_COPY_OUT)
#undef GQP_COPY_OUT
    hipLaunchKernelGGL(gqp::k_compact_scalars, dim3((cnt + 63) / 64), block, 0, s, b->D, c->D, b->d_list, cnt, 1);
}

/*
 * FULL CONDENSING of any size (option "full_dense" = 1; FULL_CONDENSING_GPU_IPM past what one condensed stage of the stage-wise
 * families may carry): one workgroup per instance condenses every state but x0, runs the IPM on the dense problem and expands
 * (dense_kernels.hpp).  The batch is worked off in slices whose workspace stays below 8 GiB; the family's own finalize kernel
 * then recovers the multipliers of the equality-flagged rows from stationarity, as after a stage-wise solve.
 */
static int dense_solve(ocp_qp_gpu_batch *b)
{
    hipStream_t s = b->stream;
    const GqpDev &D = b->D;
    if (!b->d_kd_rows)
    {
        std::vector<gqp::KdRow> rows;
        b->kd_unsupported = b->has_slack ? 1 : 0;
        for (int k = 0; k <= b->N; k++)
        {
            const GqpStage &S = b->st[k];
            const int nbg = S.nb + S.ng;
            if (S.ns > 0) b->kd_unsupported = 1;
            int sp = 0;
            for (int pv = 0; pv < 64; pv++)
                if ((S.bmask >> pv) & 1)
                {
                    const int fixed = (int) ((S.emask >> pv) & 1);
                    if (fixed && pv >= D.NU && k > 0) b->kd_unsupported = 1; /* a fixed state behind stage 0 is not a dense variable */
                    rows.push_back({k, pv, -1, S.o_ct + sp, S.o_ct + nbg + sp, sp, nbg + sp, fixed});
                    sp++;
                }
            for (int g = 0; g < S.ng; g++) rows.push_back({k, -1, S.o_g + g, S.o_ct + S.nb + g, S.o_ct + nbg + S.nb + g, S.nb + g, nbg + S.nb + g, 0});
        }
        if (b->kd_unsupported)
            fprintf(stderr, "acados_amd: full_dense: slacks and equality-flagged states behind stage 0 are not supported by the dense path (every "
                            "instance returns status 4); use the stage-wise solver (cond_N >= 1 blocks within the limits, or none)\n");
        gqp::KdDims &S = b->kd;
        const int K = b->N + 1, NX = D.NX, NU = D.NU, n = NX + NU;
        S.K = K; S.n = n; S.nvu = K * NU; S.nv = S.nvu + NX; S.R = (int) rows.size();
        size_t o = 0;
        auto take = [&](size_t cnt) { const size_t at = o; o += (cnt + 7) & ~(size_t) 7; return at; };
        S.oG = take((size_t) K * NX * S.nv); S.oc = take((size_t) K * NX); S.oH = take((size_t) K * n * n); S.og = take((size_t) K * n);
        S.orw = take((size_t) K * n); S.oM = take((size_t) S.nv * S.nv); S.orhs = take(S.nv); S.odv = take(S.nv); S.ov = take(S.nv);
        S.ow = take((size_t) K * n); S.odw = take((size_t) K * n); S.oP = take((size_t) NX * S.nv);
        S.W = o;
        b->d_kd_rows = dalloc<gqp::KdRow>(b, rows.size());
        if (!rows.empty()) HIPCHK(hipMemcpy(b->d_kd_rows, rows.data(), sizeof(gqp::KdRow) * rows.size(), hipMemcpyHostToDevice));
        const size_t budget = (size_t) 8 << 30;
        b->kd_slice = (int) std::max<size_t>(1, std::min<size_t>((size_t) b->B, budget / (S.W * sizeof(double))));
        b->d_kd_ws = dalloc<double>(b, (size_t) b->kd_slice * S.W);
    }
    const GqpOpts O = effective_opts(b->O, b);
    HIPCHK(hipEventRecord(b->ev0, s));
    for (int first = 0; first < b->B; first += b->kd_slice)
    {
        const int cnt = std::min(b->kd_slice, b->B - first);
        GQP_LAUNCH_COOP(gqp::kd_solve, dim3(cnt), dim3(GQP_KD_THREADS), 0, s, D, O, b->kd, (const gqp::KdRow *) b->d_kd_rows, b->d_kd_ws, first,
                        b->kd_unsupported);
    }
    b->launches = (b->B + b->kd_slice - 1) / b->kd_slice;
    if (!b->kd_unsupported)
    {
        GQP_IPM_LAUNCH(b, pick_kernels(b).final_, s, D);
        b->launches++;
    }
    HIPCHK(hipEventRecord(b->ev1, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, b->ev0, b->ev1));
    b->time_tot = ms * 1e-3;
    b->time_xcond = 0.0;
    b->factor_stale = true;   /* Lf of the stage-wise sweeps does not belong to this solution: the sensitivity slots refactor */
    b->sens_open = false;
    int *itv = b->h_ints, *st = b->h_ints + b->Bp;
    HIPCHK(hipMemcpy(itv, D.iter, sizeof(int) * b->B, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(st, D.status, sizeof(int) * b->B, hipMemcpyDeviceToHost));
    int mx = 0, bad = 0;
    for (int q = 0; q < b->B; q++) { mx = std::max(mx, itv[q]); bad += st[q] != 0; }
    b->last_iters = mx;
    return bad;
}

int ocp_qp_gpu_batch_solve(ocp_qp_gpu_batch *b)
try
{
    HIPCHK(hipSetDevice(b->device));
    finalize_structure(b);
    if (b->full_dense) return dense_solve(b);
    if (b->cond_N > 0 && b->cond_N < b->N)
    {
        if (b->pcond_state == 0) pcond_setup(b);
        if (b->pcond_state == 1) return pcond_solve(b, b->lhs_ready ? 2 : 3);
    }
    b->time_xcond = 0.0;
    ensure_stat(b);
    const KernelSet *ks = b->ks;
    GqpDev D = b->D;
    GqpOpts O = effective_opts(b->O, b);
    const dim3 grid((b->B + 63) / 64), block(64);
    hipStream_t s = b->stream;
    b->launches = 0;
    b->n_compactions = 0;
    b->n_tail_switches = 0;
    b->n_single_launch = 0;
    /* kernel classes: 0 init, 1 back_fact, 2 fwd_aff, 3 back_rhs, 4 fwd_corr, 5 finalize */
    b->prof_cls.clear();
    Prof prof;
    prof.root = b;

    HIPCHK(hipEventRecord(b->ev0, s));
    *b->h_nact = b->B;
    HIPCHK(hipMemcpyAsync(D.n_active, b->h_nact, sizeof(int), hipMemcpyHostToDevice, s));
    if (D.stat) HIPCHK(hipMemsetAsync(D.stat, 0, sizeof(double) * (size_t) b->stat_rows * GQP_STAT_COLS * b->stat_inst, s));
    if (O.warm_start < 2)
    {
        prof.begin(0, s);
        GQP_IPM_LAUNCH(b, ks->init, s, D, O);
        prof.end(s);
        b->launches++;
    }
    else
    {
        /* hot start: keep (ux, pi, lam, t) as they are in HBM; loop state reset as a cold start leaves it */
        const double clip = O.warm_start == 2 ? 0.1 : 0.0;
        hipLaunchKernelGGL(gqp::k_hot_start, grid, block, 0, s, D, std::max(clip, b->t0_min), std::max(clip, b->lam0_min));
        b->launches++;
    }
    HIPCHK(hipMemsetAsync(D.apend, 0, sizeof(double) * (size_t) b->Bp, s)); /* no step pending (a solve always ends behind a factor sweep; belt and braces) */
    run_ipm(b, b, prof, s, 0);
    b->n_polished = b->n_polish_reverted = 0;
    if (b->polish) polish_pass(b, prof, s);
    b->factor_stale = b->n_tail_switches + b->n_compactions > 0;
    b->sens_open = false;
    GQP_IPM_LAUNCH(b, pick_kernels(b).final_, s, D);
    b->launches++;
    HIPCHK(hipEventRecord(b->ev1, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, b->ev0, b->ev1));
    b->time_tot = ms * 1e-3;
    {
        int *itv = b->h_ints;
        HIPCHK(hipMemcpy(itv, D.iter, sizeof(int) * b->B, hipMemcpyDeviceToHost));
        int mx = 0;
        for (int q = 0; q < b->B; q++) mx = std::max(mx, itv[q]);
        b->last_iters = mx;
    }
    for (size_t q = 0; q < b->prof_cls.size(); q++)
    {
        float pm = 0.f;
        HIPCHK(hipEventElapsedTime(&pm, b->prof_ev[2 * q], b->prof_ev[2 * q + 1]));
        b->prof_ms[b->prof_cls[q]] += pm;
        b->prof_cnt[b->prof_cls[q]]++;
    }

    int *st = b->h_ints + b->Bp;
    HIPCHK(hipMemcpy(st, D.status, sizeof(int) * b->B, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < b->B; i++) bad += st[i] != 0;
    return bad;
}
catch (const gqp_hip_failure &) { return -1; }

/* One factor sweep at the final iterate with every instance awake: Lf / lf of the whole batch belong to the
 * solution afterwards (after a tail switch the root's factors of the handed-over instances are older).  Statuses
 * are restored. */
static void refactor_at_solution(ocp_qp_gpu_batch *b)
{
    hipStream_t s = b->stream;
    const dim3 g64((b->B + 63) / 64), blk(64);
    if (!b->d_saved_status) b->d_saved_status = dalloc<int>(b, b->Bp);
    GqpOpts O = effective_opts(b->O, b);
    hipLaunchKernelGGL(gqp::k_sens_prep, g64, blk, 0, s, b->D, O.tau_min, b->d_saved_status);
    GQP_FACT_LAUNCH(b, pick_kernels(b).fact, s, b->D, O, 0);
    hipLaunchKernelGGL(gqp::k_status_restore, g64, blk, 0, s, b->D, b->d_saved_status);
    HIPCHK(hipStreamSynchronize(s));
    b->factor_stale = false;
}

static int sens_begin(ocp_qp_gpu_batch *b)
{
    if (b->sens_open) return 0;
    /* after a partially condensed solve the expanded solution sits in this (full-space) batch: the factorisation at
     * the solution and the seed sweeps run here, the condensed sub-batch is not involved */
    /* a one-instance-per-lane batch hands slices of itself to a wave-per-instance sub-batch in sens_solve */
    if (b->wpi) refactor_at_solution(b);
    const GqpDev &D = b->D;
    const int n = D.NX + D.NU;
    if (!b->sfix.p) b->sfix = garr<double>(b, (size_t) (b->N + 2) * n);
    const GArr z[] = {D.rg, D.rgs, D.rb, D.rd, b->sfix};
    for (const GArr &a : z) HIPCHK(hipMemsetAsync(a.p, 0, sizeof(double) * (size_t) a.E * b->Bp, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    b->sens_open = true;
    return 0;
}

int ocp_qp_gpu_batch_sens_set(ocp_qp_gpu_batch *b, const char *f, int k, const double *data)
try
{
    HIPCHK(hipSetDevice(b->device));
    finalize_structure(b);
    if (strncmp(f, "seed_", 5) || k < 0 || k > b->N)
    {
        fprintf(stderr, "acados_amd: ocp_qp_gpu_batch_sens_set: unknown seed %s (stage %d)\n", f, k);
        return -1;
    }
    if (sens_begin(b)) return -1;
    const char *base = f + 5;
    std::vector<int> map, map2;
    GArr arr = {nullptr, 0, 0}, arr2 = {nullptr, 0, 0};
    const int len = field_map(b, base, k, map, &arr, &map2, &arr2);
    double sign = 1.0;
    GArr dst = {nullptr, 0, 0};
    if (len >= 0)
    {
        if (arr.p == b->D.rq.p) dst = b->D.rg;
        else if (arr.p == b->D.bvec.p) dst = b->D.rb;
        else if (arr.p == b->D.dvec.p) { dst = b->D.rd; sign = (base[0] == 'l' && base[1] != 'u') || !strcmp(base, "lls") ? -1.0 : 1.0; }
        else if (!strcmp(base, "zl") || !strcmp(base, "zu"))
        {
            /* gradient of the slack penalty: the slack part of the stationarity residual, indexed like sl / su */
            dst = b->D.rgs;
            const GqpStage &S = b->st[k];
            for (int j = 0; j < len; j++) map[j] = S.o_s + (base[1] == 'u' ? S.ns : 0) + j;
        }
    }
    if (len < 0 || !dst.p)
    {
        fprintf(stderr, "acados_amd: ocp_qp_gpu_batch_sens_set: %s is not a seed of this stage (seeds: seed_q seed_r seed_zl seed_zu seed_b "
                        "seed_lbu seed_ubu seed_lbx seed_ubx seed_lg seed_ug seed_lls seed_lus)\n", f);
        return -1;
    }
    if (len == 0) return 0;
    if (!strcmp(base, "lus")) sign = -1.0; /* su >= lus is a LOWER bound of the slack, like lls */
    std::vector<double> h((size_t) b->B * len);
    for (size_t e = 0; e < h.size(); e++) h[e] = sign * data[e];
    const double *src = stage_in(b, h.data(), h.size(), 0);
    int *dm = upload_map(b, map);
    hipLaunchKernelGGL(gqp::k_scatter, dim3((b->B + 63) / 64), dim3(64), 0, b->stream, src, b->B, len, dm, dst);
    HIPCHK(hipStreamSynchronize(b->stream));
    if (arr2.p)
    {
        /* equality-flagged bounds: the seed of the bound is the derivative of the variable itself */
        bool any = false;
        for (int &m : map2) { if (m >= 0) any = true; }
        if (any)
        {
            src = stage_in(b, data, (size_t) b->B * len, 0);
            dm = upload_map(b, map2);
            hipLaunchKernelGGL(gqp::k_scatter, dim3((b->B + 63) / 64), dim3(64), 0, b->stream, src, b->B, len, dm, b->sfix);
            HIPCHK(hipStreamSynchronize(b->stream));
        }
    }
    return 0;
}
catch (const gqp_hip_failure &) { return -1; }

/* the seed pass on a wave-per-instance / sixteen-lanes batch whose factor belongs to the solution */
static void sens_pass(ocp_qp_gpu_batch *b, hipStream_t s)
{
    const dim3 g64((b->B + 63) / 64), blk(64);
    GqpOpts O = effective_opts(b->O, b);
    const IpmKernels K = pick_kernels(b);
    hipLaunchKernelGGL(gqp::k_sens_fixed, g64, blk, 0, s, b->D, b->sfix, 0);
    hipLaunchKernelGGL(gqp::k_sens_prep, g64, blk, 0, s, b->D, O.tau_min, b->d_saved_status);
    GQP_SWEEP_LAUNCH(b, K.rhs, b->shmem, s, b->D, O, 2);
    GQP_SWEEP_LAUNCH(b, K.fcorr, b->shmem_fwd, s, b->D, O, 2);
    hipLaunchKernelGGL(gqp::k_sens_fixed, g64, blk, 0, s, b->D, b->sfix, 1);
    hipLaunchKernelGGL(gqp::k_status_restore, g64, blk, 0, s, b->D, b->d_saved_status);
}

/*
 * One-instance-per-lane batches: the direction-only sweeps exist in the wave-per-instance families only, so the
 * batch is walked in slices of at most GQP_SENS_SLICE instances; a slice (QP data, solution, seeds) is copied into a
 * sub-batch of the wave-per-instance family at the same padded dims (the tail switch's conversion), factorised there
 * at the solution, swept, and the directions are copied back.  ~3 extra passes over the data per call.
 */
#define GQP_SENS_SLICE 16384
static void sens_solve_sliced(ocp_qp_gpu_batch *b)
{
    hipStream_t s = b->stream;
    const char *env = getenv("ACADOS_AMD_SENS_SLICE");
    const int cap = b->sens_child ? b->sens_cap : std::min(b->B, env && atoi(env) > 0 ? atoi(env) : GQP_SENS_SLICE);
    if (!b->sens_child)
    {
        ocp_qp_gpu_batch *c = batch_create_shape(b->N, b->nx.data(), b->nu.data(), b->nbx.data(), b->nbu.data(), b->ng.data(),
                                                 b->ns.data(), cap, b->device, b->ks->NX, b->ks->NU, b->ks, true);
        if (c && b->stream_priority) set_stream_priority(c, b->stream_priority);
        if (!c) { fprintf(stderr, "acados_amd: cannot create the sensitivity sub-batch\n"); exit(1); }
        c->idxb = b->idxb; c->idxs_rev = b->idxs_rev; c->idxe = b->idxe; c->nbxe = b->nbxe;
        c->tail_max = 0;
        finalize_structure(c);
        const int n = c->D.NX + c->D.NU;
        c->sfix = garr<double>(c, (size_t) (c->N + 2) * n);
        b->sens_child = c;
        b->sens_cap = cap;
        b->d_slist = dalloc<int>(b, cap);
    }
    ocp_qp_gpu_batch *c = b->sens_child;
    c->O = b->O;
    c->tol_comp_soft_scale = b->tol_comp_soft_scale;
    std::vector<int> list(cap);
    const dim3 block(64);
    for (int i0 = 0; i0 < b->B; i0 += cap)
    {
        const int cnt = std::min(cap, b->B - i0);
        for (int j = 0; j < cnt; j++) list[j] = i0 + j;
        HIPCHK(hipMemcpyAsync(b->d_slist, list.data(), sizeof(int) * cnt, hipMemcpyHostToDevice, s));
        c->B = cnt;
        c->D.B = cnt;
#define GQP_SLICE_COPY(SRC, DST, DIR) copy_level(SRC, DST, b->d_slist, cnt, DIR, s);
#define GQP_COPY_IN(A) GQP_SLICE_COPY(b->D.A, c->D.A, 0)
        GQP_FOR_STATE_ARRAYS(GQP_COPY_IN)
        copy_level(b->D.amask, c->D.amask, b->d_slist, cnt, 0, s);
        hipLaunchKernelGGL(gqp::k_compact_scalars, dim3((cnt + 63) / 64), block, 0, s, b->D, c->D, b->d_slist, cnt, 0);
        HIPCHK(hipStreamSynchronize(s)); /* the sub-batch works on its own stream */
        refactor_at_solution(c);         /* writes the residual arrays: the seeds go in afterwards */
        GQP_COPY_IN(rg) GQP_COPY_IN(rgs) GQP_COPY_IN(rb) GQP_COPY_IN(rd)
#undef GQP_COPY_IN
        GQP_SLICE_COPY(b->sfix, c->sfix, 0)
        HIPCHK(hipStreamSynchronize(s));
        sens_pass(c, c->stream);
        HIPCHK(hipStreamSynchronize(c->stream));
#define GQP_COPY_OUT(A) GQP_SLICE_COPY(b->D.A, c->D.A, 1)
        GQP_COPY_OUT(dux) GQP_COPY_OUT(dsv) GQP_COPY_OUT(dpi) GQP_COPY_OUT(dlam) GQP_COPY_OUT(dt)
#undef GQP_COPY_OUT
#undef GQP_SLICE_COPY
        HIPCHK(hipStreamSynchronize(s));
    }
}

int ocp_qp_gpu_batch_sens_solve(ocp_qp_gpu_batch *b)
try
{
    HIPCHK(hipSetDevice(b->device));
    finalize_structure(b);
    if (sens_begin(b)) return -1; /* no seed set: all-zero seeds, zero sensitivities */
    if (!b->wpi)
    {
        sens_solve_sliced(b);
        HIPCHK(hipGetLastError());
        b->sens_open = false;
        return 0;
    }
    sens_pass(b, b->stream);
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipGetLastError());
    b->sens_open = false;
    return 0;
}
catch (const gqp_hip_failure &) { return -1; }

int ocp_qp_gpu_batch_get(ocp_qp_gpu_batch *b, const char *f, int k, double *data, int is_device)
try
{
    HIPCHK(hipSetDevice(b->device));
    finalize_structure(b);
    std::vector<int> map;
    GArr arr = {nullptr, 0, 0};
    if (b->factor_stale && !strncmp(f, "ric_", 4)) refactor_at_solution(b);
    /* sensitivities: the direction arrays of the last ocp_qp_gpu_batch_sens_solve */
    const bool is_sens = !strncmp(f, "sens_", 5);
    {
        const size_t flen = strlen(f);
        if (flen > 5 && !strcmp(f + flen - 5, "_mask"))
        {
            /* activity of the sides as 1.0 / 0.0 (equality-flagged rows: 1.0) */
            const int mlen = mask_bits(b, f, k, map);
            if (mlen < 0) { fprintf(stderr, "acados_amd: ocp_qp_gpu_batch_get: field %s not available at stage %d\n", f, k); return -1; }
            if (mlen == 0) return 0;
            const size_t mcnt = (size_t) b->B * mlen;
            double *mdst = data;
            if (!is_device)
            {
                if (mcnt > b->stage_cap) { b->stage_cap = mcnt * 2; b->d_stage = dalloc<double>(b, b->stage_cap); }
                mdst = b->d_stage;
            }
            int *dmm = upload_map(b, map);
            hipLaunchKernelGGL(gqp::k_getmask, dim3((b->B + 63) / 64), dim3(64), 0, b->stream, mdst, b->B, mlen, dmm, b->D.amask, k, b->AW);
            if (!is_device) HIPCHK(hipMemcpyAsync(data, mdst, sizeof(double) * mcnt, hipMemcpyDeviceToHost, b->stream));
            HIPCHK(hipStreamSynchronize(b->stream));
            return 0;
        }
    }
    int len;
    if (!strncmp(f, "res_", 4))
    {
        /* residual vectors of the last ocp_qp_gpu_batch_res_compute: res_g ([u; x]), res_gs ([sl; su]), res_b, res_d,
         * res_m -- same element maps as the quantity each residual belongs to */
        if (!b->R.nrm) { fprintf(stderr, "acados_amd: ocp_qp_gpu_batch_get: %s before ocp_qp_gpu_batch_res_compute\n", f); return -1; }
        const char *like = !strcmp(f, "res_g") ? "ric_l" : !strcmp(f, "res_b") ? "pi" : !strcmp(f, "res_d") || !strcmp(f, "res_m") ? "lam" : nullptr;
        if (!strcmp(f, "res_gs"))
        {
            len = 2 * b->st[k].ns;
            for (int j = 0; j < len; j++) map.push_back(b->st[k].o_s + j);
            arr = b->R.gs;
        }
        else if (like)
        {
            len = field_map(b, like, k, map, &arr);
            if (!strcmp(f, "res_b")) for (int &m : map) m -= b->ks->NX; /* pi lives in slot k+1, res_b in slot k */
            arr = !strcmp(f, "res_g") ? b->R.g : !strcmp(f, "res_b") ? b->R.b : !strcmp(f, "res_d") ? b->R.d : b->R.m;
        }
        else len = -1;
    }
    else len = field_map(b, is_sens ? f + 5 : f, k, map, &arr);
    if (len > 0 && (!strcmp(f, "Q") || !strcmp(f, "R")))
    {
        /* the packed lower triangle holds the matrix: mirror it for the reader */
        const int dim = f[0] == 'Q' ? b->nx[k] : b->nu[k];
        for (int c = 0; c < dim; c++) for (int r = 0; r < c; r++) map[c * dim + r] = map[r * dim + c];
    }
    if (is_sens && len >= 0)
    {
        if (arr.p == b->D.ux.p) arr = b->D.dux;
        else if (arr.p == b->D.sv.p) arr = b->D.dsv;
        else if (arr.p == b->D.pi.p) arr = b->D.dpi;
        else if (arr.p == b->D.lam.p) arr = b->D.dlam;
        else if (arr.p == b->D.t.p) arr = b->D.dt;
        else len = -1;
    }
    if (len < 0)
    {
        fprintf(stderr, "acados_amd: ocp_qp_gpu_batch_get: field %s not available at stage %d\n", f, k);
        return -1;
    }
    if (len == 0) return 0;
    const size_t cnt = (size_t) b->B * len;
    double *dst = data;
    if (!is_device)
    {
        if (cnt > b->stage_cap)
        {
            b->stage_cap = cnt * 2;
            b->d_stage = dalloc<double>(b, b->stage_cap);
        }
        dst = b->d_stage;
    }
    int *dm = upload_map(b, map);
    hipLaunchKernelGGL(gqp::k_gather, dim3((b->B + 63) / 64), dim3(64), 0, b->stream, dst, b->B, len, dm, arr);
    if (!is_device) HIPCHK(hipMemcpyAsync(data, dst, sizeof(double) * cnt, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return 0;
}
catch (const gqp_hip_failure &) { return -1; }

/* KKT residuals of the QP data and the iterate (ux, pi, lam, t) that are in HBM right now -- whoever put them there
 * (the solver, a warm-start set, an expansion): ocp_qp_res_compute + ocp_qp_res_compute_nrm_inf of
 * acados/ocp_qp/ocp_qp_common.c:559-667 for the whole batch in one launch.  Independent of the IPM sweeps. */
int ocp_qp_gpu_batch_res_compute(ocp_qp_gpu_batch *b)
try
{
    HIPCHK(hipSetDevice(b->device));
    finalize_structure(b);
    const GqpDev &D = b->D;
    if (!b->R.nrm)
    {
        const int n = D.NX + D.NU;
        b->R.g = garr<double>(b, (size_t) (b->N + 1) * n);
        b->R.gs = garr<double>(b, (size_t) b->ns2_tot);
        b->R.b = garr<double>(b, (size_t) (b->N + 1) * D.NX);
        b->R.d = garr<double>(b, (size_t) b->nct_tot + 16);
        b->R.m = garr<double>(b, (size_t) b->nct_tot + 16);
        b->R.nrm = dalloc<double>(b, 4 * (size_t) b->Bp);
        b->R.Bp = b->Bp;
    }
    hipLaunchKernelGGL(gqp::k_res_compute, dim3((b->B + 63) / 64), dim3(64), 0, b->stream, b->D, b->R);
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipGetLastError());
    return 0;
}
catch (const gqp_hip_failure &) { return -1; }

/* res[i * 4 + q]: inf-norms of res_g, res_b, res_d, res_m of instance i (ocp_qp_res_compute_nrm_inf) */
int ocp_qp_gpu_batch_res_nrm_inf(ocp_qp_gpu_batch *b, double *res)
try
{
    if (!b->R.nrm && ocp_qp_gpu_batch_res_compute(b) != 0) return -1;
    std::vector<double> h(4 * (size_t) b->Bp);
    HIPCHK(hipMemcpy(h.data(), b->R.nrm, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < b->B; i++)
        for (int q = 0; q < 4; q++) res[(size_t) i * 4 + q] = h[(size_t) q * b->Bp + i];
    return 0;
}
catch (const gqp_hip_failure &) { return -1; }

int ocp_qp_gpu_batch_get_info(ocp_qp_gpu_batch *b, const char *f, void *data)
try
{
    HIPCHK(hipSetDevice(b->device));
    finalize_structure(b);
    const GqpDev &D = b->D;
    const size_t B = b->B;
    if (!strcmp(f, "status")) { HIPCHK(hipMemcpy(data, D.status, sizeof(int) * B, hipMemcpyDeviceToHost)); return 0; }
    if (!strcmp(f, "iter")) { HIPCHK(hipMemcpy(data, D.iter, sizeof(int) * B, hipMemcpyDeviceToHost)); return 0; }
    const char *names[4] = {"res_stat", "res_eq", "res_ineq", "res_comp"};
    for (int q = 0; q < 4; q++)
        if (!strcmp(f, names[q]))
        {
            HIPCHK(hipMemcpy(data, D.res + (size_t) q * b->Bp, sizeof(double) * B, hipMemcpyDeviceToHost));
            return 0;
        }
    if (!strcmp(f, "mu")) { HIPCHK(hipMemcpy(data, D.mu, sizeof(double) * B, hipMemcpyDeviceToHost)); return 0; }
    if (!strcmp(f, "obj")) { HIPCHK(hipMemcpy(data, D.obj, sizeof(double) * B, hipMemcpyDeviceToHost)); return 0; }
    fprintf(stderr, "acados_amd: ocp_qp_gpu_batch_get_info: unknown field %s\n", f);
    return -1;
}
catch (const gqp_hip_failure &) { return -1; }

int ocp_qp_gpu_batch_get_stat(ocp_qp_gpu_batch *b, int inst, double *stat, int max_rows)
try
{
    HIPCHK(hipSetDevice(b->device));
    if (b->pcond_state == 1 && b->child) return ocp_qp_gpu_batch_get_stat(b->child, inst, stat, max_rows);
    if (!b->D.stat || inst < 0 || inst >= b->stat_inst) return -1;
    const int rows = std::min(max_rows, b->stat_rows);
    std::vector<double> h((size_t) b->stat_rows * GQP_STAT_COLS * b->stat_inst);
    HIPCHK(hipMemcpy(h.data(), b->D.stat, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
    for (int r = 0; r < rows; r++)
        for (int c = 0; c < GQP_STAT_COLS; c++)
            stat[r * GQP_STAT_COLS + c] = h[((size_t) r * GQP_STAT_COLS + c) * b->stat_inst + inst];
    return rows;
}
catch (const gqp_hip_failure &) { return -1; }

double ocp_qp_gpu_batch_get_scalar(ocp_qp_gpu_batch *b, const char *f)
try
{
    if (!strcmp(f, "time_tot")) return b->time_tot;
    if (!strcmp(f, "time_pack")) { double t = b->time_pack; b->time_pack = 0.0; return t; }
    if (!strcmp(f, "iter_max_batch")) return (double) b->last_iters;
    if (!strcmp(f, "launches")) return (double) b->launches;
    if (!strcmp(f, "time_xcond")) return b->time_xcond;
    if (!strcmp(f, "compactions")) return (double) b->n_compactions;
    if (!strcmp(f, "w16_tiles")) return (double) b->w16_tiles;
    if (!strcmp(f, "tail_switches")) return (double) b->n_tail_switches;
    if (!strcmp(f, "single_launch_solves")) return (double) b->n_single_launch;
    if (!strcmp(f, "cond_N_active")) return b->pcond_state == 1 ? (double) b->child->N : (double) b->N;
    if (!strcmp(f, "tol_comp_soft_scale")) return b->tol_comp_soft_scale;
    if (!strcmp(f, "polish")) return b->polish;
    if (!strcmp(f, "full_dense")) return b->full_dense;
    if (!strcmp(f, "dense_columns")) return b->kd.nv;
    if (!strcmp(f, "dense_workspace_bytes")) return (double) b->kd.W * 8.0 * b->kd_slice;
    if (!strcmp(f, "polished")) return b->child && b->pcond_state == 1 && b->cond_N > 0 && b->cond_N < b->N ? b->child->n_polished : b->n_polished;
    if (!strcmp(f, "polish_reverted")) return b->child && b->pcond_state == 1 && b->cond_N > 0 && b->cond_N < b->N ? b->child->n_polish_reverted : b->n_polish_reverted;
    if (!strcmp(f, "tol_comp_effective")) { finalize_structure(b); return effective_opts(b->O, b).tol_comp; }
    /* which condensing / expansion kernels serve the batch: 2 sixteen lanes per block, 1 one instance per lane, 0 one wave
     * per instance (meaningful once partial condensing is active) */
    if (!strcmp(f, "pcond_kernel")) return b->pcond_state != 1 ? -1.0 : (b->pcz && b->pc_rt && b->AW <= 1 && b->child->AW <= 1) ? (b->pcz_mfma ? 3.0 : 2.0) : b->pc_rt ? 0.0 : 1.0;
    if (!strcmp(f, "pexpand_kernel")) return b->pcond_state != 1 ? -1.0 : (!b->pc_rt || b->pc_lane_expand) ? 1.0 : 0.0;
    {
        /* accumulated per-kernel-class event times (ms) and launch counts since the last reset */
        const char *cls[6] = {"init", "back_fact", "fwd_aff", "back_rhs", "fwd_corr", "finalize"};
        for (int q = 0; q < 6; q++)
        {
            char nm[64];
            snprintf(nm, sizeof(nm), "prof_ms_%s", cls[q]);
            if (!strcmp(f, nm)) return b->prof_ms[q];
            snprintf(nm, sizeof(nm), "prof_cnt_%s", cls[q]);
            if (!strcmp(f, nm)) return (double) b->prof_cnt[q];
        }
        if (!strcmp(f, "prof_reset"))
        {
            for (int q = 0; q < 6; q++) { b->prof_ms[q] = 0.0; b->prof_cnt[q] = 0; }
            return 0.0;
        }
    }
    fprintf(stderr, "acados_amd: ocp_qp_gpu_batch_get_scalar: unknown field %s\n", f);
    return -1.0;
}
catch (const gqp_hip_failure &) { return NAN; } /* "tol_comp_effective" finalises the structure on the device */

int ocp_qp_gpu_batch_condense_lhs(ocp_qp_gpu_batch *b)
try
{
    HIPCHK(hipSetDevice(b->device));
    finalize_structure(b);
    b->lhs_ready = false;
    if (!(b->cond_N > 0 && b->cond_N < b->N)) return 0;
    if (b->pcond_state == 0) pcond_setup(b);
    if (b->pcond_state != 1) return 0;
    b->pmap.mode = 1;
    pcond_launch(b, false);
    HIPCHK(hipStreamSynchronize(b->stream));
    b->lhs_ready = true;
    return 0;
}
catch (const gqp_hip_failure &) { return -1; }

/* condensing-only boundary (interfaces/acados_c/condensing_interface.h:73-75): the condensed QP as an object of its
 * own, and the expansion of a solution of it */
ocp_qp_gpu_batch *ocp_qp_gpu_batch_condense(ocp_qp_gpu_batch *b)
try
{
    HIPCHK(hipSetDevice(b->device));
    finalize_structure(b);
    if (!(b->cond_N > 0 && b->cond_N < b->N)) return nullptr;
    if (b->pcond_state == 0) pcond_setup(b);
    if (b->pcond_state != 1) return nullptr;
    b->pmap.mode = 3;
    pcond_launch(b, false);
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipGetLastError());
    b->lhs_ready = false;
    return b->child;
}
catch (const gqp_hip_failure &) { return nullptr; }

/* vector part alone (gbar, bbar, bounds) on top of a resident matrix part: the `condense_rhs` slot of
 * ocp_qp_xcond_config (ocp_qp_partial_condensing.c:602-630); returns the condensed batch like _condense */
ocp_qp_gpu_batch *ocp_qp_gpu_batch_condense_rhs(ocp_qp_gpu_batch *b)
try
{
    HIPCHK(hipSetDevice(b->device));
    finalize_structure(b);
    if (b->pcond_state != 1 || !b->child) return ocp_qp_gpu_batch_condense(b); /* no lhs yet: everything */
    b->pmap.mode = 2;
    pcond_launch(b, false);
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipGetLastError());
    return b->child;
}
catch (const gqp_hip_failure &) { return nullptr; }

/* the current iterate of `b` restated in the condensed variables, written into the condensed batch: the
 * `condense_qp_out` slot (ocp_qp_partial_condensing.c:559-571) */
int ocp_qp_gpu_batch_condense_sol(ocp_qp_gpu_batch *b)
try
{
    HIPCHK(hipSetDevice(b->device));
    if (b->pcond_state != 1 || !b->child) return -1;
    hipLaunchKernelGGL(gqp::k_pcond_sol, dim3((b->B + 63) / 64), dim3(64), 0, b->stream, b->D, b->child->D, b->pmap);
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipGetLastError());
    return 0;
}
catch (const gqp_hip_failure &) { return -1; }

/* the condensed batch currently owned by `b` (after _condense / _condense_lhs), or NULL */
ocp_qp_gpu_batch *ocp_qp_gpu_batch_condensed(ocp_qp_gpu_batch *b) { return b->pcond_state == 1 ? b->child : nullptr; }

int ocp_qp_gpu_batch_expand(ocp_qp_gpu_batch *b)
try
{
    HIPCHK(hipSetDevice(b->device));
    if (b->pcond_state != 1 || !b->child) return -1;
    HIPCHK(hipStreamSynchronize(b->child->stream));
    pcond_launch(b, true);
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipGetLastError());
    return 0;
}
catch (const gqp_hip_failure &) { return -1; }

int ocp_qp_gpu_batch_get_dims(ocp_qp_gpu_batch *b, const char *f, int *out)
{
    const std::vector<int> *v = nullptr;
    if (!strcmp(f, "N")) { out[0] = b->N; return 0; }
    if (!strcmp(f, "n_batch")) { out[0] = b->B; return 0; }
    if (!strcmp(f, "nx")) v = &b->nx; else if (!strcmp(f, "nu")) v = &b->nu; else if (!strcmp(f, "nbx")) v = &b->nbx;
    else if (!strcmp(f, "nbu")) v = &b->nbu; else if (!strcmp(f, "nb")) v = &b->nb; else if (!strcmp(f, "ng")) v = &b->ng;
    else if (!strcmp(f, "ns")) v = &b->ns; else if (!strcmp(f, "nbxe")) v = &b->nbxe;
    if (!v) { fprintf(stderr, "acados_amd: ocp_qp_gpu_batch_get_dims: unknown field %s\n", f); return -1; }
    for (int k = 0; k <= b->N; k++) out[k] = (*v)[k];
    return 0;
}

int ocp_qp_gpu_batch_get_int(ocp_qp_gpu_batch *b, const char *f, int k, int *out)
{
    if (k < 0 || k > b->N) return -1;
    const std::vector<int> *v = nullptr;
    if (!strcmp(f, "idxb")) v = &b->idxb[k]; else if (!strcmp(f, "idxs_rev")) v = &b->idxs_rev[k]; else if (!strcmp(f, "idxe")) v = &b->idxe[k];
    if (!v) { fprintf(stderr, "acados_amd: ocp_qp_gpu_batch_get_int: unknown field %s\n", f); return -1; }
    for (size_t e = 0; e < v->size(); e++) out[e] = (*v)[e];
    return (int) v->size();
}

int ocp_qp_gpu_batch_condense_rhs_and_solve(ocp_qp_gpu_batch *b)
try
{
    const int bad = ocp_qp_gpu_batch_solve(b); /* uses mode 2 when the lhs is in place */
    b->lhs_ready = false;
    return bad;
}
catch (const gqp_hip_failure &) { return -1; }


/* ---- bulk pack / unpack ---- */
static const char *const k_bulk_in_fields[] = {"A", "B", "b", "Q", "S", "R", "q", "r", "lbu", "ubu", "lbx", "ubx", "lg", "ug",
                                               "C", "D", "Zl", "Zu", "zl", "zu", "lls", "lus", "lbu_mask", "ubu_mask",
                                               "lbx_mask", "ubx_mask", "lg_mask", "ug_mask", "lls_mask", "lus_mask"};
static const char *const k_bulk_out_fields[] = {"u", "x", "sl", "su", "pi", "lam", "t"};
/* the VECTOR part of the input blob (which = 2): what changes between the two halves of an RTI step -- gradient, dynamics offset,
 * bounds, masks (ocp_nlp_common.c:3119-3138 writes exactly these between condense_lhs and condense_rhs_and_solve); 12.6 KB per
 * C2-shaped QP against 85 KB for the whole blob */
static const char *const k_bulk_vec_fields[] = {"b", "q", "r", "lbu", "ubu", "lbx", "ubx", "lg", "ug", "zl", "zu", "lls", "lus", "lbu_mask", "ubu_mask",
                                                "lbx_mask", "ubx_mask", "lg_mask", "ug_mask", "lls_mask", "lus_mask"};

static void bulk_build(ocp_qp_gpu_batch *b, ocp_qp_gpu_batch::BulkMap &M, const char *const *fields, int nf)
{
    if (M.built) return;
    finalize_structure(b);
    const GqpDev &D = b->D;
    const GArr table[16] = {D.BAt, D.bvec, D.RSQ, D.rq, D.dvec, D.DCt, D.Zz, D.ux, D.sv, D.pi, D.lam, D.t,
                            {nullptr, 0, 0}, {nullptr, 0, 0}, {nullptr, 0, 0}, {nullptr, 0, 0}};
    for (int q = 0; q < 16; q++) M.T.a[q] = table[q];
    auto table_index = [&](const GArr &a) {
        for (int q = 0; q < 12; q++) if (table[q].p == a.p) return q;
        return -1;
    };
    std::vector<int> h_arr, h_elem, h_moff, h_mstage, h_mbit, h_arr_g, h_elem_g;
    for (int k = 0; k <= b->N; k++)
        for (int fi = 0; fi < nf; fi++)
        {
            const char *f = fields[fi];
            std::vector<int> map, map2;
            GArr arr = {nullptr, 0, 0}, arr2 = {nullptr, 0, 0};
            const size_t flen = strlen(f);
            int len;
            const bool is_mask = flen > 5 && !strcmp(f + flen - 5, "_mask");
            if (is_mask) len = mask_bits(b, f, k, map);
            else len = field_map(b, f, k, map, &arr, &map2, &arr2);
            if (len <= 0) continue;
            M.fields.push_back(f); M.seg_stage.push_back(k); M.seg_off.push_back((int) h_arr.size()); M.seg_len.push_back(len);
            for (int e = 0; e < len; e++)
            {
                if (is_mask)
                {
                    h_moff.push_back((int) h_arr.size()); h_mstage.push_back(k); h_mbit.push_back(map[e]);
                    h_arr.push_back(-1); h_elem.push_back(0);
                }
                else
                {
                    h_arr.push_back(map[e] >= 0 ? table_index(arr) : -1); h_elem.push_back(map[e] >= 0 ? map[e] : 0);
                }
            }
            if (nf == 30) /* the input blob: read-direction map */
            {
                const bool sym = !strcmp(f, "Q") || !strcmp(f, "R");
                const int dim = sym ? (f[0] == 'Q' ? b->nx[k] : b->nu[k]) : 0;
                for (int e = 0; e < len; e++)
                {
                    int a = h_arr[h_arr.size() - len + e], el = h_elem[h_elem.size() - len + e];
                    if (sym && a < 0)
                    {
                        const int r = e % dim, c = e / dim;       /* column-major, r < c here */
                        const int m = map[r * dim + c];           /* element (c, r) */
                        if (m >= 0) { a = table_index(arr); el = m; }
                    }
                    h_arr_g.push_back(a); h_elem_g.push_back(el);
                }
            }
            if (arr2.p)
            {
                /* equality-flagged x bounds also define the value of the variable: a second, hidden
                 * segment that re-reads the same blob entries is not possible, so those entries are
                 * written through a duplicate segment placed right after (the caller's blob carries
                 * lbx twice: once as bound, once as value -- see ocp_qp_gpu_batch_bulk_offset) */
                M.fields.push_back(std::string(f) + "#value"); M.seg_stage.push_back(k);
                M.seg_off.push_back((int) h_arr.size()); M.seg_len.push_back(len);
                for (int e = 0; e < len; e++)
                {
                    h_arr.push_back(map2[e] >= 0 ? table_index(arr2) : -1); h_elem.push_back(map2[e] >= 0 ? map2[e] : 0);
                    if (nf == 30) { h_arr_g.push_back(h_arr.back()); h_elem_g.push_back(h_elem.back()); }
                }
            }
        }
    M.len = (int) h_arr.size();
    M.nm = (int) h_moff.size();
    M.d_arr = dalloc<int>(b, M.len); M.d_elem = dalloc<int>(b, M.len);
    HIPCHK(hipMemcpy(M.d_arr, h_arr.data(), sizeof(int) * M.len, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(M.d_elem, h_elem.data(), sizeof(int) * M.len, hipMemcpyHostToDevice));
    if (nf == 30 && M.len)
    {
        M.d_arr_g = dalloc<int>(b, M.len); M.d_elem_g = dalloc<int>(b, M.len);
        HIPCHK(hipMemcpy(M.d_arr_g, h_arr_g.data(), sizeof(int) * M.len, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(M.d_elem_g, h_elem_g.data(), sizeof(int) * M.len, hipMemcpyHostToDevice));
    }
    M.d_moff = dalloc<int>(b, M.nm); M.d_mstage = dalloc<int>(b, M.nm); M.d_mbit = dalloc<int>(b, M.nm);
    if (M.nm)
    {
        HIPCHK(hipMemcpy(M.d_moff, h_moff.data(), sizeof(int) * M.nm, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(M.d_mstage, h_mstage.data(), sizeof(int) * M.nm, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(M.d_mbit, h_mbit.data(), sizeof(int) * M.nm, hipMemcpyHostToDevice));
    }
    M.built = true;
}

static int gqp_bulk_len_impl(ocp_qp_gpu_batch *b, int output) /* throws gqp_hip_failure: callers are inside a guarded entry */
{
    HIPCHK(hipSetDevice(b->device));
    auto &M = output == 2 ? b->bulk_vec : (output ? b->bulk_out : b->bulk_in);
    bulk_build(b, M, output == 2 ? k_bulk_vec_fields : (output ? k_bulk_out_fields : k_bulk_in_fields), output == 2 ? 21 : (output ? 7 : 30));
    return M.len;
}

static int gqp_bulk_offset_impl(ocp_qp_gpu_batch *b, int output, const char *field, int stage, int *len)
{
    gqp_bulk_len_impl(b, output);
    auto &M = output == 2 ? b->bulk_vec : (output ? b->bulk_out : b->bulk_in);
    for (size_t q = 0; q < M.fields.size(); q++)
        if (M.seg_stage[q] == stage && M.fields[q] == field)
        {
            if (len) *len = M.seg_len[q];
            return M.seg_off[q];
        }
    if (len) *len = 0;
    return -1;
}

/* the C-ABI entries of the layout queries: the first of them after create builds the per-stage structure and the segment
 * tables on the device (dalloc / hipMemcpy / hipFuncSetAttribute) -- where an out-of-memory error of a large batch shows up
 * first.  No exception crosses the C ABI: -1 (no valid length / offset is negative) */
int ocp_qp_gpu_batch_bulk_len(ocp_qp_gpu_batch *b, int output)
try { return gqp_bulk_len_impl(b, output); }
catch (const gqp_hip_failure &) { return -1; }

int ocp_qp_gpu_batch_bulk_offset(ocp_qp_gpu_batch *b, int output, const char *field, int stage, int *len)
try { return gqp_bulk_offset_impl(b, output, field, stage, len); }
catch (const gqp_hip_failure &) { if (len) *len = 0; return -1; }

#define GQP_MASK_SPC 4 /* stages per thread of k_bulk_masks */

/* the arrays of a bulk map -> blob (instance-major): as bulk_scatter_launch below */
static void bulk_gather_launch(ocp_qp_gpu_batch *b, double *dst, int len, const int *d_arr, const int *d_elem, const gqp::GArrTable &T)
{
    if (b->aos && !getenv("ACADOS_AMD_SCATTER_PLAIN"))
        hipLaunchKernelGGL(gqp::k_bulk_gather_aos, dim3((len + 255) / 256, b->B), dim3(256), 0, b->stream, dst, b->B, len, d_arr, d_elem, T);
    else if (!getenv("ACADOS_AMD_SCATTER_PLAIN"))
        GQP_LAUNCH_COOP(gqp::k_bulk_gather_tile, dim3((b->B + 63) / 64, (len + 63) / 64), dim3(64), 0, b->stream, dst, b->B, len, d_arr, d_elem, T);
    else
        hipLaunchKernelGGL(gqp::k_bulk_gather, dim3((b->B + 63) / 64, (len + 255) / 256), dim3(64), 0, b->stream, dst, b->B, len, d_arr, d_elem, T);
}

/* blob (instance-major) -> the arrays of a bulk map: lanes along the elements where the destinations are instance-major too, through
 * an LDS tile where they are wave-tiled (ACADOS_AMD_SCATTER_PLAIN=1: one lane per instance, the cross-check) */
static void bulk_scatter_launch(ocp_qp_gpu_batch *b, const double *src, int len, ocp_qp_gpu_batch::BulkMap &M)
{
    if (b->aos && !getenv("ACADOS_AMD_SCATTER_PLAIN"))
        hipLaunchKernelGGL(gqp::k_bulk_scatter_aos, dim3((len + 255) / 256, b->B), dim3(256), 0, b->stream, src, b->B, len, M.d_arr, M.d_elem, M.T);
    else if (!getenv("ACADOS_AMD_SCATTER_PLAIN"))
        GQP_LAUNCH_COOP(gqp::k_bulk_scatter_tile, dim3((b->B + 63) / 64, (len + 63) / 64), dim3(64), 0, b->stream, src, b->B, len, M.d_arr, M.d_elem, M.T);
    else
        hipLaunchKernelGGL(gqp::k_bulk_scatter, dim3((b->B + 63) / 64, (len + 255) / 256), dim3(64), 0, b->stream, src, b->B, len, M.d_arr, M.d_elem, M.T);
}

int ocp_qp_gpu_batch_set_bulk(ocp_qp_gpu_batch *b, const double *blob, int is_device)
try
{
    const int len = gqp_bulk_len_impl(b, 0);
    auto &M = b->bulk_in;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, b->stream));
    const double *src = stage_in(b, blob, (size_t) b->B * len, is_device);
    const dim3 block(64);
    bulk_scatter_launch(b, src, len, M);
    if (M.nm)
        hipLaunchKernelGGL(gqp::k_bulk_masks, dim3((b->B + 63) / 64, (b->N + 1 + GQP_MASK_SPC - 1) / GQP_MASK_SPC), block, 0, b->stream, src, b->B, len, M.d_moff,
                           M.d_mstage, M.d_mbit, M.nm, b->D.amask, b->AW, GQP_MASK_SPC);
    HIPCHK(hipEventRecord(e1, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    b->time_pack += ms * 1e-3;
    HIPCHK(hipEventDestroy(e0)); HIPCHK(hipEventDestroy(e1));
    return 0;
}
catch (const gqp_hip_failure &) { return -1; }

/* the vector fields alone (blob layout: ocp_qp_gpu_batch_bulk_len / _bulk_offset with which = 2): the matrices stay what the last
 * _set_bulk brought -- the host side of an RTI feedback step (condense_lhs done, new gradient / offsets / bounds) */
int ocp_qp_gpu_batch_set_bulk_vec(ocp_qp_gpu_batch *b, const double *blob, int is_device)
try
{
    const int len = gqp_bulk_len_impl(b, 2);
    auto &M = b->bulk_vec;
    HIPCHK(hipEventRecord(b->ev0, b->stream));
    const double *src = stage_in(b, blob, (size_t) b->B * len, is_device);
    const dim3 block(64);
    bulk_scatter_launch(b, src, len, M);
    if (M.nm)
        hipLaunchKernelGGL(gqp::k_bulk_masks, dim3((b->B + 63) / 64, (b->N + 1 + GQP_MASK_SPC - 1) / GQP_MASK_SPC), block, 0, b->stream, src, b->B, len, M.d_moff,
                           M.d_mstage, M.d_mbit, M.nm, b->D.amask, b->AW, GQP_MASK_SPC);
    HIPCHK(hipEventRecord(b->ev1, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, b->ev0, b->ev1));
    b->time_pack += ms * 1e-3;
    return 0;
}
catch (const gqp_hip_failure &) { return -1; }

/* ---- zero-copy gather: the device reads the instances' QP data from the caller's own (registered) host memory ----
 * Through the boundary the host pass that copies n capsules' member arrays into a pinned blob is bound by the host memory system
 * (1.8 GB/s per thread on 16 threads) while a kernel reads registered host memory at the PCIe rate (56 GB/s,
 * profiles/r06_zero_copy_probe.txt).  _host_register pins a block and maps it for the device (hipHostRegister; device pointer = host
 * pointer); _gather_tables takes the class-wide word tables once; _gather_run takes the per-instance source pointers of this call,
 * gathers into the device-side blob and scatters it as _set_bulk / _set_bulk_vec do -- byte-identical to the blob path. */
int ocp_qp_gpu_host_register(void *p, size_t bytes)
{
    if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) { (void) hipGetLastError(); return -1; }
    void *dp = nullptr; /* the gather takes HOST addresses: only where the device sees the block at the same one */
    if (hipHostGetDevicePointer(&dp, p, 0) != hipSuccess || dp != p)
    {
        (void) hipGetLastError();
        (void) hipHostUnregister(p);
        return -1;
    }
    return 0;
}

int ocp_qp_gpu_host_unregister(void *p)
{
    const hipError_t e = hipHostUnregister(p);
    if (e != hipSuccess) { (void) hipGetLastError(); return -1; }
    return 0;
}

int ocp_qp_gpu_batch_gather_tables(ocp_qp_gpu_batch *b, int which, int P, int n_words, const int *w_slot, const int *w_off, const int *w_pos,
                                   const unsigned char *w_neg)
try
{
    HIPCHK(hipSetDevice(b->device));
    const int len = gqp_bulk_len_impl(b, which == 2 ? 2 : 0);
    for (int w = 0; w < n_words; w++)
        if (w_slot[w] < 0 || w_slot[w] >= P || w_pos[w] < 0 || w_pos[w] >= len || w_off[w] < 0) return -1;
    auto &G = b->gtab[which == 2 ? 1 : 0];
    G.P = P; G.n_words = n_words;
    {
        /* do the words write EVERY position of the blob?  If not, the gather clears its buffer first: the positions no word writes
         * are zero for the scatter (the buffer is shared with the other blob and the chunk protocol) */
        std::vector<char> seen((size_t) len, 0);
        int covered = 0;
        for (int w = 0; w < n_words; w++) if (!seen[w_pos[w]]) { seen[w_pos[w]] = 1; covered++; }
        G.full = covered == len;
    }
    G.d_slot = dalloc<int>(b, n_words); G.d_off = dalloc<int>(b, n_words); G.d_pos = dalloc<int>(b, n_words); G.d_neg = dalloc<unsigned char>(b, n_words);
    if (n_words)
    {
        HIPCHK(hipMemcpy(G.d_slot, w_slot, sizeof(int) * n_words, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(G.d_off, w_off, sizeof(int) * n_words, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(G.d_pos, w_pos, sizeof(int) * n_words, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(G.d_neg, w_neg, n_words, hipMemcpyHostToDevice));
    }
    return 0;
}
catch (const gqp_hip_failure &) { return -1; }

int ocp_qp_gpu_batch_gather_run(ocp_qp_gpu_batch *b, int which, const void *const *ptrs)
try
{
    HIPCHK(hipSetDevice(b->device));
    auto &G = b->gtab[which == 2 ? 1 : 0];
    if (G.n_words <= 0 || G.P <= 0) return -1;
    const int len = gqp_bulk_len_impl(b, which == 2 ? 2 : 0);
    const size_t cnt = (size_t) b->B * gqp_bulk_len_impl(b, 0); /* (one buffer for both blobs: the full one is the longer) */
    if (cnt > b->chunks_cap)
    {
        HIPCHK(hipStreamSynchronize(b->stream));
        b->chunks_cap = cnt;
        b->d_chunks = dalloc<double>(b, b->chunks_cap);
    }
    const size_t np = (size_t) b->B * G.P;
    if (np > b->gptrs_cap) { b->gptrs_cap = np; b->d_gptrs = dalloc<const double *>(b, np); }
    hipEvent_t g0, g1;
    HIPCHK(hipEventCreate(&g0)); HIPCHK(hipEventCreate(&g1));
    const double tp = b->time_pack;
    HIPCHK(hipEventRecord(g0, b->stream));
    HIPCHK(hipMemcpyAsync(b->d_gptrs, ptrs, sizeof(void *) * np, hipMemcpyHostToDevice, b->stream));
    if (!G.full) HIPCHK(hipMemsetAsync(b->d_chunks, 0, sizeof(double) * (size_t) b->B * (size_t) len, b->stream));
    hipLaunchKernelGGL(gqp::k_gather_host, dim3((G.n_words + 255) / 256, b->B), dim3(256), 0, b->stream, (const double *const *) b->d_gptrs, G.P, b->B,
                       G.n_words, (const int *) G.d_slot, (const int *) G.d_off, (const int *) G.d_pos, (const unsigned char *) G.d_neg, b->d_chunks, len);
    b->chunks_got = 0; /* (the chunk protocol shares the buffer: a round in flight is void) */
    const int rc = which == 2 ? ocp_qp_gpu_batch_set_bulk_vec(b, b->d_chunks, 1) : ocp_qp_gpu_batch_set_bulk(b, b->d_chunks, 1);
    HIPCHK(hipEventRecord(g1, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, g0, g1));
    b->time_pack = tp + ms * 1e-3; /* gather + scatter */
    HIPCHK(hipEventDestroy(g0)); HIPCHK(hipEventDestroy(g1));
    return rc;
}
catch (const gqp_hip_failure &) { return -1; }

/* _set_bulk in pieces: a caller that fills its blob instance range by instance range (host threads unpacking n acados structs)
 * hands every finished range over at once -- the host->device copy of range j runs while range j + 1 is still being filled --
 * and scatters when the last one is in.  `blob_chunk` points at instance `first` of the caller's (pinned) blob; nothing is waited
 * for here.  _set_bulk_staged: the scatter launch (+ masks) over what the chunks brought, then the usual wait. */
int ocp_qp_gpu_batch_set_bulk_chunk(ocp_qp_gpu_batch *b, const double *blob_chunk, int first, int count)
try
{
    const int len = gqp_bulk_len_impl(b, 0);
    if (first < 0 || count < 0 || first + count > b->B) return -1;
    const size_t cnt = (size_t) b->B * len;
    if (cnt > b->chunks_cap)
    {
        HIPCHK(hipStreamSynchronize(b->stream)); /* (only ever on the first chunk of a batch: nothing of it is in flight yet) */
        b->chunks_cap = cnt; /* B * len is fixed for a batch: exactly that */
        b->d_chunks = dalloc<double>(b, b->chunks_cap);
        b->chunks_got = 0;
    }
    if (b->chunks_got == 0)
    {
        if (!b->chunks_ev) HIPCHK(hipEventCreate(&b->chunks_ev));
        HIPCHK(hipEventRecord(b->chunks_ev, b->stream));
    }
    b->chunks_got += count;
    if (count)
        HIPCHK(hipMemcpyAsync(b->d_chunks + (size_t) first * len, blob_chunk, sizeof(double) * (size_t) count * len, hipMemcpyHostToDevice, b->stream));
    return 0;
}
catch (const gqp_hip_failure &) { return -1; }

int ocp_qp_gpu_batch_set_bulk_staged(ocp_qp_gpu_batch *b)
try
{
    const int len = gqp_bulk_len_impl(b, 0);
    auto &M = b->bulk_in;
    if ((size_t) b->B * len > b->chunks_cap) return -1; /* no chunk was ever handed over */
    /* every instance must have arrived in THIS round: a missing range would scatter the previous call's data for those instances
     * (ranges are the caller's to keep disjoint: the count is what can be checked here) */
    const long got = b->chunks_got;
    b->chunks_got = 0;
    if (got != (long) b->B)
    {
        fprintf(stderr, "acados_amd: ocp_qp_gpu_batch_set_bulk_staged: %ld of %d instances were handed over since the last scatter\n", got, b->B);
        return -1;
    }
    const dim3 block(64);
    bulk_scatter_launch(b, (const double *) b->d_chunks, len, M);
    if (M.nm)
        hipLaunchKernelGGL(gqp::k_bulk_masks, dim3((b->B + 63) / 64, (b->N + 1 + GQP_MASK_SPC - 1) / GQP_MASK_SPC), block, 0, b->stream, (const double *) b->d_chunks, b->B, len, M.d_moff,
                           M.d_mstage, M.d_mbit, M.nm, b->D.amask, b->AW, GQP_MASK_SPC);
    HIPCHK(hipEventRecord(b->ev1, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, b->chunks_ev, b->ev1));
    b->time_pack += ms * 1e-3;
    return 0;
}
catch (const gqp_hip_failure &) { return -1; }

/* the QP data of the batch in the INPUT blob layout (inverse of _set_bulk): how the acados-side condensing module reads the
 * condensed QP of a child batch back into acados' containers (integration/ocp_qp_gpu_pcond.c) -- one launch (+ one for the
 * masks) and one device->host copy instead of one copy per field and stage */
int ocp_qp_gpu_batch_get_bulk_in(ocp_qp_gpu_batch *b, double *blob, int is_device)
try
{
    const int len = gqp_bulk_len_impl(b, 0);
    auto &M = b->bulk_in;
    const size_t cnt = (size_t) b->B * len;
    double *dst = blob;
    if (!is_device)
    {
        if (cnt > b->stage_cap) { b->stage_cap = cnt * 2; b->d_stage = dalloc<double>(b, b->stage_cap); }
        dst = b->d_stage;
    }
    const dim3 grid((b->B + 63) / 64, (len + 255) / 256), block(64);
    bulk_gather_launch(b, dst, len, M.d_arr_g, M.d_elem_g, M.T);
    if (M.nm)
        hipLaunchKernelGGL(gqp::k_bulk_masks_get, dim3((b->B + 63) / 64), block, 0, b->stream, dst, b->B, len, M.d_moff,
                           M.d_mstage, M.d_mbit, M.nm, b->D.amask, b->AW);
    if (!is_device) HIPCHK(hipMemcpyAsync(blob, dst, sizeof(double) * cnt, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return 0;
}
catch (const gqp_hip_failure &) { return -1; }

/* an iterate in the OUTPUT blob layout (u x sl su pi lam t) written into the batch: the starting point of a hot
 * start, one host->device copy and one launch (inverse of _get_bulk) */
int ocp_qp_gpu_batch_set_bulk_out(ocp_qp_gpu_batch *b, const double *blob, int is_device)
try
{
    const int len = gqp_bulk_len_impl(b, 1);
    auto &M = b->bulk_out;
    const double *src = stage_in(b, blob, (size_t) b->B * len, is_device);
    bulk_scatter_launch(b, src, len, M);
    HIPCHK(hipStreamSynchronize(b->stream));
    return 0;
}
catch (const gqp_hip_failure &) { return -1; }

int ocp_qp_gpu_batch_get_bulk(ocp_qp_gpu_batch *b, double *blob, int is_device)
try
{
    const int len = gqp_bulk_len_impl(b, 1);
    auto &M = b->bulk_out;
    const size_t cnt = (size_t) b->B * len;
    double *dst = blob;
    if (!is_device)
    {
        if (cnt > b->stage_cap) { b->stage_cap = cnt * 2; b->d_stage = dalloc<double>(b, b->stage_cap); }
        dst = b->d_stage;
    }
    const dim3 grid((b->B + 63) / 64, (len + 255) / 256), block(64);
    bulk_gather_launch(b, dst, len, M.d_arr, M.d_elem, M.T);
    if (!is_device) HIPCHK(hipMemcpyAsync(blob, dst, sizeof(double) * cnt, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return 0;
}
catch (const gqp_hip_failure &) { return -1; }

/* ---- bulk seeds / sensitivities: every seed of every instance in one host->device copy and one launch, every
 * direction back in one launch and one copy (the batched eval_forw_sens / eval_adj_sens of the acados-side adapter;
 * callers interfaces/acados_template/acados_template/c_templates_tera/acados_solver.in.c:3292-3337) ---- */
static const char *const k_seed_fields[] = {"r", "q", "zl", "zu", "b", "lbu", "lbx", "lg", "ubu", "ubx", "ug", "lls", "lus"};

static void seed_build(ocp_qp_gpu_batch *b)
{
    auto &M = b->bulk_seed;
    if (M.built) return;
    finalize_structure(b);
    const GqpDev &D = b->D;
    std::vector<int> h_arr, h_elem, h_sgn, h_elem2;
    for (int k = 0; k <= b->N; k++)
        for (const char *f : k_seed_fields)
        {
            std::vector<int> map, map2;
            GArr arr = {nullptr, 0, 0}, arr2 = {nullptr, 0, 0};
            const int len = field_map(b, f, k, map, &arr, &map2, &arr2);
            if (len <= 0) continue;
            const GqpStage &S = b->st[k];
            const bool slack_grad = f[0] == 'z';
            int a = -1, sgn = 1;
            if (slack_grad) a = 1;
            else if (arr.p == D.rq.p) a = 0;
            else if (arr.p == D.bvec.p) a = 2;
            else if (arr.p == D.dvec.p) { a = 3; sgn = (f[0] == 'l') ? -1 : 1; } /* lbu lbx lg lls lus: lower bounds */
            M.fields.push_back(std::string("seed_") + f); M.seg_stage.push_back(k);
            M.seg_off.push_back((int) h_arr.size()); M.seg_len.push_back(len);
            for (int e = 0; e < len; e++)
            {
                int el = map[e];
                if (slack_grad) el = S.o_s + (f[1] == 'u' ? S.ns : 0) + e; /* rgs is indexed like sv */
                h_arr.push_back(el >= 0 ? a : -1); h_elem.push_back(el >= 0 ? el : 0); h_sgn.push_back(sgn);
                h_elem2.push_back(arr2.p && map2[e] >= 0 ? map2[e] : -1);
            }
        }
    M.len = (int) h_arr.size();
    M.d_arr = dalloc<int>(b, M.len); M.d_elem = dalloc<int>(b, M.len); M.d_sgn = dalloc<int>(b, M.len); M.d_elem2 = dalloc<int>(b, M.len);
    if (M.len)
    {
        HIPCHK(hipMemcpy(M.d_arr, h_arr.data(), sizeof(int) * M.len, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(M.d_elem, h_elem.data(), sizeof(int) * M.len, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(M.d_sgn, h_sgn.data(), sizeof(int) * M.len, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(M.d_elem2, h_elem2.data(), sizeof(int) * M.len, hipMemcpyHostToDevice));
    }
    M.built = true;
}

static int gqp_sens_bulk_len_impl(ocp_qp_gpu_batch *b, int output)
{
    if (output) return gqp_bulk_len_impl(b, 1);
    HIPCHK(hipSetDevice(b->device));
    seed_build(b);
    return b->bulk_seed.len;
}

static int gqp_sens_bulk_offset_impl(ocp_qp_gpu_batch *b, int output, const char *field, int stage, int *len)
{
    if (output)
    {
        if (strncmp(field, "sens_", 5)) { if (len) *len = 0; return -1; }
        return gqp_bulk_offset_impl(b, 1, field + 5, stage, len);
    }
    gqp_sens_bulk_len_impl(b, 0);
    auto &M = b->bulk_seed;
    for (size_t q = 0; q < M.fields.size(); q++)
        if (M.seg_stage[q] == stage && M.fields[q] == field)
        {
            if (len) *len = M.seg_len[q];
            return M.seg_off[q];
        }
    if (len) *len = 0;
    return -1;
}

int ocp_qp_gpu_batch_sens_bulk_len(ocp_qp_gpu_batch *b, int output)
try { return gqp_sens_bulk_len_impl(b, output); }
catch (const gqp_hip_failure &) { return -1; }

int ocp_qp_gpu_batch_sens_bulk_offset(ocp_qp_gpu_batch *b, int output, const char *field, int stage, int *len)
try { return gqp_sens_bulk_offset_impl(b, output, field, stage, len); }
catch (const gqp_hip_failure &) { if (len) *len = 0; return -1; }

int ocp_qp_gpu_batch_sens_set_bulk(ocp_qp_gpu_batch *b, const double *blob, int is_device)
try
{
    const int len = gqp_sens_bulk_len_impl(b, 0);
    if (sens_begin(b)) return -1; /* zeroes the seed arrays, factorises at the solution where the sweeps run in place */
    if (len == 0) return 0;
    auto &M = b->bulk_seed;
    const GqpDev &D = b->D;
    const GArr table[5] = {D.rg, D.rgs, D.rb, D.rd, b->sfix};
    for (int q = 0; q < 16; q++) M.T.a[q] = q < 5 ? table[q] : GArr{nullptr, 0, 0};
    const double *src = stage_in(b, blob, (size_t) b->B * len, is_device);
    const dim3 grid((b->B + 63) / 64, (len + 255) / 256), block(64);
    hipLaunchKernelGGL(gqp::k_bulk_scatter_seed, grid, block, 0, b->stream, src, b->B, len, M.d_arr, M.d_elem, M.d_sgn, M.d_elem2, M.T);
    HIPCHK(hipStreamSynchronize(b->stream));
    return 0;
}
catch (const gqp_hip_failure &) { return -1; }

int ocp_qp_gpu_batch_sens_get_bulk(ocp_qp_gpu_batch *b, double *blob, int is_device)
try
{
    const int len = gqp_bulk_len_impl(b, 1);
    auto &M = b->bulk_out;
    gqp::GArrTable T = M.T; /* same element maps as the solution, read from the direction arrays */
    const GqpDev &D = b->D;
    T.a[7] = D.dux; T.a[8] = D.dsv; T.a[9] = D.dpi; T.a[10] = D.dlam; T.a[11] = D.dt;
    const size_t cnt = (size_t) b->B * len;
    double *dst = blob;
    if (!is_device)
    {
        if (cnt > b->stage_cap) { b->stage_cap = cnt * 2; b->d_stage = dalloc<double>(b, b->stage_cap); }
        dst = b->d_stage;
    }
    const dim3 grid((b->B + 63) / 64, (len + 255) / 256), block(64);
    bulk_gather_launch(b, dst, len, M.d_arr, M.d_elem, T);
    if (!is_device) HIPCHK(hipMemcpyAsync(blob, dst, sizeof(double) * cnt, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return 0;
}
catch (const gqp_hip_failure &) { return -1; }

/* pinned host memory for callers that stage their own blobs (the acados-side adapter is plain C and has no HIP) */
void *ocp_qp_gpu_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 8) != hipSuccess)
    {
        fprintf(stderr, "acados_amd: cannot allocate %zu bytes of pinned host memory\n", bytes);
        return nullptr;
    }
    return p;
}

void ocp_qp_gpu_host_free(void *p)
{
    if (p) (void) hipHostFree(p);
}

/* ---- multi-GPU: the ONLY collective of the path -- one gather over xGMI of the solutions and their
 * per-instance status / iteration counts plus the per-rank solve time, from device buffers on the batch's stream.
 * The communicator is a table of five transport entries (ocp_qp_gpu_comm_ops): RCCL's, bound at run time (dlopen: the copy
 * PyTorch has already loaded if there is one, so that the process holds a single RCCL instance; else the ROCm one), or a
 * table the host program supplies (ocp_qp_gpu_comm_create_from_ops: an MPI program, or the CPU test tier, which drives the
 * packing / offset / ordering logic below with torch.distributed's gloo as the transport).  One process per GPU; the RCCL
 * unique id travels between the processes by whatever means the host program has (torch.distributed broadcast in
 * bench.py, MPI_Bcast in a C harness). ---- */
struct ocp_qp_gpu_comm
{
    ocp_qp_gpu_comm_ops ops = {};
    int n = 0, rank = 0, device = 0;
    /* RCCL backing (null for an injected table) */
    void *lib = nullptr;
    void *comm = nullptr; /* ncclComm_t */
    struct uid { char internal[128]; };
    int (*get_uid)(uid *) = nullptr;
    int (*init_rank)(void **, int, uid, int) = nullptr;
    int (*nccl_all_gather)(const void *, void *, size_t, int, void *, void *) = nullptr;
    int (*nccl_send)(const void *, size_t, int, int, void *, void *) = nullptr;
    int (*nccl_recv)(void *, size_t, int, int, void *, void *) = nullptr;
    int (*nccl_group_start)() = nullptr;
    int (*nccl_group_end)() = nullptr;
    int (*destroy)(void *) = nullptr;
    const char *(*err)(int) = nullptr;
};

#if defined(__HIPCC__)
#include <dlfcn.h>
static bool rccl_bind(ocp_qp_gpu_comm *c)
{
    const char *names[] = {"librccl.so", "librccl.so.1"};
    for (const char *n : names)
        if ((c->lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break; /* already in the process (PyTorch's copy) */
    if (!c->lib)
    {
        const char *load[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : load)
            if ((c->lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    }
    if (!c->lib) { fprintf(stderr, "acados_amd: RCCL (librccl.so) cannot be loaded: %s\n", dlerror()); return false; }
    c->get_uid = (int (*)(ocp_qp_gpu_comm::uid *)) dlsym(c->lib, "ncclGetUniqueId");
    c->init_rank = (int (*)(void **, int, ocp_qp_gpu_comm::uid, int)) dlsym(c->lib, "ncclCommInitRank");
    c->nccl_all_gather = (int (*)(const void *, void *, size_t, int, void *, void *)) dlsym(c->lib, "ncclAllGather");
    c->nccl_send = (int (*)(const void *, size_t, int, int, void *, void *)) dlsym(c->lib, "ncclSend");
    c->nccl_recv = (int (*)(void *, size_t, int, int, void *, void *)) dlsym(c->lib, "ncclRecv");
    c->nccl_group_start = (int (*)()) dlsym(c->lib, "ncclGroupStart");
    c->nccl_group_end = (int (*)()) dlsym(c->lib, "ncclGroupEnd");
    c->destroy = (int (*)(void *)) dlsym(c->lib, "ncclCommDestroy");
    c->err = (const char *(*)(int)) dlsym(c->lib, "ncclGetErrorString");
    /* every entry the two gathers use is required HERE, on every rank alike: a rank that found out inside a collective
     * that it lacks ncclSend would leave while the others wait in the group (round-3 review) */
    if (!c->get_uid || !c->init_rank || !c->nccl_all_gather || !c->destroy || !c->nccl_send || !c->nccl_recv || !c->nccl_group_start
        || !c->nccl_group_end)
    {
        fprintf(stderr, "acados_amd: librccl.so lacks a collective / point-to-point entry point (ncclAllGather, ncclSend, ncclRecv, ncclGroupStart/End)\n");
        return false;
    }
    return true;
}

/* the RCCL rows of the transport table; ctx = the communicator.  ncclDataType_t: ncclInt32 = 2, ncclFloat64 = 8 (rccl.h) --
 * the table's dtype codes are those numbers */
static int rccl_report(ocp_qp_gpu_comm *c, int r, const char *what)
{
    if (r != 0) fprintf(stderr, "acados_amd: RCCL error %d (%s) in %s\n", r, c->err ? c->err(r) : "?", what);
    return r;
}
static int rccl_op_all_gather(void *ctx, const void *sendbuf, void *recvbuf, size_t count, int dtype, void *stream)
{
    ocp_qp_gpu_comm *c = (ocp_qp_gpu_comm *) ctx;
    return rccl_report(c, c->nccl_all_gather(sendbuf, recvbuf, count, dtype, c->comm, stream), "ncclAllGather");
}
static int rccl_op_send(void *ctx, const void *buf, size_t count, int dtype, int peer, void *stream)
{
    ocp_qp_gpu_comm *c = (ocp_qp_gpu_comm *) ctx;
    return rccl_report(c, c->nccl_send(buf, count, dtype, peer, c->comm, stream), "ncclSend");
}
static int rccl_op_recv(void *ctx, void *buf, size_t count, int dtype, int peer, void *stream)
{
    ocp_qp_gpu_comm *c = (ocp_qp_gpu_comm *) ctx;
    return rccl_report(c, c->nccl_recv(buf, count, dtype, peer, c->comm, stream), "ncclRecv");
}
static int rccl_op_group_start(void *ctx) { ocp_qp_gpu_comm *c = (ocp_qp_gpu_comm *) ctx; return rccl_report(c, c->nccl_group_start(), "ncclGroupStart"); }
static int rccl_op_group_end(void *ctx) { ocp_qp_gpu_comm *c = (ocp_qp_gpu_comm *) ctx; return rccl_report(c, c->nccl_group_end(), "ncclGroupEnd"); }

int ocp_qp_gpu_comm_unique_id(void *id128)
{
    ocp_qp_gpu_comm c;
    if (!rccl_bind(&c)) return -1;
    return rccl_report(&c, c.get_uid((ocp_qp_gpu_comm::uid *) id128), "ncclGetUniqueId") == 0 ? 0 : -1;
}

ocp_qp_gpu_comm *ocp_qp_gpu_comm_create(const void *id128, int n_ranks, int rank, int device)
try
{
    ocp_qp_gpu_comm *c = new ocp_qp_gpu_comm();
    if (!rccl_bind(c)) { delete c; return nullptr; }
    if (device >= 0) HIPCHK(hipSetDevice(device));
    HIPCHK(hipGetDevice(&c->device));
    c->n = n_ranks; c->rank = rank;
    ocp_qp_gpu_comm::uid id;
    memcpy(&id, id128, sizeof(id));
    const int r = c->init_rank(&c->comm, n_ranks, id, rank);
    if (r != 0)
    {
        fprintf(stderr, "acados_amd: ncclCommInitRank failed: %d (%s)\n", r, c->err ? c->err(r) : "?");
        delete c;
        return nullptr;
    }
    c->ops.ctx = c;
    c->ops.all_gather = rccl_op_all_gather;
    c->ops.send = rccl_op_send;
    c->ops.recv = rccl_op_recv;
    c->ops.group_start = rccl_op_group_start;
    c->ops.group_end = rccl_op_group_end;
    return c;
}
catch (const gqp_hip_failure &) { return nullptr; }
#else  /* host-simulation build of the CPU test tier: no RCCL; communicators come from ocp_qp_gpu_comm_create_from_ops */
int ocp_qp_gpu_comm_unique_id(void *) { return -1; }
ocp_qp_gpu_comm *ocp_qp_gpu_comm_create(const void *, int, int, int) { return nullptr; }
#endif

ocp_qp_gpu_comm *ocp_qp_gpu_comm_create_from_ops(const ocp_qp_gpu_comm_ops *ops, int n_ranks, int rank)
{
    if (!ops || !ops->all_gather || !ops->send || !ops->recv || !ops->group_start || !ops->group_end || n_ranks < 1 || rank < 0 || rank >= n_ranks)
    {
        fprintf(stderr, "acados_amd: ocp_qp_gpu_comm_create_from_ops: incomplete transport table or rank %d outside 0..%d\n", rank, n_ranks - 1);
        return nullptr;
    }
    ocp_qp_gpu_comm *c = new ocp_qp_gpu_comm();
    c->ops = *ops;
    c->n = n_ranks; c->rank = rank;
    return c;
}

void ocp_qp_gpu_comm_destroy(ocp_qp_gpu_comm *c)
{
    if (!c) return;
    if (c->comm && c->destroy) (void) c->destroy(c->comm);
    delete c;
}

/* The gather of the path.  root < 0: every rank receives everything; root >= 0: that rank only (the buffers of the other
 * ranks may be NULL).  counts: instances per rank (n_ranks entries) or NULL = b->B on every rank.  The receive buffers are
 * laid out in RANK ORDER, rank r's instances starting at instance offset sum(counts[0..r-1]):
 *     sol_all [sum counts][bulk_len(out)] doubles, info_all [sum counts][2] ints (status, iter), time_all [n_ranks] doubles.
 * Equal counts + all ranks receive: three ncclAllGather (ring over xGMI).  Otherwise -- gather to a root, or UNEVEN shards,
 * which ncclAllGather cannot express -- point-to-point sends / receives with their exact counts inside ONE group: to a
 * root every other rank moves 1x its payload instead of receiving n_ranks x (SURVEY 5 costs the 8-GPU C2 payload at 5.5 ms
 * this way against ~40 ms for the ring all-gather).  A transport error never leaves a group open: the group is always
 * ended before the error is returned (round-3 review). */
static int gather_impl(ocp_qp_gpu_batch *b, ocp_qp_gpu_comm *c, int root, const int *counts, double *sol_all, int *info_all, double *time_all)
{
    if (!c) return -1;
    HIPCHK(hipSetDevice(b->device));
    if (root >= c->n) { fprintf(stderr, "acados_amd: ocp_qp_gpu_batch_gather: root %d outside 0..%d\n", root, c->n - 1); return -1; }
    if (counts && counts[c->rank] != b->B)
    {
        fprintf(stderr, "acados_amd: ocp_qp_gpu_batch_gather: counts[%d] = %d but this rank's batch holds %d instances\n", c->rank, counts[c->rank], b->B);
        return -1;
    }
    const int len = gqp_bulk_len_impl(b, 1);
    auto &M = b->bulk_out;
    const size_t cnt = (size_t) b->B * len;
    /* send buffers: the solution blob of this rank (gather launch into the staging area), status / iter interleaved */
    const size_t need = cnt + (size_t) b->B + 8; /* doubles: blob + room for 2*B ints + the time */
    if (need > b->stage_cap) { b->stage_cap = need * 2; b->d_stage = dalloc<double>(b, b->stage_cap); }
    double *blob = b->d_stage;
    int *info = (int *) (b->d_stage + cnt);
    double *tm = b->d_stage + cnt + b->B + 1;
    const dim3 grid((b->B + 63) / 64, (len + 255) / 256), block(64);
    bulk_gather_launch(b, blob, len, M.d_arr, M.d_elem, M.T);
    hipLaunchKernelGGL(gqp::k_pack_info, dim3((b->B + 63) / 64), block, 0, b->stream, b->D, info);
    HIPCHK(hipMemcpyAsync(tm, &b->time_tot, sizeof(double), hipMemcpyHostToDevice, b->stream));
    bool even = true;
    if (counts) for (int r = 0; r < c->n; r++) even = even && counts[r] == b->B;
    const ocp_qp_gpu_comm_ops &T = c->ops;
    void *st = (void *) b->stream;
    int rc = 0;
    if (root < 0 && even)
    {
        if ((rc = T.all_gather(T.ctx, blob, sol_all, cnt, GQP_COMM_F64, st)) == 0
            && (rc = T.all_gather(T.ctx, info, info_all, 2 * (size_t) b->B, GQP_COMM_I32, st)) == 0)
            rc = T.all_gather(T.ctx, tm, time_all, 1, GQP_COMM_F64, st);
    }
    else
    {
        rc = T.group_start(T.ctx);
        if (rc != 0) return -1; /* no group was opened */
        for (int dst = 0; dst < c->n && rc == 0; dst++)
        {
            if (root >= 0 && dst != root) continue;
            if ((rc = T.send(T.ctx, blob, cnt, GQP_COMM_F64, dst, st)) == 0 && (rc = T.send(T.ctx, info, 2 * (size_t) b->B, GQP_COMM_I32, dst, st)) == 0)
                rc = T.send(T.ctx, tm, 1, GQP_COMM_F64, dst, st);
        }
        if (root < 0 || c->rank == root)
        {
            size_t off = 0; /* instances of the ranks before r */
            for (int r = 0; r < c->n && rc == 0; r++)
            {
                const size_t nr = counts ? (size_t) counts[r] : (size_t) b->B;
                if ((rc = T.recv(T.ctx, sol_all + off * len, nr * len, GQP_COMM_F64, r, st)) == 0
                    && (rc = T.recv(T.ctx, info_all + off * 2, nr * 2, GQP_COMM_I32, r, st)) == 0)
                    rc = T.recv(T.ctx, time_all + r, 1, GQP_COMM_F64, r, st);
                off += nr;
            }
        }
        const int re = T.group_end(T.ctx); /* always: an open group would swallow every later call of this process */
        if (rc == 0) rc = re;
    }
    HIPCHK(hipStreamSynchronize(b->stream));
    return rc == 0 ? 0 : -1;
}

int ocp_qp_gpu_batch_gather(ocp_qp_gpu_batch *b, ocp_qp_gpu_comm *c, double *sol_all, int *info_all, double *time_all)
try
{
    return gather_impl(b, c, -1, nullptr, sol_all, info_all, time_all);
}
catch (const gqp_hip_failure &) { return -1; }

int ocp_qp_gpu_batch_gather_root(ocp_qp_gpu_batch *b, ocp_qp_gpu_comm *c, int root, double *sol_all, int *info_all, double *time_all)
try
{
    return gather_impl(b, c, root < 0 ? 0 : root, nullptr, sol_all, info_all, time_all);
}
catch (const gqp_hip_failure &) { return -1; }

int ocp_qp_gpu_batch_gather_v(ocp_qp_gpu_batch *b, ocp_qp_gpu_comm *c, int root, const int *counts, double *sol_all, int *info_all, double *time_all)
try
{
    return gather_impl(b, c, root, counts, sol_all, info_all, time_all);
}
catch (const gqp_hip_failure &) { return -1; }

size_t ocp_qp_gpu_batch_bytes(const ocp_qp_gpu_batch *b) { return b->bytes; }
void *ocp_qp_gpu_batch_stream(ocp_qp_gpu_batch *b) { return (void *) b->stream; }
const char *ocp_qp_gpu_batch_kernel_name(const ocp_qp_gpu_batch *b) { return b->kname.c_str(); }

} /* extern "C" */

#if defined(GQP_WPI_TIMING)
/* development aid (make timing): per-phase cycle counters of the wave-per-instance factor kernel */
extern "C" void gqp_wpi_cycles_read(unsigned long long *out, int reset)
{
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(gqp::gqp_wpi_cycles), sizeof(unsigned long long) * 16));
    if (reset)
    {
        unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(gqp::gqp_wpi_cycles), z, sizeof(z)));
    }
}
#endif
