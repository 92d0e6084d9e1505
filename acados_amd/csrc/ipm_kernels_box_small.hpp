/*
 * ipm_kernels_box_small.hpp -- the one-instance-per-lane box kernels (ipm_kernels_box.hpp) for SMALL stage blocks
 * (nu + nx <= 6: the nx = 4 classes of C5), as a software pipeline over the stages.
 *
 * Same arrays, same arithmetic in the same order as kb_factor / kb_backrhs / kb_forward (bit-identical under the host
 * simulation; on the device hipcc contracts the multiply-adds of the two code shapes differently: outputs agree to
 * 1e-13, iteration counts exactly -- tests/test_gpu_parity.py::test_small_block_pipeline_gpu), different shape of the code.  The kb kernels are built around the register file of the C2
 * block (11 x 11): loads in ordered phases, W parked in LDS -- four to five dependent HBM round trips per stage.
 * That is the right trade at 65,536 instances, where HBM bandwidth is the bound.  A class of a few thousand small
 * instances is a different machine: 7,281 instances are 114 waves, one on every ninth SIMD, nothing is bandwidth- or
 * issue-bound and the launch time is (stages) x (dependent latency of a stage) -- 5.5 us per stage in kb_factor<4,1>,
 * 3.9 us in the sixteen-lanes kernel kx_factor<4,1> (which spends 650 instructions per stage and wave on 4 instances and
 * has two waves per SIMD to issue them for).  Here:
 *   - everything a stage reads from HBM is ONE record (struct of registers) whose addresses depend on nothing that
 *     is loaded: the stage structure comes through scalar loads, rows that do not exist are read through a clamped
 *     index and masked by the activity bits afterwards;
 *   - a ring of PD records is kept in flight: the record of stage s + PD is requested before stage s computes
 *     (kbs_pipeline), so a stage costs its arithmetic, not its round trips.  A wave of these kernels has the whole
 *     register file of its SIMD (512) to itself at every batch size -- blocks are single waves;
 *   - W = [B A]' Lx+ stays in registers (no LDS).
 * Ghost lanes, per-instance scalars, statuses: exactly as in ipm_kernels_box.hpp.
 */
#ifndef IPM_KERNELS_BOX_SMALL_HPP_
#define IPM_KERNELS_BOX_SMALL_HPP_

#include "ipm_kernels_box.hpp"

namespace gqp
{

typedef void (*kern_redo_fn)(GqpDev, GqpOpts, int);

/* which (NX, NU) the kernel tables serve with this file, and how many stages are requested ahead */
template <int NX, int NU>
struct KbSmall
{
#if defined(GQP_NO_KBS) /* development builds: the kb kernels for every shape */
    static constexpr bool value = false;
#else
    static constexpr bool value = NX + NU <= 6;
#endif
};
/* records of the ring: two where a record is ~60 doubles (box rows on the inputs only: a third spills the factor sweep), two with rows on every variable
 * (~85 doubles) */
#ifndef KBS_DEPTH
#define KBS_DEPTH 2
#endif
#ifndef KBS_DEPTH_XBOX
#define KBS_DEPTH_XBOX 2
#endif

/* Stages k0, k0 +- 1, ..., k1 with a ring of PD records: load(k, R) issues the loads of stage k into R, body(k, R)
 * computes and stores stage k from its record.  Stage s computes from R[s mod PD], and the record is requested again --
 * for stage s + PD -- as soon as stage s is done with it: PD - 1 stages of arithmetic lie between a request and its use.
 * The inner loop is unrolled over the ring, so every R[d] is a fixed set of registers and nothing is copied. */
template <int PD, class Rec, class LoadF, class BodyF>
__device__ static inline void kbs_pipeline(int k0, int k1, LoadF &&load, BodyF &&body)
{
    const int dir = k1 >= k0 ? 1 : -1, cnt = (k1 - k0) * dir + 1;
    Rec R[PD];
    UNROLL for (int d = 0; d < PD; d++)
        if (d < cnt) load(k0 + dir * d, R[d]);
    for (int s0 = 0; s0 < cnt; s0 += PD)
    {
        UNROLL for (int d = 0; d < PD; d++)
        {
            const int s = s0 + d;
            if (s >= cnt) break;
            body(k0 + dir * s, R[d]);
            if (s + PD < cnt) load(k0 + dir * (s + PD), R[d]);
        }
    }
}

/* the (lower, upper) values of one array for box row j: clamped to row 0 of the stage when variable j has no row */
struct KbsPair
{
    double l, u;
};
#define KBS_ROW_PAIR(dst, arr, j)                                                              \
    do {                                                                                       \
        GQP_ROW(j, has_, ib_);                                                                 \
        (void) has_;                                                                           \
        (dst).l = ACC(arr, 0).ld(S.o_ct + ib_);                                                \
        (dst).u = ACC(arr, 0).ld(S.o_ct + S.nb + ib_);                                         \
    } while (0)

/* --------------------------------------------------------------- factor */

template <int NX, int NU, bool XBOX>
__global__ void __launch_bounds__(64) kbs_factor(GqpDev D, GqpOpts O, int redo)
{
    constexpr int n = NX + NU, NP = n * (n + 1) / 2, NPX = NX * (NX + 1) / 2, NB = XBOX ? n : NU;
    const int Bp = D.Bp;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.B) return;
    const bool run = D.status[i] == GQP_RUNNING;
    if (!GQP_WAVE_ANY(run)) return;

    struct Rec
    {
        uint64_t am;
        double bv[NX], v[n], bat[n * NX], g[n], pik[NX], H[NP];
        KbsPair d[NB], lam[NB], t[NB];
    };
    auto load = [&](int k, Rec &F)
    {
        const StageU S = stage_u(D.st, k);
        const uint64_t imask = S.bmask & ~S.emask;
        F.am = GAT(D.amask, k);
        UNROLL for (int c = 0; c < NX; c++) F.bv[c] = ACC(D.bvec, 0).ldj(k * NX, c);
        UNROLL for (int j = 0; j < n; j++) F.v[j] = ACC(D.ux, 0).ldj(k * n, j);
        UNROLL for (int e = 0; e < n * NX; e++) F.bat[e] = ACC(D.BAt, 0).ldj(k * n * NX, e);
        UNROLL for (int j = 0; j < n; j++) F.g[j] = ACC(D.rq, 0).ldj(k * n, j);
        UNROLL for (int c = 0; c < NX; c++) F.pik[c] = ACC(D.pi, 0).ldj(k * NX, c);
        UNROLL for (int j = 0; j < NB; j++)
        {
            KBS_ROW_PAIR(F.d[j], D.dvec, j);
            KBS_ROW_PAIR(F.lam[j], D.lam, j);
            KBS_ROW_PAIR(F.t[j], D.t, j);
        }
        UNROLL for (int e = 0; e < NP; e++) F.H[e] = ACC(D.RSQ, 0).ldj(k * NP, e);
    };

    double Lx[NPX], lx[NX];
    UNROLL for (int e = 0; e < NPX; e++) Lx[e] = 0.0;
    UNROLL for (int c = 0; c < NX; c++) lx[c] = 0.0;
    double nrm_g = 0.0, nrm_b = 0.0, nrm_d = 0.0, nrm_m = 0.0, musum = 0.0, obj = 0.0;
    int nact = 0;
    /* x_{k+1} and pi_{k+1}: what stage k + 1 read as its own state and multiplier; slot N + 1 is zero by convention */
    double xn[NX], pin[NX];
    UNROLL for (int c = 0; c < NX; c++) { xn[c] = 0.0; pin[c] = 0.0; }

    auto body = [&](int k, const Rec &F)
    {
        const StageU S = stage_u(D.st, k);
        const uint64_t imask = S.bmask & ~S.emask;
        const uint64_t am = F.am;
        const int nbg = S.nb;
        double rb[NX], v[n], gt[n];
        UNROLL for (int c = 0; c < NX; c++) rb[c] = F.bv[c] - xn[c];
        UNROLL for (int j = 0; j < n; j++) v[j] = F.v[j];

        /* dynamics: rb += row * v_r, gt_r = row . pi+, W_r = row * Lx+ */
        double W[n * NX];
        UNROLL for (int r = 0; r < n; r++)
        {
            double row[NX];
            UNROLL for (int c = 0; c < NX; c++) row[c] = F.bat[r * NX + c];
            double a = 0.0;
            UNROLL for (int c = 0; c < NX; c++)
            {
                a += row[c] * pin[c];
                rb[c] += row[c] * v[r];
            }
            gt[r] = a;
            UNROLL for (int c = 0; c < NX; c++)
            {
                double w = 0.0;
                UNROLL for (int q = c; q < NX; q++) w += row[q] * Lx[PK(q, c)];
                W[r * NX + c] = w;
            }
        }
        double w0[NX];
        UNROLL for (int c = 0; c < NX; c++)
        {
            double a = lx[c];
            UNROLL for (int q = c; q < NX; q++) a += Lx[PK(q, c)] * rb[q];
            w0[c] = a;
        }

        /* box rows */
        double rdl[NB], rdu[NB], gadd[NB], gam[NB];
        UNROLL for (int j = 0; j < NB; j++)
        {
            GQP_ROW(j, has, ib);
            const bool al = has && ((am >> ib) & 1), au = has && ((am >> (nbg + ib)) & 1);
            const double ll = al ? F.lam[j].l : 0.0, lu = au ? F.lam[j].u : 0.0;
            const double ttl = al ? F.t[j].l : 1.0, ttu = au ? F.t[j].u : 1.0;
            rdl[j] = al ? v[j] - F.d[j].l - ttl : 0.0;
            rdu[j] = au ? F.d[j].u - v[j] - ttu : 0.0;
            const double rml = al ? ll * ttl - O.tau_min : 0.0, rmu = au ? lu * ttu - O.tau_min : 0.0;
            nacc(nrm_d, rdl[j]); nacc(nrm_d, rdu[j]); nacc(nrm_m, rml); nacc(nrm_m, rmu);
            musum += ll * ttl + lu * ttu;
            nact += (int) al + (int) au;
            gt[j] -= ll - lu;
            const double itl = frcp(ttl), itu = frcp(ttu);
            gam[j] = ll * itl + lu * itu;
            gadd[j] = (rml + ll * rdl[j]) * itl - (rmu + lu * rdu[j]) * itu;
        }

        /* Hessian rows: H v, M = H~ + W W', W w0 */
        double M[NP], hv[n], mm[n];
        UNROLL for (int e = 0; e < NP; e++) M[e] = F.H[e];
        UNROLL for (int r = 0; r < n; r++) hv[r] = 0.0;
        UNROLL for (int r = 0; r < n; r++)
        {
            UNROLL for (int c = 0; c < r; c++)
            {
                hv[r] += M[PK(r, c)] * v[c];
                hv[c] += M[PK(r, c)] * v[r];
            }
            hv[r] += M[PK(r, r)] * v[r];
            M[PK(r, r)] += O.reg_prim + (r < NB ? gam[r < NB ? r : 0] : 0.0);
            double a = 0.0;
            UNROLL for (int c = 0; c < NX; c++) a += W[r * NX + c] * w0[c];
            mm[r] = a + (r < NB ? gadd[r < NB ? r : 0] : 0.0);
            UNROLL for (int c = 0; c <= r; c++)
            {
                double sacc = 0.0;
                UNROLL for (int q = 0; q < NX; q++) sacc += W[r * NX + q] * W[c * NX + q];
                M[PK(r, c)] += sacc;
            }
        }
        UNROLL for (int r = 0; r < n; r++)
        {
            obj += (0.5 * hv[r] + F.g[r]) * v[r];
            gt[r] += hv[r] + F.g[r];
        }
        UNROLL for (int c = 0; c < NX; c++) gt[NU + c] -= F.pik[c];
        UNROLL for (int j = 0; j < n; j++)
        {
            if ((S.emask >> j) & 1) gt[j] = 0.0;
            nacc(nrm_g, gt[j]);
        }
        UNROLL for (int c = 0; c < NX; c++) nacc(nrm_b, rb[c]);
        UNROLL for (int j = 0; j < n; j++) ACC(D.rg, 0).stj(k * n, j, gt[j]);
        UNROLL for (int c = 0; c < NX; c++) ACC(D.rb, 0).stj(k * NX, c, rb[c]);
        UNROLL for (int j = 0; j < n; j++) gt[j] += mm[j];
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
        /* Cholesky (inverse diagonal kept in registers) */
        double invd[n];
        UNROLL for (int jc = 0; jc < n; jc++)
        {
            const double d = M[PK(jc, jc)];
            const bool pos = d > 0.0;
            const double r0 = frsqrt(pos ? d : 1.0);
            const double inv = pos ? r0 : 0.0;
            invd[jc] = inv;
            M[PK(jc, jc)] = pos ? d * inv : 0.0;
            UNROLL for (int r = jc + 1; r < n; r++) M[PK(r, jc)] *= inv;
            UNROLL for (int c = jc + 1; c < n; c++)
                UNROLL for (int r = c; r < n; r++) M[PK(r, c)] -= M[PK(r, jc)] * M[PK(c, jc)];
        }
        UNROLL for (int e = 0; e < NP; e++) ACC(D.Lf, 0).stj(k * NP, e, M[e]);
        UNROLL for (int r = 0; r < n; r++)
        {
            double a = gt[r];
            UNROLL for (int c = 0; c < r; c++) a -= M[PK(r, c)] * gt[c];
            gt[r] = a * invd[r];
            ACC(D.lf, 0).stj(k * n, r, gt[r]);
        }
        UNROLL for (int r = 0; r < NX; r++)
        {
            lx[r] = gt[NU + r];
            UNROLL for (int c = 0; c <= r; c++) Lx[PK(r, c)] = M[PK(NU + r, NU + c)];
        }
        UNROLL for (int c = 0; c < NX; c++) { xn[c] = v[NU + c]; pin[c] = F.pik[c]; }
        UNROLL for (int j = 0; j < NB; j++)
        {
            GQP_ROW(j, has, ib);
            if (has)
            {
                ACC(D.rd, 0).st(S.o_ct + ib, rdl[j]);
                ACC(D.rd, 0).st(S.o_ct + nbg + ib, rdu[j]);
            }
        }
    };
    kbs_pipeline<(XBOX ? KBS_DEPTH_XBOX : KBS_DEPTH), Rec>(D.N, 0, load, body);

    if (!run) return;
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
__global__ void __launch_bounds__(64) kbs_backrhs(GqpDev D, GqpOpts O, int redo)
{
    constexpr int n = NX + NU, NP = n * (n + 1) / 2, NPX = NX * (NX + 1) / 2, NB = XBOX ? n : NU;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.B) return;
    const bool run = D.status[i] == GQP_RUNNING;
    if (redo ? !(run && D.alpha[i] < 0.0) : !GQP_WAVE_ANY(run)) return;
    const double smu = D.smu[i];
    const double pscale = redo ? 0.0 : 1.0; /* redo = centering only: drop dlam_aff*dt_aff */

    struct Rec
    {
        uint64_t am;
        double rb[NX], rg[n], bat[n * NX], L[NP];
        KbsPair lam[NB], t[NB], rd[NB], pc[NB];
    };
    auto load = [&](int k, Rec &F)
    {
        const StageU S = stage_u(D.st, k);
        const uint64_t imask = S.bmask & ~S.emask;
        F.am = GAT(D.amask, k);
        UNROLL for (int c = 0; c < NX; c++) F.rb[c] = ACC(D.rb, 0).ldj(k * NX, c);
        UNROLL for (int j = 0; j < n; j++) F.rg[j] = ACC(D.rg, 0).ldj(k * n, j);
        UNROLL for (int j = 0; j < NB; j++)
        {
            KBS_ROW_PAIR(F.lam[j], D.lam, j);
            KBS_ROW_PAIR(F.t[j], D.t, j);
            KBS_ROW_PAIR(F.rd[j], D.rd, j);
            KBS_ROW_PAIR(F.pc[j], D.pcorr, j);
        }
        UNROLL for (int e = 0; e < n * NX; e++) F.bat[e] = ACC(D.BAt, 0).ldj(k * n * NX, e);
        UNROLL for (int e = 0; e < NP; e++) F.L[e] = ACC(D.Lf, 0).ldj(k * NP, e);
    };

    double Lx[NPX], lx[NX];
    UNROLL for (int e = 0; e < NPX; e++) Lx[e] = 0.0;
    UNROLL for (int c = 0; c < NX; c++) lx[c] = 0.0;

    auto body = [&](int k, const Rec &F)
    {
        const StageU S = stage_u(D.st, k);
        const uint64_t imask = S.bmask & ~S.emask;
        const uint64_t am = F.am;
        const int nbg = S.nb;
        double rb[NX], gt[n];
        UNROLL for (int c = 0; c < NX; c++) rb[c] = F.rb[c];
        UNROLL for (int j = 0; j < n; j++) gt[j] = F.rg[j];
        UNROLL for (int j = 0; j < NB; j++)
        {
            GQP_ROW(j, has, ib);
            const bool al = has && ((am >> ib) & 1), au = has && ((am >> (nbg + ib)) & 1);
            const double ll = al ? F.lam[j].l : 0.0, lu = au ? F.lam[j].u : 0.0;
            const double ttl = al ? F.t[j].l : 1.0, ttu = au ? F.t[j].u : 1.0;
            const double rml = al ? ll * ttl - O.tau_min + pscale * F.pc[j].l - smu : 0.0;
            const double rmu = au ? lu * ttu - O.tau_min + pscale * F.pc[j].u - smu : 0.0;
            const double dl = al ? F.rd[j].l : 0.0, du = au ? F.rd[j].u : 0.0;
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
        UNROLL for (int r = 0; r < n; r++)
        {
            double a = 0.0;
            UNROLL for (int c = 0; c < NX; c++) a += F.bat[r * NX + c] * y[c];
            gt[r] += a;
        }
        UNROLL for (int r = 0; r < n; r++) if ((S.emask >> r) & 1) gt[r] = 0.0;
        UNROLL for (int r = 0; r < n; r++)
        {
            double a = gt[r];
            UNROLL for (int c = 0; c < r; c++) a -= F.L[PK(r, c)] * gt[c];
            const double d = F.L[PK(r, r)];
            gt[r] = d != 0.0 ? a * frcp(d) : 0.0;
            ACC(D.lf, 0).stj(k * n, r, gt[r]);
        }
        UNROLL for (int r = 0; r < NX; r++)
        {
            lx[r] = gt[NU + r];
            UNROLL for (int c = 0; c <= r; c++) Lx[PK(r, c)] = F.L[PK(NU + r, NU + c)];
        }
    };
    kbs_pipeline<(XBOX ? KBS_DEPTH_XBOX : KBS_DEPTH), Rec>(D.N, 0, load, body);
}

/* ----------------------------------------------------------------- forward */

template <int NX, int NU, bool XBOX, bool CORR>
__global__ void __launch_bounds__(64) kbs_forward(GqpDev D, GqpOpts O, int redo)
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

    struct Rec
    {
        uint64_t am;
        double L[NP], l[n], bat[n * NX], rbn[NX];
        KbsPair lam[NB], t[NB], rd[NB], pc[NB];
    };
    auto load = [&](int k, Rec &F)
    {
        const StageU S = stage_u(D.st, k);
        const uint64_t imask = S.bmask & ~S.emask;
        F.am = GAT(D.amask, k);
        UNROLL for (int e = 0; e < NP; e++) F.L[e] = ACC(D.Lf, 0).ldj(k * NP, e);
        UNROLL for (int j = 0; j < n; j++) F.l[j] = ACC(D.lf, 0).ldj(k * n, j);
        UNROLL for (int e = 0; e < n * NX; e++) F.bat[e] = ACC(D.BAt, 0).ldj(k * n * NX, e);
        UNROLL for (int c = 0; c < NX; c++) F.rbn[c] = ACC(D.rb, 0).ldj(k * NX, c);
        UNROLL for (int j = 0; j < NB; j++)
        {
            KBS_ROW_PAIR(F.lam[j], D.lam, j);
            KBS_ROW_PAIR(F.t[j], D.t, j);
            KBS_ROW_PAIR(F.rd[j], D.rd, j);
            if (CORR) KBS_ROW_PAIR(F.pc[j], D.pcorr, j);
            else { F.pc[j].l = 0.0; F.pc[j].u = 0.0; }
        }
    };
    auto body = [&](int k, const Rec &F)
    {
        const StageU S = stage_u(D.st, k);
        const uint64_t imask = S.bmask & ~S.emask;
        const uint64_t am = F.am;
        const int nbg = S.nb;
        double invd[n];
        UNROLL for (int r = 0; r < n; r++)
        {
            const double d = F.L[PK(r, r)];
            invd[r] = d != 0.0 ? frcp(d) : 0.0;
        }
        double dv[n];
        if (CORR)
        {
            double w0[NX];
            UNROLL for (int c = 0; c < NX; c++)
            {
                double a = F.l[NU + c];
                UNROLL for (int q = c; q < NX; q++) a += F.L[PK(NU + q, NU + c)] * dx[q];
                w0[c] = a;
            }
            UNROLL for (int r = 0; r < NX; r++)
            {
                double a = 0.0;
                UNROLL for (int c = 0; c <= r; c++) a += F.L[PK(NU + r, NU + c)] * w0[c];
                if (k > 0) ACC(D.dpi, 0).stj(k * NX, r, a);
            }
        }
        const bool first = k == 0;
        UNROLL for (int r = n - 1; r >= NU; r--)
        {
            double a = -F.l[r];
            UNROLL for (int p = r + 1; p < n; p++) a -= F.L[PK(p, r)] * dv[p];
            dv[r] = first ? a * invd[r] : dx[r - NU];
        }
        UNROLL for (int r = NU - 1; r >= 0; r--)
        {
            double a = -F.l[r];
            UNROLL for (int p = r + 1; p < n; p++) a -= F.L[PK(p, r)] * dv[p];
            dv[r] = a * invd[r];
        }
        if (CORR) { UNROLL for (int j = 0; j < n; j++) ACC(D.dux, 0).stj(k * n, j, dv[j]); }
        UNROLL for (int c = 0; c < NX; c++) dx[c] = F.rbn[c];
        UNROLL for (int r = 0; r < n; r++)
            UNROLL for (int c = 0; c < NX; c++) dx[c] += F.bat[r * NX + c] * dv[r];

        double o_dll[NB], o_dlu[NB], o_dtl[NB], o_dtu[NB];
        UNROLL for (int j = 0; j < NB; j++)
        {
            GQP_ROW(j, has, ib);
            const bool al = has && ((am >> ib) & 1), au = has && ((am >> (nbg + ib)) & 1);
            const double ll = al ? F.lam[j].l : 0.0, lu = au ? F.lam[j].u : 0.0;
            const double ttl = al ? F.t[j].l : 1.0, ttu = au ? F.t[j].u : 1.0;
            const double rml = al ? ll * ttl - O.tau_min + pscale * F.pc[j].l - smu : 0.0;
            const double rmu = au ? lu * ttu - O.tau_min + pscale * F.pc[j].u - smu : 0.0;
            const double dtl = al ? dv[j] + F.rd[j].l : 0.0, dtu = au ? -dv[j] + F.rd[j].u : 0.0;
            const double dll = al ? -(rml + ll * dtl) * frcp(ttl) : 0.0;
            const double dlu = au ? -(rmu + lu * dtu) * frcp(ttu) : 0.0;
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
    };
    kbs_pipeline<(XBOX ? KBS_DEPTH_XBOX : KBS_DEPTH), Rec>(0, D.N, load, body);

    const int it = D.iter[i];
    double *st = (i < D.stat_inst && it + 1 < D.stat_rows) ? D.stat + (size_t) (it + 1) * GQP_STAT_COLS * D.stat_inst + i : nullptr;
    if (!CORR)
    {
        if (!run) return;
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
        D.alpha[i] = -alpha_aff; /* flag for the redo pair */
        return;
    }
    const double a = !run ? 0.0 : D.mu[i] > 0.0 ? gqp_step_scale(alpha) : 1.0;
    /* update pass: the iterate moves by a * direction; the same pipeline (a stage reads and writes its own slots only) */
    struct URec
    {
        double vx[n], vdx[n], vp[NX], vdp[NX];
        KbsPair vl[NB], vdl[NB], vt[NB], vdt[NB];
        uint64_t am;
    };
    auto uload = [&](int k, URec &F)
    {
        const StageU S = stage_u(D.st, k);
        const uint64_t imask = S.bmask & ~S.emask;
        F.am = GAT(D.amask, k);
        UNROLL for (int j = 0; j < n; j++) { F.vx[j] = ACC(D.ux, 0).ldj(k * n, j); F.vdx[j] = ACC(D.dux, 0).ldj(k * n, j); }
        UNROLL for (int c = 0; c < NX; c++) { F.vp[c] = ACC(D.pi, 0).ldj((k + 1) * NX, c); F.vdp[c] = ACC(D.dpi, 0).ldj((k + 1) * NX, c); }
        UNROLL for (int j = 0; j < NB; j++)
        {
            KBS_ROW_PAIR(F.vl[j], D.lam, j);
            KBS_ROW_PAIR(F.vdl[j], D.dlam, j);
            KBS_ROW_PAIR(F.vt[j], D.t, j);
            KBS_ROW_PAIR(F.vdt[j], D.dt, j);
        }
    };
    auto ubody = [&](int k, const URec &F)
    {
        const StageU S = stage_u(D.st, k);
        const uint64_t imask = S.bmask & ~S.emask;
        const uint64_t am = F.am;
        const int nbg = S.nb;
        UNROLL for (int j = 0; j < n; j++) ACC(D.ux, 0).stj(k * n, j, run ? F.vx[j] + a * F.vdx[j] : F.vx[j]);
        UNROLL for (int c = 0; c < NX; c++) ACC(D.pi, 0).stj((k + 1) * NX, c, run ? F.vp[c] + a * F.vdp[c] : F.vp[c]);
        UNROLL for (int j = 0; j < NB; j++)
        {
            GQP_ROW(j, has, ib);
            if (!has) continue;
            UNROLL for (int side = 0; side < 2; side++)
            {
                const int e = S.o_ct + side * nbg + ib;
                const bool act = run && ((am >> (side * nbg + ib)) & 1);
                const double l0 = side ? F.vl[j].u : F.vl[j].l, dl0 = side ? F.vdl[j].u : F.vdl[j].l;
                const double t0 = side ? F.vt[j].u : F.vt[j].l, dt0 = side ? F.vdt[j].u : F.vdt[j].l;
                const double lam = l0 + a * dl0;
                const double t = t0 + a * dt0;
                ACC(D.lam, 0).st(e, act ? (lam < O.lam_min ? O.lam_min : lam) : l0);
                ACC(D.t, 0).st(e, act ? (t < O.t_min ? O.t_min : t) : t0);
            }
        }
    };
    kbs_pipeline<(XBOX ? KBS_DEPTH_XBOX : KBS_DEPTH), URec>(0, D.N, uload, ubody);
    if (!run) return;
    D.alpha[i] = alpha;
    D.iter[i] = it + 1;
    if (st) { st[4 * D.stat_inst] = alpha; st[5 * D.stat_inst] = alpha; }
}

/* the kernel the tables hold for (NX, NU): this file's for the small blocks, ipm_kernels_box.hpp's otherwise */
template <int NX, int NU, bool XBOX>
constexpr kern_redo_fn kb_factor_for()
{
    if constexpr (KbSmall<NX, NU>::value) return kbs_factor<NX, NU, XBOX>;
    else return kb_factor<NX, NU, XBOX>;
}
template <int NX, int NU, bool XBOX>
constexpr kern_redo_fn kb_backrhs_for()
{
    if constexpr (KbSmall<NX, NU>::value) return kbs_backrhs<NX, NU, XBOX>;
    else return kb_backrhs<NX, NU, XBOX>;
}
template <int NX, int NU, bool XBOX, bool CORR>
constexpr kern_redo_fn kb_forward_for()
{
    if constexpr (KbSmall<NX, NU>::value) return kbs_forward<NX, NU, XBOX, CORR>;
    else return kb_forward<NX, NU, XBOX, CORR>;
}

} // namespace gqp

#endif
