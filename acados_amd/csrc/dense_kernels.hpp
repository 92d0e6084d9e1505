/*
 * dense_kernels.hpp -- FULL CONDENSING of any size (the reference's module is acados/ocp_qp/ocp_qp_full_condensing.c:468-556,
 * which hands the dense QP to a dense_qp solver -- both HPIPM-backed and absent from the reference tree).
 *
 * Correctness first, no fast-path ambition (the stage-wise Riccati IPM is 2.4 ... 45 x faster on every shape measured,
 * profiles/r03_full_condensing.txt): the condensed problem of C2 has 158 columns and 300 inequality sides, past what one condensed
 * STAGE of the stage-wise families may carry (64 variables, 128 sides), so this path has its own kernel: ONE WORKGROUP PER INSTANCE,
 * every matrix in HBM (an instance-major workspace block per workgroup), the whole IPM loop inside one launch.
 *
 * What is condensed: all states but x0.  With v = [u_0 .. u_N (padded stage inputs); x_0] the stage variables are w_k = [u_k; x_k],
 * x_k = G_k v + c_k (G_{k+1} = [B_k A_k] [E_k; G_k], c_{k+1} = A_k c_k + b_k), i.e. w_k = T_k v + [0; c_k].  The IPM is the SAME Mehrotra
 * predictor-corrector as the stage-wise kernels and the oracle (initialisation, sigma = (mu_aff / mu)^3, ratio test, step-to-boundary
 * scaling, conditional corrector, floors, exit test on the four residual norms); the only thing that differs is the elimination
 * order of the Newton system: the inequality rows are folded into the stage Hessians exactly as the Riccati sweeps do
 * (H~_k = H_k + J_k' diag(Gamma) J_k), then
 *       M = sum_k T_k' H~_k T_k   (dense, nv x nv),    M dv = - sum_k T_k' g~_k,    dw_k = T_k dv
 * by a dense Cholesky factorisation instead of the backward / forward recursion.  The dynamics hold exactly at every iterate (the
 * oracle's iterates are dynamics-infeasible until its first full step): the two paths follow different iterates to the same solution.
 * Multipliers of the dynamics come from the adjoint recursion at the end (pi_k = (H w + g - J' nu)_x,k + A_k' pi_{k+1}), with which the
 * stationarity residual of the ORIGINAL QP equals the reduced gradient the loop judged.
 *
 * Supported: hard box rows and hard general rows with one- or two-sided masks, fixed variables (idxe) anywhere among the inputs and
 * among the states of stage 0.  Not supported (status ACADOS_QP_FAILURE-class 4, loud on the host): slacks, fixed states behind stage 0.
 */
#ifndef DENSE_KERNELS_HPP_
#define DENSE_KERNELS_HPP_

#include "ipm_kernels.hpp"

namespace gqp
{

#define GQP_KD_THREADS 256

struct KdRow /* one inequality ROW of the original QP (both sides), host-built, shared by the batch */
{
    int k;      /* stage */
    int var;    /* padded variable of a box row, -1: general row */
    int g;      /* row of DCt (o_g + g) of a general row */
    int lo, up; /* element of the lower / upper side in lam / t / dvec / rd ... */
    int blo, bup; /* activity bits of the two sides inside the stage's mask word(s) */
    int fixed;  /* equality-flagged: the variable is fixed at dvec[lo], the row takes no part */
};

struct KdDims
{
    int K, n, nv, nvu, R;          /* stages N + 1, NU + NX, dense variables, of them stage inputs, inequality rows */
    size_t oG, oc, oH, og, orw, oM, orhs, odv, ov, ow, odw, oP, W; /* workspace offsets (doubles) and size per instance */
};

__device__ static inline double kd_blk_max(double v, double *sh)
{
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int s = GQP_KD_THREADS / 2; s > 0; s >>= 1)
    {
        if (t < s) { const double a = sh[t], b = sh[t + s]; sh[t] = (b > a || b != b) ? b : a; }
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}
__device__ static inline double kd_blk_min(double v, double *sh)
{
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int s = GQP_KD_THREADS / 2; s > 0; s >>= 1)
    {
        if (t < s) { const double a = sh[t], b = sh[t + s]; sh[t] = b < a ? b : a; }
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}
__device__ static inline double kd_blk_sum(double v, double *sh)
{
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int s = GQP_KD_THREADS / 2; s > 0; s >>= 1)
    {
        if (t < s) sh[t] += sh[t + s];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

/* value of inequality row r at the stage vector w (n entries of stage rows[r].k) */
#define KD_ROWVAL(r_, wk_) ((r_).var >= 0 ? (wk_)[(r_).var] : kd_genval(D, i, (r_).g, (wk_), n))
__device__ static inline double kd_genval(const GqpDev &D, int i, int g, const double *wk, int n)
{
    double a = 0.0;
    for (int j = 0; j < n; j++) a += GATL(D.DCt, g * n + j) * wk[j];
    return a;
}

/*
 * the whole solve of instance `first + blockIdx.x`.  Parent arrays are read / written through the layout-agnostic accessor.
 * Work arrays of the parent that are reused per row: rd, rm (D.rm), dlam, dt, pcorr (corrected complementarity rhs).
 */
static __global__ void __launch_bounds__(GQP_KD_THREADS) kd_solve(GqpDev D, GqpOpts O, KdDims S, const KdRow *rows, double *wsbase, int first, int unsupported)
{
    const int i = first + (int) blockIdx.x;
    if (i >= D.B) return;
    const int tid = threadIdx.x, NT = GQP_KD_THREADS;
    const int K = S.K, NX = D.NX, NU = D.NU, n = S.n, nv = S.nv, nvu = S.nvu, R = S.R;
    double *ws = wsbase + (size_t) blockIdx.x * S.W;
    double *G = ws + S.oG, *cv = ws + S.oc, *Hs = ws + S.oH, *gs = ws + S.og, *rw = ws + S.orw, *M = ws + S.oM, *rhs = ws + S.orhs;
    double *dv = ws + S.odv, *v = ws + S.ov, *w = ws + S.ow, *dw = ws + S.odw, *P = ws + S.oP;
    __shared__ double sh[GQP_KD_THREADS];
    __shared__ int s_fixed_cnt;
    if (unsupported)
    {
        if (tid == 0) { D.status[i] = 4; D.iter[i] = 0; }
        return;
    }

    /* ---- state maps: x_k = G_k v + c_k ---- */
    for (int e = tid; e < NX * nv; e += NT) { const int r = e / nv, a = e % nv; G[e] = a == nvu + r ? 1.0 : 0.0; }
    for (int e = tid; e < NX; e += NT) cv[e] = 0.0;
    __syncthreads();
    for (int k = 0; k + 1 < K; k++)
    {
        const double *Gk = G + (size_t) k * NX * nv;
        double *Gn = G + (size_t) (k + 1) * NX * nv;
        for (int e = tid; e < NX * nv; e += NT)
        {
            const int c = e / nv, a = e % nv;
            double s = 0.0;
            for (int r = 0; r < NX; r++) s += GATL(D.BAt, (k * n + NU + r) * NX + c) * Gk[(size_t) r * nv + a];
            if (a >= k * NU && a < (k + 1) * NU) s += GATL(D.BAt, (k * n + (a - k * NU)) * NX + c);
            Gn[e] = s;
        }
        for (int c = tid; c < NX; c += NT)
        {
            double s = GATL(D.bvec, k * NX + c);
            for (int r = 0; r < NX; r++) s += GATL(D.BAt, (k * n + NU + r) * NX + c) * cv[k * NX + r];
            cv[(k + 1) * NX + c] = s;
        }
        __syncthreads();
    }

    /* ---- fixed variables (idxe) as flags on the dense variables: fixed[a] in dv[] scratch is not kept; recomputed from the rows ---- */
    /* cold start (oracle init_var): v = 0, fixed variables at their value, hard box rows on inputs / x0 moved inside their bounds */
    for (int a = tid; a < nv; a += NT) { v[a] = 0.0; dv[a] = 0.0; }
    __syncthreads();
    const double thr0 = 1e-1;
    const bool heur = O.t0_init != 0 && O.t0_init != 1;
    for (int r = tid; r < R; r += NT)
    {
        const KdRow q = rows[r];
        if (q.var < 0) continue;
        const int a = q.var < NU ? q.k * NU + q.var : (q.k == 0 ? nvu + q.var - NU : -1);
        if (a < 0) continue;
        const double lo = GATL(D.dvec, q.lo), up = GATL(D.dvec, q.up);
        if (q.fixed) { v[a] = lo; continue; }
        if (!heur) continue;
        const uint64_t am0 = GATL(D.amask, q.k * D.AW + (q.blo >> 6)), am1 = GATL(D.amask, q.k * D.AW + (q.bup >> 6));
        const bool al = (am0 >> (q.blo & 63)) & 1, au = (am1 >> (q.bup & 63)) & 1;
        double x = 0.0;
        const double tl = x - lo, tu = up - x;
        if (al && au)
        {
            if (tl < thr0) x = tu < thr0 ? 0.5 * (lo + up) : lo + thr0;
            else if (tu < thr0) x = up - thr0;
        }
        else if (al) { if (tl < thr0) x = lo + thr0; }
        else if (au) { if (tu < thr0) x = up - thr0; }
        v[a] = x;
    }
    __syncthreads();

    int it = 0, status = GQP_RUNNING;
    double alpha = 1.0, mu = 0.0, nrm_g = 0.0, nrm_d = 0.0, nrm_m = 0.0;
    const double t_c = O.t0_init == 0 ? sqrt(O.mu0) : 1.0, l_c = O.t0_init == 0 ? sqrt(O.mu0) : O.mu0;
    int nact = 0;
    for (int pass = 0;; pass++)
    {
        /* ---- stage vectors from v: w_k = [u_k; G_k v + c_k] ---- */
        for (int e = tid; e < K * n; e += NT)
        {
            const int k = e / n, j = e % n;
            double s;
            if (j < NU) s = v[k * NU + j];
            else
            {
                const double *Gr = G + ((size_t) k * NX + (j - NU)) * nv;
                s = cv[k * NX + j - NU];
                for (int a = 0; a < nv; a++) s += Gr[a] * v[a];
            }
            w[e] = s;
        }
        __syncthreads();
        if (pass == 0)
        {
            /* t from the constraint residuals clipped at 0.1, lam = mu0 / t (t0_init 2); the two constant schemes otherwise */
            int cnt = 0;
            for (int r = tid; r < R; r += NT)
            {
                const KdRow q = rows[r];
                const double c = KD_ROWVAL(q, w + q.k * n);
                const uint64_t am0 = GATL(D.amask, q.k * D.AW + (q.blo >> 6)), am1 = GATL(D.amask, q.k * D.AW + (q.bup >> 6));
                const bool al = !q.fixed && ((am0 >> (q.blo & 63)) & 1), au = !q.fixed && ((am1 >> (q.bup & 63)) & 1);
                double tl = c - GATL(D.dvec, q.lo), tu = GATL(D.dvec, q.up) - c;
                if (tl < thr0) tl = thr0;
                if (tu < thr0) tu = thr0;
                GATL(D.t, q.lo) = al ? (heur ? tl : t_c) : 0.0; GATL(D.lam, q.lo) = al ? (heur ? O.mu0 / tl : l_c) : 0.0;
                GATL(D.t, q.up) = au ? (heur ? tu : t_c) : 0.0; GATL(D.lam, q.up) = au ? (heur ? O.mu0 / tu : l_c) : 0.0;
                cnt += (al ? 1 : 0) + (au ? 1 : 0);
            }
            nact = (int) (kd_blk_sum((double) cnt, sh) + 0.5);
        }

        /* ---- residuals at the iterate: rw_k = H_k w_k + g_k - J_k'(lam_l - lam_u)  (no pi: it drops out of the reduced gradient) ---- */
        for (int e = tid; e < K * n; e += NT)
        {
            const int k = e / n, r = e % n;
            const double *wk = w + k * n;
            double s = GATL(D.rq, k * n + r);
            for (int c = 0; c < n; c++) s += GATL(D.RSQ, k * (n * (n + 1) / 2) + (r >= c ? PK(r, c) : PK(c, r))) * wk[c];
            rw[e] = s;
        }
        __syncthreads();
        double md = 0.0, mm = 0.0, msum = 0.0;
        for (int r = tid; r < R; r += NT)   /* rows of one stage may touch the same entries of rw: serialised per row by atomics below */
        {
            const KdRow q = rows[r];
            const double c = KD_ROWVAL(q, w + q.k * n);
            const uint64_t am0 = GATL(D.amask, q.k * D.AW + (q.blo >> 6)), am1 = GATL(D.amask, q.k * D.AW + (q.bup >> 6));
            const bool al = !q.fixed && ((am0 >> (q.blo & 63)) & 1), au = !q.fixed && ((am1 >> (q.bup & 63)) & 1);
            const double ll = al ? GATL(D.lam, q.lo) : 0.0, lu = au ? GATL(D.lam, q.up) : 0.0;
            const double tl = al ? GATL(D.t, q.lo) : 1.0, tu = au ? GATL(D.t, q.up) : 1.0;
            const double rdl = al ? c - GATL(D.dvec, q.lo) - tl : 0.0, rdu = au ? GATL(D.dvec, q.up) - c - tu : 0.0;
            const double rml = al ? ll * tl - O.tau_min : 0.0, rmu = au ? lu * tu - O.tau_min : 0.0;
            GATL(D.rd, q.lo) = rdl; GATL(D.rd, q.up) = rdu;
            GATL(D.rm, q.lo) = rml; GATL(D.rm, q.up) = rmu;
            msum += ll * tl * (al ? 1.0 : 0.0) + lu * tu * (au ? 1.0 : 0.0);
            const double a1 = dabs(rdl), a2 = dabs(rdu), a3 = dabs(rml), a4 = dabs(rmu);
            md = (a1 > md || a1 != a1) ? a1 : md; md = (a2 > md || a2 != a2) ? a2 : md;
            mm = (a3 > mm || a3 != a3) ? a3 : mm; mm = (a4 > mm || a4 != a4) ? a4 : mm;
            const double nu_ = ll - lu;
            if (nu_ != 0.0)
            {
                if (q.var >= 0) atomicAdd(&rw[q.k * n + q.var], -nu_);
                else for (int j = 0; j < n; j++) atomicAdd(&rw[q.k * n + j], -nu_ * GATL(D.DCt, q.g * n + j));
            }
        }
        nrm_d = kd_blk_max(md, sh);
        nrm_m = kd_blk_max(mm, sh);
        mu = kd_blk_sum(msum, sh);
        mu = nact > 0 ? mu / nact : 0.0;
        /* reduced gradient rhs[a] = sum_k T_k' rw_k; fixed variables carry none */
        for (int a = tid; a < nv; a += NT)
        {
            double s = 0.0;
            if (a < nvu) s = rw[(a / NU) * n + a % NU];
            for (int k = 0; k < K; k++)
            {
                if (!(a < k * NU || a >= nvu)) continue;
                const double *Gk = G + (size_t) k * NX * nv;
                for (int r = 0; r < NX; r++) s += Gk[(size_t) r * nv + a] * rw[k * n + NU + r];
            }
            rhs[a] = s;
        }
        __syncthreads();
        /* fixed flags of the dense variables in dv[] (0 / 1) -- rebuilt every pass, cheap */
        for (int a = tid; a < nv; a += NT) dv[a] = 0.0;
        __syncthreads();
        for (int r = tid; r < R; r += NT)
        {
            const KdRow q = rows[r];
            if (!q.fixed || q.var < 0) continue;
            const int a = q.var < NU ? q.k * NU + q.var : (q.k == 0 ? nvu + q.var - NU : -1);
            if (a >= 0) dv[a] = 1.0;
        }
        __syncthreads();
        double mg = 0.0;
        for (int a = tid; a < nv; a += NT) { const double x = dv[a] != 0.0 ? 0.0 : dabs(rhs[a]); mg = (x > mg || x != x) ? x : mg; }
        nrm_g = kd_blk_max(mg, sh);

        /* ---- exit test (same order as the stage-wise factor sweeps) ---- */
        const bool bad = nrm_g != nrm_g || nrm_d != nrm_d || nrm_m != nrm_m || mu != mu;
        if (bad) status = 1;
        else if (nrm_g <= O.tol_stat && nrm_d <= O.tol_ineq && nrm_m <= O.tol_comp) status = 0;
        else if (it >= O.iter_max) status = 2;
        else if (dabs(alpha) <= O.alpha_min) status = 3;
        if (status != GQP_RUNNING) break;

        /* ---- Newton matrix: H~_k = H_k + reg + J' diag(Gamma_l + Gamma_u) J ;  M = sum_k T_k' H~_k T_k ---- */
        for (int e = tid; e < K * n * n; e += NT)
        {
            const int k = e / (n * n), r = (e / n) % n, c = e % n;
            Hs[e] = GATL(D.RSQ, k * (n * (n + 1) / 2) + (r >= c ? PK(r, c) : PK(c, r))) + (r == c ? O.reg_prim : 0.0);
        }
        __syncthreads();
        for (int r = tid; r < R; r += NT)
        {
            const KdRow q = rows[r];
            const uint64_t am0 = GATL(D.amask, q.k * D.AW + (q.blo >> 6)), am1 = GATL(D.amask, q.k * D.AW + (q.bup >> 6));
            const bool al = !q.fixed && ((am0 >> (q.blo & 63)) & 1), au = !q.fixed && ((am1 >> (q.bup & 63)) & 1);
            const double gm = (al ? GATL(D.lam, q.lo) / GATL(D.t, q.lo) : 0.0) + (au ? GATL(D.lam, q.up) / GATL(D.t, q.up) : 0.0);
            if (gm == 0.0) continue;
            double *Hk = Hs + (size_t) q.k * n * n;
            if (q.var >= 0) atomicAdd(&Hk[q.var * n + q.var], gm);
            else
                for (int a = 0; a < n; a++)
                {
                    const double da = GATL(D.DCt, q.g * n + a);
                    if (da == 0.0) continue;
                    for (int c = 0; c < n; c++) atomicAdd(&Hk[a * n + c], gm * da * GATL(D.DCt, q.g * n + c));
                }
        }
        for (int e = tid; e < nv * nv; e += NT) M[e] = 0.0;
        __syncthreads();
        for (int k = 0; k < K; k++)
        {
            const double *Hk = Hs + (size_t) k * n * n;
            const double *Gk = G + (size_t) k * NX * nv;
            /* P = Hxx G_k + Hxu E_k   (NX x nv) */
            for (int e = tid; e < NX * nv; e += NT)
            {
                const int r = e / nv, a = e % nv;
                double s = 0.0;
                if (a < k * NU || a >= nvu)
                    for (int q2 = 0; q2 < NX; q2++) s += Hk[(NU + r) * n + NU + q2] * Gk[(size_t) q2 * nv + a];
                if (a >= k * NU && a < (k + 1) * NU) s += Hk[(NU + r) * n + (a - k * NU)];
                P[e] = s;
            }
            __syncthreads();
            /* M += E_k' Huu E_k + E_k' Hux G_k + G_k' P  (lower triangle, a >= b) */
            for (int e = tid; e < nv * nv; e += NT)
            {
                const int a = e / nv, b = e % nv;
                if (b > a) continue;
                const bool ain = a < k * NU || a >= nvu, bin = b < k * NU || b >= nvu;
                const bool au_ = a >= k * NU && a < (k + 1) * NU, bu_ = b >= k * NU && b < (k + 1) * NU;
                if (!(ain || au_) || !(bin || bu_)) continue;
                double s = 0.0;
                if (ain) for (int r = 0; r < NX; r++) s += Gk[(size_t) r * nv + a] * P[(size_t) r * nv + b];
                if (au_)
                {
                    const int ja = a - k * NU;
                    if (bu_) s += Hk[ja * n + (b - k * NU)];
                    if (bin) for (int r = 0; r < NX; r++) s += Hk[ja * n + NU + r] * Gk[(size_t) r * nv + b];
                }
                M[(size_t) a * nv + b] += s;
            }
            __syncthreads();
        }
        /* fixed variables: identity rows / columns */
        for (int e = tid; e < nv * nv; e += NT)
        {
            const int a = e / nv, b = e % nv;
            if (b > a) continue;
            if (dv[a] != 0.0 || dv[b] != 0.0) M[e] = a == b ? 1.0 : 0.0;
        }
        __syncthreads();
        /* Cholesky, lower, in place (row-major M[a][b], b <= a): non-positive pivots zeroed as the stage-wise kernels do */
        for (int j = 0; j < nv; j++)
        {
            const double d = M[(size_t) j * nv + j];
            const double rinv = d > 0.0 ? 1.0 / sqrt(d) : 0.0;
            __syncthreads();
            for (int a = j + tid; a < nv; a += NT) M[(size_t) a * nv + j] *= rinv; /* (a == j: d / sqrt(d)) */
            __syncthreads();
            const int rem = nv - j - 1;
            for (int e = tid; e < rem * rem; e += NT)
            {
                const int a = j + 1 + e / rem, b = j + 1 + e % rem;
                if (b > a) continue;
                M[(size_t) a * nv + b] -= M[(size_t) a * nv + j] * M[(size_t) b * nv + j];
            }
            __syncthreads();
        }
        if (tid == 0) s_fixed_cnt = 0;

        /* ---- predictor, corrector, (conditional redo): rhs-only solves with the factor ---- */
        double sigma = 0.0, musave = mu;
        for (int phase = 0; phase < 3; phase++)
        {
            /* complementarity rhs of this phase in D.pcorr: affine rm; corrected rm + dlam dt - sigma mu; centering only rm - sigma mu */
            for (int r = tid; r < R; r += NT)
            {
                const KdRow q = rows[r];
                for (int sd = 0; sd < 2; sd++)
                {
                    const int e = sd ? q.up : q.lo;
                    const double rm = GATL(D.rm, e);
                    double x = rm;
                    if (phase == 1) x = rm + GATL(D.dlam, e) * GATL(D.dt, e) - sigma * musave;
                    if (phase == 2) x = rm - sigma * musave;
                    GATL(D.pcorr, e) = x;
                }
            }
            /* g~_k = rw_k + J'(rho_l - rho_u), rho = (rmeff + lam rd) / t */
            for (int e = tid; e < K * n; e += NT) gs[e] = rw[e];
            __syncthreads();
            for (int r = tid; r < R; r += NT)
            {
                const KdRow q = rows[r];
                const uint64_t am0 = GATL(D.amask, q.k * D.AW + (q.blo >> 6)), am1 = GATL(D.amask, q.k * D.AW + (q.bup >> 6));
                const bool al = !q.fixed && ((am0 >> (q.blo & 63)) & 1), au = !q.fixed && ((am1 >> (q.bup & 63)) & 1);
                const double rl = al ? (GATL(D.pcorr, q.lo) + GATL(D.lam, q.lo) * GATL(D.rd, q.lo)) / GATL(D.t, q.lo) : 0.0;
                const double ru = au ? (GATL(D.pcorr, q.up) + GATL(D.lam, q.up) * GATL(D.rd, q.up)) / GATL(D.t, q.up) : 0.0;
                const double nu_ = rl - ru;
                if (nu_ == 0.0) continue;
                if (q.var >= 0) atomicAdd(&gs[q.k * n + q.var], nu_);
                else for (int j = 0; j < n; j++) atomicAdd(&gs[q.k * n + j], nu_ * GATL(D.DCt, q.g * n + j));
            }
            __syncthreads();
            for (int a = tid; a < nv; a += NT)
            {
                double s = 0.0;
                if (a < nvu) s = gs[(a / NU) * n + a % NU];
                for (int k = 0; k < K; k++)
                {
                    if (!(a < k * NU || a >= nvu)) continue;
                    const double *Gk = G + (size_t) k * NX * nv;
                    for (int r = 0; r < NX; r++) s += Gk[(size_t) r * nv + a] * gs[k * n + NU + r];
                }
                rhs[a] = -s;
            }
            __syncthreads();
            /* fixed flags are in dv only before the first solve of this iteration: move them into P[0..nv) */
            if (phase == 0) { for (int a = tid; a < nv; a += NT) P[a] = dv[a]; }
            __syncthreads();
            for (int a = tid; a < nv; a += NT) if (P[a] != 0.0) rhs[a] = 0.0;
            __syncthreads();
            /* L y = rhs ; L' dv = y  (column-oriented, one column per step) */
            for (int j = 0; j < nv; j++)
            {
                const double d = M[(size_t) j * nv + j];
                if (tid == 0) rhs[j] = d != 0.0 ? rhs[j] / d : 0.0;
                __syncthreads();
                const double yj = rhs[j];
                for (int a = j + 1 + tid; a < nv; a += NT) rhs[a] -= M[(size_t) a * nv + j] * yj;
                __syncthreads();
            }
            for (int j = nv - 1; j >= 0; j--)
            {
                const double d = M[(size_t) j * nv + j];
                if (tid == 0) rhs[j] = d != 0.0 ? rhs[j] / d : 0.0;
                __syncthreads();
                const double xj = rhs[j];
                for (int b = tid; b < j; b += NT) rhs[b] -= M[(size_t) j * nv + b] * xj;
                __syncthreads();
            }
            for (int a = tid; a < nv; a += NT) dv[a] = rhs[a];
            __syncthreads();
            /* dw_k = T_k dv ; row steps ; ratio test */
            for (int e = tid; e < K * n; e += NT)
            {
                const int k = e / n, j = e % n;
                double s;
                if (j < NU) s = dv[k * NU + j];
                else
                {
                    const double *Gr = G + ((size_t) k * NX + (j - NU)) * nv;
                    s = 0.0;
                    for (int a = 0; a < nv; a++) s += Gr[a] * dv[a];
                }
                dw[e] = s;
            }
            __syncthreads();
            double amin = 1.0, s1 = 0.0, s2 = 0.0;
            for (int r = tid; r < R; r += NT)
            {
                const KdRow q = rows[r];
                const double dc = KD_ROWVAL(q, dw + q.k * n);
                const uint64_t am0 = GATL(D.amask, q.k * D.AW + (q.blo >> 6)), am1 = GATL(D.amask, q.k * D.AW + (q.bup >> 6));
                const bool on[2] = {!q.fixed && ((am0 >> (q.blo & 63)) & 1), !q.fixed && ((am1 >> (q.bup & 63)) & 1)};
                for (int sd = 0; sd < 2; sd++)
                {
                    const int e = sd ? q.up : q.lo;
                    if (!on[sd]) { GATL(D.dt, e) = 0.0; GATL(D.dlam, e) = 0.0; continue; }
                    const double lam = GATL(D.lam, e), t = GATL(D.t, e);
                    const double dt = (sd ? -dc : dc) + GATL(D.rd, e);
                    const double dl = -(GATL(D.pcorr, e) + lam * dt) / t;
                    GATL(D.dt, e) = dt; GATL(D.dlam, e) = dl;
                    if (dl < 0.0 && -lam > amin * dl) amin = -lam / dl;
                    if (dt < 0.0 && -t > amin * dt) amin = -t / dt;
                }
            }
            alpha = kd_blk_min(amin, sh);
            /* duality measure at the end of this step */
            for (int r = tid; r < R; r += NT)
            {
                const KdRow q = rows[r];
                const uint64_t am0 = GATL(D.amask, q.k * D.AW + (q.blo >> 6)), am1 = GATL(D.amask, q.k * D.AW + (q.bup >> 6));
                const bool on[2] = {!q.fixed && ((am0 >> (q.blo & 63)) & 1), !q.fixed && ((am1 >> (q.bup & 63)) & 1)};
                for (int sd = 0; sd < 2; sd++)
                {
                    const int e = sd ? q.up : q.lo;
                    if (on[sd]) s1 += (GATL(D.lam, e) + alpha * GATL(D.dlam, e)) * (GATL(D.t, e) + alpha * GATL(D.dt, e));
                }
            }
            const double mu_end = nact > 0 ? kd_blk_sum(s1, sh) / nact : 0.0;
            (void) s2;
            if (!O.pred_corr || nact == 0) break;
            if (phase == 0)
            {
                sigma = musave > 0.0 ? mu_end / musave : 0.0;
                sigma = sigma * sigma * sigma;
                continue;
            }
            if (phase == 1 && O.cond_pred_corr && mu_end > 2.0 * musave) continue; /* redo from the centering term alone */
            break;
        }

        /* ---- update (step to the boundary as HPIPM scales it), floors ---- */
        const double a_ = nact > 0 ? gqp_step_scale(alpha) : 1.0;
        for (int a = tid; a < nv; a += NT) v[a] += a_ * dv[a];
        for (int r = tid; r < R; r += NT)
        {
            const KdRow q = rows[r];
            const uint64_t am0 = GATL(D.amask, q.k * D.AW + (q.blo >> 6)), am1 = GATL(D.amask, q.k * D.AW + (q.bup >> 6));
            const bool on[2] = {!q.fixed && ((am0 >> (q.blo & 63)) & 1), !q.fixed && ((am1 >> (q.bup & 63)) & 1)};
            for (int sd = 0; sd < 2; sd++)
            {
                const int e = sd ? q.up : q.lo;
                if (!on[sd]) continue;
                const double lam = GATL(D.lam, e) + a_ * GATL(D.dlam, e), t = GATL(D.t, e) + a_ * GATL(D.dt, e);
                GATL(D.lam, e) = lam < O.lam_min ? O.lam_min : lam;
                GATL(D.t, e) = t < O.t_min ? O.t_min : t;
            }
        }
        __syncthreads();
        it++;
    }

    /* ---- results into the parent's iterate: ux, pi by the adjoint recursion, per-instance scalars ---- */
    for (int e = tid; e < K * n; e += NT) GATL(D.ux, e) = w[e];
    /* pi slot k (multiplier of the dynamics that produce x_k), k = N .. 1: pi_k = rw_x,k + A_k' pi_{k+1}; serial in k, parallel in c */
    for (int c = tid; c < NX; c += NT) { GATL(D.pi, c) = 0.0; GATL(D.pi, K * NX + c) = 0.0; }
    __syncthreads();
    for (int k = K - 1; k >= 1; k--)
    {
        for (int c = tid; c < NX; c += NT)
        {
            double s = rw[k * n + NU + c];
            if (k + 1 < K) for (int r = 0; r < NX; r++) s += GATL(D.BAt, (k * n + NU + c) * NX + r) * GATL(D.pi, (k + 1) * NX + r);
            GATL(D.pi, k * NX + c) = s;
        }
        __syncthreads();
    }
    if (tid == 0)
    {
        D.status[i] = status; D.iter[i] = it; D.mu[i] = mu; D.alpha[i] = alpha;
        D.res[0 * D.Bp + i] = nrm_g; D.res[1 * D.Bp + i] = 0.0; D.res[2 * D.Bp + i] = nrm_d; D.res[3 * D.Bp + i] = nrm_m;
    }
    (void) s_fixed_cnt;
}

} // namespace gqp

#endif
