/*
 * res_kernels.hpp -- KKT residuals of an ARBITRARY (qp_in, qp_out) pair held by a device batch.
 *
 * What it replaces (reference, /root/reference): ocp_qp_res_compute -> d_ocp_qp_res_compute and
 * ocp_qp_res_compute_nrm_inf (acados/ocp_qp/ocp_qp_common.c:559-667), reached through
 * ocp_qp_inf_norm_residuals (interfaces/acados_c/ocp_qp_interface.c:642-650) -- what the reference's unit test
 * asserts on (test/ocp_qp/test_qpsolvers.cpp:240-251).  The formulas follow the KKT system of
 * ocp_qp_clarabel.c:493-683 / acados_ocp_qp.py:24-45 with acados' conventions: lam, t >= 0 ordered
 * [lb lg ub ug ls us], pi[k] the multiplier of A x + B u + b - x+ = 0, masked sides contribute nothing.
 *
 *   res_g  = H v + g + [B A]' pi_k - [0; pi_{k-1}] - J'(lam_l - lam_u)      (and Z s + z - lam_s - lam_row for slacks)
 *   res_b  = A x + B u + b - x+
 *   res_d  = [J v + s_l - lb - t_l ; ub - J v + s_u - t_u ; s - ls - t_s]
 *   res_m  = lam .* t
 *
 * Deliberately independent of the IPM sweeps (ipm_kernels*.hpp compute their own residuals as a by-product of the
 * factor sweep): nothing is shared but the data layout, so this kernel is the second opinion on every
 * "KKT residual <= tol" statement of the solver.  Equality-flagged bounds (idxe) are treated as what they are in the
 * reference -- box rows whose two sides carry the multiplier of the fixed variable -- so the multipliers the
 * finalize kernels recover from stationarity are checked too.
 *
 * Mapping: one instance per lane, run-time dims, layout-agnostic accessor (GATL): not a hot kernel (one pass over
 * the data, called on demand).
 */
#ifndef RES_KERNELS_HPP_
#define RES_KERNELS_HPP_

#include "ipm_kernels.hpp"

namespace gqp
{

struct ResOut
{
    GArr g;   /* [N+1][n]  stationarity w.r.t. [u; x] (padded layout) */
    GArr gs;  /* [sum 2ns] stationarity w.r.t. sl then su */
    GArr b;   /* [N+1][NX] dynamics */
    GArr d;   /* [sum nct] inequalities */
    GArr m;   /* [sum nct] complementarity */
    double *nrm; /* [4][Bp] */
    int Bp;
};

static __global__ void __launch_bounds__(64) k_res_compute(GqpDev D, ResOut R)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.B) return;
    const int NX = D.NX, NU = D.NU, n = NX + NU, NP = n * (n + 1) / 2;
    double n_g = 0.0, n_b = 0.0, n_d = 0.0, n_m = 0.0;
    for (int k = 0; k <= D.N; k++)
    {
        GQP_STAGE_REF S = D.st[k];
        const int nbg = S.nb + S.ng, ns = S.ns;
        uint64_t am[2] = {GATL(D.amask, k * D.AW), D.AW > 1 ? GATL(D.amask, k * D.AW + 1) : 0};
        auto active = [&](int e) { return (am[e >> 6] >> (e & 63)) & 1; };
        /* ---- stationarity w.r.t. [u; x] ---- */
        for (int j = 0; j < n; j++)
        {
            double a = GATL(D.rq, k * n + j);
            for (int c = 0; c < n; c++)
                a += GATL(D.RSQ, k * NP + (j >= c ? PK(j, c) : PK(c, j))) * GATL(D.ux, k * n + c);
            for (int c = 0; c < NX; c++) a += GATL(D.BAt, (k * n + j) * NX + c) * GATL(D.pi, (k + 1) * NX + c);
            if (j >= NU) a -= GATL(D.pi, k * NX + (j - NU));
            if ((S.bmask >> j) & 1)
            {
                const int row = popc64_g(S.bmask & (((uint64_t) 1 << j) - 1));
                const bool fixed = (S.emask >> j) & 1;
                const double ll = GATL(D.lam, S.o_ct + row), lu = GATL(D.lam, S.o_ct + nbg + row);
                a -= (fixed || active(row) ? ll : 0.0) - (fixed || active(nbg + row) ? lu : 0.0);
            }
            for (int g = 0; g < S.ng; g++)
            {
                const int row = S.nb + g;
                const double ll = active(row) ? GATL(D.lam, S.o_ct + row) : 0.0;
                const double lu = active(nbg + row) ? GATL(D.lam, S.o_ct + nbg + row) : 0.0;
                a -= GATL(D.DCt, (S.o_g + g) * n + j) * (ll - lu);
            }
            GATL(R.g, k * n + j) = a;
            nacc(n_g, a);
        }
        /* ---- stationarity w.r.t. the slacks ---- */
        for (int q = 0; q < ns; q++)
        {
            const int e0 = S.o_ct + 2 * nbg + q, e1 = e0 + ns;
            double al = GATL(D.Zz, (S.o_s + q) * 2) * GATL(D.sv, S.o_s + q) + GATL(D.Zz, (S.o_s + q) * 2 + 1);
            double au = GATL(D.Zz, (S.o_s + ns + q) * 2) * GATL(D.sv, S.o_s + ns + q) + GATL(D.Zz, (S.o_s + ns + q) * 2 + 1);
            if (active(2 * nbg + q)) al -= GATL(D.lam, e0);
            if (active(2 * nbg + ns + q)) au -= GATL(D.lam, e1);
            for (int row = 0; row < nbg; row++)
                if (S.srev[row] == q)
                {
                    if (active(row)) al -= GATL(D.lam, S.o_ct + row);
                    if (active(nbg + row)) au -= GATL(D.lam, S.o_ct + nbg + row);
                }
            GATL(R.gs, S.o_s + q) = al;
            GATL(R.gs, S.o_s + ns + q) = au;
            nacc(n_g, al);
            nacc(n_g, au);
        }
        /* ---- dynamics ---- */
        for (int c = 0; c < NX; c++)
        {
            double a = 0.0;
            if (S.has_dyn)
            {
                a = GATL(D.bvec, k * NX + c) - GATL(D.ux, (k + 1) * n + NU + c);
                for (int j = 0; j < n; j++) a += GATL(D.BAt, (k * n + j) * NX + c) * GATL(D.ux, k * n + j);
            }
            GATL(R.b, k * NX + c) = a;
            nacc(n_b, a);
        }
        /* ---- inequalities and complementarity ---- */
        int ib = 0;
        for (int j = 0; j < n + S.ng; j++)
        {
            const bool is_row = j < n ? ((S.bmask >> j) & 1) : true;
            if (!is_row) continue;
            const int row = j < n ? ib++ : S.nb + (j - n);
            const bool fixed = j < n && ((S.emask >> j) & 1);
            double c = 0.0;
            if (j < n) c = GATL(D.ux, k * n + j);
            else
                for (int r = 0; r < n; r++) c += GATL(D.DCt, (S.o_g + (j - n)) * n + r) * GATL(D.ux, k * n + r);
            double ssl = 0.0, ssu = 0.0;
            const int sj = S.srev[row];
            if (sj >= 0) { ssl = GATL(D.sv, S.o_s + sj); ssu = GATL(D.sv, S.o_s + ns + sj); }
            const int el = S.o_ct + row, eu = S.o_ct + nbg + row;
            const bool al = fixed || active(row), au = fixed || active(nbg + row);
            const double tl = GATL(D.t, el), tu = GATL(D.t, eu);
            const double dl = al ? c + ssl - GATL(D.dvec, el) - tl : 0.0;
            const double du = au ? GATL(D.dvec, eu) - c + ssu - tu : 0.0;
            const double ml = al ? GATL(D.lam, el) * tl : 0.0, mu_ = au ? GATL(D.lam, eu) * tu : 0.0;
            GATL(R.d, el) = dl; GATL(R.d, eu) = du;
            GATL(R.m, el) = ml; GATL(R.m, eu) = mu_;
            nacc(n_d, dl); nacc(n_d, du);
            nacc(n_m, ml); nacc(n_m, mu_);
        }
        for (int q = 0; q < 2 * ns; q++)
        {
            const int e = S.o_ct + 2 * nbg + q;
            const bool a = active(2 * nbg + q);
            const double tq = GATL(D.t, e);
            const double dq = a ? GATL(D.sv, S.o_s + q) - GATL(D.dvec, e) - tq : 0.0;
            const double mq = a ? GATL(D.lam, e) * tq : 0.0;
            GATL(R.d, e) = dq; GATL(R.m, e) = mq;
            nacc(n_d, dq); nacc(n_m, mq);
        }
    }
    R.nrm[0 * (size_t) R.Bp + i] = n_g;
    R.nrm[1 * (size_t) R.Bp + i] = n_b;
    R.nrm[2 * (size_t) R.Bp + i] = n_d;
    R.nrm[3 * (size_t) R.Bp + i] = n_m;
}

} // namespace gqp

#endif
