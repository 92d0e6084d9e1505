"""Independent dense restatement of the OCP-QP (NumPy/SciPy), used only by tests.

Assembles min 1/2 w'Hw + g'w s.t. Aeq w = beq, Ain w >= bin from an AcadosOcpQp
following acados_casadi_ocp_qp.py:52-260 (the reference's own independent
restatement: stage-wise variables, dynamics equalities, two one-sided rows per
soft constraint, masks -> row dropped, slack lower bounds), and solves it with
SciPy SLSQP.  Shares no code with oracle/ or the HIP path.
"""
import numpy as np
from scipy.optimize import minimize


def assemble(qp):
    d, N = qp.dims, qp.N
    off, nw = [], 0
    for k in range(N + 1):
        off.append(nw)
        nw += d.nu[k] + d.nx[k] + 2 * d.ns[k]
    H, g = np.zeros((nw, nw)), np.zeros(nw)
    Aeq, beq, Ain, bin_ = [], [], [], []
    for k in range(N + 1):
        nu, nx, ns, nb, ng = d.nu[k], d.nx[k], d.ns[k], d.nb[k], d.ng[k]
        o = off[k]
        iu, ix = np.arange(o, o + nu), np.arange(o + nu, o + nu + nx)
        isl, isu = np.arange(o + nu + nx, o + nu + nx + ns), np.arange(o + nu + nx + ns, o + nu + nx + 2 * ns)
        H[np.ix_(iu, iu)] = qp.R[k]; H[np.ix_(iu, ix)] = qp.S[k]; H[np.ix_(ix, iu)] = qp.S[k].T
        H[np.ix_(ix, ix)] = qp.Q[k]
        g[iu], g[ix] = qp.r[k], qp.q[k]
        H[isl, isl], H[isu, isu] = qp.Zl[k], qp.Zu[k]
        g[isl], g[isu] = qp.zl[k], qp.zu[k]
        if k < N:
            nx1 = d.nx[k + 1]
            row = np.zeros((nx1, nw))
            row[:, ix], row[:, iu] = qp.A[k], qp.B[k]
            row[:, off[k + 1] + d.nu[k + 1] + np.arange(nx1)] -= np.eye(nx1)
            Aeq.append(row); beq.append(-qp.b[k])
        J = np.zeros((nb + ng, nw))
        for i in range(nb):
            J[i, o + qp.idxb[k][i]] = 1.0
        if ng:
            J[nb:, ix] = qp.C[k]
            J[nb:, iu] = qp.D[k]
        lo = np.concatenate([qp.lbu[k], qp.lbx[k], qp.lg[k]])
        up = np.concatenate([qp.ubu[k], qp.ubx[k], qp.ug[k]])
        mlo = np.concatenate([qp.lbu_mask[k], qp.lbx_mask[k], qp.lg_mask[k]])
        mup = np.concatenate([qp.ubu_mask[k], qp.ubx_mask[k], qp.ug_mask[k]])
        eq = set(int(e) for e in qp.idxe[k])
        for i in range(nb + ng):
            j = qp.idxs_rev[k][i]
            if i in eq:
                Aeq.append(J[i:i + 1]); beq.append(np.array([lo[i]]))
                continue
            if mlo[i] != 0:
                r = J[i].copy()
                if j >= 0:
                    r[isl[j]] += 1.0
                Ain.append(r[None]); bin_.append(np.array([lo[i]]))
            if mup[i] != 0:
                r = -J[i].copy()
                if j >= 0:
                    r[isu[j]] += 1.0
                Ain.append(r[None]); bin_.append(np.array([-up[i]]))
        for j in range(ns):
            if qp.lls_mask[k][j] != 0:
                r = np.zeros(nw); r[isl[j]] = 1.0
                Ain.append(r[None]); bin_.append(np.array([qp.lls[k][j]]))
            if qp.lus_mask[k][j] != 0:
                r = np.zeros(nw); r[isu[j]] = 1.0
                Ain.append(r[None]); bin_.append(np.array([qp.lus[k][j]]))
    cat = lambda L, n: np.concatenate(L) if L else np.zeros((0, n) if n else (0,))
    return H, g, cat(Aeq, nw), cat(beq, 0), cat(Ain, nw), cat(bin_, 0), off


def solve_dense(qp, w0=None):
    H, g, Aeq, beq, Ain, bin_, off = assemble(qp)
    nw = len(g)
    w0 = np.zeros(nw) if w0 is None else w0
    cons = [{"type": "eq", "fun": lambda w: Aeq @ w - beq, "jac": lambda w: Aeq}]
    if len(bin_):
        cons.append({"type": "ineq", "fun": lambda w: Ain @ w - bin_, "jac": lambda w: Ain})
    res = minimize(lambda w: 0.5 * w @ H @ w + g @ w, w0, jac=lambda w: H @ w + g, constraints=cons,
                   method="SLSQP", options={"ftol": 1e-15, "maxiter": 2000})
    return res.x, off, res


def split(qp, w, off):
    d = qp.dims
    out = {"u": [], "x": [], "sl": [], "su": []}
    for k in range(qp.N + 1):
        o, nu, nx, ns = off[k], d.nu[k], d.nx[k], d.ns[k]
        out["u"].append(w[o:o + nu]); out["x"].append(w[o + nu:o + nu + nx])
        out["sl"].append(w[o + nu + nx:o + nu + nx + ns]); out["su"].append(w[o + nu + nx + ns:o + nu + nx + 2 * ns])
    return out


# ---------------------------------------------------------------------------------------------------------------
# Independent KKT residuals (a10) and solution sensitivities (a12): plain NumPy on the stage data, acados conventions
# (acados_ocp_qp.py:24-45; multiplier order [lbu lbx lg ubu ubx ug lls lus]; pi[k] multiplies A x + B u + b - x+).
# Shares no code with oracle/ or the HIP path.
# ---------------------------------------------------------------------------------------------------------------

def _stage(qp, k):
    d = qp.dims
    nu, nx, ns, nb, ng = int(d.nu[k]), int(d.nx[k]), int(d.ns[k]), int(d.nb[k]), int(d.ng[k])
    H = np.zeros((nu + nx, nu + nx))
    H[:nu, :nu], H[:nu, nu:], H[nu:, :nu], H[nu:, nu:] = qp.R[k], qp.S[k], qp.S[k].T, qp.Q[k]
    g = np.concatenate([qp.r[k], qp.q[k]])
    J = np.zeros((nb + ng, nu + nx))
    for i in range(nb):
        J[i, qp.idxb[k][i]] = 1.0
    if ng:
        J[nb:, :nu], J[nb:, nu:] = qp.D[k], qp.C[k]
    lo = np.concatenate([qp.lbu[k], qp.lbx[k], qp.lg[k]])
    up = np.concatenate([qp.ubu[k], qp.ubx[k], qp.ug[k]])
    mlo = np.concatenate([qp.lbu_mask[k], qp.lbx_mask[k], qp.lg_mask[k]]) != 0
    mup = np.concatenate([qp.ubu_mask[k], qp.ubx_mask[k], qp.ug_mask[k]]) != 0
    eq = np.zeros(nb + ng, dtype=bool)
    for e in qp.idxe[k]:
        eq[int(e)] = True
    return dict(nu=nu, nx=nx, ns=ns, nb=nb, ng=ng, nbg=nb + ng, H=H, g=g, J=J, lo=lo, up=up, mlo=mlo, mup=mup, eq=eq,
                rev=np.asarray(qp.idxs_rev[k]).astype(int), mls=np.asarray(qp.lls_mask[k]) != 0, mus=np.asarray(qp.lus_mask[k]) != 0)


def kkt_residuals(qp, get):
    """res_g, res_b, res_d, res_m per stage of the iterate get(k, field) (x u sl su pi lam t), the way
    d_ocp_qp_res_compute defines them (ocp_qp_common.c:559-594 calls it; formulas from the KKT system in
    ocp_qp_clarabel.c:493-683): masked sides contribute nothing, equality-flagged rows are ordinary box rows"""
    N = qp.N
    out = {"res_g": [], "res_b": [], "res_d": [], "res_m": []}
    for k in range(N + 1):
        s = _stage(qp, k)
        nu, ns, nbg = s["nu"], s["ns"], s["nbg"]
        v = np.concatenate([get(k, "u"), get(k, "x")])
        sl, su, lam, t = get(k, "sl"), get(k, "su"), get(k, "lam"), get(k, "t")
        al, au = s["mlo"] | s["eq"], s["mup"] | s["eq"]
        ll, lu = np.where(al, lam[:nbg], 0.0), np.where(au, lam[nbg:2 * nbg], 0.0)
        rg = s["H"] @ v + s["g"] - s["J"].T @ (ll - lu)
        if k < N:
            pk = get(k, "pi")
            rg[:nu] += qp.B[k].T @ pk
            rg[nu:] += qp.A[k].T @ pk
        if k > 0:
            rg[nu:] -= get(k - 1, "pi")
        lls_, lus_ = np.where(s["mls"], lam[2 * nbg:2 * nbg + ns], 0.0), np.where(s["mus"], lam[2 * nbg + ns:], 0.0)
        rsl = qp.Zl[k] * sl + qp.zl[k] - lls_
        rsu = qp.Zu[k] * su + qp.zu[k] - lus_
        c = s["J"] @ v
        rd, rm = np.zeros(2 * nbg + 2 * ns), np.zeros(2 * nbg + 2 * ns)
        for i in range(nbg):
            j = s["rev"][i]
            if j >= 0:
                rsl[j] -= ll[i]
                rsu[j] -= lu[i]
            if al[i]:
                rd[i] = c[i] + (sl[j] if j >= 0 else 0.0) - s["lo"][i] - t[i]
                rm[i] = lam[i] * t[i]
            if au[i]:
                rd[nbg + i] = s["up"][i] - c[i] + (su[j] if j >= 0 else 0.0) - t[nbg + i]
                rm[nbg + i] = lam[nbg + i] * t[nbg + i]
        for j in range(ns):
            if s["mls"][j]:
                rd[2 * nbg + j] = sl[j] - qp.lls[k][j] - t[2 * nbg + j]
                rm[2 * nbg + j] = lam[2 * nbg + j] * t[2 * nbg + j]
            if s["mus"][j]:
                rd[2 * nbg + ns + j] = su[j] - qp.lus[k][j] - t[2 * nbg + ns + j]
                rm[2 * nbg + ns + j] = lam[2 * nbg + ns + j] * t[2 * nbg + ns + j]
        out["res_g"].append(np.concatenate([rg, rsl, rsu]))
        out["res_b"].append(qp.A[k] @ v[nu:] + qp.B[k] @ v[:nu] + qp.b[k] - get(k + 1, "x") if k < N else np.zeros(0))
        out["res_d"].append(rd)
        out["res_m"].append(rm)
    return out


def kkt_residual_norms(qp, get):
    r = kkt_residuals(qp, get)
    return np.array([max([np.max(np.abs(a)) if a.size else 0.0 for a in r[n]] + [0.0]) for n in ("res_g", "res_b", "res_d", "res_m")])


def sens_dense(qp, get, seeds):
    """d(solution)/d(parameter) from ONE dense solve of the linearised KKT system at the iterate get(k, field) -- what
    d_ocp_qp_ipm_sens_frw computes with the factorisation at the last IPM iterate (ocp_qp_hpipm.c:481-491):

        [ H   Aeq'  -Ain'  0 ] [dw  ]   [-dg  ]
        [ Aeq  0     0     0 ] [dnu ] = [ dbeq]
        [ Ain  0     0    -I ] [dlam]   [ dbin]
        [ 0    0     T     L ] [dt  ]   [ 0   ]          T = diag(t), L = diag(lam)

    over the inequality sides that take part (mask != 0, not equality-flagged).  seeds: {(field, k): vector}, field in
    q r zl zu b lbu ubu lbx ubx lg ug lls lus (natural-sign bounds; an equality-flagged row takes its lbx seed).  Returns a function
    (k, field) -> array for x u sl su pi lam t, lam/t zero on sides that do not take part."""
    N, d = qp.N, qp.dims
    off, nw = [], 0
    for k in range(N + 1):
        off.append(nw)
        nw += int(d.nu[k] + d.nx[k] + 2 * d.ns[k])
    H, dg = np.zeros((nw, nw)), np.zeros(nw)
    eq_rows, eq_rhs, eq_tag = [], [], []
    in_rows, in_rhs, in_tag = [], [], []
    sd = lambda f, k, n: np.asarray(seeds.get((f, k), np.zeros(n)), dtype=float).reshape(-1)
    for k in range(N + 1):
        s = _stage(qp, k)
        nu, nx, ns, nbg, nb = s["nu"], s["nx"], s["ns"], s["nbg"], s["nb"]
        o = off[k]
        H[o:o + nu + nx, o:o + nu + nx] = s["H"]
        for j in range(ns):
            H[o + nu + nx + j, o + nu + nx + j] = qp.Zl[k][j]
            H[o + nu + nx + ns + j, o + nu + nx + ns + j] = qp.Zu[k][j]
        dg[o:o + nu] = sd("r", k, nu)
        dg[o + nu:o + nu + nx] = sd("q", k, nx)
        dg[o + nu + nx:o + nu + nx + ns] = sd("zl", k, ns)
        dg[o + nu + nx + ns:o + nu + nx + 2 * ns] = sd("zu", k, ns)
        if k < N:
            nx1 = int(d.nx[k + 1])
            row = np.zeros((nx1, nw))
            row[:, o:o + nu], row[:, o + nu:o + nu + nx] = qp.B[k], qp.A[k]
            row[:, off[k + 1] + int(d.nu[k + 1]) + np.arange(nx1)] -= np.eye(nx1)
            db = sd("b", k, nx1)
            for r in range(nx1):
                eq_rows.append(row[r]); eq_rhs.append(-db[r]); eq_tag.append(("pi", k, r))
        dlo = np.concatenate([sd("lbu", k, int(d.nbu[k])), sd("lbx", k, int(d.nbx[k])), sd("lg", k, s["ng"])])
        dup = np.concatenate([sd("ubu", k, int(d.nbu[k])), sd("ubx", k, int(d.nbx[k])), sd("ug", k, s["ng"])])
        lam, t = get(k, "lam"), get(k, "t")
        for i in range(nbg):
            Ji = np.zeros(nw)
            Ji[o:o + nu + nx] = s["J"][i]
            j = s["rev"][i]
            if s["eq"][i]:
                eq_rows.append(Ji); eq_rhs.append(dlo[i]); eq_tag.append(("eq", k, i))
                continue
            if s["mlo"][i]:
                r = Ji.copy()
                if j >= 0:
                    r[o + nu + nx + j] += 1.0
                in_rows.append(r); in_rhs.append(dlo[i]); in_tag.append((k, i, lam[i], t[i]))
            if s["mup"][i]:
                r = -Ji
                if j >= 0:
                    r[o + nu + nx + ns + j] += 1.0
                in_rows.append(r); in_rhs.append(-dup[i]); in_tag.append((k, nbg + i, lam[nbg + i], t[nbg + i]))
        for j in range(ns):
            for side, m in ((0, s["mls"]), (1, s["mus"])):
                if m[j]:
                    r = np.zeros(nw)
                    r[o + nu + nx + side * ns + j] = 1.0
                    e = 2 * nbg + side * ns + j
                    in_rows.append(r); in_rhs.append(sd("lls" if side == 0 else "lus", k, ns)[j]); in_tag.append((k, e, lam[e], t[e]))
    Aeq = np.array(eq_rows).reshape(len(eq_rows), nw)
    Ain = np.array(in_rows).reshape(len(in_rows), nw)
    ne, ni = Aeq.shape[0], Ain.shape[0]
    lam_v = np.array([tg[2] for tg in in_tag])
    t_v = np.array([tg[3] for tg in in_tag])
    n = nw + ne + 2 * ni
    K, rhs = np.zeros((n, n)), np.zeros(n)
    K[:nw, :nw] = H
    K[:nw, nw:nw + ne] = Aeq.T
    K[:nw, nw + ne:nw + ne + ni] = -Ain.T
    K[nw:nw + ne, :nw] = Aeq
    K[nw + ne:nw + ne + ni, :nw] = Ain
    K[nw + ne:nw + ne + ni, nw + ne + ni:] = -np.eye(ni)
    K[nw + ne + ni:, nw + ne:nw + ne + ni] = np.diag(t_v)
    K[nw + ne + ni:, nw + ne + ni:] = np.diag(lam_v)
    rhs[:nw] = -dg
    rhs[nw:nw + ne] = eq_rhs
    rhs[nw + ne:nw + ne + ni] = in_rhs
    z = _solve_extended(K, rhs)
    dw, dnu, dlam, dt = z[:nw], z[nw:nw + ne], z[nw + ne:nw + ne + ni], z[nw + ne + ni:]
    res = {}
    for k in range(N + 1):
        o, nu, nx, ns = off[k], int(d.nu[k]), int(d.nx[k]), int(d.ns[k])
        res[(k, "u")], res[(k, "x")] = dw[o:o + nu], dw[o + nu:o + nu + nx]
        res[(k, "sl")], res[(k, "su")] = dw[o + nu + nx:o + nu + nx + ns], dw[o + nu + nx + ns:o + nu + nx + 2 * ns]
        nct = 2 * int(d.nb[k] + d.ng[k] + d.ns[k])
        res[(k, "lam")], res[(k, "t")] = np.zeros(nct), np.zeros(nct)
        if k < N:
            res[(k, "pi")] = np.zeros(int(d.nx[k + 1]))
    for q, tg in enumerate(eq_tag):
        if tg[0] == "pi":
            res[(tg[1], "pi")][tg[2]] = dnu[q]
    for q, tg in enumerate(in_tag):
        res[(tg[0], "lam")][tg[1]] = dlam[q]
        res[(tg[0], "t")][tg[1]] = dt[q]
    return _Sens(res, {(tg[0], tg[1]) for tg in in_tag})


def _solve_extended(K, rhs):
    """Gaussian elimination with partial pivoting in 80-bit extended precision + two refinement steps: the KKT matrix of
    an IPM iterate holds t ~ 1e-10 next to lam ~ 1, its condition number eats most of a double"""
    n = K.shape[0]
    A = K.astype(np.longdouble)
    A0 = A.copy()
    b0 = rhs.astype(np.longdouble)
    # row equilibration (the complementarity rows are tiny)
    sc = np.max(np.abs(A), axis=1)
    sc[sc == 0] = 1
    perm = np.arange(n)
    Lm = np.zeros((n, n), dtype=np.longdouble)
    for c in range(n):
        p = c + int(np.argmax(np.abs(A[c:, c]) / sc[perm[c:]]))
        if p != c:
            A[[c, p]] = A[[p, c]]
            Lm[[c, p]] = Lm[[p, c]]
            perm[[c, p]] = perm[[p, c]]
        piv = A[c, c]
        f = A[c + 1:, c] / piv
        Lm[c + 1:, c] = f
        A[c + 1:, c:] -= np.outer(f, A[c, c:])

    def lu_solve(b):
        y = b[perm].copy()
        for c in range(n):
            y[c + 1:] -= Lm[c + 1:, c] * y[c]
        x = np.zeros(n, dtype=np.longdouble)
        for c in range(n - 1, -1, -1):
            x[c] = (y[c] - A[c, c + 1:] @ x[c + 1:]) / A[c, c]
        return x

    x = lu_solve(b0)
    for _ in range(2):
        x = x + lu_solve(b0 - A0 @ x)
    return x.astype(np.float64)


class _Sens:
    """(k, field) -> array; .active = {(k, side index)} of the inequality sides that take part"""

    def __init__(self, res, active):
        self._res, self.active = res, active

    def __call__(self, k, f):
        return self._res[(k, f)]


# ---------------------------------------------------------------------------------------------------------------
# Exact dense solution with an optimality CERTIFICATE (round 4): the tight reference for C4- / condensed-C3-shaped
# instances, where SLSQP above is too slow (1,700 variables).  A dense log-barrier path-following method (normal
# equations, scipy LU; not the oracle's Riccati / Mehrotra scheme) brings the point close, then the active set is read
# off, the equality-constrained QP of that active set is solved by ONE dense LU with refinement, and the KKT conditions of
# the ORIGINAL QP are verified on the result: inactive rows feasible, multipliers of active rows non-negative.  A strictly
# convex QP has one KKT point -- a verified certificate makes the answer independent of how the active set was found.
# Shares no code with oracle/ or the HIP path.
# ---------------------------------------------------------------------------------------------------------------

def solve_exact(qp, mu_end=1e-13, max_rounds=20, verbose=False):
    """returns (w, off, info): w the stacked primal solution [u x sl su] per stage, info = dict(lam_in, nu_eq, active,
    cert = max violation of the KKT certificate, rounds)"""
    import scipy.linalg as sla
    H, g, Aeq, beq, Ain, bin_, off = assemble(qp)
    nw, ne, ni = len(g), len(beq), len(bin_)
    # ---- phase 1: dense primal-dual path following on  min 1/2 w'Hw + g'w, Aeq w = beq, Ain w - s = bin, s >= 0
    w = np.zeros(nw)
    s = np.maximum(Ain @ w - bin_, 1.0)
    lam = np.ones(ni)
    nu = np.zeros(ne)
    for it in range(200):
        mu = float(s @ lam) / max(ni, 1)
        rd = H @ w + g + Aeq.T @ nu - Ain.T @ lam
        rp = Aeq @ w - beq
        rs = Ain @ w - s - bin_
        if mu < mu_end and max(np.abs(rd).max(), np.abs(rp).max() if ne else 0.0, np.abs(rs).max() if ni else 0.0) < 1e-9:
            break
        sigma = 0.1 if it else 0.5
        # eliminate ds = Ain dw + rs, dlam = (sigma mu - lam s - lam ds) / s
        D = lam / s
        K = np.zeros((nw + ne, nw + ne))
        K[:nw, :nw] = H + Ain.T @ (D[:, None] * Ain)
        K[:nw, nw:] = Aeq.T
        K[nw:, :nw] = Aeq
        rc = sigma * mu / s - lam
        rhs = np.concatenate([-rd + Ain.T @ (rc - D * rs), -rp])
        lu = sla.lu_factor(K)
        z = sla.lu_solve(lu, rhs)
        dw, dnu = z[:nw], z[nw:]
        ds = Ain @ dw + rs
        dlam = rc - D * ds
        a = 1.0
        for v, dv in ((s, ds), (lam, dlam)):
            neg = dv < 0
            if neg.any():
                a = min(a, 0.995 * float(np.min(-v[neg] / dv[neg])))
        w, nu, s, lam = w + a * dw, nu + a * dnu, s + a * ds, lam + a * dlam
    # ---- phase 2: active set -> equality QP -> certificate, repaired a few times if the guess was off
    active = lam > s
    cert, rounds = np.inf, 0
    for rounds in range(1, max_rounds + 1):
        Aa = Ain[active]
        na = Aa.shape[0]
        n = nw + ne + na
        K = np.zeros((n, n))
        K[:nw, :nw] = H
        K[:nw, nw:nw + ne] = Aeq.T
        K[nw:nw + ne, :nw] = Aeq
        K[:nw, nw + ne:] = -Aa.T
        K[nw + ne:, :nw] = -Aa
        rhs = np.concatenate([-g, beq, -bin_[active]])
        lu = sla.lu_factor(K)
        z = sla.lu_solve(lu, rhs)
        for _ in range(3):
            z = z + sla.lu_solve(lu, (rhs.astype(np.longdouble) - K.astype(np.longdouble) @ z.astype(np.longdouble)).astype(np.float64))
        wq, nuq, lq = z[:nw], z[nw:nw + ne], z[nw + ne:]
        slack = Ain @ wq - bin_
        viol_p = float(np.max(-slack[~active])) if (~active).any() else 0.0        # an inactive row is violated
        viol_d = float(np.max(-lq)) if na else 0.0                                   # an active row pulls the wrong way
        cert = max(viol_p, viol_d, 0.0)
        if verbose:
            print(f"active-set round {rounds}: active {na}, primal violation {viol_p:.2e}, negative multiplier {viol_d:.2e}")
        if cert <= 1e-11:
            break
        idx_a = np.flatnonzero(active)
        if viol_d > 1e-11:
            active[idx_a[lq < -1e-11]] = False
        active[(slack < -1e-11) & ~active] = True
    lam_full = np.zeros(ni)
    lam_full[active] = lq
    stat = float(np.max(np.abs(H @ wq + g + Aeq.T @ nuq - Ain.T @ lam_full)))
    return wq, off, {"lam_in": lam_full, "nu_eq": nuq, "active": active, "cert": cert, "stationarity": stat, "rounds": rounds,
                     "barrier_iters": it}
