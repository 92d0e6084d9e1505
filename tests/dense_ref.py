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
