"""TEST INFRASTRUCTURE: random OCP-QPs with random STRUCTURE (per-stage dims, box subsets, one-sided rows through the
*_mask fields, general rows, slacks shared between rows through idxs_rev, equality-flagged x0) for the parity tests.
Strictly convex, feasible by construction (bounds are placed around a feasible trajectory; rows that could make the
QP infeasible are soft)."""
import numpy as np

from acados_amd import AcadosOcpQp


def random_structure_qp(seed, N=None, nx_max=6, nu_max=3, allow_general=True, allow_slack=True, vary_dims=True):
    g = np.random.default_rng(seed)
    N = int(g.integers(1, 7)) if N is None else N
    nx0, nu0 = int(g.integers(1, nx_max + 1)), int(g.integers(1, nu_max + 1))
    nx = [nx0] * (N + 1)
    nu = [nu0] * N + [0]
    if vary_dims and N >= 3 and g.random() < 0.5:   # state dimension switches mid-horizon
        ks = int(g.integers(1, N))
        nxs = int(g.integers(1, nx_max + 1))
        for k in range(ks + 1, N + 1):
            nx[k] = nxs
    qp = AcadosOcpQp(N)
    # a feasible trajectory to put the bounds around
    xt = [g.uniform(-1, 1, nx[0])]
    ut = []
    dyn = []
    for k in range(N):
        A = 0.7 * g.standard_normal((nx[k + 1], nx[k])) / np.sqrt(nx[k])
        B = g.standard_normal((nx[k + 1], nu[k]))
        b = 0.1 * g.standard_normal(nx[k + 1])
        u = g.uniform(-0.3, 0.3, nu[k])
        dyn.append((A, B, b))
        ut.append(u)
        xt.append(A @ xt[k] + B @ u + b)
    ut.append(np.zeros(0))
    for k in range(N + 1):
        n = nx[k] + nu[k]
        M = g.standard_normal((n, n))
        H = M @ M.T / n + 0.1 * np.eye(n)
        qp.set("R", k, H[:nu[k], :nu[k]]); qp.set("S", k, H[:nu[k], nu[k]:]); qp.set("Q", k, H[nu[k]:, nu[k]:])
        qp.set("r", k, g.standard_normal(nu[k])); qp.set("q", k, g.standard_normal(nx[k]))
        if k < N:
            A, B, b = dyn[k]
            qp.set("A", k, A); qp.set("B", k, B); qp.set("b", k, b)
        v = np.concatenate([ut[k], xt[k]])
        if k == 0 and g.random() < 0.7:   # x0 given: all states equality-flagged
            ib_u = np.sort(g.choice(nu[k], size=int(g.integers(0, nu[k] + 1)), replace=False))
            idxb = np.concatenate([ib_u, nu[k] + np.arange(nx[k])]).astype(int)
            lb = np.concatenate([v[ib_u] - g.uniform(0.05, 1.0, ib_u.size), xt[0]])
            ub = np.concatenate([v[ib_u] + g.uniform(0.05, 1.0, ib_u.size), xt[0]])
            idxe = ib_u.size + np.arange(nx[k])
            hard = np.ones(idxb.size, dtype=bool)
        else:
            ib_u = np.sort(g.choice(nu[k], size=int(g.integers(0, nu[k] + 1)), replace=False)) if nu[k] else np.zeros(0, int)
            ib_x = np.sort(g.choice(nx[k], size=int(g.integers(0, nx[k] + 1)), replace=False))
            idxb = np.concatenate([ib_u, nu[k] + ib_x]).astype(int)
            lb = v[idxb] - g.uniform(0.02, 1.0, idxb.size)
            ub = v[idxb] + g.uniform(0.02, 1.0, idxb.size)
            idxe = np.zeros(0, int)
            hard = np.ones(idxb.size, dtype=bool)
        nbu, nb = ib_u.size, idxb.size
        qp.set("idxb", k, idxb)
        qp.set("lbu", k, lb[:nbu]); qp.set("ubu", k, ub[:nbu]); qp.set("lbx", k, lb[nbu:]); qp.set("ubx", k, ub[nbu:])
        if idxe.size:
            qp.set("idxe", k, idxe)
        ng = int(g.integers(0, 4)) if allow_general and g.random() < 0.6 else 0
        if ng:
            J = g.standard_normal((ng, n))
            mid = J @ v
            qp.set("D", k, J[:, :nu[k]]); qp.set("C", k, J[:, nu[k]:])
            qp.set("lg", k, mid - g.uniform(0.02, 1.0, ng)); qp.set("ug", k, mid + g.uniform(0.02, 1.0, ng))
        # one-sided rows
        for name, cnt in (("lbu_mask", nbu), ("ubu_mask", nbu), ("lbx_mask", nb - nbu), ("ubx_mask", nb - nbu), ("lg_mask", ng), ("ug_mask", ng)):
            m = np.ones(cnt)
            if cnt and g.random() < 0.3:
                m[g.random(cnt) < 0.4] = 0.0
            if name in ("lbx_mask", "ubx_mask") and idxe.size:
                m[:] = 1.0
            qp.set(name, k, m)
        # slacks: some rows soft, some sharing one slack
        nrow = nb + ng
        rev = -np.ones(nrow, dtype=int)
        ns = 0
        if allow_slack and nrow and g.random() < 0.5:
            cand = [r for r in range(nrow) if not (idxe.size and r >= nbu and r < nb)]
            g.shuffle(cand)
            for r in cand[:int(g.integers(1, min(len(cand), 4) + 1))] if cand else []:
                if ns and g.random() < 0.3:
                    rev[r] = int(g.integers(0, ns))   # shared slack
                else:
                    rev[r] = ns
                    ns += 1
        if ns:
            qp.set("idxs_rev", k, rev)
            qp.set("Zl", k, g.uniform(0.5, 50.0, ns)); qp.set("Zu", k, g.uniform(0.5, 50.0, ns))
            qp.set("zl", k, g.uniform(0.0, 5.0, ns)); qp.set("zu", k, g.uniform(0.0, 5.0, ns))
            qp.set("lls", k, np.zeros(ns)); qp.set("lus", k, np.zeros(ns))
    qp.make_consistent()
    return qp
