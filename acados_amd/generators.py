"""Synthetic OCP-QP generators for the configurations of BASELINE.md / SURVEY.md 8(d).

* mass_spring_qp: restates `create_ocp_qp_in_mass_spring`
  (examples/c/no_interface_examples/mass_spring_model/mass_spring_qp.c:60-130 system,
  :134-205 dims, :209-516 data) -- the C1 plumbing case and the reference's unit-test QP
  (test/ocp_qp/test_qpsolvers.cpp:117-268 uses N=15).
* random_lqr_batch: configuration C2 (random stable LQR-like, u-box + x0 equality).
"""
import numpy as np
import scipy.linalg

from .ocp_qp import AcadosOcpQp, AcadosOcpQpDims


def mass_spring_system(Ts, nx, nu):
    """ZOH discretisation of nx/2 masses in a row (mass_spring_qp.c:60-130)."""
    pp = nx // 2
    T = -2.0 * np.eye(pp) + np.eye(pp, k=1) + np.eye(pp, k=-1)
    Ac = np.block([[np.zeros((pp, pp)), np.eye(pp)], [T, np.zeros((pp, pp))]])
    Bc = np.zeros((nx, nu))
    Bc[pp:pp + nu, :] = np.eye(nu)
    A = scipy.linalg.expm(Ac * Ts)
    B = np.linalg.solve(Ac, (A - np.eye(nx)) @ Bc)
    return A, B


def mass_spring_qp(N=20, nx=8, nu=3, nb=11, x0_equality=True):
    """nb = nbu + nbx with nbu = min(nu, nb); stage 0 bounds every state at x0."""
    A, B = mass_spring_system(0.5, nx, nu)
    nbu = min(nu, nb)
    nbx = max(nb - nu, 0)
    x0 = np.zeros(nx)
    x0[0] = x0[1] = 2.5
    qp = AcadosOcpQp(N)
    for k in range(N + 1):
        last = k == N
        nuk = 0 if last else nu
        qp.set("Q", k, np.eye(nx)); qp.set("q", k, 0.1 * np.ones(nx))
        qp.set("R", k, 2.0 * np.eye(nuk)); qp.set("r", k, 0.2 * np.ones(nuk))
        qp.set("S", k, np.zeros((nuk, nx)))
        if not last:
            qp.set("A", k, A); qp.set("B", k, B); qp.set("b", k, 0.1 * np.ones(nx))
        nbuk = 0 if last else nbu
        qp.set("lbu", k, -0.5 * np.ones(nbuk)); qp.set("ubu", k, 0.5 * np.ones(nbuk))
        if k == 0:
            qp.set("lbx", k, x0); qp.set("ubx", k, x0)
            qp.set("idxb", k, np.concatenate([np.arange(nbuk), nuk + np.arange(nx)]))
            if x0_equality:
                qp.set("idxe", k, nbuk + np.arange(nx))
        else:
            qp.set("lbx", k, -4.0 * np.ones(nbx)); qp.set("ubx", k, 4.0 * np.ones(nbx))
            qp.set("idxb", k, np.concatenate([np.arange(nbuk), nuk + np.arange(nbx)]))
    qp.make_consistent()
    return qp


def lqr_dims(N, nx, nu):
    d = AcadosOcpQpDims(N)
    d.nx[:] = nx
    d.nu[:N] = nu
    d.nbu[:N] = nu
    d.nbx[0] = nx
    d.nb[:] = d.nbu + d.nbx
    d.nbxe[0] = nx
    return d


_CHUNK = 4096


def _stream(seed, tag, shape, dist="uniform", first=0):
    """counter-based streams, one per (seed, field tag, chunk of 4096 instances): instance i always
    sees the same numbers whatever the batch size or the rank that generates it, and a rank only
    generates the chunks that overlap its own instance range [first, first + shape[0])."""
    n = int(np.prod(shape[1:]))
    out = np.empty((shape[0], n))
    lo, hi = first, first + shape[0]
    for c in range(lo // _CHUNK, (hi + _CHUNK - 1) // _CHUNK):
        g = np.random.Generator(np.random.Philox(key=[int(seed), (int(tag) << 32) + c]))
        a = g.random((_CHUNK, n)) * 2.0 - 1.0 if dist == "uniform" else g.standard_normal((_CHUNK, n))
        s0, s1 = max(lo, c * _CHUNK), min(hi, (c + 1) * _CHUNK)
        out[s0 - lo:s1 - lo] = a[s0 - c * _CHUNK:s1 - c * _CHUNK]
    return out.reshape(shape)


def random_lqr_batch(N=50, nx=8, nu=3, batch=1024, seed=0, first=0):
    """Configuration C2: per instance
    A = A_d + 0.02 U(-1,1) rescaled to spectral radius <= 1.05, B = B_d + 0.02 U(-1,1),
    b ~ 0.1 U(-1,1), Q = I + 0.1 GG', R = 2I + 0.1 HH', S = 0.05 N(0,1), q,r ~ 0.1 N(0,1),
    x0 ~ U(-2.5,2.5) as equality bound at stage 0, u in [-0.5, 0.5].
    Generates the global instances [first, first + batch).
    Returns dict of arrays with leading dim `batch` (matrices [batch, rows, cols])."""
    Ad, Bd = mass_spring_system(0.5, nx, nu) if nx % 2 == 0 and nu <= nx // 2 else (0.9 * np.eye(nx), np.eye(nx, nu))
    A = Ad[None] + 0.02 * _stream(seed, 1, (batch, nx, nx), first=first)
    rho = np.max(np.abs(np.linalg.eigvals(A)), axis=1)
    A = A * np.minimum(1.0, 1.05 / rho)[:, None, None]
    B = Bd[None] + 0.02 * _stream(seed, 2, (batch, nx, nu), first=first)
    b = 0.1 * _stream(seed, 3, (batch, nx), first=first)
    G = _stream(seed, 4, (batch, nx, nx), "normal", first=first) / np.sqrt(nx)
    H = _stream(seed, 5, (batch, nu, nu), "normal", first=first) / np.sqrt(nu)
    Q = np.eye(nx)[None] + 0.1 * G @ np.transpose(G, (0, 2, 1))
    R = 2.0 * np.eye(nu)[None] + 0.1 * H @ np.transpose(H, (0, 2, 1))
    S = 0.05 * _stream(seed, 6, (batch, nu, nx), "normal", first=first)
    q = 0.1 * _stream(seed, 7, (batch, nx), "normal", first=first)
    r = 0.1 * _stream(seed, 8, (batch, nu), "normal", first=first)
    x0 = 2.5 * _stream(seed, 9, (batch, nx), first=first)
    return dict(A=A, B=B, b=b, Q=Q, R=R, S=S, q=q, r=r, x0=x0,
                lbu=-0.5 * np.ones((batch, nu)), ubu=0.5 * np.ones((batch, nu)))


def lqr_instance_qp(data, i, N):
    """One instance of random_lqr_batch as an AcadosOcpQp (oracle / parity tests)."""
    nx, nu = data["A"].shape[1], data["B"].shape[2]
    qp = AcadosOcpQp(N)
    for k in range(N + 1):
        last = k == N
        qp.set("Q", k, data["Q"][i]); qp.set("q", k, data["q"][i])
        if not last:
            qp.set("R", k, data["R"][i]); qp.set("r", k, data["r"][i]); qp.set("S", k, data["S"][i])
            qp.set("A", k, data["A"][i]); qp.set("B", k, data["B"][i]); qp.set("b", k, data["b"][i])
            qp.set("lbu", k, data["lbu"][i]); qp.set("ubu", k, data["ubu"][i])
        else:
            qp.set("R", k, np.zeros((0, 0))); qp.set("S", k, np.zeros((0, nx))); qp.set("r", k, np.zeros(0))
        if k == 0:
            qp.set("lbx", k, data["x0"][i]); qp.set("ubx", k, data["x0"][i])
            qp.set("idxb", k, np.arange(nu + nx)); qp.set("idxe", k, nu + np.arange(nx))
    qp.make_consistent()
    return qp


def fill_lqr_batch(gb, data, N, xp=None):
    """Pack random_lqr_batch() data into an OcpQpGpuBatch `gb` (same block for every stage).
    `xp`: optional converter (e.g. lambda a: torch.from_numpy(a).cuda()) producing device
    tensors in the blocked C-ABI layout so that no host staging takes place."""
    nx, nu = data["A"].shape[1], data["B"].shape[2]
    gb.set_int("idxe", 0, nu + np.arange(nx))

    def blk(a):
        a = np.asarray(a, dtype=np.float64)
        if a.ndim == 3:
            a = np.transpose(a, (0, 2, 1))
        a = np.ascontiguousarray(a.reshape(a.shape[0], -1))
        return xp(a) if xp is not None else a

    for f in ("A", "B", "b", "Q", "R", "S", "q", "r", "lbu", "ubu"):
        v = blk(data[f])
        if f in ("R", "S", "r", "lbu", "ubu", "A", "B", "b"):
            for k in range(N):
                gb.set(f, k, v)
        else:
            gb.set(f, -1, v)
    x0 = blk(data["x0"])
    gb.set("lbx", 0, x0)
    gb.set("ubx", 0, x0)


def chain_soft_qp(i=0, N=40, nx=24, nu=3, seed=1, ng=4, nsx=4):
    """Configuration C4, instance i: chain-like linearisation data with
    hard input bounds, 4 soft state bounds, 4 soft general rows, ns = 8 slacks
    (idxs_rev: soft x-bounds -> slacks 0..3, general rows -> slacks 4..7), Z = 1e2, z = 1e1,
    x0 as equality bound at stage 0.  Stage 0 carries no soft rows (all 24 states are
    equality rows there; this backend addresses at most 64 inequality sides per stage)."""
    g = np.random.Generator(np.random.Philox(key=[int(seed), 1000 + int(i)]))
    T = -2.0 * np.eye(nx) + np.eye(nx, k=1) + np.eye(nx, k=-1)
    A = np.eye(nx) + 0.1 * T + 0.01 * g.standard_normal((nx, nx))
    B = np.zeros((nx, nu))
    B[nx - nu:, :] = 0.1 * np.eye(nu)
    B += 0.01 * g.standard_normal((nx, nu))
    Q = np.diag(g.uniform(0.1, 10.0, nx))
    R = np.diag(g.uniform(0.01, 1.0, nu))
    x0 = g.uniform(-1.0, 1.0, nx)
    # (ng = 4, nsx = 4: C4; ng = 8: the "ng = 8 chain class" of the round-4 review, 15 rows per stage)
    ixs = nu + (nx // nsx) * np.arange(nsx) + 1  # wall constraint on the "y positions" (every sixth state of the nx = 24 chain)
    qp = AcadosOcpQp(N)
    for k in range(N + 1):
        last = k == N
        nuk = 0 if last else nu
        qp.set("Q", k, Q); qp.set("q", k, Q @ (0.1 * g.standard_normal(nx)))
        qp.set("R", k, R[:nuk, :nuk]); qp.set("r", k, (R @ (0.1 * g.standard_normal(nu)))[:nuk])
        qp.set("S", k, np.zeros((nuk, nx)))
        if not last:
            qp.set("A", k, A); qp.set("B", k, B); qp.set("b", k, 0.01 * g.standard_normal(nx))
        qp.set("lbu", k, -1.0 * np.ones(nuk)); qp.set("ubu", k, 1.0 * np.ones(nuk))
        if k == 0:
            qp.set("lbx", k, x0); qp.set("ubx", k, x0)
            qp.set("idxb", k, np.arange(nu + nx)); qp.set("idxe", k, nu + np.arange(nx))
            continue
        C = g.standard_normal((ng, nx)) / np.sqrt(nx)
        D = g.standard_normal((ng, nuk)) / np.sqrt(nu)
        qp.set("C", k, C); qp.set("D", k, D)
        qp.set("lg", k, -0.5 * np.ones(ng)); qp.set("ug", k, 0.5 * np.ones(ng))
        qp.set("lbx", k, -0.3 * np.ones(nsx)); qp.set("ubx", k, 0.3 * np.ones(nsx))
        qp.set("idxb", k, np.concatenate([np.arange(nuk), (ixs - nu) + nuk]))
        qp.set("idxs_rev", k, np.concatenate([-np.ones(nuk, dtype=int), np.arange(nsx), nsx + np.arange(ng)]))
        ns = nsx + ng
        qp.set("Zl", k, 1e2 * np.ones(ns)); qp.set("Zu", k, 1e2 * np.ones(ns))
        qp.set("zl", k, 1e1 * np.ones(ns)); qp.set("zu", k, 1e1 * np.ones(ns))
        qp.set("lls", k, np.zeros(ns)); qp.set("lus", k, np.zeros(ns))
    qp.make_consistent()
    return qp


def chain_soft_dims(N=40, nx=24, nu=3, ng=4, nsx=4):
    """dims of the C4 class (see chain_soft_qp)"""
    d = AcadosOcpQpDims(N)
    d.nx[:] = nx
    d.nu[:N] = nu
    d.nbu[:N] = nu
    d.nbx[0] = nx
    d.nbx[1:] = nsx
    d.ng[1:] = ng
    d.ns[1:] = nsx + ng
    d.nb[:] = d.nbu + d.nbx
    d.nbxe[0] = nx
    return d


def chain_soft_batch(N=40, nx=24, nu=3, batch=1024, seed=1, first=0):
    """Configuration C4 for whole batches: the same problem class as chain_soft_qp (same structure, same
    distributions) from the chunked counter-based streams of this module, as arrays with leading dim
    `batch`; per-stage fields carry a stage axis.  Global instances [first, first + batch)."""
    ng, nsx = 4, 4
    T = -2.0 * np.eye(nx) + np.eye(nx, k=1) + np.eye(nx, k=-1)
    A = (np.eye(nx) + 0.1 * T)[None] + 0.01 * _stream(seed, 101, (batch, nx, nx), "normal", first)
    B0 = np.zeros((nx, nu))
    B0[nx - nu:, :] = 0.1 * np.eye(nu)
    B = B0[None] + 0.01 * _stream(seed, 102, (batch, nx, nu), "normal", first)
    Qd = 5.05 + 4.95 * _stream(seed, 103, (batch, nx), first=first)      # U(0.1, 10)
    Rd = 0.505 + 0.495 * _stream(seed, 104, (batch, nu), first=first)    # U(0.01, 1)
    x0 = _stream(seed, 105, (batch, nx), first=first)
    q = Qd[:, None, :] * (0.1 * _stream(seed, 106, (batch, N + 1, nx), "normal", first))
    r = Rd[:, None, :] * (0.1 * _stream(seed, 107, (batch, N, nu), "normal", first))
    b = 0.01 * _stream(seed, 108, (batch, N, nx), "normal", first)
    C = _stream(seed, 109, (batch, N, ng, nx), "normal", first) / np.sqrt(nx)      # stages 1..N
    D = _stream(seed, 110, (batch, N, ng, nu), "normal", first) / np.sqrt(nu)      # stages 1..N-1 use it
    return dict(A=A, B=B, Qd=Qd, Rd=Rd, x0=x0, q=q, r=r, b=b, C=C, D=D, ng=ng, nsx=nsx)


def chain_soft_instance_qp(data, i, N):
    """One instance of chain_soft_batch as an AcadosOcpQp (oracle / parity tests)."""
    nx, nu = data["A"].shape[1], data["B"].shape[2]
    ng, nsx = data["ng"], data["nsx"]
    ixs = nu + 6 * np.arange(nsx) + 1
    qp = AcadosOcpQp(N)
    for k in range(N + 1):
        last = k == N
        nuk = 0 if last else nu
        qp.set("Q", k, np.diag(data["Qd"][i])); qp.set("q", k, data["q"][i, k])
        qp.set("R", k, np.diag(data["Rd"][i])[:nuk, :nuk]); qp.set("r", k, data["r"][i, k] if not last else np.zeros(0))
        qp.set("S", k, np.zeros((nuk, nx)))
        if not last:
            qp.set("A", k, data["A"][i]); qp.set("B", k, data["B"][i]); qp.set("b", k, data["b"][i, k])
        qp.set("lbu", k, -1.0 * np.ones(nuk)); qp.set("ubu", k, 1.0 * np.ones(nuk))
        if k == 0:
            qp.set("lbx", k, data["x0"][i]); qp.set("ubx", k, data["x0"][i])
            qp.set("idxb", k, np.arange(nu + nx)); qp.set("idxe", k, nu + np.arange(nx))
            continue
        qp.set("C", k, data["C"][i, k - 1]); qp.set("D", k, data["D"][i, k - 1][:, :nuk])
        qp.set("lg", k, -0.5 * np.ones(ng)); qp.set("ug", k, 0.5 * np.ones(ng))
        qp.set("lbx", k, -0.3 * np.ones(nsx)); qp.set("ubx", k, 0.3 * np.ones(nsx))
        qp.set("idxb", k, np.concatenate([np.arange(nuk), (ixs - nu) + nuk]))
        qp.set("idxs_rev", k, np.concatenate([-np.ones(nuk, dtype=int), np.arange(nsx), nsx + np.arange(ng)]))
        ns = nsx + ng
        qp.set("Zl", k, 1e2 * np.ones(ns)); qp.set("Zu", k, 1e2 * np.ones(ns))
        qp.set("zl", k, 1e1 * np.ones(ns)); qp.set("zu", k, 1e1 * np.ones(ns))
        qp.set("lls", k, np.zeros(ns)); qp.set("lus", k, np.zeros(ns))
    qp.make_consistent()
    return qp


def fill_chain_soft_batch(gb, data, N):
    """Pack chain_soft_batch() data into an OcpQpGpuBatch created from chain_soft_dims(N)."""
    Bn, nx, nu = data["A"].shape[0], data["A"].shape[1], data["B"].shape[2]
    ng, nsx = data["ng"], data["nsx"]
    ns = nsx + ng
    ixs = 6 * np.arange(nsx) + 1
    ones = lambda m, v=1.0: np.full((Bn, m), v)
    colmaj = lambda a: np.ascontiguousarray(np.transpose(a, (0, 2, 1)).reshape(a.shape[0], -1))
    eye_nx = np.eye(nx)[None]
    Q = colmaj(data["Qd"][:, :, None] * eye_nx)
    R = colmaj(data["Rd"][:, :, None] * np.eye(nu)[None])
    A, Bm = colmaj(data["A"]), colmaj(data["B"])
    gb.set_int("idxe", 0, nu + np.arange(nx))
    for k in range(1, N + 1):
        nuk = nu if k < N else 0
        gb.set_int("idxb", k, np.concatenate([np.arange(nuk), ixs + nuk]))
        gb.set_int("idxs_rev", k, np.concatenate([-np.ones(nuk, dtype=int), np.arange(nsx), nsx + np.arange(ng)]))
    for k in range(N + 1):
        last = k == N
        gb.set("Q", k, Q); gb.set("q", k, np.ascontiguousarray(data["q"][:, k]))
        if not last:
            gb.set("R", k, R); gb.set("r", k, np.ascontiguousarray(data["r"][:, k]))
            gb.set("A", k, A); gb.set("B", k, Bm); gb.set("b", k, np.ascontiguousarray(data["b"][:, k]))
            gb.set("lbu", k, ones(nu, -1.0)); gb.set("ubu", k, ones(nu, 1.0))
        if k == 0:
            gb.set("lbx", 0, data["x0"]); gb.set("ubx", 0, data["x0"])
            continue
        gb.set("C", k, colmaj(data["C"][:, k - 1]))
        if not last:
            gb.set("D", k, colmaj(data["D"][:, k - 1]))
        gb.set("lg", k, ones(ng, -0.5)); gb.set("ug", k, ones(ng, 0.5))
        gb.set("lbx", k, ones(nsx, -0.3)); gb.set("ubx", k, ones(nsx, 0.3))
        gb.set("Zl", k, ones(ns, 1e2)); gb.set("Zu", k, ones(ns, 1e2))
        gb.set("zl", k, ones(ns, 1e1)); gb.set("zu", k, ones(ns, 1e1))
        gb.set("lls", k, ones(ns, 0.0)); gb.set("lus", k, ones(ns, 0.0))


C5_CLASSES = [(nx, int(np.ceil(nx / 4)), N) for nx in (4, 12, 24) for N in (20, 50, 100)]


def multiphase_qp(i=0, N=20, nx_a=12, nx_b=4, nu=3, seed=2):
    """C5 "multi-phase" class: the state dimension switches nx_a -> nx_b at k = N/2 through a
    non-square A (nx_b x nx_a); exercises per-stage dims inside one padded kernel shape."""
    g = np.random.Generator(np.random.Philox(key=[int(seed), 2000 + int(i)]))
    ks = N // 2
    qp = AcadosOcpQp(N)
    x0 = g.uniform(-1.0, 1.0, nx_a)
    for k in range(N + 1):
        nx = nx_a if k <= ks else nx_b
        nx1 = nx_a if k + 1 <= ks else nx_b
        last = k == N
        nuk = 0 if last else nu
        G = g.standard_normal((nx, nx)) / np.sqrt(nx)
        qp.set("Q", k, np.eye(nx) + 0.1 * G @ G.T); qp.set("q", k, 0.1 * g.standard_normal(nx))
        qp.set("R", k, 2.0 * np.eye(nuk)); qp.set("r", k, 0.1 * g.standard_normal(nuk))
        qp.set("S", k, 0.05 * g.standard_normal((nuk, nx)))
        if not last:
            A = 0.9 * np.eye(nx1, nx) + 0.02 * g.uniform(-1, 1, (nx1, nx))
            qp.set("A", k, A); qp.set("B", k, 0.3 * g.uniform(-1, 1, (nx1, nuk))); qp.set("b", k, 0.1 * g.uniform(-1, 1, nx1))
        qp.set("lbu", k, -0.5 * np.ones(nuk)); qp.set("ubu", k, 0.5 * np.ones(nuk))
        if k == 0:
            qp.set("lbx", k, x0); qp.set("ubx", k, x0)
            qp.set("idxb", k, np.arange(nuk + nx)); qp.set("idxe", k, nuk + np.arange(nx))
    qp.make_consistent()
    return qp


# ---- C5 "multi-phase" class as a BATCH (nx switching 12 -> 4 at k = N/2 through a non-square A) ----------
# multiphase_qp above draws every stage's blocks anew (one QP object at a time: the small parity tests); the batch form is
# time-invariant inside a phase, like random_lqr_batch, and vectorised over the instances (counter-based streams: instance
# i sees the same numbers whatever the batch size or the rank that generates it).

def multiphase_dims(N=20, nx_a=12, nx_b=4, nu=3):
    d = AcadosOcpQpDims(N)
    ks = N // 2
    d.nx[:ks + 1] = nx_a
    d.nx[ks + 1:] = nx_b
    d.nu[:N] = nu
    d.nbu[:N] = nu
    d.nbx[0] = nx_a
    d.nb[:] = d.nbu + d.nbx
    d.nbxe[0] = nx_a
    return d


def multiphase_batch(N=20, nx_a=12, nx_b=4, nu=3, batch=1024, seed=2, first=0):
    out = {"N": N, "nx_a": nx_a, "nx_b": nx_b, "nu": nu}
    tag = 100
    for ph, (nx1, nx) in (("a", (nx_a, nx_a)), ("s", (nx_b, nx_a)), ("b", (nx_b, nx_b))):
        out["A" + ph] = 0.9 * np.eye(nx1, nx)[None] + 0.02 * _stream(seed, tag + 1, (batch, nx1, nx), first=first)
        out["B" + ph] = 0.3 * _stream(seed, tag + 2, (batch, nx1, nu), first=first)
        out["b" + ph] = 0.1 * _stream(seed, tag + 3, (batch, nx1), first=first)
        tag += 10
    for ph, nx in (("a", nx_a), ("b", nx_b)):
        G = _stream(seed, tag + 1, (batch, nx, nx), "normal", first=first) / np.sqrt(nx)
        out["Q" + ph] = np.eye(nx)[None] + 0.1 * G @ np.transpose(G, (0, 2, 1))
        out["S" + ph] = 0.05 * _stream(seed, tag + 2, (batch, nu, nx), "normal", first=first)
        out["q" + ph] = 0.1 * _stream(seed, tag + 3, (batch, nx), "normal", first=first)
        tag += 10
    out["R"] = np.broadcast_to(2.0 * np.eye(nu), (batch, nu, nu)).copy()
    out["r"] = 0.1 * _stream(seed, tag + 1, (batch, nu), "normal", first=first)
    out["x0"] = _stream(seed, tag + 2, (batch, nx_a), first=first)
    out["lbu"], out["ubu"] = -0.5 * np.ones((batch, nu)), 0.5 * np.ones((batch, nu))
    return out


def _multiphase_stage(data, k):
    """names of the blocks stage k uses: (dynamics phase or None, cost phase)"""
    N, ks = data["N"], data["N"] // 2
    return (None if k == N else "a" if k < ks else "s" if k == ks else "b"), ("a" if k <= ks else "b")


def multiphase_instance_qp(data, i, N=None):
    N = data["N"]
    nu = data["nu"]
    qp = AcadosOcpQp(N)
    for k in range(N + 1):
        dyn, cost = _multiphase_stage(data, k)
        nx = data["Q" + cost].shape[1]
        qp.set("Q", k, data["Q" + cost][i]); qp.set("q", k, data["q" + cost][i])
        if dyn is not None:
            qp.set("R", k, data["R"][i]); qp.set("r", k, data["r"][i]); qp.set("S", k, data["S" + cost][i])
            qp.set("A", k, data["A" + dyn][i]); qp.set("B", k, data["B" + dyn][i]); qp.set("b", k, data["b" + dyn][i])
            qp.set("lbu", k, data["lbu"][i]); qp.set("ubu", k, data["ubu"][i])
        else:
            qp.set("R", k, np.zeros((0, 0))); qp.set("S", k, np.zeros((0, nx))); qp.set("r", k, np.zeros(0))
        if k == 0:
            qp.set("lbx", k, data["x0"][i]); qp.set("ubx", k, data["x0"][i])
            qp.set("idxb", k, np.arange(nu + nx)); qp.set("idxe", k, nu + np.arange(nx))
    qp.make_consistent()
    return qp


def fill_multiphase_batch(gb, data):
    N, nu, nx_a = data["N"], data["nu"], data["nx_a"]
    gb.set_int("idxe", 0, nu + np.arange(nx_a))

    def blk(a):
        a = np.asarray(a, dtype=np.float64)
        if a.ndim == 3:
            a = np.transpose(a, (0, 2, 1))
        return np.ascontiguousarray(a.reshape(a.shape[0], -1))

    cache = {n: blk(v) for n, v in data.items() if isinstance(v, np.ndarray)}
    for k in range(N + 1):
        dyn, cost = _multiphase_stage(data, k)
        gb.set("Q", k, cache["Q" + cost]); gb.set("q", k, cache["q" + cost])
        if dyn is not None:
            gb.set("R", k, cache["R"]); gb.set("r", k, cache["r"]); gb.set("S", k, cache["S" + cost])
            gb.set("A", k, cache["A" + dyn]); gb.set("B", k, cache["B" + dyn]); gb.set("b", k, cache["b" + dyn])
            gb.set("lbu", k, cache["lbu"]); gb.set("ubu", k, cache["ubu"])
    gb.set("lbx", 0, cache["x0"]); gb.set("ubx", 0, cache["x0"])
