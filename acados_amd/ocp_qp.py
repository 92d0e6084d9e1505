"""Host-side OCP-QP container (one problem instance, NumPy arrays).

Mirrors the reference's pure-Python QP description class
(interfaces/acados_template/acados_template/acados_ocp_qp.py:9-21 dims,
:24-45 formulation, :255-268 set(), :292-379 make_consistent(), :381-435 JSON
loading) so that a QP dumped by acados (`dump_last_qp_to_json`,
interfaces/acados_c/ocp_nlp_interface.c:2213-2372) can be loaded here unchanged:
same field names, per-stage lists, natural-sign bounds, zero-padded stage keys.

    min  sum_k 1/2 [u;x]'[R S; S' Q][u;x] + [r;q]'[u;x]
               + 1/2 sl'Zl sl + zl'sl + 1/2 su'Zu su + zu'su
    s.t. x_{k+1} = A x + B u + b
         [lbu;lbx;lg] <= [[u;x][idxb]; C x + D u] + sl[idxs_rev]
         [[u;x][idxb]; C x + D u] - su[idxs_rev] <= [ubu;ubx;ug]
         sl >= lls, su >= lus ; *_mask == 0 switches a side off ; idxe lists
         positions (in the bound list) of box rows that are equalities.
"""
import json

import numpy as np

DYNAMICS_FIELDS = ("A", "B", "b")
COST_FIELDS = ("Q", "R", "S", "q", "r", "zl", "zu", "Zl", "Zu")
CONSTRAINT_FIELDS = (
    "idxb", "lbu", "ubu", "lbx", "ubx", "C", "D", "lg", "ug", "idxs_rev", "lls", "lus",
    "lbu_mask", "ubu_mask", "lbx_mask", "ubx_mask", "lg_mask", "ug_mask", "lls_mask",
    "lus_mask", "idxe",
)
ALL_FIELDS = DYNAMICS_FIELDS + COST_FIELDS + CONSTRAINT_FIELDS
MATRIX_FIELDS = ("A", "B", "Q", "R", "S", "C", "D")
INT_FIELDS = ("idxb", "idxs_rev", "idxe")
MASK_OF = {"lbu_mask": "lbu", "ubu_mask": "ubu", "lbx_mask": "lbx", "ubx_mask": "ubx",
           "lg_mask": "lg", "ug_mask": "ug", "lls_mask": "lls", "lus_mask": "lus"}


class AcadosOcpQpDims:
    """Per-stage dimensions, arrays of length N+1 (acados_ocp_qp.py:9-21)."""

    def __init__(self, N: int):
        self.N = N
        for name in ("nx", "nu", "nbx", "nbu", "nb", "ng", "ns", "nbxe"):
            setattr(self, name, np.zeros((N + 1,), dtype=int))

    def signature(self):
        """Hashable shape-class key (used to bucket mixed batches)."""
        return (self.N,) + tuple(tuple(int(v) for v in getattr(self, n))
                                 for n in ("nx", "nu", "nbx", "nbu", "ng", "ns", "nbxe"))


class AcadosOcpQp:
    def __init__(self, N: int):
        self.__N = N
        self._f = {name: [None] * (N + 1 if name not in DYNAMICS_FIELDS else N) for name in ALL_FIELDS}
        self.__dims = AcadosOcpQpDims(N)
        self.dynamics_fields = set(DYNAMICS_FIELDS)
        self.cost_fields = set(COST_FIELDS)
        self.constraint_fields = set(CONSTRAINT_FIELDS)
        self.all_fields = set(ALL_FIELDS)

    @property
    def N(self) -> int:
        return self.__N

    @property
    def dims(self) -> AcadosOcpQpDims:
        return self.__dims

    def __getattr__(self, name):
        f = self.__dict__.get("_f")
        if f is not None and name in f:
            return f[name]
        raise AttributeError(name)

    def set(self, field_name: str, stage: int, value):
        if stage < 0 or stage > self.N:
            raise ValueError(f"Stage {stage} is out of bounds for N={self.N}.")
        if field_name in DYNAMICS_FIELDS and stage == self.N:
            raise ValueError(f"Dynamics fields cannot be set at terminal stage N={self.N}.")
        if field_name not in self._f:
            raise ValueError(f"Field name {field_name} is not recognized.")
        if field_name in MATRIX_FIELDS:
            a = np.asarray(value, dtype=float)
            if a.ndim != 2:
                a = a.reshape((a.shape[0], -1)) if a.ndim == 1 and a.size else a.reshape((0, 0))
        elif field_name in INT_FIELDS:
            a = np.asarray(value).astype(int).reshape(-1)
        else:
            a = np.asarray(value, dtype=float).reshape(-1)
        self._f[field_name][stage] = a

    def has_slacks(self) -> bool:
        return bool(np.any(self.dims.ns > 0))

    def has_masks(self) -> bool:
        return any(np.any(m == 0.0) for name in MASK_OF for m in self._f[name] if m is not None)

    def make_consistent(self, assert_dims: bool = True):
        d = self.__dims
        N = self.N
        for i in range(N + 1):
            nx = self.Q[i].shape[0] if self.Q[i] is not None else 0
            nu = self.R[i].shape[0] if self.R[i] is not None and self.R[i].size else 0
            d.nx[i], d.nu[i] = nx, nu
            for name, arr in self._f.items():
                if i == N and name in DYNAMICS_FIELDS:
                    continue
                if arr[i] is None:
                    arr[i] = np.zeros((0, 0)) if name in MATRIX_FIELDS else np.zeros((0,), dtype=int if name in INT_FIELDS else float)
            d.nbx[i], d.nbu[i] = len(self.lbx[i]), len(self.lbu[i])
            d.nb[i] = d.nbx[i] + d.nbu[i]
            d.ng[i], d.ns[i] = len(self.lg[i]), len(self.lls[i])
            d.nbxe[i] = len(self.idxe[i])
            # defaults the reference's JSON always carries explicitly
            for mname, bname in MASK_OF.items():
                if len(self._f[mname][i]) != len(self._f[bname][i]):
                    self._f[mname][i] = np.ones(len(self._f[bname][i]))
            if len(self.idxs_rev[i]) != d.nb[i] + d.ng[i]:
                self._f["idxs_rev"][i] = -np.ones(d.nb[i] + d.ng[i], dtype=int)
            if len(self.idxb[i]) != d.nb[i]:
                self._f["idxb"][i] = np.concatenate([np.arange(d.nbu[i]), nu + np.arange(d.nbx[i])]).astype(int)
            if self.S[i].size == 0:
                self._f["S"][i] = np.zeros((nu, nx))
            if self.R[i].size == 0:
                self._f["R"][i] = np.zeros((nu, nu))
            if len(self.r[i]) != nu:
                self._f["r"][i] = np.zeros(nu)
            if d.ng[i] > 0 and self.D[i].size == 0:
                self._f["D"][i] = np.zeros((d.ng[i], nu))
            if d.ng[i] == 0:
                self._f["C"][i] = np.zeros((0, nx))
                self._f["D"][i] = np.zeros((0, nu))
            for name in ("zl", "zu", "Zl", "Zu", "lus"):
                if len(self._f[name][i]) != d.ns[i]:
                    self._f[name][i] = np.zeros(d.ns[i])
            if assert_dims:
                assert self.Q[i].shape == (nx, nx) and self.q[i].shape == (nx,), f"Q/q dims at stage {i}"
                assert self.R[i].shape == (nu, nu) and self.S[i].shape == (nu, nx), f"R/S dims at stage {i}"
                assert self.C[i].shape == (d.ng[i], nx) and self.D[i].shape == (d.ng[i], nu), f"C/D dims at stage {i}"
                assert len(self.ubx[i]) == d.nbx[i] and len(self.ubu[i]) == d.nbu[i] and len(self.ug[i]) == d.ng[i]
                for e in self.idxe[i]:
                    if e < d.nbu[i] or e >= d.nb[i]:
                        raise ValueError(f"Equality constraint index {e} at stage {i} does not correspond to x bound.")
        if assert_dims:
            for i in range(N):
                nx1 = d.nx[i + 1]
                assert self.A[i].shape == (nx1, d.nx[i]), f"A dims at stage {i}"
                if d.nu[i] == 0:
                    self._f["B"][i] = np.zeros((nx1, 0))
                assert self.B[i].shape == (nx1, d.nu[i]) and self.b[i].shape == (nx1,), f"B/b dims at stage {i}"

    @classmethod
    def from_dict(cls, qp_dict) -> "AcadosOcpQp":
        N = len([k for k in qp_dict if k.startswith("Q_")]) - 1
        width = len(str(N + 1))
        bad = [k for k in qp_dict if (s := k.split("_")[-1]).isdigit() and len(s) != width]
        if bad:
            raise ValueError(f"Keys {bad} do not follow the expected format with zero-padded stage indices.")
        qp = cls(N)
        for name in ALL_FIELDS:
            for i in range(N if name in DYNAMICS_FIELDS else N + 1):
                key = f"{name}_{i:0{width}d}"
                if key in qp_dict:
                    val = qp_dict[key]
                    if name in MATRIX_FIELDS:
                        val = np.asarray(val, dtype=float)
                        if val.ndim != 2:
                            val = val.reshape((0, 0)) if val.size == 0 else np.atleast_2d(val)
                    qp.set(name, i, val)
        qp.make_consistent()
        return qp

    @classmethod
    def from_json(cls, json_file_path: str = None, json_data: dict = None) -> "AcadosOcpQp":
        if json_data is None:
            if json_file_path is None:
                raise ValueError("Either json_file_path or json_data must be provided to from_json.")
            with open(json_file_path, "r") as f:
                json_data = json.load(f)
        return cls.from_dict(json_data)

    def to_json(self, json_file_path: str):
        """write the zero-padded-key JSON format of `dump_last_qp_to_json`
        (interfaces/acados_c/ocp_nlp_interface.c:2213-2372) that from_json reads back"""
        with open(json_file_path, "w") as f:
            json.dump(self.to_dict(), f)

    def to_dict(self) -> dict:
        width = len(str(self.N + 1))
        out = {}
        for name, arr in self._f.items():
            for i, a in enumerate(arr):
                if a is not None:
                    out[f"{name}_{i:0{width}d}"] = np.asarray(a).tolist()
        return out
