"""Device-resident batch of structurally identical OCP-QPs (thin ctypes layer over
include/acados_amd/ocp_qp_gpu_batch.h).

This is the GPU counterpart of the reference's batch idiom, `AcadosOcpBatchSolver`
(interfaces/acados_template/acados_template/acados_ocp_batch_solver.py:41-) /
`<name>_acados_batch_solve` (c_templates_tera/acados_solver.in.c:3222-3243), restricted to
the QP-solve part: N_batch QPs sharing dims / idxb / idxs_rev / idxe.
"""
import ctypes as C

import numpy as np

from . import _lib

INT_FIELDS = ("idxb", "idxbx", "idxbu", "idxs_rev", "idxe")
DATA_FIELDS = ("A", "B", "b", "Q", "S", "R", "q", "r", "lbx", "ubx", "lbu", "ubu", "C", "D", "lg", "ug",
               "Zl", "Zu", "zl", "zu", "lls", "lus", "lbx_mask", "ubx_mask", "lbu_mask", "ubu_mask",
               "lg_mask", "ug_mask", "lls_mask", "lus_mask")
DYN_FIELDS = ("A", "B", "b")


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


class OcpQpGpuBatch:
    def __init__(self, dims, n_batch, device=-1, _clib=None):
        """dims: object with N and arrays nx nu nbx nbu ng ns (length N+1)."""
        self._L = _clib if _clib is not None else _lib.lib()
        self.dims, self.N, self.n_batch = dims, int(dims.N), int(n_batch)
        arrs = [np.ascontiguousarray(getattr(dims, n), dtype=np.int32) for n in ("nx", "nu", "nbx", "nbu", "ng", "ns")]
        self._nbxe = np.zeros(self.N + 1, dtype=int)
        self._h = self._L.ocp_qp_gpu_batch_create(self.N, *[_ip(a) for a in arrs], self.n_batch, int(device))
        if not self._h:
            raise RuntimeError("acados_amd: ocp_qp_gpu_batch_create failed (no GPU, or unsupported shape)")
        self._h = C.c_void_p(self._h)

    # -- structure / data -------------------------------------------------
    def set_int(self, field, stage, value):
        v = np.ascontiguousarray(np.asarray(value).reshape(-1), dtype=np.int32)
        if field == "idxe":
            self._nbxe[stage] = v.size
        if self._L.ocp_qp_gpu_batch_set_int(self._h, field.encode(), int(stage), _ip(v), int(v.size)) != 0:
            raise ValueError(f"ocp_qp_gpu_batch_set_int({field}, {stage}) failed")

    def set(self, field, stage, value):
        """value: array [n_batch, ...] (column-major flattening of matrices is done here:
        pass matrices as [n_batch, rows, cols]); or a torch CUDA tensor already in the
        blocked C-ABI layout [n_batch, len] (zero-copy, device pointer)."""
        if hasattr(value, "data_ptr") and getattr(value, "is_cuda", False):
            assert value.is_contiguous() and str(value.dtype) == "torch.float64" and value.shape[0] == self.n_batch
            rc = self._L.ocp_qp_gpu_batch_set(self._h, field.encode(), int(stage), C.c_void_p(value.data_ptr()), 1)
        else:
            a = np.asarray(value, dtype=np.float64)
            assert a.shape[0] == self.n_batch, f"{field}: leading dim {a.shape[0]} != n_batch {self.n_batch}"
            if a.ndim == 3:
                a = np.transpose(a, (0, 2, 1))  # column-major per instance
            a = np.ascontiguousarray(a.reshape(self.n_batch, -1))
            if a.shape[1] == 0:
                return
            rc = self._L.ocp_qp_gpu_batch_set(self._h, field.encode(), int(stage), a.ctypes.data_as(C.c_void_p), 0)
        if rc != 0:
            raise ValueError(f"ocp_qp_gpu_batch_set({field}, {stage}) failed")

    def opts_set(self, field, value):
        if isinstance(value, str):
            p = C.c_char_p(value.encode())
            rc = self._L.ocp_qp_gpu_batch_opts_set(self._h, field.encode(), p)
        elif isinstance(value, (bool, int, np.integer)):
            rc = self._L.ocp_qp_gpu_batch_opts_set(self._h, field.encode(), C.byref(C.c_int(int(value))))
        else:
            rc = self._L.ocp_qp_gpu_batch_opts_set(self._h, field.encode(), C.byref(C.c_double(float(value))))
        if rc != 0:
            raise ValueError(f"unknown option {field}")

    # -- solve / results ---------------------------------------------------
    def solve(self):
        """returns the number of instances with non-zero status"""
        return self._L.ocp_qp_gpu_batch_solve(self._h)

    def condense_lhs(self):
        """RTI preparation phase: matrix part of the partial condensing (no-op for cond_N == N)"""
        return self._L.ocp_qp_gpu_batch_condense_lhs(self._h)

    def condense_rhs_and_solve(self):
        """RTI feedback phase: vector part of the condensing + solve + expansion"""
        return self._L.ocp_qp_gpu_batch_condense_rhs_and_solve(self._h)

    # -- condensing-only boundary (interfaces/acados_c/condensing_interface.h:73-75) ---------------------------
    def condense(self):
        """partial condensing (opts cond_N / cond_block_size) without the solve: returns the condensed QP as a batch
        object owned by this one (None when the QP is not condensed)"""
        h = self._L.ocp_qp_gpu_batch_condense(self._h)
        if not h:
            return None
        c = object.__new__(OcpQpGpuBatch)
        c._L, c._h, c._owner = self._L, C.c_void_p(h), self   # non-owning view; keeps the parent alive
        one = np.zeros(1, dtype=np.int32)
        self._L.ocp_qp_gpu_batch_get_dims(c._h, b"N", _ip(one))
        c.N, c.n_batch = int(one[0]), self.n_batch
        from .ocp_qp import AcadosOcpQpDims
        c.dims = AcadosOcpQpDims(c.N)
        for f in ("nx", "nu", "nbx", "nbu", "nb", "ng", "ns", "nbxe"):
            v = np.zeros(c.N + 1, dtype=np.int32)
            self._L.ocp_qp_gpu_batch_get_dims(c._h, f.encode(), _ip(v))
            getattr(c.dims, f)[:] = v
        c._nbxe = np.array(c.dims.nbxe, dtype=int)
        return c

    def expand(self):
        """the condensed batch's current solution mapped back to the stages of this batch"""
        if self._L.ocp_qp_gpu_batch_expand(self._h) != 0:
            raise RuntimeError("ocp_qp_gpu_batch_expand: nothing condensed")

    def get_int(self, field, stage):
        d = self.dims
        out = np.zeros(int(d.nbx[stage] + d.nbu[stage] + d.ng[stage]) + 1, dtype=np.int32)
        n = self._L.ocp_qp_gpu_batch_get_int(self._h, field.encode(), int(stage), _ip(out))
        if n < 0:
            raise ValueError(field)
        return out[:n].astype(int)

    def to_qp(self, i):
        """instance i of the batch as an AcadosOcpQp (data read back from the device)"""
        from .ocp_qp import AcadosOcpQp
        qp = AcadosOcpQp(self.N)
        d = self.dims
        for k in range(self.N + 1):
            nx, nu = int(d.nx[k]), int(d.nu[k])
            nb, ng, ns = int(d.nbx[k] + d.nbu[k]), int(d.ng[k]), int(d.ns[k])
            if nb:
                qp.set("idxb", k, self.get_int("idxb", k))
            if ns:
                qp.set("idxs_rev", k, self.get_int("idxs_rev", k))
            if int(self._nbxe[k]):
                qp.set("idxe", k, self.get_int("idxe", k))
            for f in DATA_FIELDS:
                if k == self.N and f in DYN_FIELDS:
                    continue
                v = self.get(f, k)[i]
                if f in ("A", "B", "Q", "R", "S", "C", "D"):
                    rows = {"A": int(d.nx[min(k + 1, self.N)]), "B": int(d.nx[min(k + 1, self.N)]), "Q": nx, "R": nu, "S": nu,
                            "C": ng, "D": ng}[f]
                    v = v.reshape(-1, rows).T if rows else v.reshape(0, 0)
                    if v.size == 0:
                        continue
                elif v.size == 0:
                    continue
                qp.set(f, k, v)
        qp.make_consistent()
        return qp

    def _len(self, field, k):
        d = self.dims
        if field.startswith("sens_"):
            field = field[5:]
        nx, nu, ng, ns = int(d.nx[k]), int(d.nu[k]), int(d.ng[k]), int(d.ns[k])
        nx1 = int(d.nx[k + 1]) if k < self.N else 0
        data_len = {"A": nx1 * nx, "B": nx1 * nu, "b": nx1, "Q": nx * nx, "R": nu * nu, "S": nu * nx, "q": nx, "r": nu,
                    "lbx": int(d.nbx[k]), "ubx": int(d.nbx[k]), "lbu": int(d.nbu[k]), "ubu": int(d.nbu[k]),
                    "lbx_mask": int(d.nbx[k]), "ubx_mask": int(d.nbx[k]), "lbu_mask": int(d.nbu[k]), "ubu_mask": int(d.nbu[k]),
                    "C": ng * nx, "D": ng * nu, "lg": ng, "ug": ng, "lg_mask": ng, "ug_mask": ng,
                    "Zl": ns, "Zu": ns, "zl": ns, "zu": ns, "lls": ns, "lus": ns, "lls_mask": ns, "lus_mask": ns}
        if field in data_len:
            return data_len[field]
        if field == "x":
            return int(d.nx[k])
        if field == "u":
            return int(d.nu[k])
        if field == "pi":
            return int(d.nx[k + 1])
        if field in ("sl", "su"):
            return int(d.ns[k])
        if field in ("lam", "t"):
            return 2 * int(d.nbx[k] + d.nbu[k] + d.ng[k] + d.ns[k])
        if field == "ric_L":
            return int(d.nx[k] + d.nu[k]) ** 2
        if field in ("ric_l", "res_g"):
            return int(d.nx[k] + d.nu[k])
        if field == "res_gs":
            return 2 * int(d.ns[k])
        if field == "res_b":
            return int(d.nx[k + 1]) if k < self.N else 0
        if field in ("res_d", "res_m"):
            return 2 * int(d.nbx[k] + d.nbu[k] + d.ng[k] + d.ns[k])
        raise ValueError(field)

    def get(self, field, stage):
        n = self._len(field, stage)
        out = np.zeros((self.n_batch, n))
        if n:
            if self._L.ocp_qp_gpu_batch_get(self._h, field.encode(), int(stage), out.ctypes.data_as(C.c_void_p), 0) != 0:
                raise ValueError(f"ocp_qp_gpu_batch_get({field}, {stage}) failed")
        return out

    def riccati(self, stage):
        """P, p, K, k, Lr of the last factorisation per instance, conventions of the reference's getters
        (ocp_qp_hpipm.c:417-478; ocp_nlp_ddp.c:373-377 uses u = K x + k): arrays with leading dim n_batch"""
        d = self.dims
        nu, nx = int(d.nu[stage]), int(d.nx[stage])
        nv = nu + nx
        L = self.get("ric_L", stage).reshape(self.n_batch, nv, nv).transpose(0, 2, 1)  # column-major blocks
        l = self.get("ric_l", stage)
        Lr, Ls, Lx = L[:, :nu, :nu], L[:, nu:, :nu], L[:, nu:, nu:]
        P = Lx @ Lx.transpose(0, 2, 1)
        p = np.einsum("bij,bj->bi", Lx, l[:, nu:])
        if nu:
            LrT = Lr.transpose(0, 2, 1)
            K = -np.linalg.solve(LrT, Ls.transpose(0, 2, 1))
            k = -np.linalg.solve(LrT, l[:, :nu, None])[:, :, 0]
        else:
            K, k = np.zeros((self.n_batch, 0, nx)), np.zeros((self.n_batch, 0))
        return {"P": P, "p": p, "K": K, "k": k, "Lr": Lr}

    # -- solution sensitivities ------------------------------------------------
    def sens_set(self, field, stage, value):
        """seed = derivative of the problem data w.r.t. a parameter: seed_q seed_r seed_b seed_lbu seed_ubu seed_lbx
        seed_ubx seed_lg seed_ug (natural-sign bounds); value [n_batch, len].  The first seed after a solve opens a
        new seed set (all other seeds zero)."""
        v = np.ascontiguousarray(np.asarray(value, dtype=np.float64).reshape(self.n_batch, -1))
        if self._L.ocp_qp_gpu_batch_sens_set(self._h, field.encode(), int(stage), v.ctypes.data_as(C.c_void_p)) != 0:
            raise ValueError(f"ocp_qp_gpu_batch_sens_set({field}, {stage}) failed")

    def sens_solve(self):
        """d solution / d parameter for the seeds set since the last solve; read with get("sens_x" / "sens_u" /
        "sens_pi" / "sens_lam" / "sens_t" / "sens_sl" / "sens_su", stage)"""
        if self._L.ocp_qp_gpu_batch_sens_solve(self._h) != 0:
            raise RuntimeError("ocp_qp_gpu_batch_sens_solve failed")

    # -- KKT residuals of whatever (data, iterate) is in HBM (ocp_qp_res_compute / _nrm_inf) ----------------
    def res_compute(self):
        """residual vectors of the current iterate, one launch of a kernel independent of the IPM sweeps; read them with
        get("res_g" / "res_gs" / "res_b" / "res_d" / "res_m", stage).  Returns the inf-norms [n_batch, 4]
        (stationarity, dynamics, inequalities, complementarity) as ocp_qp_inf_norm_residuals does."""
        if self._L.ocp_qp_gpu_batch_res_compute(self._h) != 0:
            raise RuntimeError("ocp_qp_gpu_batch_res_compute failed")
        out = np.zeros((self.n_batch, 4))
        self._L.ocp_qp_gpu_batch_res_nrm_inf(self._h, out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def info(self, field):
        out = np.zeros(self.n_batch, dtype=np.int32 if field in ("status", "iter") else np.float64)
        if self._L.ocp_qp_gpu_batch_get_info(self._h, field.encode(), out.ctypes.data_as(C.c_void_p)) != 0:
            raise ValueError(field)
        return out

    def stat(self, inst=0, max_rows=1024):
        buf = np.zeros((max_rows, 20))
        rows = self._L.ocp_qp_gpu_batch_get_stat(self._h, int(inst), buf.ctypes.data_as(C.POINTER(C.c_double)), max_rows)
        if rows < 0:
            raise ValueError("statistics not available for this instance")
        return buf[:rows]

    def scalar(self, field):
        return self._L.ocp_qp_gpu_batch_get_scalar(self._h, field.encode())

    @property
    def bytes(self):
        return self._L.ocp_qp_gpu_batch_bytes(self._h)

    @property
    def stream(self):
        return self._L.ocp_qp_gpu_batch_stream(self._h)

    @property
    def kernel_name(self):
        return self._L.ocp_qp_gpu_batch_kernel_name(self._h).decode()

    def condensed_kernel_name(self):
        """kernel instantiation serving the condensed batch of the last partially condensed solve (None: not condensed)"""
        h = self._L.ocp_qp_gpu_batch_condensed(self._h)
        return self._L.ocp_qp_gpu_batch_kernel_name(C.c_void_p(h)).decode() if h else None

    def condensed_scalar(self, field):
        """`scalar(field)` of the condensed batch of the last partially condensed solve (None: not condensed)"""
        h = self._L.ocp_qp_gpu_batch_condensed(self._h)
        return self._L.ocp_qp_gpu_batch_get_scalar(C.c_void_p(h), field.encode()) if h else None

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self, "_owner", None) is None:   # a condensed view belongs to its parent
                self._L.ocp_qp_gpu_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- convenience: pack a list of AcadosOcpQp ----------------------------
    @classmethod
    def from_qps(cls, qps, device=-1, _clib=None):
        """qps: list of AcadosOcpQp with identical structure."""
        q0 = qps[0]
        sig = q0.dims.signature()
        for q in qps:
            if q.dims.signature() != sig:
                raise ValueError("all QPs of a batch must share dims (bucket by dims.signature())")
        b = cls(q0.dims, len(qps), device=device, _clib=_clib)
        N = q0.N
        for k in range(N + 1):
            for f in ("idxb", "idxs_rev", "idxe"):
                ref = np.asarray(getattr(q0, f)[k]).astype(int)
                for q in qps:
                    if not np.array_equal(np.asarray(getattr(q, f)[k]).astype(int), ref):
                        raise ValueError(f"{f} differs inside the batch at stage {k}")
                if ref.size or f == "idxe":
                    b.set_int(f, k, ref)
        for k in range(N + 1):
            for f in DATA_FIELDS:
                if k == N and f in DYN_FIELDS:
                    continue
                a = np.stack([np.asarray(getattr(q, f)[k], dtype=float) for q in qps])
                if a.size:
                    b.set(f, k, a)
        return b
