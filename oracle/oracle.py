"""ctypes front-end of the CPU oracle (oracle/ocp_qp_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package acados_amd.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# The oracle stops where the tolerances it is given say (the reference's semantics, ocp_qp_hpipm.c:104-107), and so does
# the product by default.  The product has an OPT-IN tighter exit for soft-constrained classes (option tol_comp_soft_scale
# < 1, gpu_batch.hip effective_opts); a test that switches it on passes the same scale here (soft_scale=...) so that the
# two sides are compared at the same effective tolerance.
SOFT_COMP_SCALE = 1.0


def soft_opts(opts, has_slack, soft_scale=None):
    """copy of `opts` with the product's exit rule for soft-constrained classes applied"""
    o = OqpOpts()
    C.memmove(C.byref(o), C.byref(opts), C.sizeof(OqpOpts))
    sc = SOFT_COMP_SCALE if soft_scale is None else soft_scale
    if has_slack and 0.0 < sc < 1.0:
        o.tol_comp = opts.tol_comp * sc
    return o


class OqpOpts(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("mu0", "tol_stat", "tol_eq", "tol_ineq", "tol_comp", "alpha_min",
                                           "tau_min", "lam_min", "t_min", "reg_prim")] + \
               [(n, C.c_int) for n in ("iter_max", "pred_corr", "cond_pred_corr", "warm_start", "print_level", "t0_init")]


def build(force=False):
    so = os.path.join(_HERE, "libocp_qp_oracle.so")
    src = os.path.join(_HERE, "ocp_qp_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.oqp_create.restype = C.c_void_p
        L.oqp_create.argtypes = [C.c_int] + [C.POINTER(C.c_int)] * 6
        L.oqp_free.argtypes = [C.c_void_p]
        L.oqp_clone.restype = C.c_void_p
        L.oqp_clone.argtypes = [C.c_void_p]
        L.oqp_set.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p]
        L.oqp_set_nbxe.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.oqp_solve.argtypes = [C.c_void_p, C.POINTER(OqpOpts)]
        L.oqp_get.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p]
        L.oqp_get_iter.argtypes = [C.c_void_p]
        L.oqp_get_stat.argtypes = [C.c_void_p]
        L.oqp_get_stat.restype = C.POINTER(C.c_double)
        L.oqp_compute_t.argtypes = [C.c_void_p]
        L.oqp_refactor.argtypes = [C.c_void_p, C.POINTER(OqpOpts)]
        L.oqp_res_nrm_inf.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.oqp_opts_default.argtypes = [C.POINTER(OqpOpts)]
        L.oqp_solve_batch.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(OqpOpts), C.POINTER(C.c_int), C.c_int]
        _LIB = L
    return _LIB


def default_opts(**kw):
    o = OqpOpts()
    lib().oqp_opts_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def _ia(a):
    return np.ascontiguousarray(a, dtype=np.int32)


_SET_FIELDS = ("A", "B", "b", "Q", "S", "R", "q", "r", "idxb", "lbx", "ubx", "lbu", "ubu", "lls", "lus",
               "lbx_mask", "ubx_mask", "lbu_mask", "ubu_mask", "lls_mask", "lus_mask", "ug_mask", "lg_mask",
               "C", "D", "lg", "ug", "Zl", "Zu", "zl", "zu", "idxe", "idxs_rev")
_INT = ("idxb", "idxe", "idxs_rev")
_DYN = ("A", "B", "b")


class OracleQp:
    """One QP held by the C oracle; `qp` is an acados_amd.AcadosOcpQp-like object
    (per-stage lists named as in acados_ocp_qp_solver.py:277-292)."""

    def __init__(self, qp):
        L = lib()
        d = qp.dims
        self.qp, self.N = qp, qp.N
        arrs = [_ia(getattr(d, n)) for n in ("nx", "nu", "nbx", "nbu", "ng", "ns")]
        self.h = C.c_void_p(L.oqp_create(qp.N, *[a.ctypes.data_as(C.POINTER(C.c_int)) for a in arrs]))
        for i in range(qp.N + 1):
            L.oqp_set_nbxe(self.h, i, int(d.nbxe[i]))
            for name in _SET_FIELDS:
                if i == qp.N and name in _DYN:
                    continue
                self.set(name, i, getattr(qp, name)[i])

    def set(self, name, stage, value):
        if name in _INT:
            v = np.ravel(np.asarray(value).astype(np.int32), order="F")
        else:
            v = np.ravel(np.asarray(value, dtype=np.float64), order="F")
        v = np.ascontiguousarray(v)
        if v.size:
            assert lib().oqp_set(self.h, name.encode(), stage, v.ctypes.data_as(C.c_void_p)) == 0

    @property
    def has_slack(self):
        return bool(np.any(np.asarray(self.qp.dims.ns) > 0))

    def solve(self, opts=None, soft_scale=None, **kw):
        self.opts = soft_opts(opts if opts is not None else default_opts(**kw), self.has_slack, soft_scale)
        self.status = lib().oqp_solve(self.h, C.byref(self.opts))
        return self.status

    @property
    def iter(self):
        return lib().oqp_get_iter(self.h)

    def stat(self):
        p = lib().oqp_get_stat(self.h)
        return np.ctypeslib.as_array(p, shape=((self.iter + 1), 20)).copy()

    def _dim(self, field, k):
        d = self.qp.dims
        if field == "x":
            return d.nx[k]
        if field == "u":
            return d.nu[k]
        if field == "pi":
            return d.nx[k + 1] if k < self.N else 0
        if field in ("sl", "su"):
            return d.ns[k]
        if field == "ric_L":
            return (d.nx[k] + d.nu[k]) ** 2
        if field == "ric_l":
            return d.nx[k] + d.nu[k]
        return 2 * (d.nb[k] + d.ng[k] + d.ns[k])

    def get(self, k, field):
        out = np.zeros(int(self._dim(field, k)))
        if out.size:
            lib().oqp_get(self.h, field.encode(), k, out.ctypes.data_as(C.c_void_p))
        return out

    def res(self):
        r = (C.c_double * 4)()
        lib().oqp_res_nrm_inf(self.h, r)
        return np.array(r[:])

    def refactor(self):
        lib().oqp_refactor(self.h, C.byref(self.opts))

    def compute_t(self):
        lib().oqp_compute_t(self.h)

    def __del__(self):
        try:
            lib().oqp_free(self.h)
        except Exception:
            pass


def solve_batch_handles(handles, opts=None, nthreads=1):
    """OpenMP batch solve over raw handles (OracleQp.h or clones of it)"""
    n = len(handles)
    hs = (C.c_void_p * n)(*handles)
    st = (C.c_int * n)()
    o = opts if opts is not None else default_opts()
    lib().oqp_solve_batch(hs, n, C.byref(o), st, nthreads)
    return np.array(st[:])


def clone_handle(h):
    return C.c_void_p(lib().oqp_clone(h))


def free_handle(h):
    lib().oqp_free(h)


def solve_batch(oqps, opts=None, nthreads=1):
    """OpenMP batch solve (acados_solver.in.c:3222-3243 idiom). Returns status array."""
    n = len(oqps)
    hs = (C.c_void_p * n)(*[o.h for o in oqps])
    st = (C.c_int * n)()
    o = opts if opts is not None else default_opts()
    lib().oqp_solve_batch(hs, n, C.byref(o), st, nthreads)
    return np.array(st[:])
