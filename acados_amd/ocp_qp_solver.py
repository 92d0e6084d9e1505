"""ctypes driver of the acados-shaped C API (include/acados_amd/ocp_qp_interface.h).

Same call sequence, method names and semantics as the reference's `AcadosOcpQpSolver`
(interfaces/acados_template/acados_template/acados_ocp_qp_solver.py:69-168 ctor,
:218-276 options, :277-315 data, :318-326 solve, :343-397 get incl. the stage-0 unique-dual
fold, :431-476 statistics), pointed at libacados_amd_qp.so instead of libacados.so, plus a
batch solver whose `solve()` is one device batch instead of an OpenMP loop.
"""
import ctypes as C

import numpy as np

from . import _lib
from .ocp_qp import AcadosOcpQp
from .ocp_qp_options import AcadosOcpQpOptions

_FIELDS = ("A", "B", "b", "Q", "S", "R", "q", "r", "idxb", "lbx", "ubx", "lbu", "ubu", "lls", "lus",
           "lbx_mask", "ubx_mask", "lbu_mask", "ubu_mask", "lls_mask", "lus_mask", "ug_mask", "lg_mask",
           "C", "D", "lg", "ug", "Zl", "Zu", "zl", "zu", "idxe", "idxs_rev")
_INT = ("idxb", "idxe", "idxs_rev")
_OPT_FIELDS = ("tol_stat", "tol_eq", "tol_ineq", "tol_comp", "iter_max", "cond_N", "cond_block_size", "warm_start",
               "cond_ric_alg", "ric_alg", "mu0", "t0_init", "print_level", "hpipm_mode", "tau_min", "t0_min", "lam0_min",
               "update_fact_exit", "alpha_min", "initialize_next_xcond_qp_from_qp_out", "tol_comp_soft_scale")
_DOUBLE_OPTS = ("tol_stat", "tol_eq", "tol_ineq", "tol_comp", "mu0", "tau_min", "t0_min", "lam0_min", "alpha_min", "tol_comp_soft_scale")


def _bind(L):
    vp, ci, cp = C.c_void_p, C.c_int, C.c_char_p
    sigs = {
        "ocp_qp_xcond_solver_config_create_from_name": (vp, [cp]),
        "ocp_qp_xcond_solver_config_free": (None, [vp]),
        "ocp_qp_xcond_solver_dims_create": (vp, [vp, ci]),
        "ocp_qp_xcond_solver_dims_set": (None, [vp, vp, ci, cp, C.POINTER(ci)]),
        "ocp_qp_xcond_solver_dims_free": (None, [vp]),
        "ocp_qp_xcond_solver_opts_create": (vp, [vp, vp]),
        "ocp_qp_xcond_solver_opts_set": (None, [vp, vp, cp, vp]),
        "ocp_qp_xcond_solver_opts_free": (None, [vp]),
        "ocp_qp_create": (vp, [vp, vp, vp]),
        "ocp_qp_solver_destroy": (None, [vp]),
        "ocp_qp_in_create_from_xcond_dims": (vp, [vp]),
        "ocp_qp_in_free": (None, [vp]),
        "ocp_qp_in_set": (None, [vp, vp, ci, cp, vp]),
        "ocp_qp_out_create_from_xcond_dims": (vp, [vp]),
        "ocp_qp_out_free": (None, [vp]),
        "ocp_qp_out_get": (None, [vp, ci, cp, vp]),
        "ocp_qp_solve": (ci, [vp, vp, vp]),
        "ocp_qp_condense_lhs": (ci, [vp, vp, vp]),
        "ocp_qp_condense_rhs_and_solve": (ci, [vp, vp, vp]),
        "ocp_qp_inf_norm_residuals": (None, [vp, vp, vp, C.POINTER(C.c_double)]),
        "ocp_qp_solve_batch": (ci, [vp, ci, C.POINTER(vp), C.POINTER(vp), C.POINTER(ci)]),
        "ocp_qp_xcond_solver_get_scalar": (None, [vp, vp, cp, vp]),
        "ocp_qp_solver_get_stats": (None, [vp, C.POINTER(C.c_double), cp]),
        "ocp_qp_seed_create": (vp, [vp]),
        "ocp_qp_seed_free": (None, [vp]),
        "ocp_qp_solver_eval_forw_sens": (None, [vp, vp, vp, vp]),
        "ocp_qp_solver_eval_adj_sens": (None, [vp, vp, vp, vp]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    return L


class AcadosOcpQpSolver:
    """Solve ONE OCP-QP on the GPU backend through the acados plugin surface."""

    def __init__(self, qp: AcadosOcpQp, opts: AcadosOcpQpOptions = None, verbose: bool = False, _clib=None):
        self.__created = False
        if opts is None:
            opts = AcadosOcpQpOptions()
        qp.make_consistent()
        opts.make_consistent(qp.N)
        self.qp, self.__N, self.__qp_solver_name = qp, qp.N, opts.qp_solver
        self._L = L = _bind(_clib if _clib is not None else _lib.lib())
        self.c_config = C.c_void_p(L.ocp_qp_xcond_solver_config_create_from_name(opts.qp_solver.encode()))
        if not self.c_config:
            raise RuntimeError(f"QP solver {opts.qp_solver} not available")
        self.c_dims = C.c_void_p(L.ocp_qp_xcond_solver_dims_create(self.c_config, qp.N))
        self._set_dimensions_in_c()
        self.c_opts = C.c_void_p(L.ocp_qp_xcond_solver_opts_create(self.c_config, self.c_dims))
        self._set_opts_from_class(opts)
        self.c_solver = C.c_void_p(L.ocp_qp_create(self.c_config, self.c_dims, self.c_opts))
        self.c_in = C.c_void_p(L.ocp_qp_in_create_from_xcond_dims(self.c_dims))
        self.c_out = C.c_void_p(L.ocp_qp_out_create_from_xcond_dims(self.c_dims))
        self._set_qp_data_in_c(self.c_in, qp)
        self.__created = True
        self._status = 0

    @property
    def N(self):
        return self.__N

    @property
    def qp_solver_name(self):
        return self.__qp_solver_name

    def _set_dimensions_in_c(self):
        d = self.qp.dims
        for i in range(self.qp.N + 1):
            for name in ("nx", "nu", "nbx", "nbu", "ng", "ns", "nbxe"):
                v = C.c_int(int(getattr(d, name)[i]))
                self._L.ocp_qp_xcond_solver_dims_set(self.c_config, self.c_dims, i, name.encode(), C.byref(v))

    def opts_set(self, field: str, value):
        if field not in _OPT_FIELDS:
            raise ValueError(f"AcadosOcpQpSolver.opts_set(field={field}, value={value}): '{field}' is an invalid argument."
                             f"\n Possible values are {_OPT_FIELDS}.")
        if self.__created and field in ("cond_N", "cond_block_size"):
            raise RuntimeError(f"cannot set option '{field}' after solver creation.")
        if field == "cond_block_size":
            arr = np.ascontiguousarray(value, dtype=np.intc)
            ptr = arr.ctypes.data_as(C.c_void_p)
        elif isinstance(value, str):
            ptr = C.cast(C.c_char_p(value.encode()), C.c_void_p)
        elif field in _DOUBLE_OPTS:     # the C side reads a double whatever Python type the caller used (tol_stat=1)
            self._keep = C.c_double(float(value)); ptr = C.cast(C.byref(self._keep), C.c_void_p)
        elif field == "initialize_next_xcond_qp_from_qp_out":    # a C bool (ocp_qp_xcond_solver.c:298-302)
            self._keep = C.c_bool(bool(value)); ptr = C.cast(C.byref(self._keep), C.c_void_p)
        elif isinstance(value, (bool, int, np.integer)):
            self._keep = C.c_int(int(value)); ptr = C.cast(C.byref(self._keep), C.c_void_p)
        elif isinstance(value, float):
            raise TypeError(f"option {field} is an integer option, got {value!r}")
        else:
            raise TypeError(f"unsupported type {type(value)} for option {field}")
        self._L.ocp_qp_xcond_solver_opts_set(self.c_config, self.c_opts, field.encode(), ptr)

    def _set_opts_from_class(self, opts):
        self.opts_set("hpipm_mode", opts.hpipm_mode)
        self.opts_set("t0_init", opts.t0_init)
        if opts.mu0 is not None:
            self.opts_set("mu0", float(opts.mu0))
        self.opts_set("ric_alg", opts.ric_alg)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            self.opts_set(f, float(getattr(opts, f)))
        self.opts_set("iter_max", opts.iter_max)
        self.opts_set("cond_N", opts.cond_N)
        if opts.cond_block_size is not None:
            self.opts_set("cond_block_size", opts.cond_block_size)
        self.opts_set("cond_ric_alg", opts.cond_ric_alg)
        self.opts_set("warm_start", opts.warm_start)
        self.opts_set("print_level", opts.print_level)

    def _set_qp_data_in_c(self, c_in, qp):
        for i in range(qp.N + 1):
            for name in _FIELDS:
                if i == qp.N and name in qp.dynamics_fields:
                    continue
                v = getattr(qp, name)[i]
                if name in _INT:
                    a = np.ascontiguousarray(np.ravel(np.asarray(v).astype(np.int32), order="F"))
                else:
                    a = np.ascontiguousarray(np.ravel(np.asarray(v, dtype=float), order="F"))
                if a.size:
                    self._L.ocp_qp_in_set(self.c_config, c_in, i, name.encode(), a.ctypes.data_as(C.c_void_p))

    def solve(self) -> int:
        self._status = self._L.ocp_qp_solve(self.c_solver, self.c_in, self.c_out)
        return self._status

    def condense_lhs(self) -> int:
        """RTI preparation phase: the condense_lhs slot (ocp_qp_xcond_solver.c:591-620)"""
        return self._L.ocp_qp_condense_lhs(self.c_solver, self.c_in, self.c_out)

    def condense_rhs_and_solve(self) -> int:
        """RTI feedback phase: the condense_rhs_and_solve slot (ocp_qp_xcond_solver.c:623-669)"""
        self._status = self._L.ocp_qp_condense_rhs_and_solve(self.c_solver, self.c_in, self.c_out)
        return self._status

    def set(self, stage: int, field: str, value):
        """update one field of the QP held by the solver (ocp_qp_in_set)"""
        a = np.ascontiguousarray(np.ravel(np.asarray(value).astype(np.int32) if field in _INT else np.asarray(value, dtype=float), order="F"))
        self._L.ocp_qp_in_set(self.c_config, self.c_in, int(stage), field.encode(), a.ctypes.data_as(C.c_void_p))

    def inf_norm_residuals(self) -> np.ndarray:
        """[stationarity, dynamics, inequalities, complementarity] inf-norms of the KKT residuals of (qp_in, qp_out),
        recomputed independently of the solver: ocp_qp_inf_norm_residuals (ocp_qp_interface.c:642-650)"""
        class _In(C.Structure):
            _fields_ = [("dim", C.c_void_p)]
        res = (C.c_double * 4)()
        self._L.ocp_qp_inf_norm_residuals(C.cast(self.c_in, C.POINTER(_In)).contents.dim, self.c_in, self.c_out, res)
        return np.array(res[:])

    def _dim(self, stage, field):
        d = self.qp.dims
        return {"x": d.nx[stage], "u": d.nu[stage], "pi": d.nx[stage + 1] if stage < self.N else 0,
                "lam": 2 * (d.ng[stage] + d.nbx[stage] + d.nbu[stage] + d.ns[stage]),
                "t": 2 * (d.ng[stage] + d.nbx[stage] + d.nbu[stage] + d.ns[stage]),
                "sl": d.ns[stage], "su": d.ns[stage]}[field]

    def get(self, stage_: int, field_: str, unique_duals: bool = True, _c_out=None) -> np.ndarray:
        out_fields = ["x", "u", "pi", "lam", "sl", "su", "t"]
        if field_ not in out_fields:
            raise ValueError(f"AcadosOcpQpSolver.get(stage={stage_}, field={field_}): '{field_}' is an invalid argument."
                             f"\n Possible values are {out_fields}.")
        if not isinstance(stage_, (int, np.integer)):
            raise TypeError("stage index must be an integer")
        if stage_ < 0 or stage_ > self.N:
            raise ValueError(f"stage index must be in [0, {self.N}], got: {stage_}.")
        if stage_ == self.N and field_ == "pi":
            raise KeyError(f"field '{field_}' does not exist at final stage {stage_}.")
        out = np.zeros((int(self._dim(stage_, field_)),), dtype=np.float64)
        if out.size:
            self._L.ocp_qp_out_get(_c_out if _c_out is not None else self.c_out, int(stage_), field_.encode(),
                                   out.ctypes.data_as(C.c_void_p))
        if field_ == "lam" and unique_duals and stage_ == 0:
            # same fold as acados_ocp_qp_solver.py:388-397
            d = self.qp.dims
            hard = int(d.ng[0] + d.nbx[0] + d.nbu[0])
            lam_0 = out[:2 * hard]
            unique = lam_0[hard:] - lam_0[:hard]
            out[:hard] = np.maximum(0.0, -unique)
            out[hard:2 * hard] = np.maximum(0.0, unique)
        return out

    def eval_solution_sens(self, seeds: dict, adjoint: bool = False) -> dict:
        """d(solution)/d(parameter) of the QP solved last (the eval_forw_sens / eval_adj_sens slots of the plugin,
        ocp_qp_hpipm.c:481-506).  `seeds`: {(field, stage): vector} with the derivative of the problem data w.r.t. the
        parameter, field in r q b lbu ubu lbx ubx lg ug (natural-sign bounds; for the equality-flagged x0 rows set lbx
        AND ubx, as ocp_nlp_common.c:4057-4064 does).  Returns {"x": [...per stage...], "u": ..., "pi": ..., "lam": ..., "t": ...}."""
        L = self._L

        class _Dims(C.Structure):
            _fields_ = [("N", C.c_int)] + [(n_, C.POINTER(C.c_int)) for n_ in ("nx", "nu", "nb", "nbx", "nbu", "ng", "ns", "nbxe", "nbue", "nge")]

        class _Seed(C.Structure):
            _fields_ = [("dim", C.POINTER(_Dims))] + [(n_, C.POINTER(C.POINTER(C.c_double))) for n_ in ("seed_g", "seed_b", "seed_d", "seed_m")]

        class _In(C.Structure):
            _fields_ = [("dim", C.POINTER(_Dims))]

        dim_ptr = C.cast(self.c_in, C.POINTER(_In)).contents.dim
        c_seed = C.c_void_p(L.ocp_qp_seed_create(dim_ptr))
        sd = C.cast(c_seed, C.POINTER(_Seed)).contents
        d = self.qp.dims
        for (field, k), val in seeds.items():
            v = np.asarray(val, dtype=np.float64).reshape(-1)
            nu, nbu, nbx, nb, ng = int(d.nu[k]), int(d.nbu[k]), int(d.nbx[k]), int(d.nbu[k] + d.nbx[k]), int(d.ng[k])
            arr, off = {"r": ("seed_g", 0), "q": ("seed_g", nu), "b": ("seed_b", 0), "lbu": ("seed_d", 0), "lbx": ("seed_d", nbu),
                        "lg": ("seed_d", nb), "ubu": ("seed_d", nb + ng), "ubx": ("seed_d", nb + ng + nbu), "ug": ("seed_d", 2 * nb + ng)}[field]
            p = getattr(sd, arr)[k]
            for e in range(v.size):
                p[off + e] = v[e]
        c_sens = C.c_void_p(L.ocp_qp_out_create_from_xcond_dims(self.c_dims))
        (L.ocp_qp_solver_eval_adj_sens if adjoint else L.ocp_qp_solver_eval_forw_sens)(self.c_solver, self.c_in, c_seed, c_sens)
        out = {f: [self.get(k, f, unique_duals=False, _c_out=c_sens) for k in range(self.N + (0 if f == "pi" else 1))]
               for f in ("x", "u", "pi", "lam", "t")}
        L.ocp_qp_out_free(c_sens)
        L.ocp_qp_seed_free(c_seed)
        return out

    def get_stats(self, field_: str):
        if field_ == "iter":
            v = C.c_int()
            self._L.ocp_qp_xcond_solver_get_scalar(self.c_solver, self.c_out, field_.encode(), C.byref(v))
            return v.value
        if field_ in ("tau_iter", "time_qp_solver_call", "time_qp_xcond", "time_tot"):
            v = C.c_double(0)
            self._L.ocp_qp_xcond_solver_get_scalar(self.c_solver, self.c_out, field_.encode(), C.byref(v))
            return v.value
        if field_ == "statistics":
            it = self.get_stats("iter")
            out = np.zeros((it + 1, 20))
            self._L.ocp_qp_solver_get_stats(self.c_solver, out.ctypes.data_as(C.POINTER(C.c_double)),
                                            self.qp_solver_name.encode())
            return out
        raise NotImplementedError(f"get_stats() does not support field '{field_}' yet.")

    def print_statistics(self):
        st = self.get_stats("statistics")
        print("\niter\tres_stat\tres_eq\t\tres_ineq\tres_comp\tdual_gap\talpha_prim\talpha_dual\tobj")
        for i in range(st.shape[0]):
            print(f"{i}\t{st[i, 7]:e}\t{st[i, 8]:e}\t{st[i, 9]:e}\t{st[i, 10]:e}\t{st[i, 6]:e}\t{st[i, 4]:e}\t{st[i, 5]:e}\t{st[i, 12]:e}")

    def get_cost(self) -> float:
        return float(self.get_stats("statistics")[-1, 12])

    def get_iterate(self):
        d = {}
        for field in ("x", "u", "sl", "su", "pi", "lam"):
            d[field] = [self.get(n, field) for n in range(self.N + 1) if n < self.N or field != "pi"]
        return d

    def __del__(self):
        try:
            L = self._L
            L.ocp_qp_solver_destroy(self.c_solver)
            L.ocp_qp_in_free(self.c_in); L.ocp_qp_out_free(self.c_out)
            L.ocp_qp_xcond_solver_opts_free(self.c_opts)
            L.ocp_qp_xcond_solver_dims_free(self.c_dims)
            L.ocp_qp_xcond_solver_config_free(self.c_config)
        except Exception:
            pass


class AcadosOcpQpBatchSolver(AcadosOcpQpSolver):
    """N_batch structurally identical QPs solved as ONE device batch: the GPU counterpart of
    `AcadosOcpBatchSolver.solve()` (acados_ocp_batch_solver.py:41-, acados_solver.in.c:3222-3243)."""

    def __init__(self, qps, opts: AcadosOcpQpOptions = None, _clib=None):
        super().__init__(qps[0], opts, _clib=_clib)
        self.qps = list(qps)
        self.N_batch = len(qps)
        self.c_ins, self.c_outs = [self.c_in], [self.c_out]
        for q in qps[1:]:
            q.make_consistent()
            if q.dims.signature() != qps[0].dims.signature():
                raise ValueError("all QPs of a batch must have the same dimensions")
            ci = C.c_void_p(self._L.ocp_qp_in_create_from_xcond_dims(self.c_dims))
            co = C.c_void_p(self._L.ocp_qp_out_create_from_xcond_dims(self.c_dims))
            self._set_qp_data_in_c(ci, q)
            self.c_ins.append(ci); self.c_outs.append(co)
        self.status = np.zeros(self.N_batch, dtype=np.int32)

    def solve(self, n_batch: int = None) -> int:
        n = self.N_batch if n_batch is None else n_batch
        ins = (C.c_void_p * n)(*[p.value for p in self.c_ins[:n]])
        outs = (C.c_void_p * n)(*[p.value for p in self.c_outs[:n]])
        st = (C.c_int * n)()
        worst = self._L.ocp_qp_solve_batch(self.c_solver, n, ins, outs, st)
        self.status[:n] = st[:]
        return worst

    def get_batch(self, i: int, stage_: int, field_: str, unique_duals: bool = True):
        return self.get(stage_, field_, unique_duals, _c_out=self.c_outs[i])

    def get_iter(self, i: int) -> int:
        info = C.c_void_p()
        self._L.ocp_qp_out_get(self.c_outs[i], 0, b"qp_info", C.byref(info))
        # qp_info: 4 doubles then num_iter, t_computed
        return C.cast(info.value + 32, C.POINTER(C.c_int))[0]

    def __del__(self):
        try:
            for ci, co in zip(self.c_ins[1:], self.c_outs[1:]):
                self._L.ocp_qp_in_free(ci); self._L.ocp_qp_out_free(co)
        except Exception:
            pass
        super().__del__()
