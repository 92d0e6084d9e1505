"""the oracle as the checker of a device batch: same-tolerance error and distance to THE solution (outside every timed region) (a part of bench.py)."""

import numpy as np
from benchlib.cpu import threads_allowed


TIGHT = dict(tol_stat=1e-9, tol_eq=1e-11, tol_ineq=1e-11, tol_comp=1e-12, iter_max=100)   # the "solution" the distances below refer to


def oracle_error(gb, qp_of, idx, N, tight=True, same_tol=True):
    """instances `idx` against the oracle (checker only, outside timing; the oracle solves the sample as one OpenMP batch
    over the host cores the process may use), twice:
      same_tol  the oracle at the device's effective tolerances (1e-8 x 4; soft-constrained classes: complementarity at
                1e-8 x tol_comp_soft_scale, the product's exit rule) -- "same algorithm, same stopping point";
      tight     the oracle at TIGHT (complementarity 1e-12: within ~1e-11 of the exact solution, checked against a dense
                active-set solve with an optimality certificate in tests/dense_ref.py::solve_exact) -- the DISTANCE TO THE
                SOLUTION of what the device returns; this is the number a comparison with another solver (HPIPM) at its
                own stopping point can rely on.
    Relative primal error = max over x, u of |dev - ref| / max(1, |ref|)."""
    from oracle.oracle import OracleQp, default_opts, soft_opts, solve_batch_handles
    if len(idx) == 0:
        return {"same_tol_max": 0.0, "instances": 0}
    xs = [gb.get("x", k) for k in range(N + 1)]
    us = [gb.get("u", k) for k in range(N)]
    qps = [OracleQp(qp_of(int(i))) for i in idx]
    scale = gb.scalar("tol_comp_soft_scale") if qps[0].has_slack else 1.0

    def errs():
        e = np.zeros(len(qps))
        for j, (i, o) in enumerate(zip(idx, qps)):
            for k in range(N + 1):
                r = o.get(k, "x")
                if r.size:
                    e[j] = max(e[j], float(np.max(np.abs(xs[k][i][:r.size] - r) / np.maximum(1.0, np.abs(r)))))
                if k < N:
                    r = o.get(k, "u")
                    if r.size:
                        e[j] = max(e[j], float(np.max(np.abs(us[k][i][:r.size] - r) / np.maximum(1.0, np.abs(r)))))
        return e

    hs = [q.h.value for q in qps]
    out = {"instances": len(qps)}
    if same_tol:
        st = solve_batch_handles(hs, soft_opts(default_opts(tol_stat=1e-8, tol_eq=1e-8, tol_ineq=1e-8, tol_comp=1e-8), qps[0].has_slack, scale),
                                 nthreads=threads_allowed())
        e = errs()
        out.update({"same_tol_max": float(e.max()), "same_tol_median": float(np.median(e)), "same_tol_above_1e-6": int((e > 1e-6).sum()),
                    "oracle_failures": int((st != 0).sum()), "oracle_mean_iter": float(np.mean([q.iter for q in qps]))})
    if tight:
        st = solve_batch_handles(hs, default_opts(**TIGHT), nthreads=threads_allowed())
        ok = st == 0
        e = errs()[ok]
        out["dist_to_solution"] = {"reference": "oracle at tol_stat 1e-9, tol_eq / tol_ineq 1e-11, tol_comp 1e-12 (iter_max 100)",
                                   "max": float(e.max()), "q99": float(np.quantile(e, 0.99)), "median": float(np.median(e)),
                                   "above_1e-6": int((e > 1e-6).sum()), "instances": int(ok.sum()),
                                   "reference_not_converged": int((~ok).sum())}
    return out
