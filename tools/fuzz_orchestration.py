"""Development tool: random-STRUCTURE QPs (tests/random_qp.py: per-stage dims, box subsets, one-sided rows, general rows, shared
slacks, equality-flagged x0) through the REFERENCE's compiled orchestration (ocp_qp_xcond_solver.c + ocp_qp_common.c from
/root/reference) around both plugin slots on acados' own types -- integration/ocp_qp_gpu_pcond.c + ocp_qp_gpu_ipm.c -- with a random
N2 <= N, against the CPU oracle and the reference's own residual / compute_t entries.  Needs /root/reference (runs in the build
container, host-simulation tier):  python tools/fuzz_orchestration.py 0 200"""
import os
import pathlib
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    import test_reference_orchestration as T
    from hostsim.build import build
    from random_qp import random_structure_qp
    from oracle.oracle import OracleQp, default_opts
    tmp = pathlib.Path(tempfile.mkdtemp(prefix="fuzz_orch_", dir=os.path.join(ROOT, "gpurun_out")))
    exe = T._build(build(), tmp)
    fails, n_cond, n_through = [], 0, 0
    for seed in range(lo, hi):
        g = np.random.default_rng(seed + 77)
        qp = random_structure_qp(seed, nx_max=(6, 8, 12)[seed % 3], nu_max=(3, 3, 4)[seed % 3])
        N2 = int(g.integers(1, qp.N + 1))
        flags = ["--xcond", "gpu"] + (["--cond-N", str(N2)] if N2 < qp.N else [])
        try:
            head, checks, sol, _, err = T._run(exe, qp, tmp, flags)
            full = "solving the full-space QP" in err
            n_cond += int(N2 < qp.N and not full)
            n_through += int(N2 >= qp.N or full)
            if head["status"] != 0 or head["status_mem"] != 0 or head["rti_status"] != 0:
                oq = OracleQp(qp)
                both = oq.solve(default_opts(tol_stat=1e-8, iter_max=80)) != 0
                fails.append((seed, N2, qp.N, f"status {head}" + (" -- the oracle ends at MAXITER too" if both else " -- the ORACLE CONVERGES")))
                continue
            # (t of the plugin is the iterate's, as HPIPM's: it differs from the reference's recomputed C x - d by the inequality residual)
            if checks["t_diff"] > max(1e-12, 1.01 * checks["res"][2]) or max(checks["res"]) > 1e-8 * (1 + 1e-3) + 1e-13 or checks["rti_diff"] > 1e-9:
                fails.append((seed, N2, qp.N, f"checks {checks}"))
            o = OracleQp(qp)
            if o.solve(default_opts(tol_stat=1e-8, iter_max=80)) != 0:
                fails.append((seed, N2, qp.N, f"oracle at MAXITER too (a limit cycle of the iteration); plugin status {head['status']} after {head['iter']} iterations"))
                continue
            tol = 1e-7 if N2 >= qp.N or full else 1e-4     # (condensed: another iterate path inside the same 1e-8 ball)
            for k in range(qp.N + 1):
                ref = np.concatenate([o.get(k, "u"), o.get(k, "x"), o.get(k, "sl"), o.get(k, "su")])
                if not np.allclose(sol[("ux", k)], ref, rtol=tol, atol=tol):
                    fails.append((seed, N2, qp.N, f"ux stage {k}: max diff {np.max(np.abs(sol[('ux', k)] - ref)):.2e}"))
                    break
        except AssertionError as e:
            fails.append((seed, N2, qp.N, "driver: " + str(e)[-300:]))
        except Exception as e:  # noqa: BLE001
            fails.append((seed, N2, qp.N, f"{type(e).__name__}: {str(e)[:200]}"))
        if (seed - lo) % 25 == 24:
            print(f"seed {seed}: {len(fails)} disagreements, {n_cond} condensed, {n_through} handed through", flush=True)
    print(f"{hi - lo} seeds: {n_cond} condensed on the device module, {n_through} handed through (N2 = N or the module's fall-back), {len(fails)} disagreements")
    for f in fails:
        print("  ", f)


if __name__ == "__main__":
    main()
