"""Development tool: many more seeds of the random-structure parity tests than the test tiers run (tests/random_qp.py), through
every dispatch mode, against the CPU oracle (checker).  Nothing asserts on the way: every disagreement is listed at the end.

    python tools/fuzz_parity.py hostsim 1000 1400            # CPU: the kernel sources under the host-simulation shim
    python tools/fuzz_parity.py gpu 1000 1400 [copies]       # on the GPU box (gpurun)

per seed: full space on the default dispatch, ACADOS_AMD_WPI=0 and =1; partially condensed (N2 = ceil(N/2)); dims drawn from three
size classes (nx <= 6 / <= 12 / <= 24-40) so that every kernel family is met."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def kind(f):
    """what a disagreement is: 'tolerance' (a field a few 1e-7 beyond the comparison's tolerance: another point of the same 1e-8
    ball), 'iterations' (a slowest instance two iterations from the oracle), 'maxiter' (a limit cycle of the iteration, the oracle
    has it too), or one of the kinds that mean a defect: 'hot', 'rti', 'residual', 'copies', 'error'"""
    m = f[-1]
    if f[1] in ("hot", "rti"):
        return f[1]
    if m.startswith("mismatch"):
        return "tolerance"
    if "not converged" in m or "did not converge" in m:
        return "maxiter"
    if "iterations" in m:
        return "iterations"
    if "residual" in m:
        return "residual"
    if "copies differ" in m:
        return "copies"
    return "error"


def run(tier, lo, hi, copies=None, batch=None):
    copies = copies or (3 if tier == "hostsim" else 70)
    from conftest import compare_condensed_with_oracle, compare_with_oracle
    from random_qp import random_structure_qp
    from acados_amd import OcpQpGpuBatch
    from oracle.oracle import OracleQp, default_opts
    clib = None
    if tier == "hostsim":
        import ctypes
        from hostsim.build import build
        from acados_amd import _lib
        # FUZZ_HOSTSIM_LIB: another build of the same sources, e.g. with -fsanitize=address (run under LD_PRELOAD=libasan.so):
        # the device arrays of the host simulation are heap blocks, an out-of-bounds element is reported with its source line
        clib = _lib.bind(ctypes.CDLL(os.environ.get("FUZZ_HOSTSIM_LIB") or build()))
    fails, fams, t0 = [], {}, time.time()
    sizes = [(6, 3), (12, 4), (24, 6), (40, 8)]
    for seed in range(lo, hi):
        # (FUZZ_GPU_SIZES=1: the host simulation on the GPU tier's four size classes -- to replay a seed the device tripped over)
        nxm, num = sizes[seed % 4] if tier != "hostsim" or os.environ.get("FUZZ_GPU_SIZES") else sizes[seed % 3]
        qp = random_structure_qp(seed, nx_max=nxm, nu_max=num, allow_general=(seed % 5 != 0), allow_slack=(seed % 7 != 0))
        o = OracleQp(qp)
        if o.solve(default_opts(tol_stat=1e-8, iter_max=80)) != 0:
            fails.append((seed, "oracle", "did not converge"))
            continue
        for mode in ("default", "wpi0", "wpi1", "cond"):
            if mode == "cond" and qp.N < 2:
                continue
            os.environ.pop("ACADOS_AMD_WPI", None)
            if mode in ("wpi0", "wpi1"):
                os.environ["ACADOS_AMD_WPI"] = mode[-1]
            try:
                if os.environ.get("FUZZ_VERBOSE"):
                    print("seed", seed, mode, flush=True)
                b = OcpQpGpuBatch.from_qps([qp] * copies, **({"_clib": clib} if clib is not None else {}))
                for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
                    b.opts_set(f, 1e-8)
                b.opts_set("iter_max", 80)
                if mode == "cond":
                    b.opts_set("cond_N", (qp.N + 1) // 2)
                bad = b.solve()
                name = b.kernel_name if mode != "cond" or not b.condensed_kernel_name() else "cond:" + b.condensed_kernel_name()
                fams[name.split("(")[0]] = fams.get(name.split("(")[0], 0) + 1
                if bad:
                    fails.append((seed, mode, name, f"{bad} of {copies} not converged, status {np.unique(b.info('status')).tolist()} iters {int(b.info('iter').max())} oracle {o.iter}"))
                    continue
                i = copies - 1
                if mode == "cond":
                    compare_condensed_with_oracle(lambda k, f: b.get(f, k)[i], o, qp)
                    if max(b.info(n).max() for n in ("res_stat", "res_eq", "res_ineq", "res_comp")) > 1e-8:
                        fails.append((seed, mode, name, "residual of the expanded point above 1e-8"))
                else:
                    if abs(int(b.info("iter")[i]) - o.iter) > 1:
                        fails.append((seed, mode, name, f"iterations {int(b.info('iter')[i])} vs oracle {o.iter}"))
                    compare_with_oracle(lambda k, f: b.get(f, k)[i], o, qp, 3e-7, fields=("x", "u", "sl", "su", "pi", "lam", "t"))
                    if mode == "default":
                        # hot start from the solution: converged at the first residual evaluation (or one iteration), same point;
                        # then a new right-hand side through the RTI split (condense_lhs once, condense_rhs_and_solve) = a plain solve
                        x0 = b.get("x", qp.N).copy()
                        b.opts_set("warm_start", 3)
                        bad2 = b.solve()
                        if bad2 or int(b.info("iter").max()) > 1 or np.max(np.abs(b.get("x", qp.N) - x0)) > 1e-7 * (1 + np.max(np.abs(x0))):
                            fails.append((seed, "hot", name, f"hot start: {bad2} failed, iters {int(b.info('iter').max())}, dx {np.max(np.abs(b.get('x', qp.N) - x0)):.2e}"))
                        b.opts_set("warm_start", 0)
                        if qp.N >= 2:
                            b.opts_set("cond_N", (qp.N + 1) // 2)
                            b.condense_lhs()
                            qk = b.get("q", 1)
                            b.set("q", 1, qk * 1.25 + 0.1)
                            r1 = b.condense_rhs_and_solve()
                            xa = b.get("x", qp.N).copy()
                            r2 = b.solve()
                            if r1 or r2 or np.max(np.abs(b.get("x", qp.N) - xa)) > 1e-9 * (1 + np.max(np.abs(xa))):
                                fails.append((seed, "rti", name, f"RTI split vs plain solve: {r1} {r2} dx {np.max(np.abs(b.get('x', qp.N) - xa)):.2e}"))
                        continue
                    # every copy is the same QP: the batch must agree with itself
                    for f in ("x", "u"):
                        for k in (0, qp.N):
                            a = b.get(f, k)
                            if a.size and np.max(np.abs(a - a[0])) > 1e-12:
                                fails.append((seed, mode, name, f"copies differ in {f}[{k}]"))
            except AssertionError as e:
                fails.append((seed, mode, locals().get("name", "?"), "mismatch: " + str(e)[:200]))
            except Exception as e:  # noqa: BLE001
                fails.append((seed, mode, locals().get("name", "?"), f"{type(e).__name__}: {str(e)[:200]}"))
        # a batch of DIFFERENT instances of the structure (linear cost terms perturbed per instance: other active sets, other
        # iteration counts -> the tail switch / live-instance permutation / sub-level paths), a few of them against the oracle
        # on the data read back from the device, every instance through the independent residual kernel
        if tier != "hostsim" or seed % 10 == 0 or os.environ.get("FUZZ_BATCH_ALL"):
            os.environ.pop("ACADOS_AMD_WPI", None)
            B = batch or (1536 if tier != "hostsim" or os.environ.get("FUZZ_BATCH_ALL") else 96)
            name = "?"
            try:
                if os.environ.get("FUZZ_VERBOSE"):
                    print("seed", seed, "batch", flush=True)
                g = np.random.default_rng(seed + 9000)
                b = OcpQpGpuBatch.from_qps([qp] * B, **({"_clib": clib} if clib is not None else {}))
                for k in range(qp.N + 1):
                    for f in ("q", "r"):
                        a0 = b.get(f, k)
                        if a0.shape[1]:
                            b.set(f, k, a0 * g.uniform(-2.0, 3.0, (B, 1)) + 0.3 * g.standard_normal(a0.shape))
                for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
                    b.opts_set(f, 1e-8)
                b.opts_set("iter_max", 80)
                if os.environ.get("FUZZ_VERBOSE"):
                    print("   kernel", b.kernel_name, flush=True)
                bad = b.solve()
                name = "batch:" + b.kernel_name
                fams[name.split("(")[0]] = fams.get(name.split("(")[0], 0) + 1
                it = b.info("iter")
                if bad:
                    fails.append((seed, "batch", name, f"{bad} of {B} not converged, status {np.unique(b.info('status')).tolist()} iters {int(it.min())}..{int(it.max())}"))
                elif float(b.res_compute().max()) > 1e-8 * (1 + 1e-3) + 1e-13:
                    fails.append((seed, "batch", name, f"independent residual {float(b.res_compute().max()):.3e}"))
                else:
                    for i in sorted({0, B - 1, int(np.argmax(it)), int(np.argmin(it)), B // 2, 65}):
                        qi = b.to_qp(i)
                        oi = OracleQp(qi)
                        if oi.solve(default_opts(tol_stat=1e-8, iter_max=80)) != 0:
                            fails.append((seed, "batch", name, f"oracle did not converge on instance {i}"))
                            continue
                        if abs(int(it[i]) - oi.iter) > 1:
                            fails.append((seed, "batch", name, f"instance {i}: iterations {int(it[i])} vs oracle {oi.iter}"))
                        compare_with_oracle(lambda k, f: b.get(f, k)[i], oi, qi, 3e-7, fields=("x", "u", "sl", "su", "pi", "lam", "t"))
            except AssertionError as e:
                fails.append((seed, "batch", name, "mismatch: " + str(e)[:200]))
            except Exception as e:  # noqa: BLE001
                fails.append((seed, "batch", name, f"{type(e).__name__}: {str(e)[:200]}"))
        if (seed - lo) % 25 == 24:
            print(f"seed {seed}: {len(fails)} disagreements so far, {time.time() - t0:.0f} s", flush=True)
    os.environ.pop("ACADOS_AMD_WPI", None)
    return fails, fams


def main():
    tier, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    fails, fams = run(tier, lo, hi, int(sys.argv[4]) if len(sys.argv) > 4 else None)
    print("kernel families met:", dict(sorted(fams.items(), key=lambda kv: -kv[1])))
    kinds = {}
    for f in fails:
        kinds[kind(f)] = kinds.get(kind(f), 0) + 1
    print(f"{hi - lo} seeds, {len(fails)} disagreements {kinds}")
    for f in fails:
        print("  ", kind(f), f)


if __name__ == "__main__":
    main()
