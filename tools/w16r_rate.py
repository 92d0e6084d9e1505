"""Sixteen lanes per instance with two rows per lane (ipm_kernels_w16r.hpp) against the wave-per-instance kernels on the
shapes it serves: nx=24 nu=6 (C5 classes) and the condensed C3 shape.  Avg ms per sweep launch and whole-solve rate.
Usage: python tools/w16r_rate.py [batch]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acados_amd import OcpQpGpuBatch
from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 7281


def run(tag, nx, nu, N, cond=0, batch=B):
    d = random_lqr_batch(N=N, nx=nx, nu=nu, batch=batch, seed=1)
    gb = OcpQpGpuBatch(lqr_dims(N, nx, nu), batch)
    fill_lqr_batch(gb, d, N)
    if cond:
        gb.opts_set("cond_N", cond)
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        gb.opts_set(f, 1e-8)
    gb.solve()
    gb.scalar("prof_reset"); gb.opts_set("profile", 1)
    t0 = time.perf_counter(); bad = gb.solve(); dt = time.perf_counter() - t0
    gb.opts_set("profile", 0)
    ms = {c: gb.scalar("prof_ms_" + c) / max(gb.scalar("prof_cnt_" + c), 1) for c in ("back_fact", "fwd_aff", "back_rhs", "fwd_corr")}
    t1 = time.perf_counter(); gb.solve(); dt1 = time.perf_counter() - t1
    res = gb.res_compute()
    print(f"{tag:34s} {gb.condensed_kernel_name() or gb.kernel_name:40s} {batch / dt1:9.0f} solves/s ({dt1 * 1e3:7.2f} ms)  fact {ms['back_fact']:.3f} ms  "
          f"faff {ms['fwd_aff']:.3f} rhs {ms['back_rhs']:.3f} fcor {ms['fwd_corr']:.3f}  failures {bad}  iter {np.mean(gb.info('iter')):.2f}  "
          f"kkt {float(np.max(res)):.2e}", flush=True)
    return [gb.get("x", k) for k in (0, N // 2, N)]


def run_c4(tag, batch=16384, N=40):
    from acados_amd.generators import chain_soft_batch, chain_soft_dims, fill_chain_soft_batch
    d = chain_soft_batch(N=N, batch=batch, seed=1)
    gb = OcpQpGpuBatch(chain_soft_dims(N), batch)
    fill_chain_soft_batch(gb, d, N)
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        gb.opts_set(f, 1e-8)
    gb.solve()
    gb.scalar("prof_reset"); gb.opts_set("profile", 1)
    bad = gb.solve()
    gb.opts_set("profile", 0)
    ms = {c: gb.scalar("prof_ms_" + c) / max(gb.scalar("prof_cnt_" + c), 1) for c in ("back_fact", "fwd_aff", "back_rhs", "fwd_corr")}
    t1 = time.perf_counter(); gb.solve(); dt1 = time.perf_counter() - t1
    res = gb.res_compute()
    print(f"{tag:34s} {gb.kernel_name:40s} {batch / dt1:9.0f} solves/s ({dt1 * 1e3:7.2f} ms)  fact {ms['back_fact']:.3f} ms  "
          f"faff {ms['fwd_aff']:.3f} rhs {ms['back_rhs']:.3f} fcor {ms['fwd_corr']:.3f}  failures {bad}  iter {np.mean(gb.info('iter')):.2f}  "
          f"kkt {float(np.max(res)):.2e}", flush=True)
    return [gb.get("x", k) for k in (0, N // 2, N)] + [gb.get("sl", k) for k in (1, N)]


if len(sys.argv) > 2 and sys.argv[2] == "c4":      # the C4 class: GEN two-rows-per-lane kernels against the wave-per-instance GEN kernels
    sols = {}
    for fam in ("0", "1"):
        os.environ["ACADOS_AMD_W16G"] = fam
        sols[fam] = run_c4(f"W16G={fam} C4 {B}", batch=B)
    print("   max |x(w16r-gen) - x(wpi-gen)| =", max(float(np.max(np.abs(a - b))) for a, b in zip(sols["0"], sols["1"]) if a.size), flush=True)
    sys.exit(0)
CASES = (("nx=24 nu=6 N=20", (24, 6, 20)), ("nx=24 nu=6 N=100", (24, 6, 100)), ("C3 (C2, cond_N=10) 65536", (8, 3, 50, 10, 65536)))
if len(sys.argv) > 2:      # one case, the new family only (for a profiler run): c3 | n20 | n100
    os.environ["ACADOS_AMD_W16R"] = "1"
    run(*{"c3": ("C3", 8, 3, 50, 10, 65536), "n20": ("n20", 24, 6, 20), "n100": ("n100", 24, 6, 100)}[sys.argv[2]])
    sys.exit(0)
for tag, args in CASES:
    sols = {}
    for fam in ("0", "1"):
        os.environ["ACADOS_AMD_W16R"] = fam
        sols[fam] = run(f"W16R={fam} {tag}", *args)
    print("   max |x(w16r) - x(wpi)| =", max(float(np.max(np.abs(a - b))) for a, b in zip(sols["0"], sols["1"])), flush=True)
