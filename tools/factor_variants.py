"""Factor-sweep variants of the wave-per-instance family side by side (C4 class, the nx=24 nu=6 box class, the condensed C3
shape): avg ms per full factor launch and whole-solve rate.  Usage: python tools/factor_variants.py [batch]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acados_amd import OcpQpGpuBatch
from acados_amd.generators import chain_soft_batch, chain_soft_dims, fill_chain_soft_batch, fill_lqr_batch, lqr_dims, random_lqr_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096


def run(tag, make):
    gb = make()
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        gb.opts_set(f, 1e-8)
    gb.solve()
    gb.scalar("prof_reset"); gb.opts_set("profile", 1)
    t0 = time.perf_counter(); bad = gb.solve(); dt = time.perf_counter() - t0
    gb.opts_set("profile", 0)
    ms = {c: gb.scalar("prof_ms_" + c) / max(gb.scalar("prof_cnt_" + c), 1) for c in ("back_fact", "fwd_aff", "back_rhs", "fwd_corr")}
    print(f"{tag:38s} {gb.condensed_kernel_name() or gb.kernel_name:52s} {B / dt:9.0f} solves/s  fact {ms['back_fact']:.3f} ms  "
          f"faff {ms['fwd_aff']:.3f} rhs {ms['back_rhs']:.3f} fcor {ms['fwd_corr']:.3f}  failures {bad}", flush=True)


def c4():
    d = chain_soft_batch(N=40, batch=B, seed=1)
    g = OcpQpGpuBatch(chain_soft_dims(40), B)
    fill_chain_soft_batch(g, d, 40)
    return g


def lqr(nx, nu, N, cond=0):
    def make():
        d = random_lqr_batch(N=N, nx=nx, nu=nu, batch=B, seed=1)
        g = OcpQpGpuBatch(lqr_dims(N, nx, nu), B)
        fill_lqr_batch(g, d, N)
        if cond:
            g.opts_set("cond_N", cond)
        return g
    return make


for mf, pf, ct in (("0", "0", "0"), ("0", "0", "1"), ("1", "0", "1"), ("1", "1", "1"), ("1", "2", "1")):
    os.environ["ACADOS_AMD_WPI_MFMA"], os.environ["ACADOS_AMD_WPI_MFMA_PF"], os.environ["ACADOS_AMD_WPI_CT"] = mf, pf, ct
    tag = f"MFMA={mf} PF={pf} CT={ct}"
    run(tag + " C4", c4)
    run(tag + " nx=24 nu=6 N=50", lqr(24, 6, 50))
    run(tag + " C3 (C2, cond_N=10)", lqr(8, 3, 50, 10))
