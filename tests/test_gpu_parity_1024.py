"""GPU tier: the >= 1,024-instance parity gates of every BASELINE configuration (SURVEY.md 8d asks the relative primal error
against the oracle on a >= 1,024-instance subsample; until round 3 those samples were numbers printed by bench.py, here they
are asserts).  Two references per sample (bench.oracle_error -- the same code the bench line is produced with):

  same_tol           the oracle stopped where the device stops (1e-8 x 4; with the opt-in tol_comp_soft_scale < 1 of a
                     soft-constrained class: complementarity at 1e-8 x that scale, both sides): same algorithm, same stopping point;
  dist_to_solution   the oracle at complementarity 1e-12 (pinned in the CPU tier against a dense active-set solve with an
                     optimality certificate, tests/dense_ref.py::solve_exact: <= 1e-9 from the exact solution): how far
                     from THE solution the device stops -- what agreement with another solver (HPIPM) at its own stopping
                     point can be promised from.  north_star bar: 1e-6 relative primal.

Same-tolerance bars (written here; measured values: the `oracle_check` objects of profiles/r0*_bench.json): C2 1e-8, C3 1e-6,
C5 (nine classes + the multi-phase class) 1e-8, C4 1e-6 at the opt-in tight exit and distance-aware at the default one.  Distance to the solution: an IPM stops ON the central path -- a weakly active row with
multiplier lam* sits at t = mu / lam* -- so at the acados tolerances (1e-8) even the hard-constrained C2 class has 5 % of its
instances more than 1e-6 away from the exact solution (measured: median 3e-8, max 4e-5) although every KKT residual is
<= 1e-8 and device and oracle agree to 4e-15; what is asserted is that the distance is the TOLERANCE's, not the kernels':
it shrinks with tol_comp (C2 at tol_comp 1e-10: every instance within 1e-6), and for the soft-constrained C4 class -- where
the 1e-8 ball is so flat that two runs of the same algorithm differ by 5e-6 -- the OPT-IN exit rule tol_comp_soft_scale 1e-3
(complementarity at tol_comp x 1e-3; the default is 1 = the tolerance as given, as the reference stops) brings 99 % of the
sample within 1e-6 (measured: all of it, max 8e-7)."""
import numpy as np
import pytest

from bench import oracle_error

pytestmark = pytest.mark.gpu
KKT_TOL = 1e-8 * (1.0 + 1e-3) + 1e-13


def _tols(gb):
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        gb.opts_set(f, 1e-8)
    gb.opts_set("iter_max", 50)


def _sample(B, n=1024):
    return np.unique(np.linspace(0, B - 1, n).astype(int))


def test_c2_and_c3_1024_instances_gpu(gpu_lib):
    """C2 (N=50 nx=8 nu=3, 65,536 instances, BASELINE configs[1]) and C3 (the same batch condensed to N2=10)"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, lqr_instance_qp, random_lqr_batch
    N, B = 50, 65536
    data = random_lqr_batch(N=N, batch=B, seed=0)
    gb = OcpQpGpuBatch(lqr_dims(N, 8, 3), B)
    fill_lqr_batch(gb, data, N)
    _tols(gb)
    idx = _sample(B)
    assert idx.size >= 1024
    assert gb.solve() == 0
    assert gb.res_compute().max() <= KKT_TOL
    e = oracle_error(gb, lambda i: lqr_instance_qp(data, i, N), idx, N)
    print("C2", e)
    assert e["oracle_failures"] == 0 and e["same_tol_max"] <= 1e-8, e
    d = e["dist_to_solution"]
    assert d["reference_not_converged"] == 0 and d["median"] <= 1e-6 and d["q99"] <= 1e-4 and d["max"] <= 1e-3, e
    gb.opts_set("cond_N", 10)
    assert gb.solve() == 0 and int(gb.scalar("cond_N_active")) == 10
    assert gb.res_compute().max() <= 2e-8          # expanded point in the ORIGINAL QP (condensed runs)
    e = oracle_error(gb, lambda i: lqr_instance_qp(data, i, N), idx, N)
    print("C3", e)
    d = e["dist_to_solution"]
    assert e["same_tol_max"] <= 1e-6 and d["median"] <= 1e-6 and d["q99"] <= 1e-4 and d["max"] <= 1e-3, e
    # the opt-in terminal polishing step at the plain exit: status and iteration counts unchanged, exit test still passed, the tail
    # of the distance gone (measured: 53 of 1,024 above 1e-6 without it)
    gb.opts_set("cond_N", N)
    it0, st0 = gb.info("iter").copy(), gb.info("status").copy()
    gb.opts_set("polish", 1)
    assert gb.solve() == 0 and gb.res_compute().max() <= KKT_TOL
    assert np.array_equal(gb.info("iter"), it0) and np.array_equal(gb.info("status"), st0)
    xp = oracle_error(gb, lambda i: lqr_instance_qp(data, i, N), idx, N, same_tol=False)
    print("C2 with the polishing step", xp, "polished", gb.scalar("polished"), "reverted", gb.scalar("polish_reverted"))
    assert 0 < gb.scalar("polished") < B and gb.scalar("polish_reverted") <= 0.01 * gb.scalar("polished")
    # measured (profiles/r06_polish_sweep.txt): 53 -> 13 above 1e-6, max 3.6e-5 -> 7.3e-6 at -7.6 % rate -- dominated by tol_comp 1e-10
    # (0 above, max 3.5e-7, -4.4 %), which is why the step stays an opt-in and is not what any quoted rate is run with
    assert xp["dist_to_solution"]["max"] <= 1e-5 and xp["dist_to_solution"]["above_1e-6"] <= d["above_1e-6"] // 2, xp
    gb.opts_set("polish", 0)
    # the distance is the tolerance's: two more orders on complementarity (a user's choice for a hard-constrained class)
    gb.opts_set("tol_comp", 1e-10)
    assert gb.solve() == 0
    xs = oracle_error(gb, lambda i: lqr_instance_qp(data, i, N), idx, N, same_tol=False)
    print("C2 at tol_comp 1e-10", xs, "mean iterations", gb.info("iter").mean())
    assert xs["dist_to_solution"]["max"] <= 1e-6, xs


def test_c4_1024_instances_gpu(gpu_lib):
    """C4 (chain N=40 nx=24 nu=3, soft state bounds + soft general rows, 16,384 instances, BASELINE configs[3])"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import chain_soft_batch, chain_soft_dims, chain_soft_instance_qp, fill_chain_soft_batch
    N, B = 40, 16384
    data = chain_soft_batch(N=N, batch=B, seed=1)
    gb = OcpQpGpuBatch(chain_soft_dims(N), B)
    fill_chain_soft_batch(gb, data, N)
    _tols(gb)
    idx = _sample(B)
    qp_of = lambda i: chain_soft_instance_qp(data, i, N)
    # DEFAULT: the solver stops at the tol_comp it is given (the reference's semantics, ocp_qp_hpipm.c:104-107).  NOT the leg the
    # bench line quotes C4's rate at (bench.py other_configs: the rate is quoted where the flat 1e-6 bar below holds)
    assert gb.scalar("tol_comp_soft_scale") == 1.0 and abs(gb.scalar("tol_comp_effective") - 1e-8) < 1e-20
    assert gb.solve() == 0 and gb.kernel_name.startswith("w16r-gen<NX=24,NU=3,NG=4>")
    assert gb.info("iter").max() <= 25
    assert gb.res_compute().max() <= KKT_TOL
    e1 = oracle_error(gb, qp_of, idx, N)
    print("C4 default (plain 1e-8 exit)", e1)
    d1 = e1["dist_to_solution"]
    # distance-aware: the oracle is on its PLAIN tolerances too; both iterates have KKT <= 1e-8 and lie in the flat ball a soft
    # row with a small multiplier leaves (t = mu / lam*): the two may differ by what either is away from the exact solution
    assert e1["oracle_failures"] == 0 and d1["reference_not_converged"] == 0
    assert e1["same_tol_median"] <= 1e-8 and e1["same_tol_max"] <= max(1e-6, 2.0 * d1["max"]), e1
    assert d1["median"] <= 1e-5 and d1["max"] <= 1e-3, e1
    # OPT-IN tighter exit (tol_comp_soft_scale 1e-3: complementarity at 1e-11, both sides) -- the leg that backs the rate quoted for
    # C4: FLAT 1e-6 against the oracle at the same tolerance and against the solution
    gb.opts_set("tol_comp_soft_scale", 1e-3)
    assert abs(gb.scalar("tol_comp_effective") - 1e-11) < 1e-24
    assert gb.solve() == 0 and gb.res_compute().max() <= KKT_TOL
    assert gb.info("iter").max() <= 25
    e = oracle_error(gb, qp_of, idx, N)
    print("C4 tol_comp_soft_scale 1e-3", e)
    d = e["dist_to_solution"]
    assert e["oracle_failures"] == 0 and e["same_tol_max"] <= 1e-6, e
    assert d["reference_not_converged"] == 0 and d["q99"] <= 1e-6 and d["median"] <= 1e-8 and d["max"] <= 1e-5 and d["above_1e-6"] <= idx.size // 100, e
    # (one more order -- scale 1e-4, complementarity 1e-12 -- is past what FP64 carries for this class: Gamma = lam / t of the
    #  active soft rows reaches 1e16, stationarity is lost to rounding and 38 of 16,384 instances end in MAXITER: DESIGN.md 3)
    assert d1["median"] > 10 * d["median"]
    # (the opt-in polishing step on this class: tools/polish_sweep.py, profiles/r06_polish_sweep.txt -- 276 -> 126 above 1e-6, never 0)


def test_c5_1024_instances_gpu(gpu_lib):
    """C5: the nine shape classes at the per-GPU share (7,281 instances each) plus the multi-phase class (nx 12 -> 4 at
    N/2), 114 instances of every class against the oracle (1,140 in all)"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import (C5_CLASSES, fill_lqr_batch, fill_multiphase_batch, lqr_dims, lqr_instance_qp, multiphase_batch,
                                       multiphase_dims, multiphase_instance_qp, random_lqr_batch)
    per_class = (524288 // 8) // len(C5_CLASSES)
    total = 0
    for ci, (nx, nu, N) in enumerate(C5_CLASSES):
        data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=per_class, seed=200 + ci)
        gb = OcpQpGpuBatch(lqr_dims(N, nx, nu), per_class)
        fill_lqr_batch(gb, data, N)
        _tols(gb)
        assert gb.solve() == 0, (nx, nu, N)
        assert gb.res_compute().max() <= KKT_TOL
        idx = _sample(per_class, 114)
        e = oracle_error(gb, lambda i: lqr_instance_qp(data, i, N), idx, N)
        print("C5", (nx, nu, N), gb.kernel_name, e)
        d = e["dist_to_solution"]
        assert e["oracle_failures"] == 0 and e["same_tol_max"] <= 1e-8 and d["median"] <= 1e-6 and d["max"] <= 1e-3, ((nx, nu, N), e)
        total += idx.size
        del gb
    for N in (20, 50):
        data = multiphase_batch(N=N, batch=per_class)
        gb = OcpQpGpuBatch(multiphase_dims(N), per_class)
        fill_multiphase_batch(gb, data)
        _tols(gb)
        assert gb.solve() == 0
        assert gb.res_compute().max() <= KKT_TOL
        idx = _sample(per_class, 114)
        e = oracle_error(gb, lambda i: multiphase_instance_qp(data, i), idx, N)
        print("C5 multi-phase", N, gb.kernel_name, e)
        d = e["dist_to_solution"]
        assert e["oracle_failures"] == 0 and e["same_tol_max"] <= 1e-8 and d["median"] <= 1e-6 and d["max"] <= 1e-3, e
        total += idx.size
    assert total >= 1024
