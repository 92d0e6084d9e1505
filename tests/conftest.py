import ctypes
import json
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


# The library picks the kernel family by stage-block size AND batch size (small batches go to the
# wave-per-instance kernels).  The tests run small batches but mean to cover both families, so the batch rule is
# switched off here; tests that want it (or a specific family) set ACADOS_AMD_WPI_BATCH_MAX / ACADOS_AMD_WPI.
os.environ.setdefault("ACADOS_AMD_WPI_BATCH_MAX", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu():
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def hostsim_lib():
    """g++ build of the kernel sources against the host-simulation shim (test infra only)."""
    from hostsim.build import build
    from acados_amd import _lib
    # ACADOS_AMD_HOSTSIM_LIB: another build of the same sources, e.g. tools/asan_hostsim.sh's (-fsanitize=address: the device
    # arrays of the host simulation are heap blocks, an out-of-bounds element is reported with its source line)
    return _lib.bind(ctypes.CDLL(os.environ.get("ACADOS_AMD_HOSTSIM_LIB") or build()))


@pytest.fixture(scope="session")
def gpu_lib():
    """the product library (hipcc, gfx950); tests using it are marked gpu"""
    from acados_amd import _lib
    return _lib.lib()


def load_qp(rel):
    from acados_amd import AcadosOcpQp
    return AcadosOcpQp.from_json(os.path.join(GOLDEN, rel))


def load_sol(rel):
    sol = json.load(open(os.path.join(GOLDEN, rel)))
    return {re.sub(r"_0*(\d+)$", lambda m: "_" + m.group(1), k): np.asarray(v, dtype=float).ravel() for k, v in sol.items()}


GOLDEN_PAIRS = [("qp_test/last_qp_nonuniform_pendulum.json", "qp_test/sqp_sol_nonuniform_pendulum.json"),
                ("qp_test/last_qp_one_sided_test.json", "qp_test/sqp_sol_one_sided_test.json")]
INPUT_ONLY = ["casadi_qp_tests/pendulum_qp.json", "casadi_qp_tests/pendulum_slack.json",
              "casadi_qp_tests/pend_idxs_rev_min_qp0.json"]


def fold_stage0(lam, hard):
    """unique-dual fold of acados_ocp_qp_solver.py:388-397"""
    lam = lam.copy()
    u = lam[hard:2 * hard] - lam[:hard]
    lam[:hard] = np.maximum(0.0, -u)
    lam[hard:2 * hard] = np.maximum(0.0, u)
    return lam


def compare_with_oracle(get_fn, oracle, qp, tol, fields=("x", "u", "sl", "su", "pi", "lam")):
    """max abs deviation of get_fn(stage, field) from the oracle's solution"""
    worst = 0.0
    for k in range(qp.N + 1):
        for f in fields:
            if f == "pi" and k == qp.N:
                continue
            ref = oracle.get(k, f)
            if ref.size == 0:
                continue
            got = np.asarray(get_fn(k, f))
            err = np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref)))
            assert err <= tol, f"{f} at stage {k}: {err} > {tol}\n got {got}\n ref {ref}"
            worst = max(worst, err)
    return worst


def compare_condensed_with_oracle(get_fn, oracle, qp, loose=1e-4):
    """A condensed run (or an expanded condensed solution) against the oracle's full-space solution: two iterate paths
    into the same 1e-8 KKT ball of the ORIGINAL QP (that membership is asserted by the callers with the independent
    residual kernel -- the sharp statement).  What the ball leaves open depends on the instance:
      * an inequality row that is WEAKLY active (oracle: 0 < max(lam, t) < 1e-3, i.e. lam ~ t ~ sqrt(tol)) lets the primal
        point move by ~sqrt(tol): such instances are compared at `loose` (1e-4) throughout;
      * every other instance (strict complementarity margin >= 1e-3) is compared SHARPLY: primal variables, slacks and t
        at 2e-6, multipliers of strictly INACTIVE rows (oracle t >= 0.1: lam = tau / t <= 1e-8 / t) at 1e-7, equality multipliers at
        1e-5 and multipliers of active rows at `loose` (they carry the conditioning of the active-set Jacobian and are
        not unique where active rows are dependent: measured up to 3e-5 with primal agreement at 1e-10).
    Returns True if the instance took the sharp path."""
    margin = np.inf
    for k in range(qp.N + 1):
        lam, t = oracle.get(k, "lam"), oracle.get(k, "t")
        m = np.maximum(lam, t)[(lam + t) > 0.0]          # masked sides carry lam = t = 0
        if m.size:
            margin = min(margin, float(m.min()))
    if margin < 1e-3:
        compare_with_oracle(get_fn, oracle, qp, loose, fields=("x", "u", "sl", "su", "pi", "lam", "t"))
        return False
    compare_with_oracle(get_fn, oracle, qp, 2e-6, fields=("x", "u", "sl", "su", "t"))
    compare_with_oracle(get_fn, oracle, qp, 1e-5, fields=("pi",))
    compare_with_oracle(get_fn, oracle, qp, loose, fields=("lam",))
    for k in range(qp.N + 1):
        lam, t = oracle.get(k, "lam"), oracle.get(k, "t")
        if lam.size == 0:
            continue
        got = np.asarray(get_fn(k, "lam"))
        inactive = t >= 0.1
        assert np.all(np.abs(got - lam)[inactive] <= 1e-7), f"lam of strictly inactive rows at stage {k}: {np.abs(got - lam)[inactive].max()}"
    return True


def check_finished_lanes_ride_along(clib, N, B, seed, alone):
    """One-instance-per-lane box sweeps (ipm_kernels_box.hpp, GQP_WAVE_ANY): a lane whose instance has finished keeps
    running the sweeps on its unchanged iterate while another lane of its wave iterates.  Nothing of the finished instance
    may move: the instances `alone` of a B-instance batch (iteration counts differ inside every wave) are solved once more
    as batches of one on the same kernels, where nothing rides along, and solution, multipliers, iteration count and
    the factor of the last factorisation are compared bit for bit."""
    import numpy as np
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch
    data = random_lqr_batch(N=N, batch=B, seed=seed)

    def solve(d, n):
        gb = OcpQpGpuBatch(lqr_dims(N, 8, 3), n, _clib=clib)
        fill_lqr_batch(gb, d, N)
        gb.opts_set("tol_stat", 1e-8)
        gb.opts_set("tail_max", 0)   # no hand-over to another kernel family: this is about bit identity
        assert gb.solve() == 0
        assert gb.kernel_name.startswith("1tpi-box"), gb.kernel_name
        return gb

    full = solve(data, B)
    it = full.info("iter")
    assert len(set(it.tolist())) >= 3, it          # instances really finish at different iterations
    for i in alone:
        one = solve({k: v[i:i + 1] for k, v in data.items()}, 1)
        assert int(one.info("iter")[0]) == int(it[i]), i
        for f in ("res_stat", "res_comp", "mu"):
            assert one.info(f)[0] == full.info(f)[i], (f, i)
        for k in range(N + 1):
            for f in ("x", "u", "lam", "t", "ric_L", "ric_l") + (("pi",) if k < N else ()):
                assert np.array_equal(one.get(f, k)[0], full.get(f, k)[i]), (f, k, i)
    return it


def check_polish(clib, N, B, seed, nx=8, nu=3, ratio=None):
    """The terminal polishing step (option "polish", gpu_batch.hip polish_pass): converged instances that hold a balanced pair
    min(lam, t) > 1e-3 max(lam, t) run one more iteration.  Asserted: status and iteration counts are the plain solve's; instances
    that were not selected are bit-identical; the polished point passes the exit test the solve ran with (independent KKT kernel);
    the distance to THE solution (oracle at complementarity 1e-12) of the polished instances falls by an order of magnitude in the
    maximum and never grows beyond rounding.  Returns (distance plain, distance polished, polished mask)."""
    import numpy as np
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, lqr_instance_qp, random_lqr_batch
    from oracle.oracle import OracleQp, default_opts
    data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=B, seed=seed)
    runs = []
    for pol in (0, 1):
        gb = OcpQpGpuBatch(lqr_dims(N, nx, nu), B, _clib=clib)
        fill_lqr_batch(gb, data, N)
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            gb.opts_set(f, 1e-8)
        gb.opts_set("polish", pol)
        if ratio is not None:
            gb.opts_set("polish_ratio", float(ratio))
        assert gb.solve() == 0
        assert gb.res_compute().max() <= 1e-8 * (1.0 + 1e-3) + 1e-13
        runs.append(gb)
    plain, pol = runs
    assert int(plain.scalar("polished")) == 0
    npol, nrev = int(pol.scalar("polished")), int(pol.scalar("polish_reverted"))
    for f in ("status", "iter"):
        assert np.array_equal(plain.info(f), pol.info(f)), f
    xs = [[g.get("x", k) for k in range(N + 1)] for g in runs]
    us = [[g.get("u", k) for k in range(N)] for g in runs]
    changed = np.zeros(B, dtype=bool)
    for k in range(N + 1):
        changed |= np.any(xs[0][k] != xs[1][k], axis=1)
        if k < N:
            changed |= np.any(us[0][k] != us[1][k], axis=1)
    assert changed.sum() <= npol - nrev and (npol == 0 or changed.sum() >= 1), (changed.sum(), npol, nrev)
    dist = np.zeros((2, B))
    for i in range(B):
        o = OracleQp(lqr_instance_qp(data, i, N))
        assert o.solve(default_opts(tol_stat=1e-9, tol_eq=1e-11, tol_ineq=1e-11, tol_comp=1e-12, iter_max=100)) == 0
        for r in range(2):
            for k in range(N + 1):
                ref = o.get(k, "x")
                dist[r, i] = max(dist[r, i], np.max(np.abs(xs[r][k][i] - ref) / np.maximum(1.0, np.abs(ref))))
                if k < N:
                    ref = o.get(k, "u")
                    dist[r, i] = max(dist[r, i], np.max(np.abs(us[r][k][i] - ref) / np.maximum(1.0, np.abs(ref))))
    return dist[0], dist[1], changed


def check_whole_solve_in_one_launch(clib, qp_sets):
    """Small batches of the sixteen-lanes family: the whole solve in ONE launch (kx_solve: every 16-lane row runs the IPM
    loop by itself) against the launch-per-sweep loop of the same kernels (option solve_max = 0) -- statuses, iteration
    counts, residuals, the per-iteration statistics and every solution array bit for bit."""
    import numpy as np
    from acados_amd import OcpQpGpuBatch
    used = {}
    for qps in qp_sets:
        runs = []
        for smax in (0, 256):
            b = OcpQpGpuBatch.from_qps(qps, _clib=clib)
            for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
                b.opts_set(f, 1e-8)
            b.opts_set("iter_max", 60)
            b.opts_set("solve_max", smax)
            b.solve()
            runs.append(b)
        a, o = runs
        if not o.kernel_name.startswith(("w16-box<", "w16-soft<")):
            assert int(o.scalar("single_launch_solves")) == 0, o.kernel_name
            continue
        fam = o.kernel_name.split("<")[0]
        used[fam] = used.get(fam, 0) + 1
        assert int(a.scalar("single_launch_solves")) == 0 and int(o.scalar("single_launch_solves")) == 1
        assert int(o.scalar("launches")) < int(a.scalar("launches")) and int(a.scalar("launches")) >= 12
        for f in ("status", "iter", "res_stat", "res_eq", "res_ineq", "res_comp", "mu"):
            assert np.array_equal(a.info(f), o.info(f)), (f, o.kernel_name)
        qp = qps[0]
        for k in range(qp.N + 1):
            fields = ["x", "u", "lam", "t", "ric_L", "ric_l"] + (["pi"] if k < qp.N else [])
            fields += ["sl", "su"] if int(qp.dims.ns[k]) else []
            for f in fields:
                assert np.array_equal(a.get(f, k), o.get(f, k)), (f, k, o.kernel_name)
    return used


def limit_cycle_case(clib):
    """tests/test_host_logic.py::test_conditional_corrector_ends_a_limit_cycle_hostsim and its GPU-tier twin: the five instances of
    the perturbed batch of random structure 7105 that ran into a four-cycle of the Mehrotra iteration before the conditional
    corrector applied HPIPM's test; clib = None: the product library"""
    from oracle.oracle import OracleQp, default_opts
    from acados_amd import OcpQpGpuBatch
    from random_qp import random_structure_qp
    seed, B = 7105, 1536
    qp = random_structure_qp(seed, nx_max=12, nu_max=4, allow_general=(seed % 5 != 0), allow_slack=(seed % 7 != 0))
    g = np.random.default_rng(seed + 9000)
    src = OcpQpGpuBatch.from_qps([qp] * B, _clib=clib)
    for k in range(qp.N + 1):
        for f in ("q", "r"):
            a0 = src.get(f, k)
            if a0.shape[1]:
                src.set(f, k, a0 * g.uniform(-2.0, 3.0, (B, 1)) + 0.3 * g.standard_normal(a0.shape))
    hard = [src.to_qp(i) for i in (155, 896, 1176, 1188, 1267)]     # the five instances of the batch that cycled
    for fam in ("0", "1"):
        os.environ["ACADOS_AMD_WPI"] = fam
        try:
            b = OcpQpGpuBatch.from_qps(hard, _clib=clib)
        finally:
            del os.environ["ACADOS_AMD_WPI"]
        for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
            b.opts_set(f, 1e-8)
        b.opts_set("iter_max", 80)
        assert b.solve() == 0, (b.kernel_name, b.info("status"), b.info("iter"))
        for i, q in enumerate(hard):
            o = OracleQp(q)
            assert o.solve(default_opts(tol_stat=1e-8, iter_max=80)) == 0 and o.iter <= 20
            assert abs(int(b.info("iter")[i]) - o.iter) <= 1, (b.kernel_name, i, b.info("iter"), o.iter)
            compare_with_oracle(lambda k, f: b.get(f, k)[i], o, q, 1e-7, fields=("x", "u", "pi", "lam", "t"))
        # without the conditional corrector the cycle is still there (what the test is about)
        b.opts_set("cond_pred_corr", 0)
        assert b.solve() > 0


def certified_random_structures_case(clib, seeds, bar=1e-9):
    """The device kernels against tests/dense_ref.py::solve_exact -- a dense active-set solve that carries its own optimality certificate
    and has no part in the oracle or the kernels -- on RANDOM STRUCTURES (tests/random_qp.py: general rows, slacks shared by several rows,
    one-sided and masked rows, free and fixed x0, per-stage dims): every structure class the reference's fixtures hold no answers for
    (casadi_tests store inputs only).  Tight tolerances on the device (complementarity 1e-12): what remains is the distance of an
    interior point to the vertex.  Returns {kernel family: worst relative primal distance}."""
    from acados_amd import OcpQpGpuBatch
    from dense_ref import solve_exact, split
    from random_qp import random_structure_qp
    worst = {}
    for seed in seeds:
        qp = random_structure_qp(seed)
        w, off, info = solve_exact(qp)
        assert info["cert"] <= 1e-10 and info["stationarity"] <= 1e-9, (seed, info)
        sol = split(qp, w, off)
        b = OcpQpGpuBatch.from_qps([qp] * 2, _clib=clib)
        for f, v in (("tol_stat", 1e-9), ("tol_eq", 1e-11), ("tol_ineq", 1e-11), ("tol_comp", 1e-12)):
            b.opts_set(f, v)
        b.opts_set("iter_max", 100)
        assert b.solve() == 0, (seed, b.kernel_name, b.info("status"))
        e = 0.0
        for i in range(2):
            for k in range(qp.N + 1):
                for f in ("x", "u", "sl", "su"):
                    ref = sol[f][k]
                    if ref.size:
                        e = max(e, float(np.max(np.abs(b.get(f, k)[i][:ref.size] - ref) / np.maximum(1.0, np.abs(ref)))))
        assert e <= bar, (seed, b.kernel_name, e)
        fam = b.kernel_name.split("<")[0].split("(")[0]
        worst[fam] = max(worst.get(fam, 0.0), e)
    return worst


def bulk_chunk_case(clib, seeds=(3, 22, 41)):
    """C-ABI parity of the input blob entries (include/acados_amd/ocp_qp_gpu_batch.h): the QP data of a batch read as ONE blob
    (_get_bulk_in), written into fresh batches whole (_set_bulk), in uneven chunks (_set_bulk_chunk x 3 + _set_bulk_staged) and by the
    ZERO-COPY GATHER (_host_register + _gather_tables + _gather_run: the blob's words spread over three source arrays per instance in a
    host block the caller owns, some stored negated, read by the device from there): the same blob back, the same solve bit for bit; the
    refusals of the chunk and table entries.  clib = None: the product library"""
    import ctypes as C
    from acados_amd import OcpQpGpuBatch
    from random_qp import random_structure_qp
    for seed in seeds:             # general rows + shared slacks + one-sided rows / box only / per-stage dims
        qp = random_structure_qp(seed)
        B = 37
        g = np.random.default_rng(seed)
        a = OcpQpGpuBatch.from_qps([qp] * B, _clib=clib)
        for k in range(qp.N + 1):
            for f in ("q", "r"):
                v = a.get(f, k)
                if v.shape[1]:
                    a.set(f, k, v * g.uniform(0.5, 1.5, (B, 1)))
        L = a._L
        n = L.ocp_qp_gpu_batch_bulk_len(a._h, 0)
        assert n > 0
        blob = np.zeros((B, n))
        assert L.ocp_qp_gpu_batch_get_bulk_in(a._h, blob.ctypes.data_as(C.c_void_p), 0) == 0
        assert np.isfinite(blob).all() and np.abs(blob).max() > 0
        assert a.solve() == 0
        outs = []
        for how in ("whole", "gather", "chunks"):
            b = OcpQpGpuBatch.from_qps([qp] * B, _clib=clib)     # (structure and index sets; the data is overwritten below)
            if how == "whole":
                assert L.ocp_qp_gpu_batch_set_bulk(b._h, blob.ctypes.data_as(C.c_void_p), 0) == 0
            elif how == "gather":
                P = 3
                slot = g.integers(0, P, n).astype(np.int32)
                cnt = np.bincount(slot, minlength=P)
                off = np.zeros(n, np.int32)
                for s_ in range(P):                                # every source: its words at shuffled offsets, a gap of 5 in front
                    off[slot == s_] = 5 + g.permutation(cnt[s_])
                neg = (g.uniform(size=n) < 0.3).astype(np.uint8)
                order = np.lexsort((off, slot))                    # sorted by (slot, offset) as the header asks
                pos = np.arange(n, dtype=np.int32)[order]
                slot, off, neg = slot[order], off[order], neg[order]
                stride = int(cnt.max()) + 16
                block = np.full((B, P, stride), np.nan)            # the caller's memory: NaN wherever no word lives
                for i in range(B):
                    block[i, slot, off] = np.where(neg, -blob[i, pos], blob[i, pos])
                ptrs = np.array([[block[i, s_].ctypes.data for s_ in range(P)] for i in range(B)], dtype=np.uint64)
                ip = lambda a_: a_.ctypes.data_as(C.c_void_p)
                bad = pos.copy(); bad[0] = n
                assert L.ocp_qp_gpu_batch_gather_run(b._h, 0, ip(ptrs)) == -1                              # no tables yet
                assert L.ocp_qp_gpu_batch_gather_tables(b._h, 0, P, n, ip(slot), ip(off), ip(bad), ip(neg)) == -1   # a word outside the blob
                assert L.ocp_qp_gpu_batch_gather_tables(b._h, 0, 2, n, ip(slot), ip(off), ip(pos), ip(neg)) == -1   # a source that is not there
                assert L.ocp_qp_gpu_batch_gather_tables(b._h, 0, P, n, ip(slot), ip(off), ip(pos), ip(neg)) == 0
                assert L.ocp_qp_gpu_host_register(ip(block), block.nbytes) == 0
                try:
                    # a table that leaves a field out (q of stage 1) on a buffer that has held a whole blob: those positions are ZERO
                    ln = C.c_int(0)
                    o1 = L.ocp_qp_gpu_batch_bulk_offset(b._h, 0, b"q", 1, C.byref(ln))
                    if o1 >= 0 and ln.value > 0:
                        assert L.ocp_qp_gpu_batch_set_bulk_chunk(b._h, ip(blob), 0, B) == 0 and L.ocp_qp_gpu_batch_set_bulk_staged(b._h) == 0
                        keep = (pos < o1) | (pos >= o1 + ln.value)
                        part = [np.ascontiguousarray(a_[keep]) for a_ in (slot, off, pos, neg)]
                        assert L.ocp_qp_gpu_batch_gather_tables(b._h, 0, P, int(keep.sum()), *[ip(a_) for a_ in part]) == 0
                        assert L.ocp_qp_gpu_batch_gather_run(b._h, 0, ip(ptrs)) == 0
                        hole = np.zeros((B, n))
                        assert L.ocp_qp_gpu_batch_get_bulk_in(b._h, ip(hole), 0) == 0
                        expect = blob.copy(); expect[:, o1:o1 + ln.value] = 0.0
                        assert np.array_equal(hole, expect), seed
                        assert L.ocp_qp_gpu_batch_gather_tables(b._h, 0, P, n, ip(slot), ip(off), ip(pos), ip(neg)) == 0
                    assert L.ocp_qp_gpu_batch_gather_run(b._h, 0, ip(ptrs)) == 0
                finally:
                    assert L.ocp_qp_gpu_host_unregister(ip(block)) == 0
            else:
                assert L.ocp_qp_gpu_batch_set_bulk_staged(b._h) == -1          # nothing handed over yet
                for lo, hi in ((0, 5), (5, 30), (30, 37)):
                    part = np.ascontiguousarray(blob[lo:hi])
                    assert L.ocp_qp_gpu_batch_set_bulk_chunk(b._h, part.ctypes.data_as(C.c_void_p), lo, hi - lo) == 0
                    del part                                                    # (hostsim copies at once; the GPU tier syncs below)
                assert L.ocp_qp_gpu_batch_set_bulk_chunk(b._h, blob.ctypes.data_as(C.c_void_p), 30, 8) == -1   # beyond the batch
                assert L.ocp_qp_gpu_batch_set_bulk_staged(b._h) == 0
            back = np.zeros((B, n))
            assert L.ocp_qp_gpu_batch_get_bulk_in(b._h, back.ctypes.data_as(C.c_void_p), 0) == 0
            assert np.array_equal(back, blob), (seed, how)
            assert b.solve() == 0
            assert np.array_equal(b.info("iter"), a.info("iter"))
            for k in range(qp.N + 1):
                for f in ("x", "u", "lam"):
                    assert np.array_equal(b.get(f, k), a.get(f, k)), (seed, how, f, k)
            if how == "chunks":
                # a later round in which a range never arrives is refused (it would scatter the previous round's data for those
                # instances), and the count starts again behind the refusal
                part = np.ascontiguousarray(blob[0:30])
                assert L.ocp_qp_gpu_batch_set_bulk_chunk(b._h, part.ctypes.data_as(C.c_void_p), 0, 30) == 0
                assert L.ocp_qp_gpu_batch_set_bulk_staged(b._h) == -1
                assert L.ocp_qp_gpu_batch_set_bulk_chunk(b._h, blob.ctypes.data_as(C.c_void_p), 0, B) == 0
                assert L.ocp_qp_gpu_batch_set_bulk_staged(b._h) == 0
            outs.append(b)
