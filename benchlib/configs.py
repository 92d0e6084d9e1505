"""the non-headline configurations of BASELINE.json on one GPU (C3, C4, the per-GPU share of C5) and the through-the-boundary leg (a part of bench.py)."""
import os
import subprocess
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from benchlib.cpu import threads_allowed
from benchlib.parity import oracle_error
from benchlib.roofline import MFMA_PROBES, algorithmic_bytes_dims, config_traffic, kernel_symbol, mark_stale, mfma_note_c3, mfma_util, sweep_roofline


def tol_setup(gb):
    for f in ("tol_stat", "tol_eq", "tol_ineq", "tol_comp"):
        gb.opts_set(f, 1e-8)
    gb.opts_set("iter_max", 50)
    gb.opts_set("warm_start", 0)


def run_config(name, gb, qp_of, N, dims, steps=2, check=4, extra=None, section=0, sweep_kernel_name=None):
    """one non-headline configuration: warm-up + `steps` timed solves, statistics, independent residual, oracle sample,
    dominant sweep + roofline fraction (+ the PMC traffic of the same kernel from the per-section summary)"""
    tol_setup(gb)
    gb.solve()
    gb.opts_set("profile", 1)
    gb.scalar("prof_reset")
    gb.opts_set("marker", section)       # section mark for the rocprofv3 summaries (an empty launch, outside timing)
    t0 = time.perf_counter()
    bad = 0
    for _ in range(steps):
        bad += gb.solve()
    dt = (time.perf_counter() - t0) / steps
    gb.opts_set("marker", 0)
    gb.opts_set("profile", 0)
    b_in, b_out = algorithmic_bytes_dims(dims)
    dom, prof, roof = sweep_roofline(gb, steps, b_in + b_out)
    kname = sweep_kernel_name() if sweep_kernel_name else gb.kernel_name
    tiles = bool(gb.condensed_scalar("w16_tiles") if sweep_kernel_name and gb.condensed_kernel_name() else gb.scalar("w16_tiles"))
    roof["kernel"] = f"{kernel_symbol(kname, dom, tiles)} ({dom}) of {kname}"
    tr = mark_stale(config_traffic(section, kernel_symbol(kname, dom, tiles), dom)) if section else None
    roof["traffic"] = tr["avg_main"] if tr else None
    roof["traffic_source"] = tr
    roof["traffic_over_algorithmic"] = (tr["avg_main"] / roof["bytes_per_launch"]) if tr else None
    # the same for the launches in which every instance still iterates: the two averages above are taken over different sets
    # of launches (HIP events: every root-level launch; PMC: those above 1 % of the largest) -- a class with a long tail of
    # nearly empty launches (N = 100: up to 25 iterations for a mean of 12) shows a ratio that is not re-read traffic
    roof["traffic_over_algorithmic_full_launch"] = (tr["full"] / (gb.n_batch * (b_in + b_out))) if tr else None
    roof["traffic_GBps"] = (tr["avg_main"] / (roof["avg_launch_ms"] * 1e-3) / 1e9) if tr and roof["avg_launch_ms"] > 0 else None
    it = gb.info("iter")
    res = gb.res_compute()
    out = {"workload": name, "batch": gb.n_batch, "solves_per_s": gb.n_batch / dt, "ms_per_step": dt * 1e3,
           "kernel": gb.kernel_name, "mean_iter": float(it.mean()), "max_iter": int(it.max()),
           "failures": int((gb.info("status") != 0).sum()), "max_kkt_residual_independent": float(res.max()),
           "bytes_per_instance": b_in + b_out, "roofline": roof,
           "condense_expand_ms": gb.scalar("time_xcond") * 1e3}
    if check:
        idx = np.unique(np.linspace(0, gb.n_batch - 1, check).astype(int))
        oe = oracle_error(gb, qp_of, idx, N)
        out["max_rel_primal_err_vs_oracle"] = oe["same_tol_max"]
        out["oracle_checked_instances"] = int(idx.size)
        out["oracle_check"] = oe
    if extra:
        out.update(extra)
    return out


def polish_leg(gb, qp_of, N, check):
    """the batch as it is configured + the opt-in terminal polishing step (option "polish": converged instances that hold a balanced
    pair min(lam, t) > 1e-3 max(lam, t) run one more iteration; status / iter unchanged): rate, how many instances it touched, and the
    distance to THE solution (oracle at complementarity 1e-12) it leaves -- `max_rel_primal_err_vs_oracle` of this record is that
    distance's maximum (the oracle has no polishing step: a same-tolerance comparison would measure the ORACLE's distance)"""
    gb.opts_set("polish", 1)
    gb.solve()
    t0 = time.perf_counter()
    bad = 0
    for _ in range(2):
        bad += gb.solve()
    dt = (time.perf_counter() - t0) / 2
    it = gb.info("iter")
    rec = {"batch": gb.n_batch, "solves_per_s": gb.n_batch / dt, "ms_per_step": dt * 1e3, "mean_iter": float(it.mean()), "max_iter": int(it.max()),
           "failures": int((gb.info("status") != 0).sum()), "max_kkt_residual_independent": float(gb.res_compute().max()),
           "polished": int(gb.scalar("polished")), "polish_reverted": int(gb.scalar("polish_reverted"))}
    if check:
        idx = np.unique(np.linspace(0, gb.n_batch - 1, check).astype(int))
        oe = oracle_error(gb, qp_of, idx, N, same_tol=False)
        rec.update({"oracle_check": oe, "max_rel_primal_err_vs_oracle": oe["dist_to_solution"]["max"], "oracle_checked_instances": int(idx.size),
                    "err_reference": "oracle at complementarity 1e-12 (distance to the solution)"})
    gb.opts_set("polish", 0)
    return rec


def other_configs(c2_batch, c2_data, args):
    """C3, C4 and the per-GPU share of C5 on this GPU (BASELINE.json configs[2..4])"""
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import (C5_CLASSES, chain_soft_batch, chain_soft_dims, chain_soft_instance_qp,
                                       fill_chain_soft_batch, fill_lqr_batch, lqr_dims, lqr_instance_qp, random_lqr_batch)
    out = {}
    N = 50
    # C3: the C2 batch itself with partial condensing N2 = 10 (same data, resident)
    c2_batch.opts_set("cond_N", 10)
    # (the IPM sweeps of a condensed solve run on the condensed batch's kernels)
    out["C3"] = run_config("C2 data with partial condensing to N2=10 (BASELINE configs[2]), batch 65,536", c2_batch,
                           lambda i: lqr_instance_qp(c2_data, i, N), N, lqr_dims(N, 8, 3), steps=2, check=args.check_configs,
                           section=1, sweep_kernel_name=lambda: c2_batch.condensed_kernel_name() or c2_batch.kernel_name)
    out["C3"]["cond_N_active"] = int(c2_batch.scalar("cond_N_active"))
    out["C3"]["mfma"] = mfma_note_c3(c2_batch)
    ck = c2_batch.condensed_kernel_name()
    if ck:
        pk = {3: "km_pcond", 2: "kz_pcond", 1: "k_pcond", 0: "kw_pcond"}.get(int(c2_batch.scalar("pcond_kernel")), "pcond")
        ek = {1: "k_pexpand", 0: "kw_pexpand"}.get(int(c2_batch.scalar("pexpand_kernel")), "pexpand")
        out["C3"]["kernel"] = f"{pk} + {ck} + {ek}"
    c2_batch.opts_set("cond_N", N)
    # C4
    N4, B4 = 40, args.c4_batch
    d4 = chain_soft_batch(N=N4, batch=B4, seed=1)
    g4 = OcpQpGpuBatch(chain_soft_dims(N4), B4)
    fill_chain_soft_batch(g4, d4, N4)
    out["C4"] = run_config(f"chain nx=24 nu=3, 4 soft state bounds + 4 soft general rows, ns=8, N=40 (BASELINE configs[3]), batch {B4}",
                           g4, lambda i: chain_soft_instance_qp(d4, i, N4), N4, chain_soft_dims(N4), steps=2, check=args.check_configs, section=2)
    out["C4"]["mfma"] = dict(MFMA_PROBES, used=bool(g4.scalar("w16_tiles")),
                             kernels={"kt_factor<24,3,4> (Riccati factor sweep with general rows + slacks on 4 x 4 tiles; M += A' diag(gamma) A "
                                      "of the general rows as one more chain of tile products)": bool(g4.scalar("w16_tiles"))},
                             tile_fill="n = 27 -> 28 = 7 tiles (one padding row), nx = 24 = 6 tiles: 0.96",
                             utilisation=mfma_util("_c4_c5"))
    # The rate QUOTED for C4 (configs.C4.solves_per_s) is the one at which the north-star parity bar holds on the >= 1,024-instance
    # sample (max relative primal error vs the oracle <= 1e-6): the soft-constrained class leaves so flat a 1e-8 ball that two runs
    # of ONE algorithm differ by 3e-6 inside it, so the bar can only be promised at the opt-in tighter exit
    # tol_comp_soft_scale 1e-3 (complementarity at 1e-11, both sides).  The library's DEFAULT stays the reference's stopping
    # semantics (ocp_qp_hpipm.c:104-107): that run is `plain_exit`, the secondary number.
    plain = {k: out["C4"][k] for k in ("solves_per_s", "ms_per_step", "mean_iter", "max_iter", "failures", "max_kkt_residual_independent")}
    for k in ("max_rel_primal_err_vs_oracle", "oracle_checked_instances", "oracle_check"):
        if k in out["C4"]:
            plain[k] = out["C4"].pop(k)
    plain["exit_rule"] = "tol_comp as given (1e-8): the library default = the reference's semantics"
    g4.opts_set("tol_comp_soft_scale", 1e-3)
    g4.solve()
    t0 = time.perf_counter()
    bad = 0
    for _ in range(2):
        bad += g4.solve()
    dt = (time.perf_counter() - t0) / 2
    it = g4.info("iter")
    tight = {"solves_per_s": B4 / dt, "ms_per_step": dt * 1e3, "mean_iter": float(it.mean()), "max_iter": int(it.max()),
             "failures": int((g4.info("status") != 0).sum()), "max_kkt_residual_independent": float(g4.res_compute().max())}
    if args.check_configs:
        idx = np.unique(np.linspace(0, B4 - 1, args.check_configs).astype(int))
        oe = oracle_error(g4, lambda i: chain_soft_instance_qp(d4, i, N4), idx, N4)
        tight.update({"oracle_check": oe, "max_rel_primal_err_vs_oracle": oe["same_tol_max"], "oracle_checked_instances": int(idx.size)})
    out["C4"]["plain_exit"] = plain
    out["C4"]["tight_exit"] = dict(tight)
    g4.opts_set("tol_comp_soft_scale", 1.0)
    if args.polish_legs:
        out["C4"]["polish"] = polish_leg(g4, lambda i: chain_soft_instance_qp(d4, i, N4), N4, args.check_configs)
    quoted = "tight_exit" if (not args.check_configs or plain.get("max_rel_primal_err_vs_oracle", 0.0) > 1e-6) else "plain_exit"
    out["C4"].update(tight if quoted == "tight_exit" else plain)
    out["C4"]["quoted_exit"] = quoted
    out["C4"]["exit_rule"] = {"quoted": quoted, "tol_comp_soft_scale": 1e-3 if quoted == "tight_exit" else 1.0,
                              "effective_tol_comp": 1e-11 if quoted == "tight_exit" else 1e-8,
                              "note": "solves_per_s of this record = the exit rule at which max_rel_primal_err_vs_oracle <= 1e-6 holds on the sample "
                                      "(north_star bar); plain_exit = the library default (tol_comp as given, ocp_qp_hpipm.c:104-107); tight_exit = "
                                      "opt-in tol_comp_soft_scale 1e-3 (complementarity at 1e-11, DESIGN.md 3); roofline / traffic of the record "
                                      "are the plain run's launches (same kernels, same bytes per launch)"}
    del g4, d4
    # C2 once more with complementarity at 1e-10 (a user's choice for a hard-constrained class): the distance to the solution is
    # the tolerance's -- at 1e-8 x 4 an IPM stops on the central path, t = mu / lam* on a weakly active row.  1e-10 is the cheapest
    # exit at which the whole 1,024 sample is within 1e-6 of THE solution (profiles/r06_polish_sweep.txt: 1e-9 leaves 8, 1e-10 none
    # at -4.4 % rate, 1e-11 none at -8.7 %; the opt-in polishing step needs -11 % for the same)
    c2_batch.opts_set("tol_comp", 1e-10)
    c2_batch.solve()
    t0 = time.perf_counter()
    bad = c2_batch.solve()
    dt = time.perf_counter() - t0
    it = c2_batch.info("iter")
    out["C2_tol_comp_1e-10"] = {"workload": "the headline batch with tol_comp 1e-10 (tol_stat / eq / ineq 1e-8): the rate at which every sampled instance is within 1e-6 of the solution", "batch": c2_batch.n_batch,
                                "solves_per_s": c2_batch.n_batch / dt, "ms_per_step": dt * 1e3, "mean_iter": float(it.mean()),
                                "max_iter": int(it.max()), "failures": int(bad)}
    if args.check_configs:
        idx = np.unique(np.linspace(0, c2_batch.n_batch - 1, args.check_configs).astype(int))
        out["C2_tol_comp_1e-10"]["oracle_check"] = oracle_error(c2_batch, lambda i: lqr_instance_qp(c2_data, i, N), idx, N, same_tol=False)
    c2_batch.opts_set("tol_comp", 1e-8)
    if args.polish_legs:
        out["C2_polish"] = dict(polish_leg(c2_batch, lambda i: lqr_instance_qp(c2_data, i, N), N, args.check_configs),
                                workload="the headline batch at the plain 1e-8 exit + the opt-in terminal polishing step (option polish)")
    # C5: the per-GPU share of 524,288 instances on 8 GPUs, split equally over the 9 shape classes.  Every class is one
    # device batch with its own HIP stream; the classes are solved CONCURRENTLY (one host thread per class, the solve
    # call releases the GIL) -- small, latency-bound batches overlap on the chip -- and, for reference, one after the other
    from acados_amd.shape_classes import ConcurrentClasses
    per_class = (524288 // 8) // len(C5_CLASSES)
    batches = []
    for ci, (nx, nu, Nc) in enumerate(C5_CLASSES):
        dc = random_lqr_batch(N=Nc, nx=nx, nu=nu, batch=per_class, seed=200 + ci)
        gc = OcpQpGpuBatch(lqr_dims(Nc, nx, nu), per_class)
        fill_lqr_batch(gc, dc, Nc)
        tol_setup(gc)
        gc.solve()                                     # warm-up
        batches.append((f"nx={nx} nu={nu} N={Nc}", gc, (lambda dc=dc, Nc=Nc: (lambda i: lqr_instance_qp(dc, i, Nc)))(), Nc, lqr_dims(Nc, nx, nu)))
    # ... "additionally one class with nx switching 12 -> 4 at k = N/2 via a non-square A" (SURVEY.md 8d): per-stage dims
    # inside one padded kernel shape; same share as the other classes
    from acados_amd.generators import fill_multiphase_batch, multiphase_batch, multiphase_dims, multiphase_instance_qp
    Nm = 50
    dm = multiphase_batch(N=Nm, batch=per_class)
    gm = OcpQpGpuBatch(multiphase_dims(Nm), per_class)
    fill_multiphase_batch(gm, dm)
    tol_setup(gm)
    gm.solve()
    batches.append((f"multi-phase nx=12->4 at k={Nm // 2} nu=3 N={Nm}", gm, lambda i: multiphase_instance_qp(dm, i), Nm, multiphase_dims(Nm)))
    with ConcurrentClasses([b[1] for b in batches]) as cc:
        cc.solve()                                     # warm-up of the concurrent path
        t0 = time.perf_counter()
        bad_conc = cc.solve()
        t_conc = time.perf_counter() - t0
    classes, tot_t, tot_n, bad, res_max = [], 0.0, 0, 0, 0.0
    worst_frac = None
    per_class_check = -(-args.check_configs // len(batches)) if args.check_configs else 0     # the sample is spread over the classes
    for ci, (label, gc, qp_of_c, Nc, dims_c) in enumerate(batches):
        r = run_config(label, gc, qp_of_c, Nc, dims_c, steps=1, check=per_class_check, section=3 + ci)
        classes.append({k: r[k] for k in ("workload", "batch", "solves_per_s", "ms_per_step", "kernel", "mean_iter", "failures",
                                          "max_kkt_residual_independent")}
                       | {"frac": r["roofline"]["frac"], "dominant": r["roofline"]["kernel"], "avg_launch_ms": r["roofline"]["avg_launch_ms"],
                          "traffic": r["roofline"]["traffic"], "traffic_over_algorithmic": r["roofline"]["traffic_over_algorithmic"],
                          "traffic_over_algorithmic_full_launch": r["roofline"].get("traffic_over_algorithmic_full_launch"),
                          "traffic_GBps": r["roofline"]["traffic_GBps"],
                          "max_rel_primal_err_vs_oracle": r.get("max_rel_primal_err_vs_oracle"),
                          "oracle_checked_instances": r.get("oracle_checked_instances", 0),
                          "dist_to_solution": (r.get("oracle_check") or {}).get("dist_to_solution")})
        tot_t += r["ms_per_step"] * 1e-3
        tot_n += gc.n_batch
        bad += r["failures"]
        res_max = max(res_max, r["max_kkt_residual_independent"])
        if worst_frac is None or r["ms_per_step"] > worst_frac[0]:
            worst_frac = (r["ms_per_step"], r["roofline"])
    del batches
    out["C5_share"] = {"workload": f"mixed shape classes nx in {{4,12,24}} x N in {{20,50,100}}, {per_class} instances each = per-GPU share of "
                                   f"524,288 on 8 GPUs (BASELINE configs[4]), plus the multi-phase class (nx 12 -> 4 at N/2, same share); ten device batches solved concurrently on their own streams, the longest class on a high-priority one (acados_amd/shape_classes.py)",
                       "batch": tot_n, "solves_per_s": tot_n / t_conc, "seconds": t_conc, "failures": bad + bad_conc,
                       "solves_per_s_one_after_the_other": tot_n / tot_t, "seconds_one_after_the_other": tot_t,
                       "max_kkt_residual_independent": res_max, "roofline_of_slowest_class": worst_frac[1], "classes": classes,
                       "max_rel_primal_err_vs_oracle": max((c["max_rel_primal_err_vs_oracle"] or 0.0) for c in classes),
                       "oracle_checked_instances": sum(c["oracle_checked_instances"] for c in classes)}
    ds = [c["dist_to_solution"] for c in classes if c.get("dist_to_solution")]
    if ds:      # per-class statistics pooled: max / counts exact, median and q99 = the largest class value (an upper bound of the pooled one)
        out["C5_share"]["dist_to_solution"] = {"median": max(d["median"] for d in ds), "q99": max(d["q99"] for d in ds), "max": max(d["max"] for d in ds),
                                               "above_1e-6": sum(d["above_1e-6"] for d in ds), "instances": sum(d["instances"] for d in ds),
                                               "pooled": "over the classes: max and counts exact; median / q99 = the largest class value"}
    return out


PCIE_PEAK_GBS = 63.0                    # /opt/skills/guides/MI355X_MICROARCH.md: PCIe Gen5 x16, 63 GB/s (spec)


def write_driver_qp(qp, path):
    """a QP in the text format of the C drivers under tests/mock_acados (qp_loader.h: dims, index sets, column-major fields)"""
    d = qp.dims
    with open(path, "w") as f:
        f.write(f"{qp.N}\n")
        for k in range(qp.N + 1):
            f.write(f"dims {k} {d.nx[k]} {d.nu[k]} {d.nbx[k]} {d.nbu[k]} {d.ng[k]} {d.ns[k]} {d.nbxe[k]}\n")
        for k in range(qp.N + 1):
            for name in ("idxb", "idxs_rev", "idxe"):
                v = np.asarray(getattr(qp, name)[k]).astype(int).ravel()
                if v.size:
                    f.write(f"{name} {k} {v.size} " + " ".join(str(int(x)) for x in v) + "\n")
            for name in ("A", "B", "b", "Q", "R", "S", "q", "r", "C", "D", "lbu", "lbx", "lg", "ubu", "ubx", "ug", "lls", "lus",
                         "lbu_mask", "lbx_mask", "lg_mask", "ubu_mask", "ubx_mask", "ug_mask", "lls_mask", "lus_mask", "Zl", "Zu", "zl", "zu"):
                if k == qp.N and name in ("A", "B", "b"):
                    continue
                v = np.ravel(np.asarray(getattr(qp, name)[k], dtype=float), order="F")
                if v.size:
                    f.write(f"{name} {k} {v.size} " + " ".join(repr(float(x)) for x in v) + "\n")


def pcie_rates(mb=256, reps=3):
    """what THIS box moves over PCIe from / to pinned host memory (one stream, large copies): the practical ceiling of the boundary leg,
    beside the 63 GB/s spec figure of the guide"""
    try:
        import torch
        h = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
        d = torch.empty(mb << 20, dtype=torch.uint8, device="cuda")
        out = {}
        for name, (dst, src) in (("h2d", (d, h)), ("d2h", (h, d))):
            dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
            best = 1e9
            for _ in range(reps):
                t0 = time.perf_counter(); dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            out[name + "_GBps"] = (mb << 20) / best / 1e9
        return out
    except Exception as e:       # (no torch / no device: the leg still reports its times)
        return {"error": str(e)[:80]}


def boundary_c3(n=4096, reps=4, larger=16384):
    """THROUGH THE BOUNDARY: n C3-shaped capsules (N = 50, nx = 8, nu = 3, cond_N = 10), each an acados `ocp_qp_in` / `ocp_qp_out` pair
    (HPIPM structs, panel-major BLASFEO storage: the stand-ins of tests/mock_hpipm -- the real ones are empty submodules of the
    reference) with the reference's own 22-slot solver object around the two plugin slots, ONE call of
    ocp_qp_gpu_xcond_solver_acados_evaluate_batch: the QP data of every qp_in to the device (host threads fill a pinned blob under chunked
    host->device copies, or -- smaller calls and the vector part of an RTI feedback step -- the device gathers it from the capsules' own
    registered memory), condensing + IPM + expansion on the device, one copy back, host threads write every qp_out.  `at_<larger>`: the
    same at the second size the review asked for.
    What a user of `_acados_batch_solve` gets per call, PCIe included.  The binary is built by
    integration/Makefile where the reference tree exists and travels with the snapshot."""
    exe = os.path.join(ROOT, "integration", "_ref_build", "ref_xcond_driver")
    if not os.path.exists(exe):
        return {"skipped": "integration/_ref_build/ref_xcond_driver not built (needs the reference tree at build time)"}
    import tempfile
    from acados_amd.generators import lqr_dims, lqr_instance_qp, random_lqr_batch
    N = 50
    d = tempfile.mkdtemp(prefix="boundary_")
    f = os.path.join(d, "qp.txt")
    write_driver_qp(lqr_instance_qp(random_lqr_batch(N=N, batch=1, seed=5), 0, N), f)
    threads = threads_allowed()
    env = dict(os.environ, OMP_NUM_THREADS=str(threads))
    env.pop("ACADOS_AMD_WPI_BATCH_MAX", None)
    def run(n_):
        r = subprocess.run([exe, "batch", str(n_), f, os.path.join(d, "out.bin"), "--cond-N", "10", str(reps)], capture_output=True, text=True, env=env)
        try:
            os.remove(os.path.join(d, "out.bin"))
        except OSError:
            pass
        if r.returncode != 0 or not r.stdout:
            return None, (r.stderr or r.stdout)[-300:]
        h = r.stdout.splitlines()[0].split()
        return {h[i]: float(h[i + 1]) for i in range(1, len(h) - 1, 2)}, None
    info, err = run(n)
    if info is None:
        return {"error": err}
    big = None
    if larger and larger > n:
        bi, berr = run(larger)
        big = ({"batch": larger, "solves_per_s": larger / (bi["ms_per_call"] * 1e-3), "ms_per_step": bi["ms_per_call"], "failures": int(bi.get("status", 0) != 0),
                "rti_feedback_solves_per_s": larger / (bi["rti_feedback_ms"] * 1e-3) if bi.get("rti_feedback_ms") else None,
                "rti_feedback_ms": bi.get("rti_feedback_ms"), "max_kkt_residual_reference_entry": bi.get("res_max")} if bi else {"error": berr})
    b_in, b_out = algorithmic_bytes_dims(lqr_dims(N, 8, 3))
    t = info["ms_per_call"] * 1e-3
    pcie = n * (b_in + b_out)
    return {"workload": f"{n} C3-shaped capsules in acados structs (panel-major stand-ins), reference 22-slot solver objects, one fused batch call "
                        f"(ocp_qp_gpu_xcond_solver_acados_evaluate_batch), best of {reps}",
            "batch": n, "solves_per_s": n / t, "ms_per_step": t * 1e3, "failures": int(info.get("status", 0) != 0), "host_threads": int(info.get("threads", threads)),
            "max_kkt_residual_reference_entry": info.get("res_max"), "fused_vs_per_capsule_orchestration": info.get("fused_vs_orchestrated"),
            "phases_ms": {k: info[k] for k in ("unpack_in_ms", "copy_and_device_ms", "device_solve_ms", "pack_out_ms") if k in info},
            "zero_copy_gather": int(info.get("zero_copy", 0)), f"at_{larger}": big,
            # the same capsules as the two halves of an RTI step: preparation (everything to the device, matrices condensed there) is off the
            # control loop's critical path; the FEEDBACK call reads and sends only the vector members of every qp_in
            "rti_feedback": ({"solves_per_s": n / (info["rti_feedback_ms"] * 1e-3), "ms_per_step": info["rti_feedback_ms"],
                              "preparation_ms": info.get("rti_preparation_ms"), "failures": int(info.get("rti_status", 0) != 0),
                              "vs_one_call": info.get("rti_vs_one_call"), "upload_bytes_per_qp": 8 * int(info.get("rti_feedback_upload_doubles", 0)),
                              "phases_ms": {k: info["fb_" + k] for k in ("unpack_in_ms", "copy_and_device_ms", "pack_out_ms") if "fb_" + k in info},
                              "pcie_bytes": n * (8 * int(info.get("rti_feedback_upload_doubles", 0)) + b_out)}
                             if info.get("rti_feedback_ms") else None),
            "pcie_bytes": pcie, "pcie_GBps": pcie / t / 1e9, "pcie_peak_GBps": PCIE_PEAK_GBS, "pcie_frac": pcie / t / 1e9 / PCIE_PEAK_GBS,
            "pcie_measured": pcie_rates(),
            "pcie_note": "algorithmic bytes of SURVEY 8d per QP (85,336 in + 12,720 out; the blob also carries 0/1 masks) over the WHOLE call; "
                         "every call re-reads every member array of every qp_in (ocp_nlp_common.c:2797-2894: they alias ocp_nlp memory)"}
