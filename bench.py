#!/usr/bin/env python3
"""bench.py -- OCP-QP solves/sec on MI355X (BASELINE.json metric), contract of the driver.

A "step" = one cold-start solve of the whole batch of synthetic OCP-QPs (every IPM iteration of
every instance), inputs already packed and resident in HBM when the timed region starts.
Workload at N=1 GPU: BASELINE.json configs[1] (C2) -- random LQR-like OCP-QP, N=50, nx=8, nu=3,
batch=65,536, u-box + x0 equality, tolerances 1e-8, iter_max 50 (SURVEY.md 8d "C2 input").
N>1: one process per GPU (torch.distributed.run), the batch is sharded by instance -- every rank
solves its own 65,536 instances (weak scaling), no data-path collective; AFTER the timed region the
full solution payload {ux, pi, lam, t, status, iter, solve time} of every rank is gathered with ONE
RCCL all-gather over xGMI from device buffers through the library's own collective entry
(ocp_qp_gpu_batch_gather; its time is reported as gather_ms).

Extra objects on the JSON line:
  roofline     dominant kernel (by accumulated HIP-event time inside the timed region, events on the
               stream the kernels are launched on): algorithmic bytes per launch / avg duration vs the
               8 TB/s HBM peak.  Algorithmic bytes per launch = instances the launch still processes x
               98,056 B (unique QP input 85,336 B + iterate/solution 12,720 B per solve).
               `traffic` = HBM bytes per launch of the SAME kernel over the SAME set of launches from the
               rocprofv3 PMC passes of tools/profile_round.sh (FETCH_SIZE / WRITE_SIZE, separate passes,
               gfx950 correction of MI355X_MICROARCH.md), read from the newest profiles/*_pmc_traffic.json
               whose recorded commit is reported next to it; `full_launch` repeats both for the launches in
               which every instance is still iterating.
  configs      the other single-GPU configurations of BASELINE.json (C3, C4, the per-GPU share of C5),
               each with solves/s, iterations, failures, independently recomputed KKT residual, oracle
               error on a sample, dominant kernel and its roofline fraction (rank 0, N=1 only).
  cpu_baseline the oracle (restated CPU port, NOT HPIPM: its sources are absent from the reference
               tree) on a bounded sample of the same workload on the host cores, rank 0, N=1 only:
               one-thread rate, best thread count of a sweep, cores visible / allowed.
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0                   # same guide: what a float4 copy sustains on this part (the practical ceiling of a streaming kernel)
SWEEPS = ("back_fact", "fwd_aff", "back_rhs", "fwd_corr")
CLASSES = SWEEPS + ("init", "finalize")


def algorithmic_bytes_dims(d):
    """unique QP input read once + solution written once, sizes as colmaj_ocp_qp_in_calculate_size
    (ocp_qp_common_frontend.c:67-86) + ux, pi, lam, t.  C2: 85,336 + 12,720 = 98,056 B."""
    N = int(d.N)
    nx, nu, nb, ng, ns = (np.asarray(getattr(d, n), dtype=np.int64) for n in ("nx", "nu", "nb", "ng", "ns"))
    nx1 = nx[1:]
    dbl_in = int(np.sum(nx1 * nx[:N] + nx1 * nu[:N] + nx1)
                 + np.sum(nx * nx + nu * nx + nu * nu + nx + nu + 2 * nb + ng * (nx + nu) + 2 * ng + 4 * ns + 2 * ns))
    int_in = int(np.sum(nb + np.where(ns > 0, nb + ng, 0)))      # idxb; idxs_rev where a stage has slacks
    dbl_out = int(np.sum(nx + nu + 2 * ns) + np.sum(nx1) + 2 * np.sum(2 * (nb + ng + ns)))
    return 8 * dbl_in + 4 * int_in, 8 * dbl_out


def git_head():
    """commit of the benched tree: git where there is a checkout, else the stamp __graft_entry__.build() leaves next to
    the library (the GPU box receives a snapshot without .git)"""
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        try:
            return open(os.path.join(ROOT, "acados_amd", "csrc", "BUILD_COMMIT")).read().strip()
        except Exception:
            return None


LINE_LIMIT = 4096      # bytes; the driver keeps an 8 KB tail of stdout and parses its LAST line (round 4's 22 KB line was lost)


def _r(v, sig=5):
    """floats at `sig` significant digits (the detail file keeps full precision)"""
    if isinstance(v, float):
        return float(f"{v:.{sig}g}") if np.isfinite(v) else None
    return v


def _pick(d, keys, sig=5):
    return {k: _r(d[k], sig) for k in keys if isinstance(d, dict) and k in d and not isinstance(d[k], (dict, list))}


DIST_KEYS = ("median", "q99", "max", "above_1e-6", "instances")


def _dist(rec):
    """{median, q99, max, above_1e-6, instances} of the relative primal distance between what the device returns and THE solution
    (oracle at complementarity 1e-12; oracle_error's `dist_to_solution`) of one configuration record, or None"""
    d = rec.get("dist_to_solution") or (rec.get("oracle_check") or {}).get("dist_to_solution")
    return _pick(d, DIST_KEYS, 3) if d else None


def compact_line(out, detail_path=None):
    """The ONE line the driver parses: headline fields, `config`, `ipm`, `roofline` (scalars + the source of the PMC traffic),
    `cpu_baseline` (value, cores, kind, sample, one_thread) and one short record per other configuration.  Everything else
    (per-class tables, launch histograms, MFMA probe notes, oracle distance statistics, gather detail) goes to the detail file
    named in `detail`.  Asserted < LINE_LIMIT by tests/test_bench_line.py on the committed round-4 line."""
    head = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: _r(out[k], 7) for k in head if k in out}
    cfg = out.get("config", {})
    line["config"] = {k: cfg[k] for k in ("workload", "batch_per_gpu", "global_batch", "parallelism", "kernel", "commit") if k in cfg}
    if "ipm" in out:
        line["ipm"] = _pick(out["ipm"], ("mean_iter", "max_iter", "failures", "max_kkt_residual_independent", "max_rel_primal_err_vs_oracle",
                                         "oracle_checked_instances", "launches_per_step", "wave_max_iter_mean"), 4)
        if out["ipm"].get("iter_hist"):
            line["ipm"]["iter_hist"] = out["ipm"]["iter_hist"]
        d = _dist(out["ipm"])
        if d:
            line["ipm"]["dist_to_solution"] = d
    ro = out.get("roofline")
    if ro:
        r = _pick(ro, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "bytes_per_launch",
                       "avg_launch_ms", "launches_timed", "whole_solve_frac", "traffic_GBps", "traffic_frac_of_sustained_copy"))
        ts = ro.get("traffic_source")
        r["traffic_source"] = _pick(ts, ("file", "commit", "stale")) if ts else None
        fl = ro.get("full_launch")
        if fl:
            r["full_launch_traffic_over_algorithmic"] = _r(fl.get("traffic_over_algorithmic"))
        mf = ro.get("mfma")
        if mf:
            r["mfma_utilisation"] = mf.get("mfma_utilisation")
        line["roofline"] = r
    cb = out.get("cpu_baseline")
    if cb:
        c = _pick(cb, ("value", "unit", "cores", "kind", "one_thread", "mean_iter"))
        c["sample"] = f"{cb.get('unique', '')} C2 instances (seed 0), OpenMP over instances, restated CPU oracle (not HPIPM: sources absent)".strip()
        line["cpu_baseline"] = c
    if out.get("gather"):
        line["gather"] = _pick(out["gather"], ("ms", "ranks", "GBps_received_per_rank", "slice_matches_getters", "gather_to_root_ms", "error"), 4)
    if "configs" in out:
        cs = {}
        for name, c in out["configs"].items():
            ro_c = c.get("roofline") or c.get("roofline_of_slowest_class") or {}
            rec = _pick(c, ("batch", "solves_per_s", "ms_per_step", "mean_iter", "failures", "max_rel_primal_err_vs_oracle",
                            "condense_expand_ms", "solves_per_s_one_after_the_other", "polished"), 4)
            rec.update({"frac": _r(ro_c.get("frac"), 3), "traffic_over_algorithmic": _r(ro_c.get("traffic_over_algorithmic"), 3)})
            mf = c.get("mfma") or {}
            u = (mf.get("utilisation") or {}).get("kernels") if isinstance(mf.get("utilisation"), dict) else None
            if u:
                rec["mfma_utilisation"] = _r(max((k.get("mfma_utilisation") or 0.0) for k in u.values()), 3)
            d = _dist(c)
            if d:
                rec["dist_to_solution"] = d
            if "quoted_exit" in c:
                rec["quoted_exit"] = c["quoted_exit"]
            for leg in ("plain_exit", "tight_exit", "polish"):   # C4: the other exit rules beside the one the record's rate is quoted at
                if leg in c:
                    rec[leg + "_solves_per_s"] = _r(c[leg].get("solves_per_s"), 4)
                    if c[leg].get("max_rel_primal_err_vs_oracle") is not None:
                        rec[leg + "_err_vs_oracle"] = _r(c[leg]["max_rel_primal_err_vs_oracle"], 3)
            if "classes" in c:          # C5: one number per class, in the order of the detail file
                rec["class_solves_per_s"] = [_r(k["solves_per_s"], 3) for k in c["classes"]]
                rec["class_frac"] = [_r(k["frac"], 2) for k in c["classes"]]
            cs[name] = {k: v for k, v in rec.items() if v is not None}
        line["configs"] = cs
    for k in ("failures", "pack_s", "hbm_bytes_per_gpu"):
        if k in out and k not in line:
            line[k] = _r(out[k], 4)
    if detail_path:
        line["detail"] = detail_path
    s = json.dumps(line, separators=(",", ":"))
    if len(s) >= LINE_LIMIT:           # never lose the line: drop the optional parts in order of weight
        for k in ("configs", "gather", "ipm"):
            if k == "configs" and "configs" in line:
                line["configs"] = {n: _pick(c, ("solves_per_s", "frac", "failures"), 4) for n, c in line["configs"].items()}
            else:
                line.pop(k, None)
            s = json.dumps(line, separators=(",", ":"))
            if len(s) < LINE_LIMIT:
                break
    return s


def emit(out, args):
    """full object -> detail file, compact object -> the last line of stdout"""
    path = getattr(args, "detail_file", None) or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
    rel = None
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
        rel = os.path.relpath(path, ROOT)
    except OSError:
        pass
    # the JSON line is the LAST thing on stdout: RCCL prints a version banner through C stdio when its first communicator
    # comes up -- push that out first
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(compact_line(out, rel), flush=True)


def kernel_symbol(name, cls, tiles=False):
    """profile class -> kernel function of the family `name` (a batch's kernel_name) runs on (what rocprofv3 lists);
    tiles: the factor sweep of the two-rows family runs on 4 x 4 MFMA tiles (kt_factor, scalar "w16_tiles")"""
    if tiles and name.startswith("w16r") and cls == "back_fact":
        return "kt_factor"
    fam = ("kbs" if name.startswith("1tpi-pipe") else "kb" if name.startswith("1tpi-box") else "ky" if name.startswith("w16r") else "kx" if name.startswith("w16")
           else "kw" if name.startswith("wpi") else "k")
    table = {"kbs": {"back_fact": "kbs_factor", "fwd_aff": "kbs_forward", "back_rhs": "kbs_backrhs", "fwd_corr": "kbs_forward"},
             "kb": {"back_fact": "kb_factor", "fwd_aff": "kb_forward", "back_rhs": "kb_backrhs", "fwd_corr": "kb_forward"},
             "kx": {"back_fact": "kx_factor", "fwd_aff": "kx_fwd", "back_rhs": "kx_backrhs", "fwd_corr": "kx_fwd"},
             "ky": {"back_fact": "ky_factor", "fwd_aff": "ky_fwd", "back_rhs": "ky_backrhs", "fwd_corr": "ky_fwd"},
             "kw": {"back_fact": "kw_factor", "fwd_aff": "kw_fwd", "back_rhs": "kw_backrhs", "fwd_corr": "kw_fwd"},
             "k": {"back_fact": "k_backward", "fwd_aff": "k_forward", "back_rhs": "k_backward", "fwd_corr": "k_forward"}}
    return table[fam].get(cls, cls)


def sweep_roofline(gb, steps, bytes_per_instance):
    """roofline object of the dominant sweep of the solves profiled since the last prof_reset"""
    B = gb.n_batch
    prof = {c: (gb.scalar(f"prof_ms_{c}"), int(gb.scalar(f"prof_cnt_{c}"))) for c in CLASSES}
    dom = max(SWEEPS, key=lambda c: prof[c][0])
    dom_ms, dom_cnt = prof[dom]
    iters = gb.info("iter")
    # units one launch processes: launch j of the factor kernel sees the instances that have not converged before
    # iteration j (iter >= j), the other sweeps those with iter > j; only root-level launches are timed (the last
    # survivors of a one-instance-per-lane batch continue on a small sub-batch)
    per_solve = max(dom_cnt // max(steps, 1), 1)
    hist = np.bincount(iters, minlength=per_solve + 1)
    still = B - np.cumsum(hist)                      # still[j] = instances with iter > j
    units = [(B if j == 0 else int(still[j - 1])) if dom == "back_fact" else int(still[j]) for j in range(per_solve)]
    avg_s = dom_ms * 1e-3 / max(dom_cnt, 1)
    per_launch = float(np.mean(units)) * bytes_per_instance
    achieved = per_launch / avg_s / 1e9 if avg_s > 0 else 0.0
    return dom, prof, {"bound": "hbm", "kernel": f"{kernel_symbol(gb.kernel_name, dom, bool(gb.scalar('w16_tiles')))} ({dom}) of {gb.kernel_name}", "sweep": dom,
                       "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                       "bytes_per_launch": per_launch, "units_per_launch": units, "avg_launch_ms": avg_s * 1e3,
                       "launches_timed": dom_cnt, "kernel_ms_share": {c: prof[c][0] for c in CLASSES}}


def pmc_traffic(dom, nx, nu, B, N):
    """HBM bytes per launch of the dominant C2 kernel from the newest PMC summary under profiles/ (rocprofv3 cannot
    run inside this process; tools/profile_round.sh regenerates the file for the commit it is run on)"""
    import re

    def tag(f):     # rNN_vM_pmc_traffic.json -> (NN, M): the newest generation by NAME (mtimes do not survive the snapshot)
        m = re.match(r"r(\d+)_v(\d+)_pmc_traffic\.json$", os.path.basename(f))
        return (int(m.group(1)), int(m.group(2))) if m else None
    files = sorted((f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")) if tag(f)), key=tag)
    if not files or (B, N, nx, nu) != (65536, 50, 8, 3):
        return None
    want = {"back_fact": f"kb_factor<{nx}, {nu}, false>", "back_rhs": f"kb_backrhs<{nx}, {nu}, false>",
            "fwd_aff": f"kb_forward<{nx}, {nu}, false, false>", "fwd_corr": f"kb_forward<{nx}, {nu}, false, true>"}.get(dom)
    try:
        pmc = json.load(open(files[-1]))
        e = pmc[want]
        return {"file": os.path.relpath(files[-1], ROOT), "commit": pmc.get("_commit"), "src_hash": pmc.get("_src_hash"),
                "avg_main": e.get("hbm_bytes_per_launch_avg_main", e["hbm_bytes_per_launch_avg"]),
                "full": e["hbm_bytes_per_launch_full"], "kernel": want}
    except Exception:
        return None


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


def config_traffic(section, symbol, sweep):
    """HBM bytes per launch of a configuration's dominant kernel from the newest per-section PMC summary under profiles/
    (tools/profile_round.sh <tag> <commit> full; sections are cut by the marker launches run_config brackets its timed
    solves with)"""
    import re

    def tag(f):
        m = re.match(r"r(\d+)_v(\d+)_config_pmc_traffic\.json$", os.path.basename(f))
        return (int(m.group(1)), int(m.group(2))) if m else None
    files = sorted((f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_config_pmc_traffic.json")) if tag(f)), key=tag)
    if not files:
        return None
    try:
        pmc = json.load(open(files[-1]))
        sec = pmc["sections"][str(section)]
        cand = {k: v for k, v in sec.items() if k.startswith(symbol + "<")}
        if not cand:
            return None
        # the two forward sweeps share a symbol: the corrector sweep (update pass included) moves more bytes
        pick = (min if sweep == "fwd_aff" else max)(cand, key=lambda k: cand[k]["hbm_bytes_per_launch_avg_main"])
        e = cand[pick]
        return {"file": os.path.relpath(files[-1], ROOT), "commit": pmc.get("_commit"), "src_hash": pmc.get("_src_hash"), "kernel": pick, "section": section,
                "avg_main": e["hbm_bytes_per_launch_avg_main"], "full": e["hbm_bytes_per_launch_full"], "launches": e["launches"]}
    except Exception:
        return None


def mark_stale(tr):
    """the PMC summary was collected on another build than the one being benched: say so (counter passes cannot run inside
    this process; tools/profile_round.sh regenerates the summary for the commit it is run on)"""
    if tr is not None:
        head = (git_head() or "").replace("+dirty", "")
        tr["benched_commit"] = git_head()
        if tr.get("src_hash"):
            # the summary records a hash of the library's sources: a later commit that touches only documents, tests or tools
            # leaves it valid
            sys.path.insert(0, os.path.join(ROOT, "profiles"))
            from summarize import kernel_src_hash
            tr["benched_src_hash"] = kernel_src_hash()
            tr["stale"] = tr["src_hash"] != tr["benched_src_hash"]
        else:
            tr["stale"] = bool(tr.get("commit")) and bool(head) and not (str(tr["commit"]).startswith(head) or head.startswith(str(tr["commit"])))
    return tr


def mfma_util(suffix=""):
    """newest profiles/rNN_vM_mfma_util<suffix>.json (tools/profile_mfma.sh: rocprofv3 PMC SQ_INSTS_MFMA / SQ_VALU_MFMA_BUSY_CYCLES /
    GRBM_GUI_ACTIVE over one C3 solve; suffix "_c4_c5": tools/profile_mfma_c4_c5.sh, the C4 batch and the nx=24 nu=6 N=50 class):
    matrix-pipe utilisation of the kernels that issue MFMAs"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_mfma_util{suffix}.json")))
    if not files:
        return None
    try:
        j = json.load(open(files[-1]))
    except Exception:
        return None
    return mark_stale({"file": os.path.relpath(files[-1], ROOT), "commit": j.get("_commit"), "src_hash": j.get("_src_hash"),
            "kernels": {k: {f: v.get(f) for f in ("mfma_utilisation", "mfma_TFLOPs", "frac_of_measured_mfma_peak_73.2", "avg_us", "mfma_instructions_per_launch")}
                        for k, v in j.get("kernels", {}).items()}})


MFMA_PROBES = {"v_mfma_f64_4x4x4_4b_TFLOPs_measured": 73.2, "v_mfma_f64_16x16x4_TFLOPs_measured": 47.6, "v_fma_f64_TFLOPs_measured": 69.3,
               "dpp_broadcast_plus_2_fma_cycles": 14.3, "mfma_4x4x4_4b_cycles": 17.2,
               "probes": ["profiles/r04_mfma4x4x4_probe.txt", "profiles/r04_mfma4x4x4_layout.txt", "profiles/r02_mfma_f64_probe.txt"]}
# C2 (the headline): one instance per lane, every sweep HBM-bound at 5-6 TB/s of real traffic -- no matrix product to offload
MFMA_NOTE_C2 = dict(MFMA_PROBES, used=False, mfma_utilisation=0.0,
                    why="the C2 sweeps run one instance per lane and are bound by HBM traffic (roofline.bound = hbm); the 11 x 11 stage "
                        "blocks live in the lanes' registers.  Where the path has matrix products between lanes -- the partial condensing "
                        "contraction and the Riccati factor sweep of the condensed / nx = 24 QPs (configs.C3, configs.C5_share) -- they run on "
                        "v_mfma_f64_4x4x4_4b_f64, the one FP64 MFMA shape whose tiles nx = 8 fills and the one that beats the vector pipe "
                        "on gfx950 (73.2 vs 69.3 TFLOP/s measured; the 16x16x4 shape: 47.6)")


def mfma_note_c3(batch):
    u = mfma_util()
    pk, tiles = int(batch.scalar("pcond_kernel")), None
    try:
        tiles = int(batch.condensed_scalar("w16_tiles"))
    except Exception:
        pass
    return dict(MFMA_PROBES, used=(pk == 3 or bool(tiles)),
                kernels={"km_pcond (partial condensing, Z'HZ / [B A]Z on 4 x 4 tiles, pcond_kernels_mfma.hpp)": pk == 3,
                         "kt_factor (Riccati factor sweep of the condensed QP: W = [B A]'Lx+, M += WW', blocked Cholesky, ipm_kernels_w16t.hpp)": tiles},
                tile_fill="nx = 8: 2 x 2 tiles, nc = 23 + 1 vector column = 6 tiles: 1.0 (zero tiles of the block's later inputs skipped at compile time)",
                utilisation=u,
                mfma_utilisation=(max((k.get("mfma_utilisation") or 0.0) for k in u["kernels"].values()) if u and u["kernels"] else None),
                note="utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x active cycles) from the committed PMC summary named in utilisation.file "
                     "(counter passes cannot run inside this process); both kernels are bound by memory latency / dependent chains at one or two "
                     "waves per SIMD, not by the matrix pipe (DESIGN.md 4.5, 4.6)")


def cpu_caps():
    caps = {"logical": os.cpu_count() or 1, "affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None}
    try:
        import psutil
        caps["physical"] = psutil.cpu_count(logical=False)
    except Exception:
        caps["physical"] = None
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        caps["cgroup_quota_cpus"] = None if q[0] == "max" else float(q[0]) / float(q[1])
    except Exception:
        caps["cgroup_quota_cpus"] = None
    return caps


def threads_allowed():
    caps = cpu_caps()
    allowed = caps["affinity"] or caps["logical"]
    if caps["cgroup_quota_cpus"]:
        allowed = max(1, min(allowed, int(round(caps["cgroup_quota_cpus"]))))
    return allowed


def cpu_baseline(data, N, unique, budget_s=25.0):
    """the oracle on the host cores: `unique` instances of the same workload built once, cloned so that every thread of
    every probe has >= 64 independent solves; thread counts swept in powers of two up to the cores the process may use"""
    from acados_amd.generators import lqr_instance_qp
    from oracle.oracle import OracleQp, clone_handle, default_opts, free_handle, solve_batch_handles
    t_begin = time.perf_counter()
    qps = [OracleQp(lqr_instance_qp(data, i, N)) for i in range(unique)]
    opts = default_opts(tol_stat=1e-8, tol_eq=1e-8, tol_ineq=1e-8, tol_comp=1e-8, iter_max=50)
    caps = cpu_caps()
    allowed = caps["affinity"] or caps["logical"]
    if caps["cgroup_quota_cpus"]:
        allowed = max(1, min(allowed, int(round(caps["cgroup_quota_cpus"]))))
    handles = [q.h.value for q in qps]
    clones = []

    def pool(n):
        while len(handles) + len(clones) < n:
            clones.append(clone_handle(qps[len(clones) % unique].h))
        return (handles + [c.value for c in clones])[:n]

    # one thread: 512 solves (~0.5 s)
    def run(threads, n, reps):
        hs = pool(n)
        best = 1e300
        for _ in range(reps):
            t0 = time.perf_counter()
            st = solve_batch_handles(hs, opts, nthreads=threads)
            best = min(best, time.perf_counter() - t0)
        assert np.all(st == 0)
        return n / best

    one = run(1, min(512, max(unique, 64)), 2)
    sweep = {1: one}
    th = 2
    while th <= allowed and time.perf_counter() - t_begin < budget_s:
        n = max(64 * th, 1024)                       # >= 64 QPs per thread
        sweep[th] = run(th, n, 2)
        th *= 2
    if allowed not in sweep and time.perf_counter() - t_begin < budget_s:
        sweep[allowed] = run(allowed, max(64 * allowed, 1024), 2)
    best_t = max(sweep, key=lambda k: sweep[k])
    # every unique instance solved at least once (the probes above may have touched only the first ones): one pass over
    # the whole sample with the best thread count -- its solutions are the parity sample
    t0 = time.perf_counter()
    st = solve_batch_handles(handles, opts, nthreads=best_t)
    full_pass = unique / (time.perf_counter() - t0)
    assert np.all(st == 0)
    iters = float(np.mean([q.iter for q in qps]))
    for c in clones:
        free_handle(c)
    cpu_baseline.solved = qps     # the same solutions double as the parity sample
    return {"value": sweep[best_t], "unit": "OCP-QP solves/s", "cores": best_t, "kind": "port", "unique": unique,
            "kind_note": "port = this repository's restated CPU oracle (plain C, scalar loops, no BLASFEO micro-kernels); HPIPM + BLASFEO sources "
                         "are absent from the reference tree, so the reference itself cannot be timed here.  Expect HPIPM on the same cores to be "
                         "several times faster than this port (its dpotrf / dsyrk / dtrmm run on AVX-512 panel-major kernels): the GPU / CPU "
                         "ratio of this line would shrink by that factor and says nothing about kernel quality -- the roofline fraction does",
            "sample": f"{unique} instances of the same workload (seed 0, first instances) built once and cloned to >= 64 "
                      f"independent solves per thread, min of 2 repeats per thread count, OpenMP over instances as "
                      f"acados_solver.in.c:3232 does; restated CPU oracle, not HPIPM",
            "one_thread": one, "full_sample_pass": full_pass, "thread_sweep": {str(k): v for k, v in sorted(sweep.items())},
            "host": caps, "threads_allowed": allowed, "mean_iter": iters, "seconds": time.perf_counter() - t_begin}


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


GATHER_LIMIT_S = 240.0


def guarded_gather(fn, world, limit_s=None):
    """The solutions gather is the one step of an N > 1 run that has never met more than one device (RCCL has only
    ever run as one rank here).  It runs after everything the line needs has been measured; with more than one rank it runs on a
    helper thread under a time limit, so that a collective that never completes costs the gather record, not the run: returns
    (result or None, error string or None).  The caller emits its line and, if the error says the collective is still in flight, leaves
    the process with os._exit (the helper thread cannot be joined)."""
    if world <= 1:
        return fn(), None
    import threading
    import torch
    limit_s = float(os.environ.get("ACADOS_AMD_GATHER_LIMIT_S", GATHER_LIMIT_S)) if limit_s is None else limit_s
    box, dev_idx = {}, torch.cuda.current_device() if torch.cuda.is_available() else None

    def run():
        try:
            if dev_idx is not None:
                torch.cuda.set_device(dev_idx)      # the current device is per thread
            box["v"] = fn()
        except Exception as e:      # noqa: BLE001 -- reported on the line
            box["e"] = f"{type(e).__name__}: {e}"
    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(limit_s)
    if th.is_alive():
        return None, f"in flight after {limit_s:.0f} s"
    return box.get("v"), box.get("e")


def leave_after_stuck_gather(err):
    """a collective still in flight holds the stream and a thread: no destroy_process_group, no interpreter shutdown"""
    if err and err.startswith("in flight"):
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


def relaunch(n):
    """N ranks of this script on one node (the command line the driver uses for N > 1): rank 0's JSON line is the last line of
    stdout, the exit code is the launcher's"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def dry_run(args):
    """the launch / sharding path without a GPU: a gloo group of the ranks that were started"""
    import torch.distributed as dist
    from acados_amd.generators import C5_CLASSES
    from acados_amd.sharding import shard_range
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    if args.config == "c5":
        per_class = args.c5_total // len(C5_CLASSES)
        # every class (the nine shapes + the multi-phase one) is split the same way: one range per rank says it all; gather_counts is
        # what main_c5 hands to the library's exact-count gather (uneven shards: 58,254 = 6 x 7,282 + 2 x 7,281)
        mine = {"classes": len(C5_CLASSES) + 1, "per_class": list(shard_range(per_class, rank, max(world, 8))),
                "gather_counts": [hi - lo for lo, hi in (shard_range(per_class, r, max(world, 8)) for r in range(world))]}
    else:
        mine = {"instances": [rank * args.batch, (rank + 1) * args.batch]}
    mine.update(rank=rank, local_rank=int(os.environ.get("LOCAL_RANK", "0")), pid=os.getpid())
    ranks = [mine]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "config": args.config, "ranks": ranks,
                          "gather": {"ranks": world, "collective": "ocp_qp_gpu_batch_gather (RCCL) after the timed region"}},
                         separators=(",", ":")), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main_c5(args):
    """BASELINE configs[4]: the nine shape classes nx in {4,12,24} x N in {20,50,100} plus the multi-phase class, --c5-total
    instances split evenly over the classes and every class over the ranks (identical work per rank: ranks finish together);
    each rank solves its share of every class as one device batch, the classes concurrently (acados_amd/shape_classes.py).
    Timed region = `steps` solves of everything a rank holds, data resident; MAX over ranks; afterwards every class's
    solutions are gathered through the library's collective (ocp_qp_gpu_batch_gather, RCCL)."""
    import torch
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import (C5_CLASSES, fill_lqr_batch, fill_multiphase_batch, lqr_dims, multiphase_batch, multiphase_dims,
                                       random_lqr_batch)
    from acados_amd.shape_classes import ConcurrentClasses
    from acados_amd.sharding import gather_solutions, reduce_max, shard_range
    ranks_total = max(world, 8)          # weak scaling: a rank holds the share of the 8-GPU job whatever the number of ranks present
    per_class = args.c5_total // len(C5_CLASSES)      # split equally over the nine classes; the multi-phase class comes on top
    lo, hi = shard_range(per_class, rank, ranks_total)
    batches = []
    for ci, (nx, nu, N) in enumerate(C5_CLASSES):
        data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=hi - lo, seed=200 + ci, first=lo)
        gb = OcpQpGpuBatch(lqr_dims(N, nx, nu), hi - lo, device=local_rank)
        fill_lqr_batch(gb, data, N, xp=lambda a: torch.from_numpy(a).to(dev))
        batches.append((f"nx={nx} nu={nu} N={N}", gb))
    dm = multiphase_batch(N=50, batch=hi - lo, first=lo)
    gm = OcpQpGpuBatch(multiphase_dims(50), hi - lo, device=local_rank)
    fill_multiphase_batch(gm, dm)
    batches.append(("multi-phase nx=12->4 at k=25 nu=3 N=50", gm))
    for _, gb in batches:
        tol_setup(gb)
        gb.solve()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with ConcurrentClasses([gb for _, gb in batches]) as cc:
        for _ in range(max(args.warmup, 1)):
            cc.solve()
        barrier()
        t0 = time.perf_counter()
        bad = 0
        for _ in range(args.steps):
            bad += cc.solve()
        barrier()
        elapsed = reduce_max(time.perf_counter() - t0, dist, dev)
    count = sum(gb.n_batch for _, gb in batches)
    per = [{"class": c, "instances": gb.n_batch, "kernel": gb.kernel_name, "ms": gb.scalar("time_tot") * 1e3,
            "iters_mean": float(gb.info("iter").mean()), "failures": int((gb.info("status") != 0).sum()),
            "max_kkt_residual_independent": float(gb.res_compute().max())} for c, gb in batches]
    # shards of a class are uneven when per_class is not a multiple of the rank count (58,254 = 6 x 7,282 + 2 x 7,281): the
    # library's exact-count gather (ocp_qp_gpu_batch_gather_v) needs every rank's count
    counts = [hi_r - lo_r for lo_r, hi_r in (shard_range(per_class, r, ranks_total) for r in range(world))]
    tot = torch.tensor([count, bad], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tot)
    gathers, gerr = guarded_gather(lambda: [gather_solutions(gb, dist, rank, world, counts=counts) for _, gb in batches], world)
    if rank == 0:
        ok = [g for g in (gathers or []) if g]
        out = {"metric": "OCP-QP solves/sec, mixed shape classes (BASELINE configs[4])", "value": float(tot[0]) * args.steps / elapsed,
               "unit": "OCP-QP solves/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1),
               "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"ten shape classes (nx in {{4,12,24}} x N in {{20,50,100}} + multi-phase nx 12->4), "
                                      f"{args.c5_total} instances per 8 GPUs over the nine classes + the same share of the multi-phase class, {count} on this rank, "
                                      f"classes solved concurrently",
                          "global_batch": int(tot[0]), "parallelism": f"every class instance-sharded x{world}", "commit": git_head()},
               "failures": int(tot[1]), "per_class_rank0": per,
               "gather": {"ranks": world, "ms": sum(g["ms"] for g in ok), "ms_all_classes": sum(g["ms"] for g in ok), "classes_gathered": len(ok),
                          "slice_matches_getters": all(g["slice_matches_getters"] for g in ok) if ok else None,
                          "instances_per_rank": counts,
                          "collective": ok[0]["collective"] if ok else None}}
        if gerr:
            out["gather"]["error"] = gerr
        emit(out, args)
    leave_after_stuck_gather(gerr)
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=65536, help="instances per GPU")
    ap.add_argument("--horizon", type=int, default=50)
    ap.add_argument("--nx", type=int, default=8)
    ap.add_argument("--nu", type=int, default=3)
    ap.add_argument("--cpu-sample", type=int, default=2048, help="unique instances built for the CPU baseline / parity sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip C3 / C4 / C5 (the headline line only)")
    ap.add_argument("--c4-batch", type=int, default=16384)
    ap.add_argument("--check-configs", type=int, default=1024,
                    help="instances per configuration checked against the oracle (OpenMP batch on the host, outside timing; SURVEY 8d asks >= 1,024)")
    ap.add_argument("--polish-legs", action="store_true",
                    help="also time the opt-in terminal polishing step on C2 / C4 (measured and dominated by a tighter tol_comp: profiles/r06_polish_sweep.txt)")
    ap.add_argument("--compact-min", type=int, default=None, help="override the library default of the compaction threshold")
    ap.add_argument("--check", type=int, default=8, help="instances per rank checked against the oracle (outside timing)")
    ap.add_argument("--config", choices=("c2", "c5"), default="c2",
                    help="c2: the headline workload (BASELINE configs[1]); c5: the mixed-shape-class batch of configs[4], 524,288 instances "
                         "split over the ranks present (a rank of a smaller job still holds the share of an 8-GPU job: weak scaling)")
    ap.add_argument("--c5-total", type=int, default=524288)
    ap.add_argument("--detail-file", default=None,
                    help="where the full result object goes (default gpurun_out/bench_detail.json); stdout's last line is the compact one")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch path only: every rank joins a gloo group, reports the instance ranges it would own, rank 0 prints one JSON "
                         "line; no GPU is touched (the CPU tier checks that --gpus N starts N ranks)")
    args = ap.parse_args()

    # `python bench.py --gpus N` started as ONE process: become N ranks (one process per GPU) under torch.distributed.run.
    # Started by the driver under torch.distributed.run (WORLD_SIZE set), the ranks are already there.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch(args.gpus))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus and int(os.environ.get("RANK", "0")) == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: the launcher's world size is what runs", file=sys.stderr)
    if args.dry_run:
        return dry_run(args)
    if args.config == "c5":
        return main_c5(args)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, lqr_instance_qp, random_lqr_batch

    N, nx, nu, B = args.horizon, args.nx, args.nu, args.batch
    # instance ids are global: rank r owns [r*B, (r+1)*B) of one counter-based stream
    data = random_lqr_batch(N=N, nx=nx, nu=nu, batch=B, seed=0, first=rank * B)
    dims = lqr_dims(N, nx, nu)
    gb = OcpQpGpuBatch(dims, B, device=local_rank)

    def to_dev(a):
        t = torch.from_numpy(a).to(dev)
        torch.cuda.synchronize()
        return t

    t0 = time.perf_counter()
    fill_lqr_batch(gb, data, N, xp=to_dev)
    t_pack = time.perf_counter() - t0
    tol_setup(gb)
    if args.compact_min is not None:
        gb.opts_set("compact_min", args.compact_min)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        gb.solve()
    gb.opts_set("profile", 1)
    gb.scalar("prof_reset")
    barrier()
    t0 = time.perf_counter()
    bad = 0
    for _ in range(args.steps):
        bad += gb.solve()
    barrier()
    elapsed = time.perf_counter() - t0
    gb.opts_set("profile", 0)

    from acados_amd.sharding import gather_solutions, reduce_max
    elapsed = reduce_max(elapsed, dist, dev)      # MAX over ranks

    iters = gb.info("iter")
    status = gb.info("status")
    res_max = max(float(gb.info(n).max()) for n in ("res_stat", "res_eq", "res_ineq", "res_comp"))
    res_indep = float(gb.res_compute().max())

    if dist is not None:
        stats = torch.tensor([[float(iters.mean()), float(iters.max()), float((status != 0).sum()), res_max, res_indep]],
                             dtype=torch.float64, device=dev)
        s_all = [torch.empty_like(stats) for _ in range(world)]
        dist.all_gather(s_all, stats)
        s_all = torch.cat(s_all).cpu().numpy()
        mean_iter, max_iter = float(s_all[:, 0].mean()), int(s_all[:, 1].max())
        failures, res_max, res_indep = int(s_all[:, 2].sum()), float(s_all[:, 3].max()), float(s_all[:, 4].max())
    else:
        mean_iter, max_iter, failures = float(iters.mean()), int(iters.max()), int((status != 0).sum())

    # ---- parity spot check against the oracle (checker only, outside timing) ----
    err = None
    if args.check > 0:
        err = oracle_error(gb, lambda i: lqr_instance_qp(data, i, N), np.linspace(0, B - 1, args.check).astype(int), N, tight=False)["same_tol_max"]

    # ---- gather of the full solution payload + statistics over RCCL/xGMI, outside the timed region, LAST: everything the line
    # reports is in hand by then (guarded_gather) ----
    def the_gather():
        return gather_solutions(gb, dist, rank, world)       # device buffers, library collective (no host bounce)

    if rank != 0:
        _, gerr = guarded_gather(the_gather, world)
        leave_after_stuck_gather(gerr)
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (HIP events recorded on the launch stream) ----
    b_in, b_out = algorithmic_bytes_dims(dims)
    dom, prof, roof = sweep_roofline(gb, args.steps, b_in + b_out)
    solves_per_s = world * B * args.steps / elapsed
    tr = mark_stale(pmc_traffic(dom, nx, nu, B, N))
    roof["mfma"] = MFMA_NOTE_C2
    roof["traffic"] = tr["avg_main"] if tr else None
    roof["traffic_source"] = tr
    roof["traffic_note"] = ("HBM bytes per launch from the committed rocprofv3 PMC summary named in traffic_source (counter passes "
                            "cannot run inside this process); the time it is divided by is measured live in this run")
    roof["traffic_over_algorithmic"] = (tr["avg_main"] / roof["bytes_per_launch"]) if tr else None
    # the REAL rate of the dominant sweep (PMC bytes of the committed summary over the launch time measured here) against the copy
    # rate the part sustains: how much of the gap between `frac` and 1 is bytes the sweep also carries, how much is rate
    roof["traffic_GBps"] = (tr["avg_main"] / (roof["avg_launch_ms"] * 1e-3) / 1e9) if tr and roof.get("avg_launch_ms") else None
    roof["traffic_frac_of_sustained_copy"] = (roof["traffic_GBps"] / HBM_COPY_GBS) if roof["traffic_GBps"] else None
    full_units = [u for u in roof["units_per_launch"] if u == B]
    roof["full_launch"] = {"bytes": float(B * (b_in + b_out)), "traffic": tr["full"] if tr else None,
                           "traffic_over_algorithmic": (tr["full"] / (B * (b_in + b_out))) if tr else None,
                           "launches_per_solve_with_every_instance_active": len(full_units)}
    roof["whole_solve_GBps"] = solves_per_s / world * (b_in + b_out) / 1e9
    roof["whole_solve_frac"] = roof["whole_solve_GBps"] / HBM_PEAK_GBS
    out = {
        "metric": "OCP-QP solves/sec (batch) at N=50 nx=8 nu=3",
        "value": solves_per_s,
        "unit": "OCP-QP solves/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"random LQR OCP-QP (BASELINE configs[1]): N={N} nx={nx} nu={nu}, u-box + x0 equality, "
                               f"full-space Riccati IPM, cold start, tol 1e-8, iter_max 50",
                   "batch_per_gpu": B, "global_batch": world * B, "parallelism": f"instance-sharded x{world}",
                   "kernel": gb.kernel_name, "commit": git_head()},
        "ipm": {"mean_iter": mean_iter, "max_iter": max_iter, "failures": failures, "max_kkt_residual": res_max,
                "max_kkt_residual_independent": res_indep,
                "max_rel_primal_err_vs_oracle": err, "oracle_checked_instances": args.check,
                "launches_per_step": int(gb.scalar("launches")),
                # the structural waste of one instance per lane: a wave streams its tiles until its LAST lane has converged.
                # iter_hist[j] = instances (this rank) that took j iterations; wave_max_iter_mean = mean over the 64-instance tiles
                # of the tile's largest count (what every sweep's traffic is proportional to) against mean_iter
                "iter_hist": np.bincount(iters).tolist(),
                "wave_max_iter_mean": float(iters[:(iters.size // 64) * 64].reshape(-1, 64).max(axis=1).mean()) if iters.size >= 64 else float(iters.max())},
        "roofline": roof,
        "pack_s": t_pack,
        "hbm_bytes_per_gpu": gb.bytes,
        "gather_ms": None,
        "gather": None,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(data, N, min(args.cpu_sample, B))
        if args.check > 0:
            # max relative primal error of the GPU solution against the oracle over the whole CPU sample
            xs = [gb.get("x", k) for k in range(N + 1)]
            us = [gb.get("u", k) for k in range(N)]
            err_s = 0.0
            for i, o in enumerate(cpu_baseline.solved):
                for k in range(N + 1):
                    r = o.get(k, "x")
                    err_s = max(err_s, float(np.max(np.abs(xs[k][i] - r) / np.maximum(1.0, np.abs(r)))))
                    if k < N:
                        r = o.get(k, "u")
                        err_s = max(err_s, float(np.max(np.abs(us[k][i] - r) / np.maximum(1.0, np.abs(r)))))
            out["ipm"]["max_rel_primal_err_vs_oracle"] = max(err, err_s)
            out["ipm"]["oracle_checked_instances"] = args.check + len(cpu_baseline.solved)
        cpu_baseline.solved = None
    if world == 1 and args.check_configs and not args.no_configs:
        idx = np.unique(np.linspace(0, B - 1, args.check_configs).astype(int))
        oe = oracle_error(gb, lambda i: lqr_instance_qp(data, i, N), idx, N)
        out["ipm"]["oracle_check"] = oe
        out["ipm"]["max_rel_primal_err_vs_oracle"] = max(out["ipm"]["max_rel_primal_err_vs_oracle"] or 0.0, oe["same_tol_max"])
        out["ipm"]["oracle_checked_instances"] = int(out["ipm"]["oracle_checked_instances"]) + int(idx.size)
    if world == 1 and not args.no_configs:
        out["configs"] = other_configs(gb, data, args)
    gather, gerr = guarded_gather(the_gather, world)
    out["gather_ms"] = gather["ms"] if gather else None
    out["gather"] = gather if gather or not gerr else {"ranks": world, "error": gerr}
    emit(out, args)
    leave_after_stuck_gather(gerr)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
