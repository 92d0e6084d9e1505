"""the ONE line the driver parses (compact, < 4 KB) and the detail file beside it (a part of bench.py)."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


DIST_KEYS = ("median", "q99", "max", "above_1e-6")


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
                            "condense_expand_ms", "solves_per_s_one_after_the_other", "polished", "host_threads", "pcie_GBps", "pcie_frac", "skipped", "error"), 4)
            if isinstance(c.get("pcie_measured"), dict) and "h2d_GBps" in c["pcie_measured"]:
                rec["pcie_h2d_d2h_GBps"] = [_r(c["pcie_measured"]["h2d_GBps"], 3), _r(c["pcie_measured"].get("d2h_GBps"), 3)]
            if c.get("rti_feedback"):
                rec["rti_feedback_solves_per_s"] = _r(c["rti_feedback"]["solves_per_s"], 4)
                rec["rti_feedback_ms"] = _r(c["rti_feedback"]["ms_per_step"], 4)
            for k_, v_ in c.items():
                if k_.startswith("at_") and isinstance(v_, dict) and "solves_per_s" in v_:
                    rec[k_] = [_r(v_["solves_per_s"], 4), _r(v_.get("rti_feedback_solves_per_s"), 4)]   # [one call, RTI feedback half] QP/s
            if "phases_ms" in c:
                rec["phases_ms"] = [_r(c["phases_ms"].get(k), 3) for k in ("unpack_in_ms", "copy_and_device_ms", "device_solve_ms", "pack_out_ms")]
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
