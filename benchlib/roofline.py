"""roofline of the dominant sweep: algorithmic bytes, HIP-event launch times, PMC traffic summaries under profiles/, MFMA notes (a part of bench.py)."""
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from benchlib.line import git_head


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
