#!/usr/bin/env python3
"""Turn rocprofv3 (rocpd sqlite) outputs into the small text summaries kept under profiles/.
usage: summarize.py kernel <results.db>   -> per-kernel calls / total / average (kernel-trace --stats)
       summarize.py pmc <results.db>      -> per-kernel counter sums and per-launch averages
       summarize.py traffic <fetch.db> <write.db> -> JSON: HBM bytes per launch of every gqp kernel
       summarize.py sections <fetch.db> <write.db> -> the same per marker-delimited section (the configuration legs)
       summarize.py mfma <pmc.db> <trace.db> [commit] -> JSON: matrix-pipe utilisation of every kernel that issues MFMAs
                                                  (--pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES;
                                                   durations from a --kernel-trace run of the same command)"""
import sqlite3
import sys


def kernel_src_hash():
    """sha256 over the kernel / host sources of the library (acados_amd/csrc): what a PMC summary was measured on.  bench.py
    compares it with the tree it runs in -- a later commit that touches only documents or tests does not make a summary stale"""
    import glob, hashlib, os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "acados_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "*.hpp")) + glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h")) + glob.glob(os.path.join(root, "*.cpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def kernel(db):
    cur = sqlite3.connect(db).cursor()
    print(f"{'kernel':80s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
    for name, calls, tot, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(f"{name[:80]:80s} {calls:6d} {tot:12.1f} {avg:10.1f} {pct:6.2f}")
    print("\nper-dispatch durations of the IPM kernels (us), in launch order, first 60:")
    rows = cur.execute("select name,duration,vgpr_count,accum_vgpr_count,sgpr_count,scratch_size from kernels "
                       "where name like '%gqp::k_backward%' or name like '%gqp::k_forward%' or name like '%gqp::kb_%' "
                       "or name like '%gqp::kw_%' or name like '%gqp::kx_%' or name like '%gqp::ky_%' or name like '%gqp::kz_%' order by start limit 60").fetchall()
    for name, dur, v, a, s, sc in rows:
        short = name.split("gqp::")[1].split("(")[0]
        print(f"  {short:40s} {dur / 1e3:10.1f}  vgpr {v} agpr {a} sgpr {s} scratch {sc}")


def timeline(db, which=-1):
    """every launch of ONE solve in order (`which`-th window between two init launches, default the last complete one): offset from
    the window's first launch, duration, the gap to the previous launch's end -- where a latency-bound small batch spends its time"""
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name,start,end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "_init<" in r[0] or "k_init" in r[0]]
    if len(marks) < 2:
        print("fewer than two solves in the trace"); return
    which = int(which)
    lo, hi = (marks[which - 1], marks[which]) if which < 0 else (marks[which], marks[which + 1])
    # back up to the uploads in front of the init launch
    while lo > 0 and rows[lo - 1][1] > rows[lo][1] - 3e6 and "finalize" not in rows[lo - 1][0] and "pexpand" not in rows[lo - 1][0]:
        lo -= 1
    t0, prev_end, busy = rows[lo][1], rows[lo][1], 0.0
    print(f"{'offset_us':>10s} {'dur_us':>9s} {'gap_us':>8s}  kernel")
    for name, st, en in rows[lo:hi]:
        short = name.split("gqp::")[1].split("(")[0] if "gqp::" in name else name[:40]
        print(f"{(st - t0) / 1e3:10.1f} {(en - st) / 1e3:9.1f} {(st - prev_end) / 1e3:8.1f}  {short}")
        busy += (en - st) / 1e3
        prev_end = max(prev_end, en)
    print(f"window {(prev_end - t0) / 1e3:.1f} us, kernels busy {busy:.1f} us, launches {hi - lo}")


def traffic(fetch_db, write_db, commit=None):
    import json
    out = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes over `bench.py --steps 1 --warmup 0 "
                    "--no-cpu-baseline --no-configs`. Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports exactly "
                    "half of the bytes of a wide coalesced streaming read, so bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024. "
                    "_avg: all launches of that kernel in the solve; _avg_main: the launches that carry work (> 1 % of the "
                    "largest one -- leaves out the conditional redo launches, which touch flagged instances only, i.e. the "
                    "same set of launches bench.py times with HIP events); _full: the largest launch (every instance active).",
           "_commit": commit, "_src_hash": kernel_src_hash()}
    q = "select kernel_name, value from counters_collection where counter_name=? order by dispatch_id"
    def per_kernel(db, counter):
        d = {}
        for k, v in sqlite3.connect(db).cursor().execute(q, (counter,)):
            d.setdefault(k, []).append(v)
        return d
    try:
        f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    except sqlite3.OperationalError:   # older rocpd schema: no dispatch_id column
        q = "select kernel_name, value from counters_collection where counter_name=?"
        f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    for k in sorted(f):
        if "gqp::" not in k or k not in w or len(f[k]) != len(w[k]):
            continue
        short = k.split("gqp::")[1].split("(")[0]
        tot = [(2 * a + b) * 1024 for a, b in zip(f[k], w[k])]      # the two passes replay the same launch sequence
        mx = max(tot)
        main = [t for t in tot if t > 0.01 * mx] or [0.0]
        out[short] = {"launches": len(tot), "launches_main": len(main), "fetch_kib_avg": sum(f[k]) / len(f[k]),
                      "write_kib_avg": sum(w[k]) / len(w[k]), "fetch_kib_max": max(f[k]), "write_kib_max": max(w[k]),
                      "hbm_bytes_per_launch_avg": sum(tot) / len(tot), "hbm_bytes_per_launch_avg_main": sum(main) / len(main),
                      "hbm_bytes_per_launch_full": mx}
    print(json.dumps(out, indent=1))


def sections(fetch_db, write_db, commit=None):
    """HBM bytes per launch of every gqp kernel, per SECTION of the run: bench.py brackets the timed solves of each
    configuration with marker launches (`k_marker<ID>`, option "marker"), the dispatch sequence is cut at them.
    Sections: 1 = C3, 2 = C4, 3.. = the C5 classes in generator order; 0 closes a section."""
    import json
    import re
    out = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate counter-only passes over `bench.py --steps 1 "
                    "--warmup 0 --no-cpu-baseline --check 0 --check-configs 0` (configuration legs included); bytes = "
                    "(2*FETCH_SIZE + WRITE_SIZE) * 1024 per MI355X_MICROARCH.md (HBM section).  Per section (marker id) and "
                    "kernel: launches, average over the launches that carry work (> 1 % of the largest) and the largest.",
           "_commit": commit, "_src_hash": kernel_src_hash(), "sections": {}}
    q = "select kernel_name, value from counters_collection where counter_name=? order by dispatch_id"

    def cut(db, counter):
        """label -> kernel -> [counter value of every launch, in order]; launches outside a section are dropped (the
        concurrent C5 leg interleaves its nine streams differently from run to run: it carries no marker)"""
        label, d = None, {}
        for k, v in sqlite3.connect(db).cursor().execute(q, (counter,)):
            if "gqp::" not in k:
                continue
            short = k.split("gqp::")[1].split("(")[0]
            m = re.match(r"k_marker<(\d+)>", short)
            if m:
                label = int(m.group(1)) or None
            elif label is not None:
                d.setdefault(str(label), {}).setdefault(short, []).append(v)
        return d
    f, w = cut(fetch_db, "FETCH_SIZE"), cut(write_db, "WRITE_SIZE")
    acc = {}
    for lab in f:
        for short, fv in f[lab].items():
            wv = w.get(lab, {}).get(short)
            if wv is None or len(wv) != len(fv):
                out.setdefault("_mismatch", []).append(f"section {lab} {short}: {len(fv)} vs {0 if wv is None else len(wv)} launches")
                continue
            acc.setdefault(lab, {})[short] = [((2 * a + b) * 1024, a, b) for a, b in zip(fv, wv)]
    for lab, ks in acc.items():
        sec = out["sections"].setdefault(lab, {})
        for short, rows in ks.items():
            tot = [r[0] for r in rows]
            mx = max(tot)
            main = [t for t in tot if t > 0.01 * mx] or [0.0]
            sec[short] = {"launches": len(tot), "launches_main": len(main), "fetch_kib_avg": sum(r[1] for r in rows) / len(rows),
                          "write_kib_avg": sum(r[2] for r in rows) / len(rows), "hbm_bytes_per_launch_avg": sum(tot) / len(tot),
                          "hbm_bytes_per_launch_avg_main": sum(main) / len(main), "hbm_bytes_per_launch_full": mx}
    print(json.dumps(out, indent=1))


def mfma(pmc_db, trace_db, commit=None):
    """per kernel: MFMA instructions, busy cycles of the matrix pipe, utilisation = busy cycles / (SIMDs x cycles the GPU
    was active for the launch).  GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (a 2.9 ms launch reports 50.6 M = 8 x
    2.2 GHz x 2.9 ms), SQ_VALU_MFMA_BUSY_CYCLES summed over the 1,024 SIMDs in cycles (16 per v_mfma_f64_4x4x4_4b_f64)."""
    import json
    cur = sqlite3.connect(pmc_db).cursor()
    vals = {}
    for k, c, n, s_ in cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
        vals.setdefault(k, {})[c] = (n, s_)
    dur = {}
    for name, calls, tot in sqlite3.connect(trace_db).cursor().execute("select name,total_calls,total_duration from top_kernels"):
        dur[name] = (calls, tot)
    out = {"_note": "rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES over one C3 solve "
                    "(tools/c3_once.py 65536 1), durations from a --kernel-trace --stats pass of the same command; "
                    "utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); TFLOPs = MFMA instructions x 512 "
                    "flops / kernel time (v_mfma_f64_4x4x4_4b_f64: four 4x4x4 products; measured peak of that instruction 73.2 TFLOP/s, "
                    "profiles/r04_mfma4x4x4_probe.txt)", "_commit": commit, "_src_hash": kernel_src_hash(), "kernels": {}}
    for k, v in vals.items():
        if "SQ_INSTS_MFMA" not in v or v["SQ_INSTS_MFMA"][1] <= 0 or "gqp::" not in k:
            continue
        short = k.split("gqp::")[1].split("(")[0]
        n = v["SQ_INSTS_MFMA"][0]
        insts, busy, gui = v["SQ_INSTS_MFMA"][1], v.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[1], v.get("GRBM_GUI_ACTIVE", (0, 0))[1]
        d = dur.get(k)
        e = {"launches": n, "mfma_instructions_per_launch": insts / n, "mfma_busy_cycles_per_launch": busy / n,
             "gpu_active_cycles_per_launch_per_xcd": gui / 8 / n,
             "mfma_utilisation": busy / (1024.0 * gui / 8) if gui else None}
        if d:
            e["avg_us"] = d[1] / d[0]           # (top_kernels.total_duration is in microseconds)
            secs = d[1] * 1e-6
            e["mfma_TFLOPs"] = insts * (d[0] / n) * 512.0 / secs / 1e12 if secs > 0 else None
            e["frac_of_measured_mfma_peak_73.2"] = e["mfma_TFLOPs"] / 73.2 if e["mfma_TFLOPs"] else None
        out["kernels"][short] = e
    print(json.dumps(out, indent=1))


def pmc(db):
    cur = sqlite3.connect(db).cursor()
    q = ("select kernel_name, counter_name, count(*), sum(value), avg(value), max(value) from counters_collection "
         "group by kernel_name, counter_name order by sum(value) desc")
    print(f"{'kernel':70s} {'counter':12s} {'n':>5s} {'sum':>16s} {'avg/launch':>14s} {'max':>14s}   (FETCH_SIZE/WRITE_SIZE in KiB)")
    for k, c, n, s, a, m in cur.execute(q):
        print(f"{k[:70]:70s} {c:12s} {n:5d} {s:16.1f} {a:14.1f} {m:14.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "mfma":
        mfma(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None)
    elif sys.argv[1] in ("traffic", "sections"):
        {"traffic": traffic, "sections": sections}[sys.argv[1]](sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None)
    else:
        {"kernel": kernel, "pmc": pmc, "timeline": timeline}[sys.argv[1]](*sys.argv[2:])
