#!/usr/bin/env python3
"""Turn rocprofv3 (rocpd sqlite) outputs into the small text summaries kept under profiles/.
usage: summarize.py kernel <results.db>   -> per-kernel calls / total / average (kernel-trace --stats)
       summarize.py pmc <results.db>      -> per-kernel counter sums and per-launch averages"""
import sqlite3
import sys


def kernel(db):
    cur = sqlite3.connect(db).cursor()
    print(f"{'kernel':80s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
    for name, calls, tot, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(f"{name[:80]:80s} {calls:6d} {tot:12.1f} {avg:10.1f} {pct:6.2f}")
    print("\nper-dispatch durations of the IPM kernels (us), in launch order, first 60:")
    rows = cur.execute("select name,duration,vgpr_count,accum_vgpr_count,sgpr_count,scratch_size from kernels "
                       "where name like '%gqp::k_backward%' or name like '%gqp::k_forward%' or name like '%gqp::kb_%' "
                       "or name like '%gqp::kw_%' order by start limit 60").fetchall()
    for name, dur, v, a, s, sc in rows:
        short = name.split("gqp::")[1].split("(")[0]
        print(f"  {short:40s} {dur / 1e3:10.1f}  vgpr {v} agpr {a} sgpr {s} scratch {sc}")


def pmc(db):
    cur = sqlite3.connect(db).cursor()
    q = ("select kernel_name, counter_name, count(*), sum(value), avg(value), max(value) from counters_collection "
         "group by kernel_name, counter_name order by sum(value) desc")
    print(f"{'kernel':70s} {'counter':12s} {'n':>5s} {'sum':>16s} {'avg/launch':>14s} {'max':>14s}   (FETCH_SIZE/WRITE_SIZE in KiB)")
    for k, c, n, s, a, m in cur.execute(q):
        print(f"{k[:70]:70s} {c:12s} {n:5d} {s:16.1f} {a:14.1f} {m:14.1f}")


if __name__ == "__main__":
    {"kernel": kernel, "pmc": pmc}[sys.argv[1]](sys.argv[2])
