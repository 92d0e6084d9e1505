"""The driver keeps an 8 KB tail of bench.py's stdout and parses its LAST line: round 4's 22 KB line was lost
(BENCH_r04.json parsed = null).  bench.compact_line is what is printed now; the full object goes to a detail file."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

HEAD = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def strict_loads(s):
    def bad(c):
        raise ValueError(f"non-standard JSON constant {c}")
    return json.loads(s, parse_constant=bad)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[3456]_v*_bench.json"))))
def test_committed_lines_compact_below_the_limit(path):
    full = json.load(open(path))
    if "roofline" not in full:
        pytest.skip("not a headline line")
    line = bench.compact_line(full, "gpurun_out/bench_detail.json")
    assert "\n" not in line and len(line) < bench.LINE_LIMIT <= 4096
    j = strict_loads(line)
    for k in HEAD + ("roofline", "cpu_baseline"):
        assert k in j, k
    assert j["value"] == pytest.approx(full["value"], rel=1e-6) and j["dtype"] == "f64"
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(j["roofline"])
    assert j["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-3)
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(j["cpu_baseline"])
    if "configs" in full:
        assert set(j["configs"]) == set(full["configs"])
        for name, c in j["configs"].items():
            if "solves_per_s" in full["configs"][name]:       # (a leg that was skipped on the box carries its reason instead)
                assert c["solves_per_s"] == pytest.approx(full["configs"][name]["solves_per_s"], rel=1e-3)


def test_the_round4_line_that_was_lost():
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_v6_bench.json")))
    assert len(json.dumps(full)) > 8192          # what the driver could not read
    assert len(bench.compact_line(full)) < 4096


def test_oversized_input_still_yields_a_line_below_the_limit():
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_v6_bench.json")))
    full["configs"] = {f"cls{i}": dict(full["configs"]["C5_share"]) for i in range(40)}
    line = bench.compact_line(full)
    assert len(line) < bench.LINE_LIMIT
    j = strict_loads(line)
    assert "roofline" in j and "cpu_baseline" in j and j["value"] > 0


def test_nan_and_inf_do_not_reach_the_line():
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_v6_bench.json")))
    full["roofline"]["traffic_over_algorithmic"] = float("nan")
    full["ipm"]["max_rel_primal_err_vs_oracle"] = float("inf")
    strict_loads(bench.compact_line(full))


def test_gather_guard_reports_instead_of_hanging():
    """N > 1: the solutions gather (the one step that has never met a second device) runs last and under a time limit --
    a collective that never completes costs the gather record, not the line"""
    import time
    import bench
    v, e = bench.guarded_gather(lambda: {"ms": 1.0}, world=1)
    assert v == {"ms": 1.0} and e is None
    v, e = bench.guarded_gather(lambda: {"ms": 2.0}, world=2, limit_s=5.0)
    assert v == {"ms": 2.0} and e is None
    t0 = time.time()
    v, e = bench.guarded_gather(lambda: time.sleep(30), world=2, limit_s=0.5)
    assert v is None and e.startswith("in flight") and time.time() - t0 < 5.0

    def boom():
        raise RuntimeError("ncclCommInitRank failed")
    v, e = bench.guarded_gather(boom, world=2, limit_s=5.0)
    assert v is None and "ncclCommInitRank" in e
    line = bench.compact_line({"metric": "m", "value": 1.0, "gather": {"ranks": 2, "error": e}})
    assert json.loads(line)["gather"]["error"] == e
