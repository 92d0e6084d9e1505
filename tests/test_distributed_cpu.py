"""CPU tier: the N>1 path (instance sharding + gather + max-reduction) with gloo, world_size 2.
Each rank solves its shard through the C-ABI (kernel sources under the host-simulation shim) and
the gathered result must equal the single-process solve of the whole batch, instance by instance."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT

WORKER = r"""
import ctypes, os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, os.environ["REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["REPO_ROOT"], "tests"))
from acados_amd import OcpQpGpuBatch, _lib
from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch
from acados_amd.sharding import shard_range, gather_instances, reduce_max
from hostsim.build import build
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
L = _lib.bind(ctypes.CDLL(build()))
N, TOTAL = 6, 37
data = random_lqr_batch(N=N, batch=TOTAL, seed=4)
lo, hi = shard_range(TOTAL, rank, world)
local = {k: np.ascontiguousarray(v[lo:hi]) for k, v in data.items()}
gb = OcpQpGpuBatch(lqr_dims(N, 8, 3), hi - lo, _clib=L)
fill_lqr_batch(gb, local, N)
assert gb.solve() == 0
u0 = gather_instances(gb.get("u", 0), TOTAL, dist)
it = gather_instances(gb.info("iter").astype(np.float64)[:, None], TOTAL, dist)
tmax = reduce_max(float(rank + 1), dist)
if rank == 0:
    np.save(os.environ["OUT_FILE"], np.concatenate([u0, it], axis=1))
    assert tmax == float(world)
dist.destroy_process_group()
"""


def test_sharded_solve_equals_single_process(tmp_path, hostsim_lib):
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch
    from acados_amd.sharding import shard_range
    assert [shard_range(37, r, 2) for r in range(2)] == [(0, 19), (19, 37)]
    assert [shard_range(8, r, 3) for r in range(3)] == [(0, 3), (3, 6), (6, 8)]
    out = tmp_path / "gathered.npy"
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, REPO_ROOT=ROOT, OUT_FILE=str(out), MASTER_ADDR="127.0.0.1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)], env=env, timeout=300)
    got = np.load(out)
    N, TOTAL = 6, 37
    data = random_lqr_batch(N=N, batch=TOTAL, seed=4)
    gb = OcpQpGpuBatch(lqr_dims(N, 8, 3), TOTAL, _clib=hostsim_lib)
    fill_lqr_batch(gb, data, N)
    assert gb.solve() == 0
    assert np.array_equal(got[:, :3], gb.get("u", 0))
    assert np.array_equal(got[:, 3], gb.info("iter").astype(np.float64))


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` started as ONE process must become two ranks (round-3 review: the flag was parsed and never
    used); --dry-run walks the launch + sharding path on gloo without touching a GPU and prints the one JSON line"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for extra, key in (([], "instances"), (["--config", "c5"], "per_class")):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"] + extra, env=env, timeout=300,
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout.decode().strip().splitlines()[-1]
        line = json.loads(out)
        assert line["dry_run"] and line["n_gpus"] == 2 and line["gather"]["ranks"] == 2
        assert [r["rank"] for r in line["ranks"]] == [0, 1] and line["ranks"][0]["pid"] != line["ranks"][1]["pid"]
        assert all(key in r for r in line["ranks"])
    # the driver's largest job: 8 ranks on the mixed-class configuration.  Shards of a class are UNEVEN (58,254 = 6 x 7,282 + 2 x 7,281):
    # every rank reports the same per-rank counts -- what main_c5 passes to the exact-count gather (round-4 advice: it assumed even
    # shards) -- they tile the class without gap or overlap, and the line stays below the 4 KB the driver can read
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--config", "c5"], env=env, timeout=600,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout.decode().strip().splitlines()[-1]
    line = json.loads(out)
    assert len(out) < 4096 and line["n_gpus"] == 8 and [r["rank"] for r in line["ranks"]] == list(range(8))
    per_class = 524288 // 9
    counts = line["ranks"][0]["gather_counts"]
    assert all(r["gather_counts"] == counts for r in line["ranks"]) and sum(counts) == per_class and sorted(set(counts)) == [7281, 7282]
    edges = [r["per_class"] for r in line["ranks"]]
    assert edges[0][0] == 0 and edges[-1][1] == per_class and all(edges[i][1] == edges[i + 1][0] for i in range(7))
    assert [hi - lo for lo, hi in edges] == counts
    # one rank: no launcher in between
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"], env=env, timeout=120, stdout=subprocess.PIPE,
                         check=True).stdout.decode().strip().splitlines()[-1]
    assert json.loads(out)["n_gpus"] == 1
