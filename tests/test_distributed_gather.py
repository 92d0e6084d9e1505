"""CPU tier, world size 2: the PRODUCT collective (ocp_qp_gpu_batch_gather / _gather_root / _gather_v in gpu_batch.hip) --
not a Python re-implementation of it -- driven through a host-supplied transport table (ocp_qp_gpu_comm_create_from_ops) with
gloo underneath: packing of the solution blob, rank order, instance offsets of UNEVEN shards, gather to either root, the
three-all-gather fast path of even shards, and that a transport error inside a group still ends the group.  The gathered
payload must equal, bit for bit, the output blob of a single-process solve of the whole batch."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT

WORKER = r"""
import ctypes as C, os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, os.environ["REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["REPO_ROOT"], "tests"))
from acados_amd import OcpQpGpuBatch, _lib
from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch
from acados_amd.sharding import shard_range
from gloo_transport import GlooTransport
from hostsim.build import build
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
L = _lib.bind(C.CDLL(build()))
N = 6
out = {}
for TOTAL in (37, 38):                      # uneven (19 + 18) and even (19 + 19) shards
    data = random_lqr_batch(N=N, batch=TOTAL, seed=4)
    lo, hi = shard_range(TOTAL, rank, world)
    gb = OcpQpGpuBatch(lqr_dims(N, 8, 3), hi - lo, _clib=L)
    fill_lqr_batch(gb, {k: np.ascontiguousarray(v[lo:hi]) for k, v in data.items()}, N)
    assert gb.solve() == 0
    tr = GlooTransport()
    comm = C.c_void_p(L.ocp_qp_gpu_comm_create_from_ops(C.byref(tr.ops), world, rank))
    assert comm.value
    Lout = L.ocp_qp_gpu_batch_bulk_len(gb._h, 1)
    counts = np.array([shard_range(TOTAL, r, world)[1] - shard_range(TOTAL, r, world)[0] for r in range(world)], dtype=np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    def bufs():
        return np.full((TOTAL, Lout), np.nan), np.full((TOTAL, 2), -7, dtype=np.int32), np.full(world, np.nan)
    # every rank receives everything, exact counts
    sol, info, tm = bufs()
    assert L.ocp_qp_gpu_batch_gather_v(gb._h, comm, -1, p(counts), p(sol), p(info), p(tm)) == 0
    if TOTAL % world == 0:
        assert tr.calls["all_gather"] == 3 and tr.calls["group_start"] == 0       # even shards: the ring path
        sol2, info2, tm2 = bufs()
        assert L.ocp_qp_gpu_batch_gather(gb._h, comm, p(sol2), p(info2), p(tm2)) == 0
        assert np.array_equal(sol2, sol) and np.array_equal(info2, info)
    else:
        assert tr.calls["all_gather"] == 0 and tr.calls["group_start"] == 1 and tr.calls["group_end"] == 1
        assert tr.calls["send"] == 3 * world and tr.calls["recv"] == 3 * world
    assert not np.isnan(sol).any() and not np.isnan(tm).any() and (info[:, 0] == 0).all()
    # gather to either root: the other rank's buffers may be NULL
    for root in range(world):
        s_r, i_r, t_r = bufs()
        null = C.c_void_p(0)
        rc = L.ocp_qp_gpu_batch_gather_v(gb._h, comm, root, p(counts), p(s_r) if rank == root else null, p(i_r) if rank == root else null,
                                         p(t_r) if rank == root else null)
        assert rc == 0
        if rank == root:
            assert np.array_equal(s_r, sol) and np.array_equal(i_r, info) and np.array_equal(t_r, tm)
        else:
            assert np.isnan(s_r).all()
    # this rank's slice sits at its instance offset and is what the getters return
    n0 = C.c_int(0)
    off = L.ocp_qp_gpu_batch_bulk_offset(gb._h, 1, b"u", 0, C.byref(n0))
    assert np.array_equal(sol[lo:hi, off:off + n0.value], gb.get("u", 0))
    assert np.array_equal(info[lo:hi, 1], gb.info("iter"))
    # a count that is not this rank's batch size is refused before anything is sent
    bad = counts.copy(); bad[rank] += 1
    before = dict(tr.calls)
    assert L.ocp_qp_gpu_batch_gather_v(gb._h, comm, -1, p(bad), p(sol), p(info), p(tm)) == -1 and tr.calls == before
    assert L.ocp_qp_gpu_batch_gather_v(gb._h, comm, world, p(counts), p(sol), p(info), p(tm)) == -1       # root out of range
    L.ocp_qp_gpu_comm_destroy(comm)
    # a transport that fails in the middle of a group: -1 comes back AND the group was ended (on every rank alike)
    trf = GlooTransport(fail_send_after=1)
    commf = C.c_void_p(L.ocp_qp_gpu_comm_create_from_ops(C.byref(trf.ops), world, rank))
    s_f, i_f, t_f = bufs()
    assert L.ocp_qp_gpu_batch_gather_v(gb._h, commf, 0, p(counts), p(s_f), p(i_f), p(t_f)) == -1
    assert trf.calls["group_start"] == 1 and trf.calls["group_end"] == 1 and not trf.open
    L.ocp_qp_gpu_comm_destroy(commf)
    # an incomplete table is refused
    from gloo_transport import Ops
    assert not L.ocp_qp_gpu_comm_create_from_ops(C.byref(Ops()), world, rank)
    dist.barrier()
    if rank == 0:
        out[f"sol{TOTAL}"], out[f"info{TOTAL}"] = sol, info
if rank == 0:
    np.savez(os.environ["OUT_FILE"], **out)
dist.destroy_process_group()
"""


def test_product_gather_world_size_2(tmp_path, hostsim_lib):
    import ctypes as C
    from acados_amd import OcpQpGpuBatch
    from acados_amd.generators import fill_lqr_batch, lqr_dims, random_lqr_batch
    out = tmp_path / "gathered.npz"
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, REPO_ROOT=ROOT, OUT_FILE=str(out), MASTER_ADDR="127.0.0.1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)], env=env, timeout=600)
    got = np.load(out)
    N = 6
    for TOTAL in (37, 38):
        data = random_lqr_batch(N=N, batch=TOTAL, seed=4)
        gb = OcpQpGpuBatch(lqr_dims(N, 8, 3), TOTAL, _clib=hostsim_lib)
        fill_lqr_batch(gb, data, N)
        assert gb.solve() == 0
        Lout = hostsim_lib.ocp_qp_gpu_batch_bulk_len(gb._h, 1)
        blob = np.zeros((TOTAL, Lout))
        assert hostsim_lib.ocp_qp_gpu_batch_get_bulk(gb._h, blob.ctypes.data_as(C.c_void_p), 0) == 0
        assert np.array_equal(got[f"sol{TOTAL}"], blob)            # rank order == instance order, no padding, bit for bit
        assert np.array_equal(got[f"info{TOTAL}"][:, 1], gb.info("iter")) and np.array_equal(got[f"info{TOTAL}"][:, 0], gb.info("status"))
