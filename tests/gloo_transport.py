"""TEST INFRASTRUCTURE: a transport table (ocp_qp_gpu_comm_ops, include/acados_amd/ocp_qp_gpu_batch.h) over torch.distributed
-- gloo in the CPU tier -- so that the library's OWN collective (ocp_qp_gpu_batch_gather / _gather_root / _gather_v: packing,
offsets, rank order, uneven shards, error handling inside a group) runs with world size > 1 without a GPU.  The buffers the
host-simulation build hands to the table are host pointers; with RCCL they are device pointers and the table is the
library's own (gpu_batch.hip rccl_op_*)."""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

_AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
_SR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p)
_GR = C.CFUNCTYPE(C.c_int, C.c_void_p)


class Ops(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("all_gather", _AG), ("send", _SR), ("recv", _SR), ("group_start", _GR), ("group_end", _GR)]


_NP = {2: np.int32, 8: np.float64}       # ncclInt32, ncclFloat64


def _view(ptr, count, dtype):
    t = _NP[dtype]
    buf = (C.c_char * (count * np.dtype(t).itemsize)).from_address(ptr)
    return torch.from_numpy(np.frombuffer(buf, dtype=t))


class GlooTransport:
    """fail_send_after: the n-th send of a group reports an error (the library must still end the group)"""

    def __init__(self, fail_send_after=None):
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.calls = {"all_gather": 0, "send": 0, "recv": 0, "group_start": 0, "group_end": 0}
        self.open = False
        self.failed = False
        self.pending = []
        self.fail_send_after = fail_send_after
        self.ops = Ops(None, _AG(self._all_gather), _SR(self._send), _SR(self._recv), _GR(self._start), _GR(self._end))

    def _all_gather(self, ctx, send, recv, count, dtype, stream):
        self.calls["all_gather"] += 1
        out = _view(recv, count * self.world, dtype)
        dist.all_gather([out[r * count:(r + 1) * count] for r in range(self.world)], _view(send, count, dtype).clone())
        return 0

    def _send(self, ctx, buf, count, dtype, peer, stream):
        self.calls["send"] += 1
        if not self.open:
            return 5
        if self.fail_send_after is not None and self.calls["send"] > self.fail_send_after:
            self.failed = True        # a transport in an error state drops what the group had queued (ncclCommAbort semantics)
            return 1
        self.pending.append(("s", _view(buf, count, dtype), peer))
        return 0

    def _recv(self, ctx, buf, count, dtype, peer, stream):
        self.calls["recv"] += 1
        if not self.open:
            return 5
        self.pending.append(("r", _view(buf, count, dtype), peer))
        return 0

    def _start(self, ctx):
        self.calls["group_start"] += 1
        self.open, self.pending, self.failed = True, [], False
        return 0

    def _end(self, ctx):
        self.calls["group_end"] += 1
        self.open = False
        if self.failed:
            self.pending = []
            return 1
        # transfers to oneself are matched in order and copied; the others go out as one batch of isend / irecv
        own_s = [t for k, t, p in self.pending if k == "s" and p == self.rank]
        own_r = [t for k, t, p in self.pending if k == "r" and p == self.rank]
        for s, r in zip(own_s, own_r):
            r.copy_(s)
        reqs = [dist.P2POp(dist.isend if k == "s" else dist.irecv, t, p) for k, t, p in self.pending if p != self.rank]
        self.pending = []
        if reqs:
            for w in dist.batch_isend_irecv(reqs):
                w.wait()
        return 0
