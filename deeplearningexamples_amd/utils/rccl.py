"""RCCL through the C ABI (csrc/rccl_comm.hip: dle_rccl_*), the `DLE_COMM=rccl` alternative to torch's ProcessGroupNCCL.

SURVEY.md 8 row b4: "C-ABI wrappers over librccl.so (ncclCommInitRank, ncclAllReduce, grouped ncclSend / ncclRecv) on dedicated HIP
streams with event fencing against the compute stream; unique-id exchange through the existing MASTER_ADDR/PORT env rendezvous".
The rendezvous is torch.distributed's (whatever backend the process group was initialised with moves the 128-byte id, once);
afterwards every collective of utils/comm.py on a device tensor is ONE call into the library, enqueued on torch's CURRENT stream --
the engines already issue their exchanges inside `with torch.cuda.stream(comm_stream)` blocks fenced with wait_stream / events
(utils/buckets.py, dlrm/engine.py), so the stream discipline is theirs and no host synchronisation happens here.

Reference call sites this stands in for: Classification/ConvNets/image_classification/training.py:78-84,
LanguageModeling/BERT/run_pretraining.py:461-470, Recommendation/DLRM/dlrm/model/distributed.py:68,95.
"""
import ctypes
import os

import torch
import torch.distributed as dist

from .. import _cabi as C

_DT = {torch.float32: C.F32, torch.float16: C.F16, torch.bfloat16: C.BF16, torch.int32: 100, torch.int64: 101, torch.uint8: 102}
SUM, MAX, AVG = 0, 2, 4
_comms = {}


def enabled():
    return os.environ.get("DLE_COMM", "torch").lower() == "rccl"


class RcclComm:
    """One RCCL communicator over the ranks of a torch process group (used for the rendezvous only)."""

    def __init__(self, group=None):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        lib = C.lib()
        if not lib.dle_rccl_available():
            raise C.DleError("DLE_COMM=rccl: " + lib.dle_last_error().decode("utf-8", "replace"))
        ident = ctypes.create_string_buffer(128)
        if self.rank == 0:
            C.check(lib.dle_rccl_unique_id(ident), "dle_rccl_unique_id")
        box = [bytes(ident.raw)]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ident = ctypes.create_string_buffer(box[0], 128)
        handle = ctypes.c_void_p()
        C.check(lib.dle_rccl_init(ident, self.rank, self.world, ctypes.byref(handle)), "dle_rccl_init")
        self.handle = handle

    def count(self):
        """Ranks of the communicator as RCCL itself reports them (ncclCommCount)."""
        return int(C.lib().dle_rccl_count(self.handle))

    def allreduce_(self, t, op):
        if not t.is_contiguous():
            raise ValueError("rccl all-reduce: contiguous tensor expected")
        C.check(C.lib().dle_rccl_allreduce(self.handle, C.ptr(t), t.numel(), _DT[t.dtype], op, C.stream()), "dle_rccl_allreduce")
        return t

    def broadcast_(self, t, src=0):
        if not t.is_contiguous():
            raise ValueError("rccl broadcast: contiguous tensor expected")
        C.check(C.lib().dle_rccl_broadcast(self.handle, C.ptr(t), t.numel() * t.element_size(), src, C.stream()), "dle_rccl_broadcast")
        return t

    def all_to_all_single(self, out, inp, out_splits, in_splits):
        if not (out.is_contiguous() and inp.is_contiguous()) or out.dtype != inp.dtype:
            raise ValueError("rccl all-to-all: contiguous tensors of one dtype expected")
        if len(out_splits) != self.world or len(in_splits) != self.world:
            raise ValueError("rccl all-to-all: one split per rank expected")
        e = inp.element_size()
        sb = (ctypes.c_int64 * self.world)(*[int(n) * e for n in in_splits])
        rb = (ctypes.c_int64 * self.world)(*[int(n) * e for n in out_splits])
        C.check(C.lib().dle_rccl_alltoallv(self.handle, C.ptr(inp), sb, C.ptr(out), rb, self.world, C.stream()), "dle_rccl_alltoallv")
        return out

    def destroy(self):
        if self.handle:
            C.lib().dle_rccl_destroy(self.handle)
            self.handle = None


def comm_for(group=None):
    """The (lazily created) communicator of a process group."""
    key = id(group) if group is not None else 0
    if key not in _comms:
        _comms[key] = RcclComm(group)
    return _comms[key]


def destroy_all():
    for c in _comms.values():
        c.destroy()
    _comms.clear()
