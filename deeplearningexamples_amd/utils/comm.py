"""Collectives of the train steps: one process per GPU, torch.distributed with backend "nccl" (= RCCL over xGMI).

Replaces what the reference gets from torch DDP / torch.distributed over NCCL
(Classification/ConvNets/image_classification/training.py:78-84, LanguageModeling/BERT/run_pretraining.py:455-475,
Recommendation/DLRM/dlrm/model/distributed.py:68,95).  Every exchange of the engines goes through these helpers.

The `gloo` branch exists for the tests only: the CPU suite (world_size 2, CPU tensors) and the two-process run on a
ONE-GPU box (RCCL refuses two ranks on one device), where device tensors are staged through host memory around
the gloo call.  It moves bytes; no arithmetic of the product path runs on the CPU.
"""
import torch
import torch.distributed as dist

from . import rccl as _rccl


def _staged(t, group):
    return t.is_cuda and dist.get_backend(group) != "nccl"


def _direct(t, group):
    """DLE_COMM=rccl: device tensors go through the C-ABI RCCL wrappers (utils/rccl.py, csrc/rccl_comm.hip) on the current stream
    instead of torch's ProcessGroupNCCL -- SURVEY.md 8 row b4.  The default (`torch`) keeps the reference's own boundary."""
    return _rccl.comm_for(group) if (t.is_cuda and _rccl.enabled()) else None


def ranks_seen(device=None, group=None):
    """How many ranks actually take part in a collective of this process group: a SUM all-reduce of ones over the path the train
    steps use (RCCL's own ncclCommCount when the direct path is on).  bench.py prints it next to the world size."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    dev = device if (device is not None and dist.get_backend(group) == "nccl") else "cpu"
    t = torch.ones(1, dtype=torch.float32, device=dev)
    if t.is_cuda and _rccl.enabled():
        c = _rccl.comm_for(group)
        c.allreduce_(t, _rccl.SUM)
        return min(int(t.item()), c.count())
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())


def allreduce_mean_(t: torch.Tensor, group=None):
    """In-place mean over the ranks.  RCCL has a native AVG; gloo sums and scales."""
    c = _direct(t, group)
    if c is not None:
        return c.allreduce_(t, _rccl.AVG)
    if dist.get_backend(group) == "nccl":
        dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group)
    elif t.is_cuda:
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h.div_(dist.get_world_size(group)))
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t.div_(dist.get_world_size(group))
    return t


def allreduce_sum_(t: torch.Tensor, group=None):
    """In-place sum over the ranks (16-bit wire buffers of the gradient buckets: pre-divided by the world size)."""
    c = _direct(t, group)
    if c is not None:
        return c.allreduce_(t, _rccl.SUM)
    if _staged(t, group):
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def allreduce_max_(t: torch.Tensor, group=None):
    """In-place maximum over the ranks (the found-inf flag of model-parallel parts)."""
    c = _direct(t, group)
    if c is not None:
        return c.allreduce_(t, _rccl.MAX)
    if _staged(t, group):
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t


def broadcast_(t: torch.Tensor, src=0, group=None):
    c = _direct(t, group)
    if c is not None and t.is_contiguous():
        return c.broadcast_(t, src)
    if _staged(t, group):
        h = t.detach().cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)
    return t


def _pairwise_exchange_host(ho, hi, out_splits, in_splits, group=None):
    """The all-to-all on HOST tensors by pairwise send / recv of the raw bytes: gloo has no all-to-all for every dtype
    and its scatter wants equal blocks (the blocks of a table-wise DLRM placement differ from rank to rank)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    outs = list(ho.view(-1).split(list(out_splits)))
    ins = [c.contiguous() for c in hi.view(-1).split(list(in_splits))]
    reqs = []
    for r in range(world):
        if r == rank:
            outs[r].copy_(ins[r])
            continue
        if ins[r].numel():
            reqs.append(dist.isend(ins[r].view(torch.uint8), dst=r, group=group))
        if outs[r].numel():
            reqs.append(dist.irecv(outs[r].view(torch.uint8), src=r, group=group))
    for q in reqs:
        q.wait()
    return ho


def all_to_all_single(out: torch.Tensor, inp: torch.Tensor, out_splits, in_splits, group=None):
    """torch.distributed.all_to_all_single with element split lists (DLRM bottom -> top exchange and its reverse)."""
    c = _direct(out, group)
    if c is not None:
        return c.all_to_all_single(out, inp, out_splits, in_splits)
    if _staged(out, group):
        ho = torch.empty(out.shape, dtype=out.dtype)
        out.copy_(_pairwise_exchange_host(ho, inp.detach().cpu(), out_splits, in_splits, group))
    else:
        dist.all_to_all_single(out, inp, list(out_splits), list(in_splits), group=group)
    return out


def broadcast_parameters_(tensors, src=0, group=None):
    """Make every data-parallel replica start from rank `src`'s values (what torch DDP does when it wraps a module:
    ConvNets training.py:78-84, BERT run_pretraining.py:455-460, DLRM dlrm/scripts/main.py:463-466).  `tensors`:
    parameters and buffers; integer buffers (num_batches_tracked) included."""
    for t in tensors:
        broadcast_(t.data if isinstance(t, torch.nn.Parameter) else t, src=src, group=group)
