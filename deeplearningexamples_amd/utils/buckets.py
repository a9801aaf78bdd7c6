"""Gradient buckets over one flat fp32 buffer + mean all-reduce (the data-parallel exchange of the train step).

Replaces torch DDP's reducer (Classification/ConvNets/image_classification/training.py:78-84: 25 MB buckets;
LanguageModeling/BERT/run_pretraining.py:455-475: one bucket + pre-divide hook).  Parameters are laid out in the
flat buffer in the order their gradients become final during backward, buckets are cut at parameter
boundaries, and a bucket is all-reduced (mean) as soon as its last gradient exists -- on a side stream when a
CUDA stream is given, so RCCL traffic over xGMI overlaps the rest of the backward pass.
"""
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def cut_buckets(named_numels: Sequence[Tuple[str, int]], bucket_bytes: int) -> List[Tuple[int, int, str]]:
    """-> [(start, end, name of the parameter that completes the bucket)], element offsets into the flat buffer."""
    out, start, pos, lim = [], 0, 0, max(bucket_bytes // 4, 1)
    last = None
    for name, n in named_numels:
        pos += n
        last = name
        if pos - start >= lim:
            out.append((start, pos, name))
            start = pos
    if start < pos:
        out.append((start, pos, last))
    return out


from .comm import allreduce_mean_, allreduce_sum_  # noqa: E402,F401  (re-exported: the engines import it from here)


class GradBuckets:
    """`named_numels` lists the parameters in the order they lie in `flat`.  reverse=False: gradients complete in that
    order (ResNet: the flat buffer is laid out in backward-completion order).  reverse=True: they complete from the
    END of the buffer towards its start (BERT: the buffer follows named_parameters(), backward runs heads -> layer 23
    -> ... -> embeddings), buckets are cut walking backwards and fire on the parameter with the LOWEST offset."""

    def __init__(self, flat: torch.Tensor, named_numels, bucket_mb=25, group=None, comm_stream=None, reverse=False,
                 wire_dtype=None):
        """wire_dtype (torch.float16 / torch.bfloat16, optional): the buckets travel in 16 bits -- each rank pre-divides its
        fp32 gradients by the world size, rounds them into a 16-bit staging buffer, the SUM all-reduce runs on that buffer and
        the result is widened back into the flat fp32 buffer (half the bytes over xGMI; what the reference gets from
        --allreduce_post_accumulation_fp16 with an fp16 model, run_pretraining.py:416-417,461-475)."""
        self.flat, self.group, self.stream = flat, group, comm_stream
        self.wire_dtype = wire_dtype
        self._wire = None
        self.wire_divisor = 1.0     # extra pre-division of the 16-bit wire copy (gradient-accumulation count: the reference's
                                    # micro-step losses are already divided by it, run_pretraining.py:521), undone after widening
        self.extra_streams = []
        if reverse:
            total = sum(n for _, n in named_numels)
            cut = cut_buckets(list(reversed(list(named_numels))), bucket_mb * (1 << 20))
            self.buckets = [(total - e, total - s, name) for (s, e, name) in cut]
            assert self.buckets[0][1] == flat.numel() and self.buckets[-1][0] == 0
        else:
            self.buckets = cut_buckets(named_numels, bucket_mb * (1 << 20))
            assert self.buckets[-1][1] == flat.numel() and self.buckets[0][0] == 0
        self._by_last = {b[2]: i for i, b in enumerate(self.buckets)}
        self.fired = 0

    def grad_ready(self, name):
        """Call when the gradient of `name` has been written; fires the bucket that this parameter completes."""
        i = self._by_last.get(name)
        if i is None:
            return False
        s, e, _ = self.buckets[i]
        self.fired += 1
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())
            for extra in self.extra_streams:          # streams that write gradients beside the current one (weight-gradient stream)
                self.stream.wait_stream(extra)
            with torch.cuda.stream(self.stream):
                self._reduce(s, e)
        else:
            self._reduce(s, e)
        return True

    def _reduce(self, s, e):
        if self.wire_dtype is None:
            allreduce_mean_(self.flat[s:e], self.group)
            return
        if self._wire is None:          # one staging buffer of the largest bucket (buckets are reduced one after the other
            n = max(b[1] - b[0] for b in self.buckets)                                    # on the communication stream)
            self._wire = torch.empty(n, dtype=self.wire_dtype, device=self.flat.device)
        w = self._wire[:e - s]
        torch.div(self.flat[s:e], float(dist.get_world_size(self.group)) * float(self.wire_divisor), out=self.flat[s:e])
        w.copy_(self.flat[s:e])
        allreduce_sum_(w, self.group)
        self.flat[s:e].copy_(w)
        if self.wire_divisor != 1.0:    # the optimizer's scale bookkeeping expects the undivided sum of the micro-steps
            self.flat[s:e].mul_(float(self.wire_divisor))

    def wait(self):
        """Make the compute stream wait for every bucket; all of them must have been launched by now."""
        assert self.fired == len(self.buckets), "only %d of %d gradient buckets were reduced" % (self.fired, len(self.buckets))
        self.fired = 0
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
