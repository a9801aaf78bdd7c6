"""Whole-step HIP-graph capture and replay.

Mirrors the reference's CudaGraphWrapper (Recommendation/DLRM/dlrm/scripts/main.py:194-274) and the two-graph scheme of
BERT (LanguageModeling/BERT/run_pretraining.py:602-640): the train step runs a few times eagerly on a side stream
(allocator warm-up), is captured ONCE into a hipGraph with its inputs in static device buffers, and every later step is a
copy into those buffers plus one graph launch.  The trainers qualify by construction: a fixed kernel sequence on the
current stream, every per-step scalar (learning rate, loss scale, found-inf, LAMB step) a device tensor, no host
synchronisation inside the step (BertTrainer: construct it with max_predictions_per_seq so that the masked-row
selection has a static shape).  torch.cuda.CUDAGraph is hipGraph on ROCm.
"""
from typing import Callable, List, Optional

import torch


class GraphedStep:
    """step_fn(*tensors) -> loss tensor.  Call the wrapper like step_fn; the returned tensor is the graph's static
    output (valid until the next call)."""

    def __init__(self, step_fn: Callable, enabled: bool = True, warmup_steps: int = 3, stream=None):
        """stream: wrappers whose steps depend on each other (BERT: first micro-step / accumulating micro-step / optimizer
        step) should share ONE side stream; every warm-up call is additionally ordered after the caller's current stream."""
        self._fn, self.enabled, self.warmup_steps = step_fn, enabled, warmup_steps
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.stream = (stream or torch.cuda.Stream()) if enabled else None
        self.static_args: Optional[List] = None
        self.loss = None
        self.step = -1

    def _copy_inputs(self, args):
        if len(args) != len(self.static_args):
            raise ValueError("expected %d arguments to the train step, got %d" % (len(self.static_args), len(args)))
        for src, dst in zip(args, self.static_args):
            if dst is None:
                continue
            if not isinstance(dst, torch.Tensor):
                if src != dst:
                    raise ValueError("non-tensor arguments of a captured step must not change")
                continue
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)

    def __call__(self, *args):
        self.step += 1
        if not self.enabled:
            self.loss = self._fn(*args)
            return self.loss
        if self.step == 0:
            self.static_args = [a.clone() if isinstance(a, torch.Tensor) else a for a in args]
        if self.step < self.warmup_steps:
            # every warm-up step: the side stream starts after whatever the caller's stream has enqueued (the batch being
            # produced, another wrapper's step -- e.g. the optimizer step that rewrites the weights this step reads)
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self._copy_inputs(args)
                self.loss = self._fn(*self.static_args)
            torch.cuda.current_stream().wait_stream(self.stream)
            return self.loss
        if self.graph is None:
            torch.cuda.synchronize()
            self._copy_inputs(args)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.loss = self._fn(*self.static_args)
            # the capture did not execute the step: fall through to the first replay
        self._copy_inputs(args)
        self.graph.replay()
        return self.loss
