"""16-bit STORAGE emulation for the step oracles (TEST INFRASTRUCTURE ONLY).

The AMP path keeps activations, GEMM weights and activation gradients in fp16 / bf16 between kernels while every
kernel accumulates in fp32.  `q(t, dtype)` rounds a tensor to that type where the HIP path stores it and rounds the
gradient flowing back through the same point (identity derivative), so that an oracle run with storage_dtype set
measures the precision floor of 16-bit storage on a given network -- the part of |loss_hip - loss_fp32| that no kernel
can remove.  The parity bars of the per-step loss tests are 1e-3 (BASELINE.json north_star) + this measured floor.
"""
import torch


class RoundSTE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.dtype = dtype
        return x.to(dtype).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).float(), None


def q(t, dtype):
    return t if dtype is None else RoundSTE.apply(t, dtype)
