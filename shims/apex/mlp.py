"""apex.mlp.{MLP, MlpFunction} on the MI355X GEMM (apex/mlp/mlp.py; used by Recommendation/DLRM/dlrm/nn/mlps.py:38-43).

Same module surface: `MLP(mlp_sizes, bias=True, activation='relu')`, `.weights` / `.biases` ParameterLists, weight i is
[mlp_sizes[i+1], mlp_sizes[i]], the activation follows EVERY layer.  Forward = dle_gemm with the bias + ReLU epilogue,
backward = data-gradient GEMMs with the ReLU mask fused into their epilogue, split-K weight gradients, column-sum
bias gradients."""
import math

import torch
from torch import nn

from deeplearningexamples_amd import _cabi as C
from deeplearningexamples_amd import functional as F


class MlpFunction(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.half)
    def forward(ctx, bias, activation, input, *args):
        n = len(args) // 2 if bias else len(args)
        weights, biases = args[:n], (args[n:] if bias else [None] * n)
        if activation not in (0, 1):
            raise ValueError("apex.mlp shim: activation must be 'none' or 'relu' (sigmoid is not on the DLRM path)")
        act = C.ACT_RELU if activation == 1 else C.ACT_NONE
        x = input.reshape(-1, input.shape[-1]).contiguous()
        acts = [x]
        for w, b in zip(weights, biases):
            y, _ = F.linear_fwd(acts[-1], w.contiguous(), b.float().contiguous() if b is not None else None, act)
            acts.append(y)
        ctx.save_for_backward(*acts, *weights)
        ctx.n, ctx.bias, ctx.act, ctx.in_shape = n, bias, act, input.shape
        return acts[-1].reshape(*input.shape[:-1], acts[-1].shape[-1])

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_o):
        n = ctx.n
        acts, weights = ctx.saved_tensors[:n + 1], ctx.saved_tensors[n + 1:]
        g = grad_o.reshape(-1, grad_o.shape[-1]).contiguous().to(acts[-1].dtype)
        if ctx.act == C.ACT_RELU:
            g = F.relu_bwd(g, acts[-1])
        gw, gb = [None] * n, [None] * n
        gx = None
        for i in range(n - 1, -1, -1):
            gw[i] = F.linear_wgrad(g, acts[i]).to(weights[i].dtype)
            if ctx.bias:
                gb[i] = F.colsum(g).to(weights[i].dtype)
            if i > 0:
                g = F.linear_dgrad(g, weights[i].contiguous(), mask_src=acts[i] if ctx.act == C.ACT_RELU else None)
            else:
                gx = F.linear_dgrad(g, weights[i].contiguous()).reshape(ctx.in_shape)
        return (None, None, gx, *gw, *(gb if ctx.bias else []))


mlp_function = MlpFunction.apply


class MLP(nn.Module):
    def __init__(self, mlp_sizes, bias=True, activation="relu"):
        super().__init__()
        self.num_layers = len(mlp_sizes) - 1
        self.mlp_sizes = list(mlp_sizes)
        self.bias = 1 if bias else 0
        if activation not in ("none", "relu", "sigmoid"):
            raise TypeError("activation must be relu or none.")
        self.activation = {"none": 0, "relu": 1, "sigmoid": 2}[activation]
        self.weights = nn.ParameterList()
        self.biases = nn.ParameterList()
        for i in range(self.num_layers):
            self.weights.append(nn.Parameter(torch.empty(mlp_sizes[i + 1], mlp_sizes[i])))
            if self.bias:
                self.biases.append(nn.Parameter(torch.empty(mlp_sizes[i + 1])))
        self.reset_parameters()

    def reset_parameters(self):
        for w in self.weights:
            nn.init.normal_(w, 0.0, math.sqrt(2.0 / float(w.size(0) + w.size(1))))
        for b in self.biases:
            nn.init.normal_(b, 0.0, math.sqrt(1.0 / float(b.size(0))))

    def forward(self, input):
        return mlp_function(self.bias, self.activation, input, *self.weights, *self.biases)

    def extra_repr(self):
        return "MLP sizes: %s, Bias=%s, activation=%s" % (self.mlp_sizes, self.bias, self.activation)
