"""DLRM modules on the MI355X kernels, with the reference's module tree and state_dict key names.

Mirrors (Recommendation/DLRM/):
    dlrm/nn/mlps.py:78-114           TorchMlp        -> Mlp            (layers.{0,2,..}.weight/bias)
    dlrm/nn/embeddings.py:163-224    FusedJointEmbedding -> JointEmbedding (weight [sum N, D], offsets)
    dlrm/nn/interactions.py:40-101   DotInteraction / CudaDotInteraction -> DotInteraction
    dlrm/nn/parts.py:27-136          DlrmBottom / DlrmTop
    dlrm/model/distributed.py:104-179 DistributedDlrm
Parameters are fp32 nn.Parameters (checkpoint compatible); every layer also keeps a 16-bit working copy
that the optimizer kernel refreshes in the same pass that updates the master weight, replacing autocast's
per-forward weight casts.  Compute is explicit forward()/backward() over the C-ABI kernels: the step is a
fixed kernel sequence on one HIP stream (graph-capturable), not an autograd tape.
The autograd-Function form of the individual ops (the reference's dlrm.cuda_ext boundary) lives in
cuda_ext.py.
"""
import math
import os
from contextlib import nullcontext as _nullcontext
from typing import List, Optional, Sequence

import torch
from torch import nn

from .. import _cabi as C
from .. import functional as F


def _ceil_to(v, m):
    return (v + m - 1) // m * m


class Mlp(nn.Module):
    """(Linear + ReLU) x L.  K of the first layer is zero-padded to a multiple of 8 in the 16-bit copies so
    every operand row is 16-byte aligned for the MFMA GEMM loaders."""

    def __init__(self, input_dim: int, sizes: Sequence[int], device="cuda", compute_dtype=torch.float16):
        super().__init__()
        layers = []
        d = input_dim
        for out_d in sizes:
            layers.append(nn.Linear(d, out_d, device=device))
            layers.append(nn.ReLU(inplace=True))   # placeholder module: keeps the reference's key numbering
            d = out_d
        self.layers = nn.Sequential(*layers)
        self.input_dim = input_dim
        self.sizes = list(sizes)
        self.compute_dtype = compute_dtype
        self._initialize_weights()
        self._w16: List[torch.Tensor] = []
        self._saved = None
        self._bits = None
        # the ReLU masks of the hidden layers as ONE BIT per element, written by the forward GEMM's epilogue and read by the masked
        # data gradient instead of the 16-bit activation (csrc/gemm8_kernel.h ACT_RELU_BITS / ACT_RELU_BWD_BITS);
        # DLE_DLRM_RELU_BITS=0 keeps the activation as the mask source
        self.relu_bits = os.environ.get("DLE_DLRM_RELU_BITS", "1") != "0"

    def _initialize_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight.data, 0., math.sqrt(2. / (m.in_features + m.out_features)))
                nn.init.normal_(m.bias.data, 0., math.sqrt(1. / m.out_features))

    @property
    def linears(self) -> List[nn.Linear]:
        return [m for m in self.layers if isinstance(m, nn.Linear)]

    @property
    def weights(self):
        return [m.weight for m in self.linears]

    @property
    def biases(self):
        return [m.bias for m in self.linears]

    def load_state(self, weights, biases):
        """AbstractMlp.load_state (dlrm/nn/mlps.py:69-75): take over checkpointed fp32 weights / biases.  The 16-bit
        working copies are rebuilt by refresh_working_copies() (the checkpoint loader calls it on the whole model)."""
        for new_w, w, new_b, b in zip(weights, self.weights, biases, self.biases):
            w.data.copy_(new_w.data.to(w.device))
            b.data.copy_(new_b.data.to(b.device))

    # ---- 16-bit working copies ---------------------------------------------------------------
    def k_padded(self, i):
        return _ceil_to(self.linears[i].in_features, 8)

    def refresh_working_copies(self):
        """(Re)build the 16-bit weights from the fp32 masters (after init / checkpoint load).  Existing copies are
        refreshed IN PLACE: the trainer's multi-tensor SGD tables hold their addresses (it rewrites them in the pass that
        updates the masters), so handing out new tensors would leave the forward pass reading stale weights."""
        if self._w16:
            for i, lin in enumerate(self.linears):
                F.cast_rows(lin.weight.data, self.compute_dtype, cols_out=self.k_padded(i), out=self._w16[i])
            return self._w16
        for i, lin in enumerate(self.linears):
            self._w16.append(F.cast_rows(lin.weight.data, self.compute_dtype, cols_out=self.k_padded(i)))
        return self._w16

    def working_copies(self):
        if not self._w16:
            self.refresh_working_copies()
        return self._w16

    # ---- explicit forward / backward ------------------------------------------------------------
    def forward(self, x16: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x16 [B, k_padded(0)] 16-bit.  `out` (optional, may be a strided view) receives the last layer."""
        w16 = self.working_copies()
        acts = [x16]
        bits = [None]                               # bits[i]: keep bits of acts[i] (None: the backward masks with acts[i] itself)
        h = x16
        lins = self.linears
        for i, lin in enumerate(lins):
            m, k = h.shape[0], w16[i].shape[1]
            n = lin.out_features
            dst = out if (i == len(lins) - 1 and out is not None) else None
            r = None
            # (only a layer whose mask is consumed by the NEXT layer's masked data gradient: not the last one)
            if self.relu_bits and dst is None and i < len(lins) - 1 and h.is_cuda:
                r = F.gemm_relu_bits(h, w16[i], m, n, k, lin.bias.data)
            if r is not None:
                h, b = r
            else:
                h, b = F.gemm(h, w16[i], m, n, k, True, True, out=dst, out_dtype=self.compute_dtype,
                              bias=lin.bias.data, act=C.ACT_RELU), None
            acts.append(h)
            bits.append(b)
        self._saved = acts
        self._bits = bits
        return h

    def backward(self, gy: torch.Tensor, need_input_grad: bool = False, grads=None, masked: bool = False, defer_wgrad: bool = False,
                 skip_last_bias: bool = False):
        """gy: gradient w.r.t. the (post-ReLU) output of the last layer, 16-bit, may be a strided view;
        masked=True when the producer already applied the last ReLU's mask in its epilogue.
        Writes fp32 weight/bias gradients into `grads` [(gw, gb), ...] (views of a flat bucket) or into
        .grad.  Returns the input gradient when asked.
        defer_wgrad: run the data-gradient chain only and return (input gradient, finish) -- finish() launches the weight / bias
        gradients afterwards (the multi-rank step starts the gradient all-to-all as soon as the chain is through and overlaps
        it with them).  skip_last_bias: the producer of gy also wrote the last layer's bias gradient (its column sums)."""
        acts, w16, lins = self._saved, self.working_copies(), self.linears
        g = gy
        pending = []
        bias_done = {len(lins) - 1} if skip_last_bias else set()

        def wgrad(i, g):
            lin, x = lins[i], acts[i]
            m = x.shape[0]
            gw, gb = grads[i] if grads is not None else (_grad_buf(lin.weight), _grad_buf(lin.bias))
            kp = w16[i].shape[1]
            # dW[n, k] = g[m, n]^T x[m, k]   (fp32, split-K)
            gw_full = gw if kp == lin.in_features else torch.empty((lin.out_features, kp), dtype=torch.float32,
                                                                   device=gw.device)
            F.gemm(g, x, lin.out_features, kp, m, False, False, out=gw_full,
                   splitk=F.pick_splitk(lin.out_features, kp, m))
            if gw_full is not gw:
                gw.copy_(gw_full[:, :lin.in_features])
            if i not in bias_done:           # (written by the producer of g: the fused head / the data-gradient GEMM above)
                F.colsum(g, out=gb)

        gin = None
        for i in range(len(lins) - 1, -1, -1):
            lin = lins[i]
            y, x = acts[i + 1], acts[i]
            m = x.shape[0]
            if not masked:
                g = F.relu_bwd(g, y)
            kp = w16[i].shape[1]
            if defer_wgrad:
                pending.append((i, g))
            else:
                wgrad(i, g)
            if i > 0:
                # dX = g W, masked by the ReLU of the previous layer in the epilogue -- which also leaves the column sums of dX,
                # the bias gradient of the layer below (one pass less over every dX)
                gb_prev = grads[i - 1][1] if grads is not None else _grad_buf(lins[i - 1].bias)
                gn = None
                if kp == lins[i - 1].out_features:
                    if self._bits is not None and self._bits[i] is not None:
                        gn = F.gemm_colsum_bits(g, w16[i], m, kp, lin.out_features, self._bits[i], gb_prev)
                    if gn is None:
                        gn = F.gemm_colsum(g, w16[i], m, kp, lin.out_features, acts[i], gb_prev)
                if gn is not None:
                    bias_done.add(i - 1)
                    g = gn
                else:
                    g = F.gemm(g, w16[i], m, kp, lin.out_features, True, False, out_dtype=self.compute_dtype,
                               act=C.ACT_RELU_BWD, mask_src=acts[i])
                masked = True
            elif need_input_grad:
                gin = F.gemm(g, w16[i], m, kp, lin.out_features, True, False, out_dtype=self.compute_dtype)
        if defer_wgrad:
            def finish():
                for i, gi in pending:
                    wgrad(i, gi)
            return gin, finish
        return gin


def _grad_buf(p: torch.Tensor) -> torch.Tensor:
    if p.grad is None:
        p.grad = torch.empty_like(p)
    return p.grad


class JointEmbedding(nn.Module):
    """All local tables stacked in one fp32 matrix [sum N_t, D] + int64 offsets (FusedJointEmbedding layout).
    forward() = row-id arithmetic (idx (mod N_t) + offset_t, int64) + gather; the update is the
    duplicate-free sparse SGD applied straight from the 16-bit upstream gradient."""

    def __init__(self, categorical_feature_sizes: Sequence[int], embedding_dim: int, device="cuda",
                 hash_indices: bool = False, out_dtype=torch.float16):
        super().__init__()
        self._categorical_feature_sizes = list(categorical_feature_sizes)
        self.embedding_dim = embedding_dim
        self.hash_indices = hash_indices
        self.out_dtype = out_dtype
        off = torch.tensor([0] + list(categorical_feature_sizes), dtype=torch.int64).cumsum(0)
        self.register_buffer("offsets", off.to(device))
        self._offsets_host = off.numpy().copy()
        self.weight = nn.Parameter(torch.empty((int(off[-1]), embedding_dim), device=device), requires_grad=True)
        self._sizes_dev = torch.tensor(list(categorical_feature_sizes), dtype=torch.int64, device=device) \
            if hash_indices else None
        self._ws = None
        self._rows = None

    @property
    def num_tables(self):
        return len(self._categorical_feature_sizes)

    @property
    def weights(self):
        o = self._offsets_host
        return [self.weight.data[int(o[t]):int(o[t + 1])] for t in range(self.num_tables)]

    def load_weights(self, weights):
        for dst, src in zip(self.weights, weights):
            dst.copy_(src)

    def workspace(self):
        if self._ws is None:
            self._ws = F.EmbUpdateWorkspace(self._offsets_host, self.embedding_dim, self.weight.device)
        return self._ws

    def forward(self, categorical_inputs: torch.Tensor, out=None, out_batch_stride=0):
        """categorical_inputs int64 [B, T] -> rows gathered into `out` (or a fresh [B,T,D])."""
        self._rows = F.emb_offset_indices(categorical_inputs, self.offsets, self._sizes_dev)
        return F.emb_gather_fwd(self.weight.data, self._rows, out_dtype=self.out_dtype, out=out,
                                out_batch_stride=out_batch_stride)

    def apply_sparse_sgd(self, grad, lr, inv_scale=None, skip_flag=None, grad_batch_stride=0):
        F.emb_sgd_dedup_(self.weight.data, self._rows, grad, self.workspace(), lr, scale=inv_scale,
                         skip_flag=skip_flag, grad_batch_stride=grad_batch_stride)


class DotInteraction(nn.Module):
    """[x0 | strict lower triangle of X X^T | zero pad to a multiple of 8]  (interactions.py:40-101)."""

    def __init__(self, embedding_num: int, embedding_dim: int):
        super().__init__()
        self._num_interaction_inputs = embedding_num + 1
        self._embedding_dim = embedding_dim
        self._raw_num_interactions = (self._num_interaction_inputs * (self._num_interaction_inputs - 1) // 2
                                      + embedding_dim)

    @property
    def num_interactions(self) -> int:
        n = self._raw_num_interactions
        return n + (((n - 1) // 8 + 1) * 8 - n)

    def interact(self, bottom_output, bottom_mlp_output=None):
        self._x = bottom_output
        return F.dot_interact_fwd(bottom_output)

    def backward(self, upstream, grad_out=None, found_inf=None):
        """-> grad [B, R, D] with the bottom-MLP slice already folded into row 0 (found_inf: see F.dot_interact_bwd)."""
        g, _ = F.dot_interact_bwd(self._x, upstream, fuse_mlp_grad=True, grad_out=grad_out, found_inf=found_inf)
        return g


class DlrmBottom(nn.Module):
    def __init__(self, num_numerical_features, categorical_feature_sizes, bottom_mlp_sizes=None, embedding_type="multi_table",
                 embedding_dim=128, hash_indices=False, use_cpp_mlp=False, fp16=False, device="cuda", compute_dtype=None):
        """The reference's argument list and order (nn/parts.py:25-35); embedding_type / use_cpp_mlp select among the reference's
        implementations of the same functions and are accepted for that reason only (one implementation here: the HIP kernels);
        the 16-bit compute type is compute_dtype, else fp16."""
        super().__init__()
        compute_dtype = compute_dtype or torch.float16
        assert bottom_mlp_sizes is None or embedding_dim == bottom_mlp_sizes[-1], \
            "The last bottom MLP layer must have same size as embedding."
        self._embedding_dim = embedding_dim
        self._categorical_feature_sizes = list(categorical_feature_sizes)
        self.compute_dtype = compute_dtype
        self.embeddings = (JointEmbedding(categorical_feature_sizes, embedding_dim, device, hash_indices,
                                          out_dtype=compute_dtype)
                           if len(categorical_feature_sizes) > 0 else None)
        self.mlp = Mlp(num_numerical_features, bottom_mlp_sizes, device, compute_dtype) if bottom_mlp_sizes else None
        if self.embeddings is not None:
            for size, w in zip(categorical_feature_sizes, self.embeddings.weights):
                nn.init.uniform_(w, -math.sqrt(1. / size), math.sqrt(1. / size))

    @property
    def num_categorical_features(self):
        return len(self._categorical_feature_sizes)

    @property
    def num_feature_vectors(self):
        return self.num_categorical_features + int(self.mlp is not None)

    def forward(self, numerical_input, categorical_inputs, batch=None):
        """-> [B, n_local, D] 16-bit: bottom-MLP output in slot 0 (if this rank owns it), then the tables."""
        n_vec = self.num_feature_vectors
        d = self._embedding_dim
        b = numerical_input.shape[0] if numerical_input is not None else categorical_inputs.shape[0]
        dev = self.embeddings.weight.device if self.embeddings is not None else numerical_input.device
        out = torch.empty((b, n_vec, d), dtype=self.compute_dtype, device=dev)
        slot = 1 if self.mlp is not None else 0
        # the gather (HBM-bound, no LDS) and the bottom MLP (MFMA tiles) write disjoint slots of `out` and want different
        # resources: they run side by side on two streams and join before anything reads `out`
        side = self._side_stream(dev) if (self.mlp is not None and self.embeddings is not None) else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
        if self.embeddings is not None:
            with (torch.cuda.stream(side) if side is not None else _nullcontext()):
                self.embeddings(categorical_inputs, out=out[:, slot:, :], out_batch_stride=n_vec * d)
        if self.mlp is not None:
            x16 = F.cast_rows(numerical_input, self.compute_dtype, cols_out=self.mlp.k_padded(0))
            self.mlp(x16, out=out[:, 0, :])
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        self._out = out
        return out

    def _side_stream(self, dev):
        """Second stream of this module (None on the CPU test doubles or with DLE_DLRM_TWO_STREAMS=0)."""
        if dev.type != "cuda" or os.environ.get("DLE_DLRM_TWO_STREAMS", "1") == "0":
            return None
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=dev)
        return self._side

    def backward(self, grad_out, emb_lr, inv_scale=None, skip_flag=None, mlp_grads=None, freeze_embeddings=False):
        """grad_out [B, n_local, D] 16-bit.  Embedding rows are updated in place (fused sparse SGD);
        bottom-MLP gradients are produced for the dense optimizer."""
        n_vec, d = self.num_feature_vectors, self._embedding_dim
        slot = 1 if self.mlp is not None else 0
        both = self.embeddings is not None and not freeze_embeddings and self.mlp is not None
        side = self._side_stream(grad_out.device) if both else None
        if side is not None:                       # the sparse update (HBM-bound row read-modify-writes) beside the MLP backward
            side.wait_stream(torch.cuda.current_stream())
        if self.embeddings is not None and not freeze_embeddings:
            with (torch.cuda.stream(side) if side is not None else _nullcontext()):
                self.embeddings.apply_sparse_sgd(grad_out[:, slot:, :], emb_lr, inv_scale, skip_flag,
                                                 grad_batch_stride=n_vec * d)
        if self.mlp is not None:
            self.mlp.backward(grad_out[:, 0, :], grads=mlp_grads)
        if side is not None:                       # joined here: the caller may now raise found_inf for the dense gradients
            torch.cuda.current_stream().wait_stream(side)


class DlrmTop(nn.Module):
    def __init__(self, top_mlp_sizes, interaction: DotInteraction, use_cpp_mlp=False, device="cuda", compute_dtype=torch.float16):
        """(nn/parts.py:95-100: top_mlp_sizes, interaction, use_cpp_mlp -- the last accepted and unused, see DlrmBottom.)"""
        super().__init__()
        self.interaction = interaction
        self.mlp = Mlp(interaction.num_interactions, top_mlp_sizes[:-1], device, compute_dtype)
        self.out = nn.Linear(top_mlp_sizes[-2], top_mlp_sizes[-1], device=device)
        self.compute_dtype = compute_dtype
        # weight facing the zero-padded interaction column stays zero for the whole training (parts.py:121-125)
        nn.init.zeros_(self.mlp.weights[0][:, -1].data)
        self._out_w16 = None

    def out_working_copy(self):
        if self._out_w16 is None:
            self._out_w16 = F.cast_rows(self.out.weight.data, self.compute_dtype)
        return self._out_w16

    def refresh_working_copies(self):
        self.mlp.refresh_working_copies()
        if self._out_w16 is not None:                       # in place: the optimizer table holds this buffer's address
            F.cast_rows(self.out.weight.data, self.compute_dtype, out=self._out_w16)
        else:
            self.out_working_copy()

    def forward(self, bottom_output, bottom_mlp_output=None):
        z = self.interaction.interact(bottom_output, bottom_mlp_output)
        h = self.mlp(z)
        self._h = h
        w = self.out_working_copy()
        return F.gemm(h, w, h.shape[0], w.shape[0], w.shape[1], True, True, out_dtype=self.compute_dtype,
                      bias=self.out.bias.data)

    def head_fusable(self):
        """The fused head (csrc/dlrm_head.hip) covers out_features == 1 over a 16-bit hidden layer of <= 512 columns."""
        return (self.out.out_features == 1 and self.out.in_features % 8 == 0 and self.out.in_features <= 512
                and self.compute_dtype in (torch.float16, torch.bfloat16))

    def forward_loss_backward_head(self, bottom_output, target, grad_scale=None, grads=None, out_grads=None,
                                   bottom_mlp_output=None):
        """forward() + BCEWithLogitsLoss(mean) + the backward of the loss and the `out` layer, the last three in ONE pass over the
        last hidden activation: -> loss fp32 [1].  The gradient w.r.t. that activation (ReLU mask applied) is kept for
        backward(None, ...); `out`'s weight / bias gradients and the bias gradient of the last MLP layer are written here."""
        z = self.interaction.interact(bottom_output, bottom_mlp_output)
        h = self.mlp(z)
        self._h = h
        w = self.out_working_copy()
        gw, gb = out_grads if out_grads is not None else (_grad_buf(self.out.weight), _grad_buf(self.out.bias))
        last = len(self.mlp.linears) - 1
        gprev = grads[last][1] if grads is not None else _grad_buf(self.mlp.linears[last].bias)
        ws = getattr(self, "_head_ws", None)
        if ws is None or ws.m != h.shape[0] or ws.k != h.shape[1] or ws.buf.device != h.device:
            ws = self._head_ws = F.HeadWorkspace(h.shape[0], h.shape[1], h.device)
        loss, self._gh, _ = F.head_bce_fwd_bwd(h, w.view(-1), self.out.bias.data, target, grad_scale, gw.view(-1), gb, ws,
                                               gprev_bias=gprev)
        return loss

    def backward(self, dlogits, grads=None, out_grads=None, grad_x_out=None, found_inf=None, defer_wgrad=False):
        """dlogits [B, 1] 16-bit -> gradient of the interaction input [B, R, D]; found_inf (optional fp32 [1]) is set by the
        interaction backward when that gradient holds an inf / nan.  defer_wgrad: -> (gradient, finish): the data-gradient chain
        first, finish() launches every weight / bias gradient of the top model.  dlogits = None: after
        forward_loss_backward_head (the head's share of the backward is done)."""
        h, w = self._h, self.out_working_copy()
        m, n, k = h.shape[0], w.shape[0], w.shape[1]
        fused = dlogits is None
        if fused:
            gh, self._gh = self._gh, None

            def out_wgrad():
                pass
        else:
            gw, gb = out_grads if out_grads is not None else (_grad_buf(self.out.weight), _grad_buf(self.out.bias))

            def out_wgrad():
                F.gemm(dlogits, h, n, k, m, False, False, out=gw, splitk=F.pick_splitk(n, k, m))
                F.colsum(dlogits, out=gb)
            if not defer_wgrad:
                out_wgrad()
            gh = F.gemm(dlogits, w, m, k, n, True, False, out_dtype=self.compute_dtype, act=C.ACT_RELU_BWD,
                        mask_src=h)
        if not defer_wgrad:
            gz = self.mlp.backward(gh, need_input_grad=True, grads=grads, masked=True, skip_last_bias=fused)
            return self.interaction.backward(gz, grad_out=grad_x_out, found_inf=found_inf)
        gz, mlp_finish = self.mlp.backward(gh, need_input_grad=True, grads=grads, masked=True, defer_wgrad=True,
                                           skip_last_bias=fused)
        gx = self.interaction.backward(gz, grad_out=grad_x_out, found_inf=found_inf)

        def finish():
            out_wgrad()
            mlp_finish()
        return gx, finish


class DistributedDlrm(nn.Module):
    """Same constructor surface as the reference's DistributedDlrm (model/distributed.py:106-159); the
    embedding/interaction/MLP implementations are fixed to the HIP kernels."""

    def __init__(self, num_numerical_features, categorical_feature_sizes, bottom_mlp_sizes, top_mlp_sizes,
                 vectors_per_gpu=None, embedding_device_mapping=None, world_num_categorical_features=None,
                 embedding_type="joint_fused", embedding_dim=128, interaction_op="cuda_dot", hash_indices=False,
                 use_cpp_mlp=True, fp16=True, bottom_features_ordered=False, device="cuda",
                 compute_dtype=None, world_size=1):
        super().__init__()
        if interaction_op not in ("cuda_dot", "dot"):
            raise ValueError("only the dot interaction is on the MI355X hot path (got %r)" % interaction_op)
        self.distributed = world_size > 1
        self._vectors_per_gpu = vectors_per_gpu
        self._embedding_dim = embedding_dim
        self._interaction_op = interaction_op
        self._hash_indices = hash_indices
        cd = compute_dtype or (torch.float16 if fp16 else torch.bfloat16)
        self.compute_dtype = cd
        if self.distributed:
            order = torch.tensor([-1] + [i for bucket in embedding_device_mapping for i in bucket],
                                 dtype=torch.long, device=device) + 1
            self._device_feature_order = order if bottom_features_ordered else None
            self._feature_order = order.argsort() if bottom_features_ordered else None
        else:
            if world_num_categorical_features is None:      # (the row-sharded placement keeps ONE joint matrix for all T tables)
                world_num_categorical_features = len(categorical_feature_sizes)
            self._device_feature_order = self._feature_order = None
        interaction = DotInteraction(world_num_categorical_features, embedding_dim)
        self.bottom_model = DlrmBottom(num_numerical_features, categorical_feature_sizes, bottom_mlp_sizes,
                                       embedding_type, embedding_dim, hash_indices=hash_indices, use_cpp_mlp=use_cpp_mlp,
                                       fp16=fp16, device=device, compute_dtype=cd)
        self.top_model = DlrmTop(top_mlp_sizes, interaction, use_cpp_mlp=use_cpp_mlp, device=device, compute_dtype=cd)

    def extra_repr(self):
        return f"interaction_op={self._interaction_op}, hash_indices={self._hash_indices}"

    @classmethod
    def from_dict(cls, obj_dict, **kwargs):
        """(model/distributed.py:155-158)"""
        return cls(**obj_dict, **kwargs)

    def forward(self, numerical_input, categorical_inputs, batch_sizes_per_gpu=None):
        """Inference-mode forward (model/distributed.py:160-180): numerical_input [B, num_numerical] (None on a rank without the
        bottom MLP), categorical_inputs int64 [B, tables of this rank] -> logits [B, 1] in the compute type.  Nothing is kept for a
        backward pass: the train step is DlrmTrainer.train_step (explicit forward / backward / update).  With several ranks the
        bottom -> top all-to-all lives in the trainer (engine.DlrmTrainer.evaluate runs this forward around it)."""
        if self.distributed:
            raise RuntimeError("DistributedDlrm.forward: with world_size > 1 the bottom -> top exchange belongs to DlrmTrainer "
                               "(train_step / evaluate)")
        return self.top_model(self.bottom_model(numerical_input, categorical_inputs))

    def refresh_working_copies(self):
        if self.bottom_model.mlp is not None:
            self.bottom_model.mlp.refresh_working_copies()
        self.top_model.refresh_working_copies()
