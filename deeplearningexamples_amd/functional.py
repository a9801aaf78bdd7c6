"""Thin tensor-level wrappers over the C ABI (allocation + argument marshalling only).

Every function here launches hand-written HIP kernels from libdle_mi355x.so on the current
torch stream; there is no eager fallback.  torch is used for memory, streams and autograd plumbing.
"""
import os

import torch

from . import _cabi as C


# ------------------------------------------------------------------ DLRM dot interaction
def dot_interact_out_width(rows, cols):
    return C.lib().dle_dot_interact_out_width(rows, cols)


def dot_interact_fwd(x, force_generic=False):
    """x [B,R,C] -> [B, ceil8(R(R-1)/2 + C)]  (dotBasedInteractFwd)."""
    C.require_cuda(x)
    if x.dim() != 3:
        raise ValueError("dot_interact: expected [batch, rows, cols], got %s" % (tuple(x.shape),))
    x = x.contiguous()
    b, r, c = x.shape
    out = torch.empty((b, dot_interact_out_width(r, c)), dtype=x.dtype, device=x.device)
    C.annotate(bytes=float(b) * (r * c + out.shape[1]) * x.element_size())
    C.call("dle_dot_interact_fwd", C.ptr(x), C.ptr(out), b, r, c, C.dt(x), int(force_generic), C.stream())
    return out


def dot_interact_bwd(x, upstream, force_generic=False, fuse_mlp_grad=False, grad_out=None, found_inf=None):
    """-> (grad [B,R,C], mlp_grad [B,C])  (dotBasedInteractBwd).  fuse_mlp_grad: mlp_grad is added onto
    grad[:,0,:] in the kernel and None is returned in its place.  found_inf (fp32 [1], optional): set to 1 by the kernel
    when the gradient it stores holds an inf / nan (GradScaler's check without a pass over the tensor)."""
    C.require_cuda(x, upstream)
    x = x.contiguous()
    b, r, c = x.shape
    ow = dot_interact_out_width(r, c)
    if tuple(upstream.shape) != (b, ow):
        raise ValueError("dot_interact_bwd: upstream grad must be [%d, %d], got %s" % (b, ow, tuple(upstream.shape)))
    upstream = upstream.to(x.dtype).contiguous()
    grad = torch.empty_like(x) if grad_out is None else grad_out
    mlp_grad = None if fuse_mlp_grad else torch.empty((b, c), dtype=x.dtype, device=x.device)
    C.annotate(bytes=float(b) * (2 * r * c + ow + (0 if fuse_mlp_grad else c)) * x.element_size())
    if found_inf is not None:
        C.require_cuda(found_inf)
        C.call("dle_dot_interact_bwd_checked", C.ptr(x), C.ptr(upstream), C.ptr(grad), C.ptr(mlp_grad), b, r, c,
               C.dt(x), int(force_generic), C.ptr(found_inf), C.stream())
    else:
        C.call("dle_dot_interact_bwd", C.ptr(x), C.ptr(upstream), C.ptr(grad), C.ptr(mlp_grad), b, r, c,
               C.dt(x), int(force_generic), C.stream())
    return grad, mlp_grad


# ------------------------------------------------------------------ DLRM embeddings
def _i64(t, name):
    if t is None:
        return None
    if t.dtype != torch.int64:
        raise ValueError("%s must be int64 (got %s)" % (name, t.dtype))
    return t.contiguous()


def emb_gather_fwd(weight, indices, offsets=None, hash_sizes=None, out_dtype=torch.float32, out=None,
                   out_batch_stride=0):
    """out[b,t,:] = W[(idx[b,t] mod size_t) + offsets[t], :].  `out` (+ out_batch_stride in elements) lets
    the rows land inside a wider [B, 1+T, D] buffer."""
    C.require_cuda(weight, indices, offsets, hash_sizes)
    if weight.dtype != torch.float32 or weight.dim() != 2:
        raise ValueError("embedding table must be a 2-D fp32 tensor")
    if indices.dim() != 2:
        raise ValueError("indices must be [batch, tables]")
    weight = weight.contiguous() if not weight.is_contiguous() else weight
    indices, offsets, hash_sizes = _i64(indices, "indices"), _i64(offsets, "offsets"), _i64(hash_sizes, "hash_sizes")
    b, t = indices.shape
    d = weight.shape[1]
    if offsets is not None and offsets.numel() < t:
        raise ValueError("offsets has %d entries for %d tables" % (offsets.numel(), t))
    if out is None:
        out = torch.empty((b, t, d), dtype=out_dtype, device=weight.device)
    C.annotate(bytes=float(b) * t * (d * 4 + d * out.element_size() + 8))
    C.call("dle_emb_gather_fwd", C.ptr(weight), C.ptr(indices), C.ptr(offsets), C.ptr(hash_sizes), C.ptr(out),
           b, t, d, C.dt(out.dtype), out_batch_stride, C.stream())
    return out


def emb_offset_indices(indices, offsets=None, hash_sizes=None):
    C.require_cuda(indices, offsets, hash_sizes)
    indices, offsets, hash_sizes = _i64(indices, "indices"), _i64(offsets, "offsets"), _i64(hash_sizes, "hash_sizes")
    b, t = indices.shape
    rows = torch.empty_like(indices)
    C.call("dle_emb_offset_indices", C.ptr(indices), C.ptr(offsets), C.ptr(hash_sizes), C.ptr(rows), b, t, C.stream())
    return rows


def emb_grad_values(grad, scale=None):
    """fp32 COO values of the sparse embedding gradient (optionally * device scalar)."""
    C.require_cuda(grad, scale)
    grad = grad.contiguous()
    values = torch.empty(grad.shape, dtype=torch.float32, device=grad.device)
    C.call("dle_emb_grad_values", C.ptr(grad), C.ptr(values), C.ptr(scale), grad.numel(), C.dt(grad), C.stream())
    return values


def emb_sparse_sgd_(weight, rows, grad, lr, scale=None, skip_flag=None):
    """In place: W[rows[i]] -= lr * scale * grad[i] (duplicates accumulate). lr: float or device tensor."""
    C.require_cuda(weight, rows, grad, scale, skip_flag)
    if weight.dtype != torch.float32 or not weight.is_contiguous():
        raise ValueError("embedding table must be contiguous fp32")
    rows = _i64(rows.reshape(-1), "rows")
    d = weight.shape[1]
    grad = grad.reshape(-1, d).contiguous()
    if grad.shape[0] != rows.numel():
        raise ValueError("sparse sgd: %d gradient rows for %d indices" % (grad.shape[0], rows.numel()))
    lr_dev = lr if isinstance(lr, torch.Tensor) else None
    lr_host = 0.0 if lr_dev is not None else float(lr)
    C.call("dle_emb_sparse_sgd", C.ptr(weight), C.ptr(rows), C.ptr(grad), C.ptr(lr_dev), lr_host, C.ptr(scale),
           C.ptr(skip_flag), rows.numel(), 1, d, 0, C.dt(grad), C.stream())
    return weight


class EmbUpdateWorkspace:
    """Persistent state of the duplicate-free sparse SGD: head[total_rows] (all -1 between calls),
    next[batch*tables] scratch, the small-table mask and the host copy of the table offsets."""

    def __init__(self, table_offsets, dim, device):
        import ctypes
        import numpy as np
        off = np.ascontiguousarray(np.asarray(table_offsets, dtype=np.int64))
        self.offsets_host = off
        self.tables = off.size - 1
        self.dim = dim
        mask = np.zeros(self.tables, dtype=np.uint8)
        C.call("dle_emb_small_table_mask", off.ctypes.data_as(ctypes.c_void_p), self.tables, dim,
               mask.ctypes.data_as(ctypes.c_void_p))
        self.mask_host = mask
        self.is_small = torch.from_numpy(mask).to(device)
        self.head = torch.full((int(off[-1]),), -1, dtype=torch.int32, device=device)
        self.next = None
        # scratch of the update (one-hot partial blocks of the tiny tables, sub-lists of the mid tables), sized per batch
        sizes = off[1:] - off[:-1]
        self.n_onehot = int((sizes <= 128).sum()) if dim == 128 else 0
        self.scratch = None

    def scratch_for(self, batch, device):
        import ctypes
        need = int(C.lib().dle_emb_sgd_workspace_bytes(self.offsets_host.ctypes.data_as(ctypes.c_void_p), self.tables, self.dim,
                                                       batch))
        if need == 0:
            return None
        if self.scratch is None or self.scratch.numel() * 4 < need:
            self.scratch = torch.empty((need + 3) // 4, dtype=torch.float32, device=device)
        return self.scratch

    def next_for(self, n, device):
        if self.next is None or self.next.numel() < n:
            self.next = torch.empty(n, dtype=torch.int32, device=device)
        return self.next


def emb_sgd_dedup_(weight, rows, grad, ws, lr, scale=None, skip_flag=None, grad_batch_stride=0):
    """In place W[rows[b,t]] -= lr*scale*grad[b,t] with duplicate rows summed first (no float atomics).
    rows int64 [B,T]; grad: tensor whose element (b,t,0) sits at data_ptr + b*grad_batch_stride + t*dim."""
    import ctypes
    C.require_cuda(weight, rows, grad, scale, skip_flag)
    if weight.dtype != torch.float32 or not weight.is_contiguous():
        raise ValueError("embedding table must be contiguous fp32")
    rows = _i64(rows, "rows")
    b, t = rows.shape
    if t != ws.tables or weight.shape[1] != ws.dim:
        raise ValueError("workspace was built for %d tables x dim %d" % (ws.tables, ws.dim))
    lr_dev = lr if isinstance(lr, torch.Tensor) else None
    lr_host = 0.0 if lr_dev is not None else float(lr)
    nxt = ws.next_for(b * t, weight.device)
    # algorithmic: grad row read + table row read-modify-write + row id
    C.annotate(bytes=float(b) * t * (ws.dim * grad.element_size() + 2 * ws.dim * 4 + 8))
    oh = ws.scratch_for(b, weight.device) if grad.dtype in (torch.float16, torch.bfloat16) else None
    C.call("dle_emb_sgd_dedup_ws", C.ptr(weight), C.ptr(rows), C.ptr(grad), C.ptr(ws.head), C.ptr(nxt),
           C.ptr(ws.is_small), ws.offsets_host.ctypes.data_as(ctypes.c_void_p), C.ptr(lr_dev), lr_host,
           C.ptr(scale), C.ptr(skip_flag), b, t, ws.dim, grad_batch_stride, C.dt(grad), C.ptr(oh),
           oh.numel() * 4 if oh is not None else 0, C.stream())
    return weight


# ------------------------------------------------------------------ small train-step kernels
def cast_rows(x, out_dtype, cols_out=None, out=None):
    """2-D cast with optional zero padding of the trailing columns (K padded to a multiple of 8/16)."""
    C.require_cuda(x, out)
    if x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("cast_rows expects a 2-D tensor with unit inner stride")
    r, c = x.shape
    co = c if cols_out is None else cols_out
    if out is None:
        out = torch.empty((r, co), dtype=out_dtype, device=x.device)
    C.call("dle_cast_rows", C.ptr(x), C.ptr(out), r, c, co, x.stride(0), out.stride(0), C.dt(x), C.dt(out),
           C.stream())
    return out


def a2a_blocks(blocks, x, rows, widths, pack):
    """blocks: flat concatenation over ranks of [rows, widths[s]] matrices; x: [rows, sum(widths)] contiguous.  pack=False:
    x <- blocks (after the forward all-to-all); pack=True: blocks <- x (before the backward one).  One launch."""
    import ctypes
    C.require_cuda(blocks, x)
    if not (blocks.is_contiguous() and x.is_contiguous()) or blocks.dtype != x.dtype or blocks.numel() != x.numel():
        raise ValueError("a2a_blocks: contiguous buffers of one dtype and size")
    if x.numel() != rows * sum(widths):
        raise ValueError("a2a_blocks: x must hold rows x sum(widths) elements")
    w = (ctypes.c_int * len(widths))(*[int(v) for v in widths])
    C.call("dle_a2a_blocks", C.ptr(blocks), C.ptr(x), int(rows), len(widths), ctypes.cast(w, ctypes.c_void_p), x.element_size(),
           int(bool(pack)), C.stream())
    return blocks if pack else x


def transpose_cast(x, out_dtype, out=None):
    """out[c, r] = (out_dtype) x[r, c] for a 2-D fp32 / 16-bit matrix with unit inner stride: the transposed 16-bit working copy."""
    C.require_cuda(x, out)
    if x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("transpose_cast: 2-D input with unit inner stride")
    r, c = x.shape
    if out is None:
        out = torch.empty((c, r), dtype=out_dtype, device=x.device)
    if out.shape != (c, r) or out.stride(1) != 1 or out.dtype != out_dtype:
        raise ValueError("transpose_cast: output must be [cols, rows] of the requested dtype")
    C.call("dle_transpose_cast", C.ptr(x), C.ptr(out), r, c, x.stride(0), out.stride(0), C.dt(x), C.dt(out), C.stream())
    return out


def cast(x, out_dtype, out=None):
    """Flat dtype cast of a contiguous tensor."""
    C.require_cuda(x, out)
    x = x.contiguous()
    if out is None:
        out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    n = x.numel()
    C.call("dle_cast_rows", C.ptr(x), C.ptr(out), 1, n, n, n, n, C.dt(x), C.dt(out), C.stream())
    return out


def bce_with_logits(logits, target, grad_scale=None, want_grad=True, ld_logits=1):
    """-> (loss fp32 [1], dlogits or None).  dlogits = d(mean BCE)/dlogits * grad_scale."""
    C.require_cuda(logits, target, grad_scale)
    if target.dtype != torch.float32:
        raise ValueError("BCE targets must be fp32")
    n = target.numel()
    loss = torch.empty(1, dtype=torch.float32, device=target.device)
    dl = torch.empty(n, dtype=logits.dtype, device=logits.device) if want_grad else None
    C.call("dle_bce_logits", C.ptr(logits), C.ptr(target), C.ptr(loss), C.ptr(dl), C.ptr(grad_scale), n,
           ld_logits, C.dt(logits), C.stream())
    return loss, dl


class HeadWorkspace:
    """Scratch of head_bce_fwd_bwd for one (batch, K): one partial row per workgroup."""

    def __init__(self, m, k, device):
        self.m, self.k = int(m), int(k)
        nbytes = int(C.lib().dle_head_bce_workspace_bytes(self.m, self.k))
        self.buf = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)


def head_bce_fwd_bwd(h, w16, bias, target, grad_scale, gw, gb, ws, gprev_bias=None, dh=None, want_logits=False):
    """Last linear layer (out_features = 1) + BCEWithLogitsLoss(mean) + the backward of both in one pass over h [M, K] (16-bit):
    -> (loss fp32 [1], dh [M, K] masked by the ReLU of h, logits [M] or None); gw [K], gb [1] and (optional) gprev_bias [K] =
    column sums of dh are written in place (fp32)."""
    C.require_cuda(h, w16, bias, target, grad_scale, gw, gb, gprev_bias, dh)
    if h.dim() != 2 or h.stride(1) != 1 or target.dtype != torch.float32 or h.dtype != w16.dtype:
        raise ValueError("head_bce_fwd_bwd: h [M, K] 16-bit with unit inner stride, w16 of the same dtype, fp32 targets")
    m, k = h.shape
    if target.numel() != m or w16.numel() != k or gw.numel() != k or ws.m != m or ws.k != k:
        raise ValueError("head_bce_fwd_bwd: shape mismatch")
    for t in (gw, gb, gprev_bias):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
            raise ValueError("head_bce_fwd_bwd: gradients are contiguous fp32 buffers")
    if dh is None:
        dh = torch.empty((m, k), dtype=h.dtype, device=h.device)
    loss = torch.empty(1, dtype=torch.float32, device=h.device)
    logits = torch.empty(m, dtype=h.dtype, device=h.device) if want_logits else None
    C.annotate(bytes=2.0 * m * k * h.element_size() + 6.0 * m, tag="%dx%d" % (m, k))
    C.call("dle_head_bce_fwd_bwd", C.ptr(h), C.ptr(w16), C.ptr(bias), C.ptr(target), C.ptr(grad_scale), C.ptr(loss),
           C.ptr(logits), C.ptr(dh), C.ptr(gw), C.ptr(gb), C.ptr(gprev_bias), C.ptr(ws.buf), ws.buf.numel() * 4, m, k,
           h.stride(0), dh.stride(0), C.dt(h), C.stream())
    return loss, dh, logits


def amp_update_scale_(scale, growth_tracker, found_inf, inv_scale=None, growth_factor=2.0, backoff_factor=0.5,
                      growth_interval=2000, clear_found_inf=True):
    C.require_cuda(scale, growth_tracker, found_inf, inv_scale)
    C.call("dle_amp_update_scale", C.ptr(scale), C.ptr(growth_tracker), C.ptr(found_inf), C.ptr(inv_scale),
           float(growth_factor), float(backoff_factor), int(growth_interval), int(clear_found_inf), C.stream())


def axpby_(x, y, out, a=1.0, b=1.0):
    """out = a * x + b * y on flat fp32 tensors (out may be x or y; y may be None when b == 0)."""
    C.require_cuda(x, y, out)
    if x.dtype != torch.float32 or out.dtype != torch.float32 or not (x.is_contiguous() and out.is_contiguous()):
        raise ValueError("axpby_ expects contiguous fp32 tensors")
    if out.numel() != x.numel() or (y is not None and y.numel() != x.numel()):
        raise ValueError("axpby_: size mismatch")
    C.call("dle_axpby_f32", C.ptr(x), C.ptr(y), C.ptr(out), float(a), float(b if y is not None else 0.0), x.numel(), C.stream())
    return out


def check_nonfinite_(x, found_inf):
    C.require_cuda(x, found_inf)
    C.call("dle_check_nonfinite", C.ptr(x), C.ptr(found_inf), x.numel(), C.dt(x), C.stream())


# ------------------------------------------------------------------ GEMM
def gemm(a, b, m, n, k, a_kc, b_kc, out=None, out_dtype=None, bias=None, act=C.ACT_NONE, aux=None,
         mask_src=None, splitk=1, accumulate=False, alpha=1.0, lda=None, ldb=None):
    """C[m,n] = act(alpha * A(m,k) B(n,k) + bias).  a/b are 2-D 16-bit tensors, row-major with
    arbitrary leading dimension; a_kc/b_kc say whether the contraction dim is the contiguous one."""
    C.require_cuda(a, b, out, bias, aux, mask_src)
    if a.dtype != b.dtype or a.dtype not in (torch.float16, torch.bfloat16):
        raise ValueError("gemm inputs must both be f16 or both bf16 (got %s, %s)" % (a.dtype, b.dtype))
    def _ld(t):
        if t.shape[-1] != 1 and t.stride(-1) != 1:
            raise ValueError("gemm operands must have unit inner stride")
        return t.stride(0) if (t.shape[0] > 1 and t.stride(0) >= t.shape[-1]) else t.shape[-1]
    lda = _ld(a) if lda is None else lda
    ldb = _ld(b) if ldb is None else ldb
    if out is None:
        out = torch.empty((m, n), dtype=out_dtype or a.dtype, device=a.device)
    if out.stride(-1) != 1:
        raise ValueError("gemm output must have unit inner stride")
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != n):
        raise ValueError("gemm bias must be fp32 [n]")
    extra = (float(m * n) * mask_src.element_size() if mask_src is not None else 0.0) + \
            (float(aux.numel()) * aux.element_size() if aux is not None else 0.0)
    C.annotate(flops=2.0 * m * n * k,
               bytes=float(m * k + n * k) * a.element_size() + float(m * n) * out.element_size() + extra,
               tag="%dx%dx%d%s%s" % (m, n, k, "+src" if mask_src is not None else "", "+aux" if aux is not None else ""),
               replay=not accumulate)
    ws = splitk_workspace(a.device, splitk * m * n * 4) if splitk > 1 else None
    C.call("dle_gemm", C.ptr(a), C.ptr(b), C.ptr(out), C.ptr(aux), C.ptr(bias), C.ptr(mask_src), m, n, k,
           lda, ldb, out.stride(0) if out.dim() == 2 else n, int(a_kc), int(b_kc), C.dt(a), C.dt(out), act,
           splitk, int(accumulate), float(alpha), C.ptr(ws), ws.numel() * 4 if ws is not None else 0, C.stream())
    return out


_splitk_ws = {}


def splitk_workspace(device, nbytes):
    """fp32 scratch for split-K partial slabs, reduction partials and the like: one buffer per (device, CURRENT STREAM), grown on
    demand -- launches of one stream use it one after the other; two streams running kernels side by side (the weight-gradient
    stream of the ResNet engine) must not share it."""
    key = (device, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    w = _splitk_ws.get(key)
    if w is None or w.numel() * 4 < nbytes:
        w = torch.empty(max(nbytes // 4, 1 << 22), dtype=torch.float32, device=device)
        _splitk_ws[key] = w
    return w


def gemm_colsum(g, w, m, n, k, src, colsum_out, act=C.ACT_RELU_BWD, accumulate=False):
    """dX [m, n] = f(g [m, k] w [k, n], src [m, n]) and colsum_out [n] (fp32) (+)= column sums of the rounded dX: the data gradient
    of a linear layer through the activation derivative of the layer below + that layer's bias gradient in one launch (+ a small
    fold).  act = ACT_RELU_BWD (dX = product where src > 0) or ACT_MUL (dX = product * src).  -> dX, or None outside the kernel's
    envelope (the caller runs gemm + colsum)."""
    C.require_cuda(g, w, src, colsum_out)
    if (g.dtype not in (torch.float16, torch.bfloat16) or w.dtype != g.dtype or src.dtype != g.dtype
            or g.stride(1) != 1 or w.stride(1) != 1 or src.stride(1) != 1 or act not in (C.ACT_RELU_BWD, C.ACT_MUL)
            or colsum_out.dtype != torch.float32 or not colsum_out.is_contiguous() or colsum_out.numel() != n):
        return None
    out = torch.empty((m, n), dtype=g.dtype, device=g.device)
    if src.stride(0) != out.stride(0):
        return None
    ws = splitk_workspace(g.device, ((m + 127) // 128) * n * 4)
    by = float(m) * (k + 2 * n) * g.element_size() + float(k) * n * w.element_size()
    C.annotate(bytes=by, flops=2.0 * m * n * k, tag="%dx%dx%d+src+colsum" % (m, n, k))
    # (recorded in the family of the GEMM it replaces)
    rc = _timed_optional("dle_gemm", C.lib().dle_gemm_colsum,
                         (C.ptr(g), C.ptr(w), C.ptr(out), C.ptr(src), C.ptr(colsum_out), m, n, k, g.stride(0), w.stride(0),
                          out.stride(0), C.dt(g), int(act), int(accumulate), C.ptr(ws), ws.numel() * 4, C.stream()),
                         replayable=not accumulate)
    if rc > 1:
        C.check(rc - 1000 if rc > 1000 else -1, "dle_gemm_colsum")
    return out if rc == 1 else None


def gemm_relu_bits(x, w, m, n, k, bias):
    """y [m, n] = relu(x [m, k] w [n, k]^T + bias) together with the KEEP BITS of y (uint8 [m n / 8]: bit (i n + j) & 7 of byte
    (i n + j) >> 3 = y[i, j] > 0) -- one (Linear + ReLU) layer whose backward reads 1 bit per element instead of y
    (csrc/gemm8_kernel.h ACT_RELU_BITS; dlrm/nn/mlps.py:38-43).  -> (y, bits), or None outside the ping-pong kernel's envelope
    (the caller runs gemm(act=ACT_RELU) and keeps y as the mask source).  y is bit-identical to gemm(act=ACT_RELU)."""
    C.require_cuda(x, w, bias)
    if (x.dtype not in (torch.float16, torch.bfloat16) or w.dtype != x.dtype or x.stride(1) != 1 or w.stride(1) != 1
            or n % 16 != 0 or bias is None or bias.dtype != torch.float32):
        return None
    y = torch.empty((m, n), dtype=x.dtype, device=x.device)
    bits = torch.empty(m * n // 8, dtype=torch.uint8, device=x.device)
    C.annotate(bytes=float(m) * (k + n) * x.element_size() + float(k) * n * w.element_size() + bits.numel(), flops=2.0 * m * n * k,
               tag="%dx%dx%d+bits" % (m, n, k))
    rc = _timed_optional("dle_gemm", C.lib().dle_gemm8_relu_bits_try,
                         (C.ptr(x), C.ptr(w), C.ptr(y), C.ptr(bits), C.ptr(bias), m, n, k, x.stride(0), w.stride(0), C.dt(x), C.stream()))
    if rc > 1:
        C.check(rc - 1000 if rc > 1000 else -1, "dle_gemm8_relu_bits_try")
    return (y, bits) if rc == 1 else None


def gemm_colsum_bits(g, w, m, n, k, bits, colsum_out, accumulate=False):
    """gemm_colsum(act=ACT_RELU_BWD) with the mask given as the keep bits gemm_relu_bits left (1 bit per element) instead of the
    16-bit activation: dX [m, n] = (g [m, k] w [k, n]) where the bit is set, colsum_out [n] (+)= column sums of the rounded dX.
    -> dX, or None outside the envelope (the caller uses gemm_colsum with the activation)."""
    C.require_cuda(g, w, bits, colsum_out)
    if (g.dtype not in (torch.float16, torch.bfloat16) or w.dtype != g.dtype or g.stride(1) != 1 or w.stride(1) != 1
            or bits.dtype != torch.uint8 or bits.numel() != m * n // 8 or n % 16 != 0 or m % 256 != 0
            or colsum_out.dtype != torch.float32 or not colsum_out.is_contiguous() or colsum_out.numel() != n):
        return None
    out = torch.empty((m, n), dtype=g.dtype, device=g.device)
    ws = splitk_workspace(g.device, ((m + 127) // 128) * n * 4)
    by = float(m) * (k + n) * g.element_size() + float(k) * n * w.element_size() + bits.numel()
    C.annotate(bytes=by, flops=2.0 * m * n * k, tag="%dx%dx%d+bits+colsum" % (m, n, k))
    rc = _timed_optional("dle_gemm", C.lib().dle_gemm_colsum_bits,
                         (C.ptr(g), C.ptr(w), C.ptr(out), C.ptr(bits), C.ptr(colsum_out), m, n, k, g.stride(0), w.stride(0), C.dt(g),
                          int(accumulate), C.ptr(ws), ws.numel() * 4, C.stream()), replayable=not accumulate)
    if rc > 1:
        C.check(rc - 1000 if rc > 1000 else -1, "dle_gemm_colsum_bits")
    return out if rc == 1 else None


def colsum(x, out=None, accumulate=False):
    """fp32 column sums of a 2-D tensor (bias gradient)."""
    C.require_cuda(x, out)
    if x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("colsum expects a 2-D tensor with unit inner stride")
    m, n = x.shape
    if out is None:
        out = torch.empty((n,), dtype=torch.float32, device=x.device)
        accumulate = False
    ws = splitk_workspace(x.device, 2048 * n * 4)
    C.annotate(bytes=float(m) * n * x.element_size(), tag="%dx%d" % (m, n))
    C.call("dle_colsum", C.ptr(x), C.ptr(out), m, n, x.stride(0) if m > 1 else n, C.dt(x), int(accumulate),
           C.ptr(ws), ws.numel() * 4, C.stream())
    return out


class ColsumTable:
    """Device table of (matrix, fp32 destination) pairs for colsum_batched; the tensors are kept alive here (their ADDRESSES are
    in the table: they must be persistent buffers)."""

    def __init__(self, entries):
        self.entries = list(entries)
        src0 = self.entries[0][0]
        for x, out in self.entries:
            C.require_cuda(x, out)
            if x.shape != src0.shape or x.dtype != src0.dtype or not x.is_contiguous() or x.dim() != 2:
                raise ValueError("ColsumTable: contiguous 2-D matrices of one shape and dtype")
            if out.dtype != torch.float32 or out.numel() != x.shape[1] or not out.is_contiguous():
                raise ValueError("ColsumTable: fp32 destinations of N elements")
        self.n = len(self.entries)
        self.dev = torch.tensor([[x.data_ptr(), o.data_ptr()] for x, o in self.entries], dtype=torch.int64).to(src0.device)


def colsum_batched(table, m, n, ld, dtype):
    """fp32 column sums of every matrix of a ColsumTable ([m, n], row pitch ld) into its destination: one launch pair."""
    need = int(C.lib().dle_colsum_batched_workspace_bytes(table.n, m, n))
    ws = splitk_workspace(table.dev.device, need)
    C.annotate(bytes=float(table.n) * m * n * 2, tag="b%dx%dx%d" % (table.n, m, n))
    C.call("dle_colsum_batched", C.ptr(table.dev), table.n, m, n, ld, C.dt(dtype), C.ptr(ws), ws.numel() * 4, C.stream())


def pick_splitk(m_out, n_out, k, target_blocks=512):
    """Split the contraction so that a skinny wgrad still fills 256 CUs (K tiles of 64).  Outputs of at least 256x256
    run on the 256x256 tile (gemm_dma.hip launch_gemm): ONE slice per CU (256 // tiles, each >= 4 K tiles deep) -- the big
    tile issues half the LDS-DMA pieces per MFMA of the 128x128 one, which is what bounds the split weight gradients."""
    ktiles = (k + 63) // 64
    if m_out >= 256 and n_out >= 256 and "DLE_SPLITK_TARGET" not in os.environ:
        tiles_big = ((m_out + 255) // 256) * ((n_out + 255) // 256)
        if tiles_big >= 160:
            return 1
        s = min(ktiles // 4, max(256 // tiles_big, 1))
        if tiles_big <= 2:
            s = min(s, 64)      # (256x512x65536: 46 us at 64 slices, 56 at 128 -- each slice writes a whole fp32 slab of the output)
        if s >= 2 and tiles_big * s >= 128:
            return s
    target_blocks = int(os.environ.get("DLE_SPLITK_TARGET", target_blocks))      # tuning knob (tools/, not the product default)
    tiles = ((m_out + 127) // 128) * ((n_out + 127) // 128)
    s = max(1, min(ktiles, target_blocks // max(tiles, 1)))
    return s


def linear_fwd(x, w, bias=None, act=C.ACT_NONE, want_pre=False):
    """y = act(x @ w.T + bias); x [M,K], w [N,K] (both 16-bit).  Returns (y, pre or None)."""
    m, k = x.shape
    n = w.shape[0]
    pre = torch.empty((m, n), dtype=x.dtype, device=x.device) if want_pre else None
    y = gemm(x, w, m, n, k, True, True, bias=bias, act=act, aux=pre)
    return y, pre


def linear_dgrad(gy, w, mask_src=None):
    """dx[M,K] = gy[M,N] @ w[N,K]  (optionally * (mask_src > 0))."""
    m, n = gy.shape
    k = w.shape[1]
    return gemm(gy, w, m, k, n, True, False, act=C.ACT_RELU_BWD if mask_src is not None else C.ACT_NONE,
                mask_src=mask_src)


def linear_wgrad(gy, x, out=None, accumulate=False):
    """dw[N,K] (fp32) = gy[M,N].T @ x[M,K]."""
    m, n = gy.shape
    k = x.shape[1]
    sk = pick_splitk(n, k, m)
    if out is None:
        out = torch.empty((n, k), dtype=torch.float32, device=x.device)
        accumulate = False
    return gemm(gy, x, n, k, m, False, False, out=out, splitk=sk, accumulate=accumulate)


def relu_bwd(g, y, out=None):
    """g * (y > 0) for 16-bit 2-D (possibly row-strided) views."""
    C.require_cuda(g, y, out)
    if g.dim() != 2 or g.shape != y.shape or g.stride(1) != 1 or y.stride(1) != 1:
        raise ValueError("relu_bwd expects matching 2-D views with unit inner stride")
    r, c = g.shape
    if out is None:
        out = torch.empty((r, c), dtype=g.dtype, device=g.device)
    C.call("dle_relu_bwd", C.ptr(g), C.ptr(y), C.ptr(out), r, c, g.stride(0), y.stride(0), out.stride(0), C.dt(g),
           C.stream())
    return out


def copy_rows(src, dst):
    """dst[r, :] = src[r, :] for 2-D views with unit inner stride (row strides may differ)."""
    C.require_cuda(src, dst)
    if src.shape != dst.shape or src.dtype != dst.dtype:
        raise ValueError("copy_rows: shape/dtype mismatch")
    r, c = src.shape
    C.call("dle_cast_rows", C.ptr(src), C.ptr(dst), r, c, c, src.stride(0), dst.stride(0), C.dt(src), C.dt(dst),
           C.stream())
    return dst


# ------------------------------------------------------------------ convolutions (NHWC / KRSC, implicit GEMM)
def _conv_out(h, w, r, s, stride, pad):
    return (h + 2 * pad - r) // stride + 1, (w + 2 * pad - s) // stride + 1


def conv2d_fwd(x, w, stride=1, pad=0, bias=None, act=C.ACT_NONE, out=None, out_dtype=None):
    """x [N,H,W,C], w [Ko,R,S,C] (both 16-bit, contiguous) -> y [N,P,Q,Ko]."""
    C.require_cuda(x, w, bias, out)
    n, h, wd, c = x.shape
    ko, r, s, c2 = w.shape
    if c2 != c or not x.is_contiguous() or not w.is_contiguous() or x.dtype != w.dtype:
        raise ValueError("conv2d_fwd: x must be NHWC-contiguous and w KRSC-contiguous with matching C/dtype")
    p, q = _conv_out(h, wd, r, s, stride, pad)
    if out is None:
        out = torch.empty((n, p, q, ko), dtype=out_dtype or x.dtype, device=x.device)
    C.annotate(flops=2.0 * n * p * q * ko * r * s * c, tag="fwd %dx%dx%dx%d k%d %dx%d s%d" % (n, h, wd, c, ko, r, s, stride),
               bytes=float(x.numel() + w.numel() + out.numel()) * 2)
    C.call("dle_conv2d_fwd", C.ptr(x), C.ptr(w), C.ptr(out), C.ptr(bias), n, h, wd, c, ko, r, s, stride, pad,
           C.dt(x), C.dt(out), act, C.stream())
    return out


def conv2d_dgrad(dy, w, in_hw, stride=1, pad=0, addend=None, out=None):
    """dy [N,P,Q,Ko], w [Ko,R,S,C] -> dx [N,H,W,C] (+ addend)."""
    C.require_cuda(dy, w, addend, out)
    n, p, q, ko = dy.shape
    ko2, r, s, c = w.shape
    h, wd = in_hw
    if ko2 != ko or not dy.is_contiguous() or not w.is_contiguous() or (p, q) != _conv_out(h, wd, r, s, stride, pad):
        raise ValueError("conv2d_dgrad: shape mismatch")
    if out is None:
        out = torch.empty((n, h, wd, c), dtype=dy.dtype, device=dy.device)
    if addend is not None and (addend.shape != out.shape or not addend.is_contiguous() or addend.dtype != dy.dtype):
        raise ValueError("conv2d_dgrad: addend must match dx")
    if r == 1 and s == 1 and pad == 0 and stride > 1 and addend is None and c % 8 == 0:
        # 1x1 stride-s (downsample branch): plain GEMM on the P x Q grid, then zero-stuffing -- the gather form would
        # run 1 - 1/s^2 of its tiles on rows that are identically zero
        m = n * p * q
        compact = gemm(dy.view(m, ko), w.view(ko, c), m, c, ko, True, False)
        C.annotate(bytes=float(out.numel() + compact.numel()) * 2, tag="N%dx%dx%dxC%d s%d" % (n, h, wd, c, stride))
        C.call("dle_upsample_zero", C.ptr(compact), C.ptr(out), n, p, q, h, wd, c, stride, C.dt(dy), C.stream())
        return out
    C.annotate(flops=2.0 * n * p * q * ko * r * s * c, tag="dgrad %dx%dx%dx%d k%d %dx%d s%d" % (n, h, wd, c, ko, r, s, stride),
               bytes=float(dy.numel() + w.numel() + out.numel()) * 2)
    if (r == 3 and s == 3 and stride == 2 and pad == 1 and addend is None and h % 2 == 0 and wd % 2 == 0
            and ko % 64 == 0 and c % 8 == 0):
        # four parity-class correlations with 1 + 2 + 2 + 4 taps instead of 9 taps over 3/4 zeros
        ws = splitk_workspace(dy.device, 9 * ko * c * 2)
        C.call("dle_conv2d_dgrad_s2", C.ptr(dy), C.ptr(w), C.ptr(out), n, h, wd, c, ko, C.ptr(ws), ws.numel() * 4,
               C.dt(dy), C.stream())
        return out
    C.call("dle_conv2d_dgrad", C.ptr(dy), C.ptr(w), C.ptr(out), C.ptr(addend), n, h, wd, c, ko, r, s, stride, pad,
           C.dt(dy), C.stream())
    return out


def upsample_zero(compact, in_hw, stride):
    """compact [N,P,Q,C] -> [N,H,W,C] with compact at the pixels (stride * i, stride * j) and zeros elsewhere."""
    C.require_cuda(compact)
    n, p, q, c = compact.shape
    h, wd = in_hw
    out = torch.empty((n, h, wd, c), dtype=compact.dtype, device=compact.device)
    C.annotate(bytes=float(out.numel() + compact.numel()) * 2, tag="N%dx%dx%dxC%d s%d" % (n, h, wd, c, stride))
    C.call("dle_upsample_zero", C.ptr(compact), C.ptr(out), n, p, q, h, wd, c, stride, C.dt(compact), C.stream())
    return out


def _timed_optional(name, fn, args, replayable=True):
    """A C-ABI call that may DECLINE (return 0) cannot go through C.call (which raises on non-zero): run it, and when bench.py's
    kernel timer is installed record it like C.call does (event pair on the launch stream, replayable) if it launched.  -> rc
    replayable=False: the launch ADDS into a live buffer (an accumulating bias-gradient / weight-gradient epilogue): it is
    timed, but KernelTimer.replay must never re-launch it."""
    tm = C._timer
    if tm is None:
        return fn(*args)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    rc = fn(*args)
    e.record()
    meta, tm.meta = tm.meta, None
    if rc == 1:
        tm.records.append((name, s, e, meta))
        if replayable:
            tm.last[(name, meta.get("tag") if meta else None)] = (fn, args, torch.cuda.current_stream())
    return rc


def conv1x1_s2_dgrad_compact(dy, w):
    """The data gradient of a 1x1 stride-2 convolution ON ITS OWN GRID: dy [N,P,Q,Ko], w [Ko,1,1,C] -> [N,P,Q,C] (the non-zero
    pixels of dx; conv2d_dgrad zero-stuffs it to [N,2P,2Q,C], gemm_add_upsampled2 adds it without materialising that)."""
    n, p, q, ko = dy.shape
    c = w.shape[-1]
    m = n * p * q
    return gemm(dy.view(m, ko), w.view(ko, c), m, c, ko, True, False).view(n, p, q, c)


def gemm_add_upsampled2(a, b, compact, hw):
    """out [M, N] = a [M, K] @ b [K, N] + zero_stuffed(compact [n, H/2, W/2, N]) for M = n*H*W rows (the bottleneck's conv1 data
    gradient + the stride-2 downsample branch's, csrc/gemm_expand.hip).  Returns None when the shape is outside the streaming
    kernel's envelope (the caller materialises the zero-stuffed tensor instead)."""
    C.require_cuda(a, b, compact)
    m, k = a.shape
    k2, nn = b.shape
    h, wd = hw
    if k2 != k or a.stride(1) != 1 or b.stride(1) != 1 or not compact.is_contiguous() or compact.shape[-1] != nn or \
            a.dtype != b.dtype or compact.dtype != a.dtype:
        raise ValueError("gemm_add_upsampled2: shape / layout mismatch")
    out = torch.empty((m, nn), dtype=a.dtype, device=a.device)
    C.annotate(flops=2.0 * m * nn * k, tag="%dx%dx%d+up2" % (m, nn, k), bytes=float(a.numel() + b.numel() + out.numel() + compact.numel()) * 2)
    rc = _timed_optional("dle_gemm", C.lib().dle_gemm_expand_add_up2,
                         (C.ptr(a), C.ptr(b), C.ptr(out), C.ptr(compact), m, nn, k, a.stride(0), b.stride(0), nn, nn, 0, h, wd, C.dt(a),
                          C.stream()))
    if rc == 0:
        return None
    if rc != 1:
        C.check(rc - 1000 if rc > 1000 else rc, "dle_gemm_expand_add_up2")
    return out


def wgrad1x1(dy2d, x2d, out, accumulate=False):
    """out [Ko, C] fp32 (+)= dy2d [M, Ko]^T x2d [M, C] on the streaming weight-gradient kernel (csrc/wgrad1x1.hip).  Returns False
    when the shape is outside its envelope (the caller then uses gemm() with split-K)."""
    C.require_cuda(dy2d, x2d, out)
    m, ko = dy2d.shape
    c = x2d.shape[1]
    if x2d.shape[0] != m or not dy2d.is_contiguous() or not x2d.is_contiguous() or not out.is_contiguous() or \
            out.numel() != ko * c or out.dtype != torch.float32 or dy2d.dtype != x2d.dtype:
        return False
    need = int(C.lib().dle_wgrad1x1_workspace_for(m, ko, c))     # 0: outside the envelope -- no 64 MB scratch for a declined shape
    if need == 0:
        return False
    ws = splitk_workspace(dy2d.device, need)
    C.annotate(flops=2.0 * m * ko * c, tag="%dx%dx%d" % (ko, c, m), bytes=float(dy2d.numel() + x2d.numel()) * 2 + out.numel() * 4.0)
    # (recorded in the family of the split-K GEMM it replaces: the 1x1 gradients)
    rc = _timed_optional("dle_gemm", C.lib().dle_wgrad1x1_try,
                         (C.ptr(dy2d), C.ptr(x2d), C.ptr(out), m, ko, c, C.dt(dy2d), int(accumulate), C.ptr(ws), ws.numel() * 4, C.stream()),
                         replayable=not accumulate)
    if rc > 1:
        C.check(rc - 1000 if rc > 1000 else -1, "dle_wgrad1x1_try")
    return rc == 1


def conv2d_wgrad(dy, x, rs, stride=1, pad=0, out=None, accumulate=False, splitk=None):
    """dy [N,P,Q,Ko], x [N,H,W,C] -> dw [Ko,R,S,C] fp32."""
    C.require_cuda(dy, x, out)
    n, p, q, ko = dy.shape
    n2, h, wd, c = x.shape
    r, s = rs
    if n2 != n or not dy.is_contiguous() or not x.is_contiguous() or (p, q) != _conv_out(h, wd, r, s, stride, pad):
        raise ValueError("conv2d_wgrad: shape mismatch")
    if out is None:
        out = torch.empty((ko, r, s, c), dtype=torch.float32, device=x.device)
        accumulate = False
    m_out, n_out, k = ko, r * s * c, n * p * q
    if splitk is None:
        splitk = pick_splitk(m_out, n_out, k, target_blocks=1024)
    need = splitk * m_out * n_out * 4 if splitk > 1 else 0
    if (r, s, stride, pad) == (3, 3, 1, 1):                 # the halo-tile kernel's partial blocks (csrc/conv3x3_wgrad.hip)
        need = max(need, int(C.lib().dle_conv3x3_wgrad_workspace()))
    ws = splitk_workspace(x.device, need) if need else None
    C.annotate(flops=2.0 * n * p * q * ko * r * s * c, tag="wgrad %dx%dx%dx%d k%d %dx%d s%d" % (n, h, wd, c, ko, r, s, stride),
               bytes=float(dy.numel() + x.numel()) * 2 + out.numel() * 4.0)
    C.call("dle_conv2d_wgrad", C.ptr(dy), C.ptr(x), C.ptr(out), n, h, wd, c, ko, r, s, stride, pad, C.dt(x), splitk,
           int(accumulate), C.ptr(ws), ws.numel() * 4 if ws is not None else 0, C.stream())
    return out


# ------------------------------------------------------------------ ResNet HBM-bound ops (NHWC, 16-bit)
def nchw_to_nhwc(x, out_dtype, c_padded=None):
    """fp32 [N,C,H,W] -> 16-bit [N,H,W,Cp] with zero-filled padding channels."""
    C.require_cuda(x)
    if x.dtype != torch.float32 or x.dim() != 4 or not x.is_contiguous():
        raise ValueError("nchw_to_nhwc expects a contiguous fp32 NCHW tensor")
    n, c, h, w = x.shape
    cp = c_padded or (c + 7) // 8 * 8
    y = torch.empty((n, h, w, cp), dtype=out_dtype, device=x.device)
    C.call("dle_nchw_to_nhwc", C.ptr(x), C.ptr(y), n, c, h * w, cp, C.dt(y), C.stream())
    return y


def u8_nchw_normalize_nhwc(x, mean, std, out_dtype, c_padded=None):
    """uint8 [N,C,H,W] -> ((x - mean[c]) / std[c]) as 16-bit [N,H,W,Cp] (PrefetchedWrapper's normalisation + layout)."""
    C.require_cuda(x, mean, std)
    if x.dtype != torch.uint8 or x.dim() != 4 or not x.is_contiguous():
        raise ValueError("u8_nchw_normalize_nhwc expects a contiguous uint8 NCHW tensor")
    n, c, h, w = x.shape
    cp = c_padded or (c + 7) // 8 * 8
    y = torch.empty((n, h, w, cp), dtype=out_dtype, device=x.device)
    C.call("dle_u8_nchw_normalize_nhwc", C.ptr(x), C.ptr(y), C.ptr(mean), C.ptr(std), n, c, h * w, cp, C.dt(y), C.stream())
    return y


def _bn_ws(x2d):
    m, c = x2d.shape
    nbytes = C.lib().dle_bn_workspace_bytes(m, c)
    return splitk_workspace(x2d.device, nbytes)


def bn_fwd(x, gamma, beta, running_mean=None, running_var=None, eps=1e-5, momentum=0.1, residual=None, relu=True,
           out=None, want_mask=False):
    """Training-mode BatchNorm over the leading dims of NHWC x (+ residual, + ReLU).  -> (y, mean, rstd) or, with
    want_mask (and relu), (y, mean, rstd, relu_mask): the bit-packed y > 0 mask the backward pass reads instead of y."""
    C.require_cuda(x, gamma, beta, running_mean, running_var, residual, out)
    c = x.shape[-1]
    m = x.numel() // c
    x2 = x.reshape(m, c)
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    rstd = torch.empty(c, dtype=torch.float32, device=x.device)
    ws = _bn_ws(x2)
    C.annotate(bytes=float(x.numel()) * 2, tag="M%dxC%d" % (x.numel() // c, c))
    C.call("dle_bn_fwd_stats", C.ptr(x), m, c, eps, momentum, C.ptr(mean), C.ptr(rstd), C.ptr(running_mean),
           C.ptr(running_var), C.ptr(ws), ws.numel() * 4, C.dt(x), C.stream())
    y = torch.empty_like(x) if out is None else out
    mask = torch.empty(x.numel() // 8, dtype=torch.uint8, device=x.device) if (want_mask and relu) else None
    C.annotate(bytes=float(x.numel()) * (2 * (3 if residual is not None else 2) + (0.125 if mask is not None else 0)),
               tag="M%dxC%d%s" % (x.numel() // c, c, "+res" if residual is not None else ""))
    C.call("dle_bn_fwd_apply", C.ptr(x), C.ptr(residual), C.ptr(y), C.ptr(mask), C.ptr(mean), C.ptr(rstd), C.ptr(gamma),
           C.ptr(beta), m, c, int(relu), C.dt(x), C.stream())
    if want_mask:
        return y, mean, rstd, mask
    return y, mean, rstd


def conv2d_fwd_bnstats(x, w, stride=1, pad=0, running_mean=None, running_var=None, eps=1e-5, momentum=0.1):
    """conv (no bias / activation) + the training-mode BatchNorm statistics of its output in ONE pass over the
    activation: the convolution epilogue leaves per-tile column sums, a tiny deterministic fold turns them into
    mean / rstd / running stats.  x [N,H,W,C], w [Ko,R,S,C] -> (y [N,P,Q,Ko], mean [Ko], rstd [Ko])."""
    C.require_cuda(x, w, running_mean, running_var)
    n, h, wd, c = x.shape
    ko, r, s, c2 = w.shape
    if c2 != c or not x.is_contiguous() or not w.is_contiguous() or x.dtype != w.dtype:
        raise ValueError("conv2d_fwd_bnstats: x must be NHWC-contiguous and w KRSC-contiguous with matching C/dtype")
    p, q = _conv_out(h, wd, r, s, stride, pad)
    m = n * p * q
    y = torch.empty((n, p, q, ko), dtype=x.dtype, device=x.device)
    # partial rows: one per 128-row tile, or one per workgroup group of the streaming 1x1 kernel (gemm_expand.hip: at most
    # 1032, at most one per 64 rows rounded up to a multiple of 8); the call reports how many it wrote
    groups = max((m + 127) // 128, min(1032, (m + 63) // 64 + 8))
    ws = splitk_workspace(x.device, (groups + 32) * 2 * ko * 4)
    part, fold = ws[:groups * 2 * ko], ws[groups * 2 * ko:(groups + 32) * 2 * ko]
    import ctypes
    g_out = ctypes.c_int(0)
    C.annotate(flops=2.0 * m * ko * r * s * c, tag="fwd+stats %dx%dx%dx%d k%d %dx%d s%d" % (n, h, wd, c, ko, r, s, stride),
               bytes=float(x.numel() + w.numel() + y.numel()) * 2)
    C.call("dle_conv2d_fwd_colstats", C.ptr(x), C.ptr(w), C.ptr(y), n, h, wd, c, ko, r, s, stride, pad, C.dt(x),
           C.ptr(part), part.numel() * 4, ctypes.byref(g_out), C.stream())
    mean = torch.empty(ko, dtype=torch.float32, device=x.device)
    rstd = torch.empty(ko, dtype=torch.float32, device=x.device)
    C.call("dle_bn_stats_from_partials", C.ptr(part), g_out.value, m, ko, float(eps), float(momentum), C.ptr(mean),
           C.ptr(rstd), C.ptr(running_mean), C.ptr(running_var), C.ptr(fold), fold.numel() * 4, C.stream())
    return y, mean, rstd


def conv1x1_bnload_fwd(t, res, w, mean, rstd, gamma, beta, running_mean=None, running_var=None, eps=1e-5, momentum=0.1, res_bn=None):
    """The producer's BatchNorm-apply (+ residual) + ReLU on the operand load of a 1x1 convolution (csrc/conv_bnload.hip).
    t [.., K] pre-BatchNorm activations, res like t or None, w [N, 1, 1, K]; mean / rstd / gamma / beta: the PRODUCER's BatchNorm;
    running_mean / running_var / eps / momentum: THIS convolution's BatchNorm (its batch statistics come out of the epilogue).
    res_bn = (mean_r, rstd_r, gamma_r, beta_r): `res` is the downsample branch's CONVOLUTION output and its BatchNorm is applied on
    the residual's load (bit-identical to bn_fwd_apply(res, relu=False) in front).
    -> (out [.., N], y [.., K], bits, mean_out, rstd_out), or None when the shape is outside the kernel's envelope."""
    C.require_cuda(t, res, w, mean, rstd, gamma, beta, running_mean, running_var, *(res_bn or ()))
    k = t.shape[-1]
    n = w.shape[0]
    m = t.numel() // k
    if not t.is_contiguous() or not w.is_contiguous() or w.numel() != n * k or w.dtype != t.dtype or \
            (res is not None and (res.shape != t.shape or not res.is_contiguous() or res.dtype != t.dtype)):
        raise ValueError("conv1x1_bnload_fwd: dense operands of one 16-bit dtype expected")
    groups = C.lib().dle_conv1x1_bnload_groups(m, n, k)
    if groups == 0:
        return None
    out = torch.empty(t.shape[:-1] + (n,), dtype=t.dtype, device=t.device)
    y = torch.empty_like(t)
    bits = torch.empty(t.numel() // 8, dtype=torch.uint8, device=t.device)
    ws = splitk_workspace(t.device, (groups + 32) * 2 * n * 4)
    part, fold = ws[:groups * 2 * n], ws[groups * 2 * n:(groups + 32) * 2 * n]
    C.annotate(flops=2.0 * m * n * k, tag="bn+conv %dx%dx%d%s" % (m, n, k, "+res" if res is not None else ""),
               bytes=float(t.numel() * (3 if res is not None else 2) + out.numel() + w.numel()) * 2 + bits.numel())
    if res_bn is not None:
        if res is None:
            raise ValueError("conv1x1_bnload_fwd: res_bn without a residual")
        C.annotate(flops=2.0 * m * n * k, tag="bn+conv %dx%dx%d+bnres" % (m, n, k),
                   bytes=float(t.numel() * 3 + out.numel() + w.numel()) * 2 + bits.numel())
        rc = _timed_optional("dle_conv1x1_bnload_fwd", C.lib().dle_conv1x1_bnload_fwd2,
                             (C.ptr(t), C.ptr(res), C.ptr(w), C.ptr(out), C.ptr(y), C.ptr(bits), C.ptr(mean), C.ptr(rstd), C.ptr(gamma),
                              C.ptr(beta), C.ptr(res_bn[0]), C.ptr(res_bn[1]), C.ptr(res_bn[2]), C.ptr(res_bn[3]),
                              C.ptr(part), part.numel() * 4, m, n, k, C.dt(t), C.stream()))
    else:
        rc = _timed_optional("dle_conv1x1_bnload_fwd", C.lib().dle_conv1x1_bnload_fwd,
                             (C.ptr(t), C.ptr(res), C.ptr(w), C.ptr(out), C.ptr(y), C.ptr(bits), C.ptr(mean), C.ptr(rstd), C.ptr(gamma),
                              C.ptr(beta), C.ptr(part), part.numel() * 4, m, n, k, C.dt(t), C.stream()))
    if rc == 0:
        return None
    if rc != 1:
        C.check(rc - 1000 if rc > 1000 else -1, "dle_conv1x1_bnload_fwd")
    mean_o = torch.empty(n, dtype=torch.float32, device=t.device)
    rstd_o = torch.empty(n, dtype=torch.float32, device=t.device)
    C.call("dle_bn_stats_from_partials", C.ptr(part), groups, m, n, float(eps), float(momentum), C.ptr(mean_o), C.ptr(rstd_o),
           C.ptr(running_mean), C.ptr(running_var), C.ptr(fold), fold.numel() * 4, C.stream())
    return out, y, bits, mean_o, rstd_o


# ---- the ResNet stem on its own kernels (csrc/stem.hip): 4-channel image, packed weights
def stem_pack_weight(w_master, out_dtype, out=None):
    """fp32 master of the stem convolution, a channels_last [64, 3, 7, 7] parameter (memory order [64][7][7][3]) -> the packed
    16-bit [64, 7, 8, 4] operand of the stem kernels (tap 7 / channel 3 zero)."""
    C.require_cuda(w_master, out)
    ko, ci, r, s = w_master.shape
    phys = w_master.permute(0, 2, 3, 1)
    if (ko, ci, r, s) != (64, 3, 7, 7) or not phys.is_contiguous() or w_master.dtype != torch.float32:
        raise ValueError("stem_pack_weight expects a channels_last fp32 [64, 3, 7, 7] weight")
    if out is None:
        out = torch.empty((64, 7, 8, 4), dtype=out_dtype, device=w_master.device)
    C.call("dle_stem_pack_weight", C.ptr(w_master), C.ptr(out), C.dt(out), C.stream())
    return out


def stem_conv_fwd(x4, w2, want_stats=True):
    """x4 [N, H, W, 4] 16-bit, w2 packed [64, 7, 8, 4] -> (y [N, P, Q, 64], partial [groups, 2, 64] fp32 or None)."""
    C.require_cuda(x4, w2)
    n, h, wd, c = x4.shape
    if c != 4 or not x4.is_contiguous() or tuple(w2.shape) != (64, 7, 8, 4) or w2.dtype != x4.dtype or not w2.is_contiguous():
        raise ValueError("stem_conv_fwd: x4 must be [N, H, W, 4] and w2 the packed [64, 7, 8, 4] weight of the same dtype")
    p, q = (h - 1) // 2 + 1, (wd - 1) // 2 + 1
    y = torch.empty((n, p, q, 64), dtype=x4.dtype, device=x4.device)
    part = None
    if want_stats:
        groups = C.lib().dle_stem_conv7_groups(n, h)
        part = splitk_workspace(x4.device, (groups + 32) * 2 * 64 * 4)[:(groups + 32) * 2 * 64]
    C.annotate(flops=2.0 * n * p * q * 64 * 147, tag="stem fwd%s %dx%dx%d" % ("+stats" if want_stats else "", n, h, wd),
               bytes=float(x4.numel() + y.numel()) * 2)
    C.call("dle_stem_conv7_fwd", C.ptr(x4), C.ptr(w2), C.ptr(y), C.ptr(part), part.numel() * 4 if part is not None else 0, n, h, wd,
           C.dt(x4), C.stream())
    return y, part


def stem_conv_fwd_bnstats(x4, w2, running_mean=None, running_var=None, eps=1e-5, momentum=0.1):
    """Stem convolution + the training-mode BatchNorm statistics of its output (the contract of conv2d_fwd_bnstats)."""
    y, part = stem_conv_fwd(x4, w2, want_stats=True)
    n, p, q, ko = y.shape
    groups = C.lib().dle_stem_conv7_groups(n, x4.shape[1])
    mean = torch.empty(ko, dtype=torch.float32, device=y.device)
    rstd = torch.empty(ko, dtype=torch.float32, device=y.device)
    fold = part[groups * 2 * ko:]
    C.call("dle_bn_stats_from_partials", C.ptr(part), groups, n * p * q, ko, float(eps), float(momentum), C.ptr(mean),
           C.ptr(rstd), C.ptr(running_mean), C.ptr(running_var), C.ptr(fold), fold.numel() * 4, C.stream())
    return y, mean, rstd


def stem_conv_wgrad(dy, x4, out, accumulate=False):
    """dy [N, P, Q, 64], x4 [N, H, W, 4] -> out: fp32 [64 * 7 * 7 * 3] in the master's KRSC memory order (the flat gradient view of
    the channels_last [64, 3, 7, 7] parameter)."""
    C.require_cuda(dy, x4, out)
    n, h, wd, c = x4.shape
    if c != 4 or not x4.is_contiguous() or not dy.is_contiguous() or dy.shape[0] != n or dy.shape[3] != 64 or \
            out.numel() != 64 * 147 or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("stem_conv_wgrad: shape mismatch")
    ws = splitk_workspace(x4.device, int(C.lib().dle_stem_conv7_wgrad_workspace(n, h)))
    C.annotate(flops=2.0 * dy.numel() * 147, tag="stem wgrad %dx%dx%d" % (n, h, wd), bytes=float(dy.numel() + x4.numel()) * 2)
    C.call("dle_stem_conv7_wgrad", C.ptr(dy), C.ptr(x4), C.ptr(out), C.ptr(ws), ws.numel() * 4, n, h, wd, C.dt(x4), int(accumulate),
           C.stream())
    return out


def bn_fwd_apply(x, mean, rstd, gamma, beta, residual=None, relu=True, want_mask=False, residual_bn=None):
    """y = act((x - mean) * rstd * gamma + beta (+ residual)) with given statistics -> (y, relu_mask or None).
    residual_bn = (mean_r, rstd_r, gamma_r, beta_r): `residual` is the downsample branch's convolution output, its BatchNorm (no
    ReLU) is applied on load (csrc/convnet.hip bn_apply2_pf_kernel; needs relu and want_mask)."""
    C.require_cuda(x, mean, rstd, gamma, beta, residual, *(residual_bn or ()))
    c = x.shape[-1]
    m = x.numel() // c
    y = torch.empty_like(x)
    if residual_bn is not None:
        if residual is None or not relu or not want_mask or residual.shape != x.shape or residual.dtype != x.dtype \
                or not (x.is_contiguous() and residual.is_contiguous()):
            raise ValueError("bn_fwd_apply: residual_bn needs a dense residual of x's shape / dtype, relu and want_mask")
        mask = torch.empty(x.numel() // 8, dtype=torch.uint8, device=x.device)
        C.annotate(bytes=float(x.numel()) * (6 + 0.125), tag="M%dxC%d+bnres" % (m, c))
        C.call("dle_bn_fwd_apply2", C.ptr(x), C.ptr(residual), C.ptr(y), C.ptr(mask), C.ptr(mean), C.ptr(rstd), C.ptr(gamma), C.ptr(beta),
               C.ptr(residual_bn[0]), C.ptr(residual_bn[1]), C.ptr(residual_bn[2]), C.ptr(residual_bn[3]), m, c, C.dt(x), C.stream())
        return y, mask
    mask = torch.empty(x.numel() // 8, dtype=torch.uint8, device=x.device) if (want_mask and relu) else None
    C.annotate(bytes=float(x.numel()) * (2 * (3 if residual is not None else 2) + (0.125 if mask is not None else 0)),
               tag="M%dxC%d%s" % (m, c, "+res" if residual is not None else ""))
    C.call("dle_bn_fwd_apply", C.ptr(x), C.ptr(residual), C.ptr(y), C.ptr(mask), C.ptr(mean), C.ptr(rstd), C.ptr(gamma),
           C.ptr(beta), m, c, int(relu), C.dt(x), C.stream())
    return y, mask


def bn_relu_maxpool_fwd(t, mean, rstd, gamma, beta):
    """maxpool3x3/2/pad1(relu(bn(t))) in one pass (the ResNet stem) -> (pooled [N,H/2,W/2,C], argmax uint8, relu_mask of the
    never-materialised activation); bit-identical to bn_fwd_apply + maxpool_fwd.  H, W even."""
    C.require_cuda(t, mean, rstd, gamma, beta)
    n, h, w, c = t.shape
    if h % 2 or w % 2 or not t.is_contiguous():
        raise ValueError("bn_relu_maxpool_fwd: contiguous NHWC input with even H, W expected")
    y = torch.empty((n, h // 2, w // 2, c), dtype=t.dtype, device=t.device)
    am = torch.empty((n, h // 2, w // 2, c), dtype=torch.uint8, device=t.device)
    mask = torch.empty(t.numel() // 8, dtype=torch.uint8, device=t.device)
    C.annotate(bytes=float(t.numel() + y.numel()) * 2 + am.numel() + mask.numel(), tag="N%dx%dx%dxC%d" % (n, h, w, c))
    C.call("dle_bn_relu_maxpool_fwd", C.ptr(t), C.ptr(y), C.ptr(am), C.ptr(mask), C.ptr(mean), C.ptr(rstd), C.ptr(gamma), C.ptr(beta),
           n, h, w, c, C.dt(t), C.stream())
    return y, am, mask


def bn_bwd(dy, y, x, mean, rstd, gamma, dgamma, dbeta, want_skip_grad=False, dx_out=None, relu_mask=None, reduce_done=False):
    """-> (dx, g) ; y = saved post-ReLU output or relu_mask = its bit-packed y > 0 mask (both None when the BN had no
    ReLU); g = dy*(y>0) if requested."""
    C.require_cuda(dy, y, x, mean, rstd, gamma, dgamma, dbeta, relu_mask)
    c = x.shape[-1]
    m = x.numel() // c
    ws = _bn_ws(x.reshape(m, c))
    if relu_mask is not None:
        y = None
    act_bytes = 0.125 if relu_mask is not None else (2.0 if y is not None else 0.0)
    relu_tag = "+relu" if (y is not None or relu_mask is not None) else ""
    if not reduce_done:                      # (reduce_done: the producer of dy already left dgamma / dbeta: gemm_masked_add_bnred)
        C.annotate(bytes=float(x.numel()) * (4 + act_bytes), tag="M%dxC%d%s" % (x.numel() // c, c, relu_tag))
        C.call("dle_bn_bwd_reduce", C.ptr(dy), C.ptr(y), C.ptr(relu_mask), C.ptr(x), C.ptr(mean), C.ptr(rstd), C.ptr(dgamma),
               C.ptr(dbeta), m, c, 0, C.ptr(ws), ws.numel() * 4, C.dt(x), C.stream())
    dx = torch.empty_like(x) if dx_out is None else dx_out
    g = torch.empty_like(x) if want_skip_grad else None
    C.annotate(bytes=float(x.numel()) * (6 + act_bytes + 2 * int(want_skip_grad)),
               tag="M%dxC%d%s%s" % (x.numel() // c, c, relu_tag, "+skip" if want_skip_grad else ""))
    C.call("dle_bn_bwd_apply", C.ptr(dy), C.ptr(y), C.ptr(relu_mask), C.ptr(x), C.ptr(dx), C.ptr(g), C.ptr(mean),
           C.ptr(rstd), C.ptr(gamma), C.ptr(dgamma), C.ptr(dbeta), m, c, C.dt(x), C.stream())
    return dx, g


def bn_bwd_reduce2(dy, relu_mask, x1, mean1, rstd1, dgamma1, dbeta1, x2, mean2, rstd2, dgamma2, dbeta2):
    """The backward reduction (dgamma, dbeta) of TWO BatchNorms fed by the same gradient dy under the same keep bits -- bn3 and the
    downsample branch's BatchNorm of a bottleneck's first block -- in one pass over dy and the mask (csrc/convnet.hip
    bn_reduce2_kernel); bit-identical to two bn_bwd reductions.  The callers then run their apply passes with reduce_done=True."""
    C.require_cuda(dy, relu_mask, x1, mean1, rstd1, dgamma1, dbeta1, x2, mean2, rstd2, dgamma2, dbeta2)
    c = x1.shape[-1]
    m = x1.numel() // c
    if x2.shape != x1.shape or dy.shape != x1.shape or not (dy.is_contiguous() and x1.is_contiguous() and x2.is_contiguous()) \
            or x1.dtype != dy.dtype or x2.dtype != dy.dtype:
        raise ValueError("bn_bwd_reduce2: dy, x1, x2 must be dense tensors of one shape / dtype")
    ws = splitk_workspace(x1.device, 2 * int(C.lib().dle_bn_workspace_bytes(m, c)))
    C.annotate(bytes=float(x1.numel()) * (6 + 0.125), tag="M%dxC%d+relu x2" % (m, c))
    C.call("dle_bn_bwd_reduce2", C.ptr(dy), C.ptr(relu_mask), C.ptr(x1), C.ptr(mean1), C.ptr(rstd1), C.ptr(dgamma1), C.ptr(dbeta1),
           C.ptr(x2), C.ptr(mean2), C.ptr(rstd2), C.ptr(dgamma2), C.ptr(dbeta2), m, c, C.ptr(ws), ws.numel() * 4, C.dt(x1), C.stream())


def gemm_masked_add_bnred(g2, w, m, n, k, addend, bits, t2, bits2, mean2, rstd2, dgamma2, dbeta2):
    """dx [m, n] = g2 [m, k] w [k, n] + addend under `bits` (the masked residual gradient, as gemm(act=ACT_ADD_MASKED)) AND the
    backward reduction of the BatchNorm that dx flows into: dgamma2 / dbeta2 (fp32 [n], overwritten) from g = dx under bits2 and
    xhat = (t2 - mean2) rstd2 -- what bn_bwd's first pass computes from a re-read of dx, t2 and the mask.  -> dx, or None outside
    the streaming kernel's envelope (nothing launched: run gemm and let the unit reduce for itself)."""
    C.require_cuda(g2, w, addend, bits, t2, bits2, mean2, rstd2, dgamma2, dbeta2)
    if (g2.dtype not in (torch.float16, torch.bfloat16) or m < 4096 or k not in (64, 128, 256) or n % 128 != 0 or n < 2 * k
            or (k == 256 and os.environ.get("DLE_GEMM_BNRED_K256", "0") != "1")
            or not (g2.is_contiguous() and w.is_contiguous() and addend.is_contiguous() and t2.is_contiguous())
            or t2.numel() != m * n or os.environ.get("DLE_RN50_FUSE_BNRED", "1") == "0"):
        return None
    groups = int(C.lib().dle_gemm_expand_groups(m, n, k))
    if groups <= 0:
        return None
    ws = splitk_workspace(g2.device, groups * 2 * n * 4)
    out = torch.empty((m, n), dtype=g2.dtype, device=g2.device)
    by = float(m) * (k + 3 * n) * g2.element_size() + float(m) * n * 0.25 + float(k) * n * 2
    C.annotate(bytes=by, flops=2.0 * m * n * k, tag="%dx%dx%d+src+aux+bnred" % (m, n, k))
    rc = _timed_optional("dle_gemm", C.lib().dle_gemm_expand_masked_bnred,
                         (C.ptr(g2), C.ptr(w), C.ptr(out), C.ptr(addend), C.ptr(bits), C.ptr(t2), C.ptr(bits2), C.ptr(mean2),
                          C.ptr(rstd2), C.ptr(ws), ws.numel() * 4, m, n, k, g2.stride(0), w.stride(0), out.stride(0), 0, C.dt(g2),
                          C.stream()))
    if rc > 1:
        C.check(rc - 1000 if rc > 1000 else -1, "dle_gemm_expand_masked_bnred")
    if rc != 1:
        return None
    C.call("dle_bn_bwd_finish", C.ptr(ws), groups, n, C.ptr(dgamma2), C.ptr(dbeta2), 0, C.stream())
    return out


def bn_bwd_conv1x1_dgrad(dy, x, mean, rstd, gamma, dgamma, dbeta, w, relu_mask=None, reduce_done=False, bnred=None):
    """BatchNorm backward of a conv + BN unit whose convolution is 1x1 / stride 1, with the unit's data gradient in the same
    pass: the reduction (dgamma, dbeta) as in bn_bwd, then ONE kernel that applies the backward on the operand load of
    dx = dt W (csrc/conv_bnbwd.hip).  x: the convolution output [.., K]; w: the 16-bit weight [K, N] (n contiguous).
    bnred = (t2 [.., N], bits2, mean2, rstd2, dgamma2, dbeta2): dx is the gradient that enters a second BatchNorm (bn2 of the
    bottleneck, behind a ReLU with keep bits bits2); its backward reduction is taken from dx in the same kernel and left in
    dgamma2 / dbeta2 (the caller's second unit then skips its own first pass).
    -> (dt, dx [m, N], bnred taken) or None outside the kernel's envelope (nothing has been launched then: run bn_bwd + gemm).
    (When the C side declines AFTER the reduction was launched the unfused apply + GEMM run here: same triple, taken = False.)"""
    C.require_cuda(dy, x, mean, rstd, gamma, dgamma, dbeta, w, relu_mask)
    k = x.shape[-1]
    m = x.numel() // k
    n = w.shape[-1]
    if (m < 4096 or k != 256 or n != 64 or w.numel() != k * n or not (dy.is_contiguous() and x.is_contiguous() and w.is_contiguous())
            or x.dtype not in (torch.float16, torch.bfloat16) or dy.dtype != x.dtype or w.dtype != x.dtype
            or os.environ.get("DLE_CONV_BNBWD", "1") == "0"
            or any(t is not None and t.data_ptr() % 16 for t in (dy, x, w, relu_mask))):     # (a view with a storage offset)
        return None
    ws = _bn_ws(x.reshape(m, k))
    act_bytes = 0.125 if relu_mask is not None else 0.0
    relu_tag = "+relu" if relu_mask is not None else ""
    if not reduce_done:                      # (reduce_done: the producer of dy already left dgamma / dbeta: gemm_masked_add_bnred)
        C.annotate(bytes=float(x.numel()) * (4 + act_bytes), tag="M%dxC%d%s" % (m, k, relu_tag))
        C.call("dle_bn_bwd_reduce", C.ptr(dy), None, C.ptr(relu_mask), C.ptr(x), C.ptr(mean), C.ptr(rstd), C.ptr(dgamma),
               C.ptr(dbeta), m, k, 0, C.ptr(ws), ws.numel() * 4, C.dt(x), C.stream())
    dt = torch.empty_like(x)
    dx = torch.empty((m, n), dtype=x.dtype, device=x.device)
    t2 = bits2 = mean2 = rstd2 = part = None
    groups = 0
    if bnred is not None and bnred[0].numel() == m * n and bnred[0].is_contiguous() and bnred[1] is not None \
            and os.environ.get("DLE_RN50_FUSE_BNRED", "1") != "0" and os.environ.get("DLE_RN50_FUSE_BNRED2", "1") != "0":
        t2, bits2, mean2, rstd2 = bnred[:4]
        groups = int(C.lib().dle_conv1x1_bnbwd_groups(m))
        part = splitk_workspace(x.device, groups * 2 * n * 4)
    C.annotate(bytes=float(x.numel()) * (6 + act_bytes) + dx.numel() * (2.0 + (2.125 if t2 is not None else 0.0)),
               flops=2.0 * m * n * k, tag="bn_bwd+dgrad %dx%dx%d%s%s" % (m, n, k, relu_tag, "+bnred" if t2 is not None else ""))
    rc = _timed_optional("dle_conv1x1_bnbwd_dgrad", C.lib().dle_conv1x1_bnbwd_dgrad,
                         (C.ptr(dy), C.ptr(x), C.ptr(relu_mask), C.ptr(w), C.ptr(dt), C.ptr(dx), C.ptr(mean), C.ptr(rstd),
                          C.ptr(gamma), C.ptr(dgamma), C.ptr(dbeta), C.ptr(t2), C.ptr(bits2), C.ptr(mean2), C.ptr(rstd2),
                          C.ptr(part), part.numel() * 4 if part is not None else 0, m, n, k, C.dt(x), C.stream()))
    if rc > 1:
        C.check(rc - 1000 if rc > 1000 else -1, "dle_conv1x1_bnbwd_dgrad")
    if rc != 1:
        # the C side declined (a condition the envelope above cannot see, e.g. its statically cached DLE_CONV_BNBWD pin): the
        # reduction has been launched and dgamma / dbeta are final -- continue HERE with the unfused apply + GEMM, so that a
        # caller only ever sees None (nothing launched) or the (dt, dx, taken) triple
        dt, _ = bn_bwd(dy, None, x, mean, rstd, gamma, dgamma, dbeta, relu_mask=relu_mask, reduce_done=True, dx_out=dt)
        gemm(dt.view(m, k), w, m, n, k, True, False, out=dx)
        return dt, dx, False
    if t2 is not None:
        C.call("dle_bn_bwd_finish", C.ptr(part), groups, n, C.ptr(bnred[4]), C.ptr(bnred[5]), 0, C.stream())
    return dt, dx, t2 is not None


def maxpool_fwd(x, ksize=3, stride=2, pad=1):
    C.require_cuda(x)
    n, h, w, c = x.shape
    p, q = (h + 2 * pad - ksize) // stride + 1, (w + 2 * pad - ksize) // stride + 1
    y = torch.empty((n, p, q, c), dtype=x.dtype, device=x.device)
    am = torch.empty((n, p, q, c), dtype=torch.uint8, device=x.device)
    C.annotate(bytes=float(x.numel() + y.numel()) * 2 + am.numel())
    C.call("dle_maxpool_fwd", C.ptr(x), C.ptr(y), C.ptr(am), n, h, w, c, ksize, stride, pad, C.dt(x), C.stream())
    return y, am


def maxpool_bwd(dy, argmax, in_hw, ksize=3, stride=2, pad=1):
    C.require_cuda(dy, argmax)
    n, p, q, c = dy.shape
    h, w = in_hw
    dx = torch.empty((n, h, w, c), dtype=dy.dtype, device=dy.device)
    C.annotate(bytes=float(dx.numel() + dy.numel()) * 2 + argmax.numel())
    C.call("dle_maxpool_bwd", C.ptr(dy), C.ptr(argmax), C.ptr(dx), n, h, w, c, ksize, stride, pad, C.dt(dy), C.stream())
    return dx


def pool_bn_bwd(dy, argmax, x, mean, rstd, gamma, dgamma, dbeta, relu_mask):
    """The stem's backward chain MaxPool2d(3, 2, 1) -> ReLU -> BatchNorm in two passes that gather the pooling gradient from dy
    [N, H/2, W/2, C] + argmax themselves (csrc/convnet.hip pool_bn_bwd_kernel): the full-resolution pooling gradient is never
    written.  x [N, H, W, C]: the BatchNorm's input; dgamma / dbeta are written.  -> dz, or None outside the envelope (nothing
    launched: maxpool_bwd + bn_bwd)."""
    C.require_cuda(dy, argmax, x, mean, rstd, gamma, dgamma, dbeta, relu_mask)
    n, h, w, c = x.shape
    if relu_mask is None or dy.shape != (n, h // 2, w // 2, c) or not (dy.is_contiguous() and x.is_contiguous()) or dy.dtype != x.dtype:
        return None
    nbytes = int(C.lib().dle_pool_bn_bwd_workspace_bytes(n, h, w, c))
    if nbytes == 0:
        return None
    ws = splitk_workspace(x.device, nbytes)
    dz = torch.empty_like(x)
    # two passes: (pooled gradient + argmax + x + keep bits) twice, dz once
    C.annotate(bytes=2.0 * (dy.numel() * 3 + x.numel() * 2.125) + x.numel() * 2.0, tag="N%dx%dx%dxC%d" % (n, h, w, c))
    C.call("dle_pool_bn_bwd", C.ptr(dy), C.ptr(argmax), C.ptr(relu_mask), C.ptr(x), C.ptr(dz), C.ptr(mean), C.ptr(rstd), C.ptr(gamma),
           C.ptr(dgamma), C.ptr(dbeta), n, h, w, c, C.ptr(ws), ws.numel() * 4, C.dt(x), C.stream())
    return dz


def avgpool_fwd(x):
    C.require_cuda(x)
    n, h, w, c = x.shape
    y = torch.empty((n, c), dtype=x.dtype, device=x.device)
    C.call("dle_avgpool_fwd", C.ptr(x), C.ptr(y), n, h * w, c, C.dt(x), C.stream())
    return y


def avgpool_bwd(dy, hw):
    C.require_cuda(dy)
    n, c = dy.shape
    h, w = hw
    dx = torch.empty((n, h, w, c), dtype=dy.dtype, device=dy.device)
    C.call("dle_avgpool_bwd", C.ptr(dy), C.ptr(dx), n, h * w, c, C.dt(dy), C.stream())
    return dx


def softmax_xent(logits, target, smoothing=0.0, ignore_index=-100, grad_scale=None, grad_dtype=None, ld_out=None):
    """logits fp32 [rows, classes(+pad)], target int64 [rows] -> (mean loss [1], dlogits or None)."""
    C.require_cuda(logits, target, grad_scale)
    if logits.dtype != torch.float32 or logits.dim() != 2 or logits.stride(1) != 1 or target.dtype != torch.int64:
        raise ValueError("softmax_xent expects fp32 [rows, classes] logits and int64 targets")
    rows, classes = logits.shape
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    scratch = torch.empty(1, dtype=torch.int32, device=logits.device)
    dl = None
    ldo = ld_out or classes
    if grad_dtype is not None:
        dl = torch.empty((rows, ldo), dtype=grad_dtype, device=logits.device)
    C.call("dle_softmax_xent", C.ptr(logits), C.ptr(target), C.ptr(loss), C.ptr(dl), C.ptr(grad_scale), C.ptr(scratch),
           rows, classes, logits.stride(0), ldo, float(smoothing), int(ignore_index),
           C.dt(grad_dtype) if grad_dtype is not None else 0, C.stream())
    return loss, dl


# ------------------------------------------------------------------ BERT ops (hidden states [tokens, H], 16-bit)
def gemm_batched(a, b, c, m, n, k, lda, ldb, ldc, a_kc, b_kc, batch, batch_inner, sa, sb, sc, alpha=1.0):
    """C[z] = alpha * A[z](m,k) B[z](n,k) for z = (zo, zi); sa/sb/sc = (outer, inner) element strides."""
    C.require_cuda(a, b, c)
    C.annotate(flops=2.0 * m * n * k * batch, tag="b%dx%dx%dx%d" % (batch, m, n, k),
               bytes=float(batch) * (m * k + n * k + m * n) * 2)
    C.call("dle_gemm_batched", C.ptr(a), C.ptr(b), C.ptr(c), m, n, k, lda, ldb, ldc, int(a_kc), int(b_kc), C.dt(a),
           C.dt(c), float(alpha), batch, batch_inner, sa[0], sa[1], sb[0], sb[1], sc[0], sc[1], C.stream())
    return c


def layernorm_fwd(x, gamma, beta, residual=None, eps=1e-12, write_z=True):
    """y = LN(x + residual).  -> (y, z, mean, rstd); z is x itself when there is no residual."""
    C.require_cuda(x, gamma, beta, residual)
    rows, h = x.shape
    y = torch.empty_like(x)
    z = torch.empty_like(x) if (residual is not None and write_z) else None
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    C.annotate(bytes=float(x.numel()) * 2 * (4 if residual is not None else 2), tag="R%dxH%d" % (x.numel() // h, h))
    C.call("dle_layernorm_fwd", C.ptr(x), C.ptr(residual), C.ptr(z), C.ptr(y), C.ptr(gamma), C.ptr(beta), C.ptr(mean),
           C.ptr(rstd), rows, h, float(eps), C.dt(x), C.stream())
    return y, (z if z is not None else x), mean, rstd


def layernorm_bwd(dy, z, mean, rstd, gamma, dgamma, dbeta, accumulate=False):
    C.require_cuda(dy, z, mean, rstd, gamma, dgamma, dbeta)
    rows, h = z.shape
    dz = torch.empty_like(z)
    ws = splitk_workspace(z.device, C.lib().dle_layernorm_workspace_bytes(h))
    C.annotate(bytes=float(z.numel()) * 2 * 3, tag="R%dxH%d" % (rows, h))
    C.call("dle_layernorm_bwd", C.ptr(dy), C.ptr(z), C.ptr(mean), C.ptr(rstd), C.ptr(gamma), C.ptr(dz), C.ptr(dgamma),
           C.ptr(dbeta), rows, h, int(accumulate), C.ptr(ws), ws.numel() * 4, C.dt(z), C.stream())
    return dz


def dropout_add_layernorm_bwd(dy, z, mean, rstd, gamma, mask, p, dgamma, dbeta, dbias=None, accumulate=False):
    """Backward of dropout_add_layernorm_fwd in one pass -> (dz, dx): dz feeds the residual branch, dx = dz * keep / (1 - p)
    the dense layer; dbias (+)= column sums of dx."""
    C.require_cuda(dy, z, mean, rstd, gamma, mask, dgamma, dbeta, dbias)
    rows, h = z.shape
    dz, dx = torch.empty_like(z), torch.empty_like(z)
    ws = splitk_workspace(z.device, C.lib().dle_layernorm_workspace_bytes(h))
    C.annotate(bytes=float(z.numel()) * 2 * 4 + z.numel() / 8, tag="R%dxH%d+drop" % (rows, h))
    C.call("dle_dropout_add_layernorm_bwd", C.ptr(dy), C.ptr(z), C.ptr(mean), C.ptr(rstd), C.ptr(gamma), C.ptr(mask), float(p),
           C.ptr(dz), C.ptr(dx), C.ptr(dgamma), C.ptr(dbeta), C.ptr(dbias), rows, h, int(accumulate), C.ptr(ws),
           ws.numel() * 4, C.dt(z), C.stream())
    return dz, dx


def embed_sum(word, pos, typ, ids, token_type, seq_len, out_dtype):
    C.require_cuda(word, pos, typ, ids, token_type)
    t = ids.numel()
    h = word.shape[1]
    z = torch.empty((t, h), dtype=out_dtype, device=word.device)
    C.call("dle_embed_sum", C.ptr(word), C.ptr(pos), C.ptr(typ), C.ptr(ids), C.ptr(token_type), C.ptr(z), t, seq_len, h,
           C.dt(z), C.stream())
    return z


def embed_scatter_add_(grad_word, dz, ids):
    C.require_cuda(grad_word, dz, ids)
    C.call("dle_embed_scatter_add", C.ptr(dz), C.ptr(ids), C.ptr(grad_word), ids.numel(), dz.shape[1], C.dt(dz), C.stream())


def rows_select_sum(x, sel, k, out, accumulate=False):
    C.require_cuda(x, sel, out)
    rows, h = x.shape
    ws = splitk_workspace(x.device, 128 * k * h * 4)
    C.call("dle_rows_select_sum", C.ptr(x), C.ptr(sel), C.ptr(out), rows, h, k, int(accumulate), C.ptr(ws),
           ws.numel() * 4, C.dt(x), C.stream())
    return out


def rows_gather(src, idx):
    C.require_cuda(src, idx)
    dst = torch.empty((idx.numel(), src.shape[1]), dtype=src.dtype, device=src.device)
    C.call("dle_rows_gather", C.ptr(src), C.ptr(idx), C.ptr(dst), idx.numel(), src.shape[1], C.dt(src), C.stream())
    return dst


def rows_scatter_(dst, src, idx, accumulate=False):
    C.require_cuda(dst, src, idx)
    C.call("dle_rows_scatter", C.ptr(src), C.ptr(idx), C.ptr(dst), idx.numel(), src.shape[1], int(accumulate), C.dt(src),
           C.stream())
    return dst


def softmax_fwd_(scores, mask_add, rows_per_batch, scale):
    """In place over scores [..., L]."""
    C.require_cuda(scores, mask_add)
    l = scores.shape[-1]
    rows = scores.numel() // l
    C.annotate(bytes=float(scores.numel()) * 4, tag="R%dxL%d" % (rows, l))
    C.call("dle_softmax_fwd", C.ptr(scores), C.ptr(mask_add), rows, l, rows_per_batch, float(scale), C.dt(scores), C.stream())
    return scores


def softmax_bwd_(probs, dprobs, scale):
    C.require_cuda(probs, dprobs)
    l = probs.shape[-1]
    rows = probs.numel() // l
    C.annotate(bytes=float(probs.numel()) * 6, tag="R%dxL%d" % (rows, l))
    C.call("dle_softmax_bwd", C.ptr(probs), C.ptr(dprobs), rows, l, float(scale), C.dt(probs), C.stream())
    return dprobs


def act_bwd(g, src, act):
    """g * act'(src): act = C.ACT_GELU_BWD (src = pre-activation) or C.ACT_TANH_BWD (src = tanh output)."""
    C.require_cuda(g, src)
    out = torch.empty_like(g)
    C.call("dle_act_bwd", C.ptr(g), C.ptr(src), C.ptr(out), g.numel(), act, C.dt(g), C.stream())
    return out


# ------------------------------------------------------------------ dropout (BERT training mode)
def dropout_fwd(x, p, seed, offset, offset_base=None):
    """y = dropout(x) with the C-ABI's counter-based RNG.  Returns (y, mask) -- mask is bit-packed uint8 [numel/8].
    offset_base (all dropout wrappers): optional int64 DEVICE tensor [1] added to `offset` by the kernel -- the RNG
    advance of a HIP-graph-captured step lives in device memory."""
    C.require_cuda(x)
    n = x.numel()
    y = torch.empty_like(x)
    mask = torch.empty(n // 8, dtype=torch.uint8, device=x.device)
    C.annotate(bytes=float(n) * 2 * 2 + n / 8, tag="N%d" % n)
    C.call("dle_dropout_fwd", C.ptr(x), C.ptr(y), C.ptr(mask), n, float(p), int(seed), int(offset), C.ptr(offset_base),
           C.dt(x), C.stream())
    return y, mask


def dropout_bwd(dy, mask, p):
    C.require_cuda(dy, mask)
    dx = torch.empty_like(dy)
    n = dy.numel()
    C.annotate(bytes=float(n) * 2 * 2 + n / 8, tag="N%d" % n)
    C.call("dle_dropout_bwd", C.ptr(dy), C.ptr(mask), C.ptr(dx), n, float(p), C.dt(dy), C.stream())
    return dx


def dropout_add_layernorm_fwd(x, gamma, beta, residual, p, seed, offset, eps=1e-12, offset_base=None):
    """y = LayerNorm(dropout(x) + residual).  Returns (y, z, mean, rstd, mask)."""
    C.require_cuda(x, gamma, beta, residual)
    rows, h = x.shape
    y = torch.empty_like(x)
    z = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    mask = torch.empty(x.numel() // 8, dtype=torch.uint8, device=x.device)
    C.annotate(bytes=float(x.numel()) * 2 * 4 + x.numel() / 8, tag="R%dxH%d+drop" % (rows, h))
    C.call("dle_dropout_add_layernorm_fwd", C.ptr(x), C.ptr(residual), C.ptr(z), C.ptr(y), C.ptr(mask), C.ptr(gamma),
           C.ptr(beta), C.ptr(mean), C.ptr(rstd), rows, h, float(eps), float(p), int(seed), int(offset),
           C.ptr(offset_base), C.dt(x), C.stream())
    return y, z, mean, rstd, mask


def softmax_dropout_fwd_(scores, mask_add, rows_per_batch, scale, p, seed, offset, offset_base=None):
    """scores -> probs in place; returns (dropped = dropout(probs), mask)."""
    C.require_cuda(scores, mask_add)
    l = scores.shape[-1]
    rows = scores.numel() // l
    dropped = torch.empty_like(scores)
    mask = torch.empty(scores.numel() // 8, dtype=torch.uint8, device=scores.device)
    C.annotate(bytes=float(scores.numel()) * 6 + scores.numel() / 8, tag="R%dxL%d+drop" % (rows, l))
    C.call("dle_softmax_dropout_fwd", C.ptr(scores), C.ptr(dropped), C.ptr(mask), C.ptr(mask_add), rows, l,
           rows_per_batch, float(scale), float(p), int(seed), int(offset), C.ptr(offset_base), C.dt(scores), C.stream())
    return dropped, mask


def softmax_dropout_bwd_(probs, dprobs, mask, scale, p):
    C.require_cuda(probs, dprobs, mask)
    l = probs.shape[-1]
    rows = probs.numel() // l
    C.annotate(bytes=float(probs.numel()) * 6 + probs.numel() / 8, tag="R%dxL%d+drop" % (rows, l))
    C.call("dle_softmax_dropout_bwd", C.ptr(probs), C.ptr(dprobs), C.ptr(mask), rows, l, float(scale), float(p),
           C.dt(probs), C.stream())
    return dprobs


def attention_supported(seq_len, head_dim):
    """True when the fused attention kernels cover (seq_len, head_dim); otherwise use the batched-GEMM path."""
    return bool(C.lib().dle_attention_supported(int(seq_len), int(head_dim)))


def attention_fwd(qkv, mask_add, batch, seq_len, heads, scale, p=0.0, seed=0, offset=0, want_mask=False,
                  offset_base=None):
    """context = dropout(softmax(q k^T * scale + mask_add)) v for every (sequence, head) of qkv [T, 3H] in ONE kernel
    (BertSelfAttention.forward, modeling.py:340-384).  -> (ctx [T, H], stats [B*heads, S, 2 or 4] fp32, keep mask or None).
    seq_len 128, or a multiple of 128 up to 1024 (K / V then stream through LDS in 128-key blocks: csrc/attention.hip)."""
    C.require_cuda(qkv, mask_add)
    t, h3 = qkv.shape
    h = h3 // 3
    d = h // heads
    if not qkv.is_contiguous() or t != batch * seq_len or heads * d * 3 != h3:
        raise ValueError("attention_fwd: qkv must be a contiguous [batch * seq_len, 3 * heads * head_dim] tensor")
    ctx = torch.empty((t, h), dtype=qkv.dtype, device=qkv.device)
    stats = torch.empty((batch * heads, seq_len, int(C.lib().dle_attention_stats_floats(int(seq_len)))), dtype=torch.float32,
                        device=qkv.device)
    mask = torch.empty(batch * heads * seq_len * seq_len // 8, dtype=torch.uint8, device=qkv.device) \
        if (want_mask and p > 0) else None
    bh = batch * heads
    C.annotate(flops=4.0 * bh * seq_len * seq_len * d, bytes=float(t) * h * 2 * 4 + stats.numel() * 4,
               tag="B%dxh%dxS%dxd%d" % (batch, heads, seq_len, d))
    C.call("dle_attention_fwd", C.ptr(qkv), C.ptr(mask_add), C.ptr(ctx), C.ptr(stats), C.ptr(mask), batch, seq_len,
           heads, d, float(scale), float(p), int(seed), int(offset), C.ptr(offset_base), C.dt(qkv), C.stream())
    return ctx, stats, mask


def attention_bwd(qkv, dctx, mask_add, stats, batch, seq_len, heads, scale, p=0.0, seed=0, offset=0, offset_base=None,
                  colsum_partial=None, keep_mask=None):
    """dqkv [T, 3H] from dctx [T, H]: probabilities and dropout mask are recomputed from (qkv, stats, seed, offset).
    colsum_partial (optional fp32 [batch * seq_len / 128, 3H]) receives the column sums of dqkv per (sequence, 128-row block): the
    QKV bias gradient partials."""
    C.require_cuda(qkv, dctx, mask_add, stats)
    t, h3 = qkv.shape
    h = h3 // 3
    d = h // heads
    if not qkv.is_contiguous() or not dctx.is_contiguous() or dctx.shape != (t, h):
        raise ValueError("attention_bwd: qkv [T, 3H] and dctx [T, H] must be contiguous")
    dqkv = torch.empty_like(qkv)
    bh = batch * heads
    if colsum_partial is not None and (colsum_partial.numel() != batch * (seq_len // 128) * h3 or colsum_partial.dtype != torch.float32
                                       or not colsum_partial.is_contiguous()):
        raise ValueError("attention_bwd: colsum_partial must be fp32 [batch * seq_len / 128, 3H]")
    C.annotate(flops=10.0 * bh * seq_len * seq_len * d, bytes=float(t) * h * 2 * 7 + stats.numel() * 4,
               tag="B%dxh%dxS%dxd%d" % (batch, heads, seq_len, d))
    C.require_cuda(colsum_partial, keep_mask)
    if keep_mask is not None:
        # the keep mask attention_fwd(want_mask=True) returned: read by the S = 128 kernel instead of re-drawing it (same bits)
        C.call("dle_attention_bwd_keep", C.ptr(qkv), C.ptr(dctx), C.ptr(mask_add), C.ptr(stats), C.ptr(keep_mask), C.ptr(dqkv),
               C.ptr(colsum_partial), batch, seq_len, heads, d, float(scale), float(p), int(seed), int(offset), C.ptr(offset_base),
               C.dt(qkv), C.stream())
        return dqkv
    C.call("dle_attention_bwd", C.ptr(qkv), C.ptr(dctx), C.ptr(mask_add), C.ptr(stats), C.ptr(dqkv), C.ptr(colsum_partial),
           batch, seq_len,
           heads, d, float(scale), float(p), int(seed), int(offset), C.ptr(offset_base), C.dt(qkv), C.stream())
    return dqkv


def unpack_dropout_mask(mask, shape):
    """Bit-packed keep mask -> bool tensor of `shape` (bit k of byte i <-> flat element 8 i + k)."""
    bits = (mask.to(torch.int32).unsqueeze(1) >> torch.arange(8, device=mask.device, dtype=torch.int32)) & 1
    return bits.reshape(shape).bool()
