"""Tensor-level wrappers of the WaveGlow entry points of the C ABI (include/dle_mi355x.h, csrc/waveglow.hip): allocation and
argument marshalling only.  No CPU path: every function raises on a CPU tensor or a missing library (_cabi.require_cuda / lib()).

Reference pieces they stand in for (SpeechSynthesis/Tacotron2/waveglow/): model.py:34-41 (gate), :44-85 (invertible 1x1
convolution), :95-136 (weight_norm'd Conv1d weights), :160-231 (upsampling, grouping, affine coupling), loss_function.py:30-48.
"""
import torch

from .. import _cabi as C


def _row_view(t, what):
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError("%s: expected a 2-D view with unit inner stride" % what)
    return t.stride(0) if t.shape[0] > 1 else t.shape[1]


def taps(x, batch, steps, ntaps, dilation, left, out=None):
    """col[b*T+t, k*C+c] = x[b*T + t + (k-left)*dilation, c] inside a sample, 0 outside.  x [batch*steps, C] 16-bit, possibly a
    column slice of a wider matrix (row-strided view)."""
    C.require_cuda(x, out)
    ld_x = _row_view(x, "taps x")
    if x.shape[0] != batch * steps:
        raise ValueError("taps: x must have batch*steps rows")
    ch = x.shape[1]
    if out is None:
        out = torch.empty((batch * steps, ntaps * ch), dtype=x.dtype, device=x.device)
    if not out.is_contiguous() or out.numel() != batch * steps * ntaps * ch:
        raise ValueError("taps: out must be a contiguous [batch*steps, ntaps*C] matrix")
    C.annotate(bytes=float(x.numel() + out.numel()) * x.element_size(), tag="%dx%dx%d" % (batch * steps, ch, ntaps))
    C.call("dle_wg_taps", C.ptr(x), C.ptr(out), batch, steps, ch, ntaps, dilation, left, ld_x, C.dt(x), C.stream())
    return out


def taps_bwd(dcol, batch, steps, ch, ntaps, dilation, left, out, addend=None):
    """out[b*T+t, c] = sum_k dcol[b*T + t - (k-left)*dilation, k*C+c] (+ addend); out / addend are 2-D row-strided views
    (out may BE addend)."""
    C.require_cuda(dcol, out, addend)
    if not dcol.is_contiguous() or dcol.shape != (batch * steps, ntaps * ch):
        raise ValueError("taps_bwd: dcol must be a contiguous [batch*steps, ntaps*C] matrix")
    ld_out = _row_view(out, "taps_bwd out")
    ld_add = _row_view(addend, "taps_bwd addend") if addend is not None else 0
    C.annotate(bytes=float(dcol.numel() + out.numel()) * dcol.element_size(), tag="%dx%dx%d" % (batch * steps, ch, ntaps))
    C.call("dle_wg_taps_bwd", C.ptr(dcol), C.ptr(addend), C.ptr(out), batch, steps, ch, ntaps, dilation, left, ld_add,
           ld_out, C.dt(dcol), C.stream())
    return out


def gate_fwd(s, nc, out=None):
    """tanh(s[:, :nc]) * sigmoid(s[:, nc:2nc]); s is a [M, 2nc] row-strided view."""
    C.require_cuda(s, out)
    ld = _row_view(s, "gate_fwd s")
    m = s.shape[0]
    if out is None:
        out = torch.empty((m, nc), dtype=s.dtype, device=s.device)
    C.annotate(bytes=float(3 * m * nc) * s.element_size(), tag="%dx%d" % (m, nc))
    C.call("dle_wg_gate_fwd", C.ptr(s), C.ptr(out), m, nc, ld, C.dt(s), C.stream())
    return out


def gate_bwd(dacts, s, ds):
    """ds (a [M, 2nc] row-strided view) = gradient of the gate with respect to s."""
    C.require_cuda(dacts, s, ds)
    m, nc = dacts.shape
    if not dacts.is_contiguous():
        raise ValueError("gate_bwd: dacts must be contiguous")
    C.annotate(bytes=float(5 * m * nc) * s.element_size(), tag="%dx%d" % (m, nc))
    C.call("dle_wg_gate_bwd", C.ptr(dacts), C.ptr(s), C.ptr(ds), m, nc, _row_view(s, "gate_bwd s"),
           _row_view(ds, "gate_bwd ds"), C.dt(s), C.stream())
    return ds


def _state(t, what):
    if t.dtype != torch.float32 or t.dim() != 2 or t.shape[1] != 8 or not t.is_contiguous():
        raise ValueError("%s: flow-state tensors are contiguous fp32 [M, 8]" % what)
    return t.shape[0]


def invconv_fwd(x, w, c, dtype):
    """y = diag(I, W) x on the fp32 flow state; a0 = the first c/2 mixed channels, zero padded, 16-bit [M, 8]."""
    C.require_cuda(x, w)
    m = _state(x, "invconv_fwd")
    y = torch.empty_like(x)
    a0 = torch.empty((m, 8), dtype=dtype, device=x.device)
    C.call("dle_wg_invconv_fwd", C.ptr(x), C.ptr(w), C.ptr(y), C.ptr(a0), m, c, C.dt(dtype), C.stream())
    return y, a0


def logdet_inv(w, c, logdet_out, sign_out):
    """log|det W| -> logdet_out[0], sign -> sign_out[0]; returns W^-T [c, c] (fp32)."""
    C.require_cuda(w, logdet_out, sign_out)
    winv_t = torch.empty((c, c), dtype=torch.float32, device=w.device)
    C.call("dle_wg_logdet_inv", C.ptr(w), C.ptr(logdet_out), C.ptr(winv_t), C.ptr(sign_out), c, C.stream())
    return winv_t


def invconv_bwd(dy, da0, x, w, winv_t, dw_out, scale, logdet_coef, c):
    """dx = diag(I, W)^T (dy [+ da0 on the first c/2 active channels]); dw_out (c*c floats) = sum_rows g x^T - scale *
    logdet_coef * W^-T."""
    C.require_cuda(dy, da0, x, w, winv_t, dw_out, scale)
    m = _state(dy, "invconv_bwd")
    dx = torch.empty_like(dy)
    parts = C.lib().dle_wg_invconv_bwd_partials(m)
    ws = torch.empty((parts, 64), dtype=torch.float32, device=dy.device)
    C.call("dle_wg_invconv_bwd", C.ptr(dy), C.ptr(da0), C.ptr(x), C.ptr(w), C.ptr(winv_t), C.ptr(dx), C.ptr(dw_out),
           C.ptr(scale), float(logdet_coef), C.ptr(ws), m, c, C.stream())
    return dx


def coupling_partials(m):
    return int(C.lib().dle_wg_coupling_partials(m))


def coupling_fwd(y, o, c, logs_partial):
    """z = y with its upper active half replaced by exp(log_s) * y1 + b, (b | log_s) = o[:, :c]; logs_partial: fp32 slots."""
    C.require_cuda(y, o, logs_partial)
    m = _state(y, "coupling_fwd")
    _state(o, "coupling_fwd o")
    z = torch.empty_like(y)
    C.call("dle_wg_coupling_fwd", C.ptr(y), C.ptr(o), C.ptr(z), C.ptr(logs_partial), m, c, C.stream())
    return z


def coupling_bwd(dz, y, o, scale, logs_coef, c, dtype):
    """-> (dy fp32 [M, 8], d_o 16-bit [M, 8])."""
    C.require_cuda(dz, y, o, scale)
    m = _state(dz, "coupling_bwd")
    dy = torch.empty_like(dz)
    d_o = torch.empty((m, 8), dtype=dtype, device=dz.device)
    C.call("dle_wg_coupling_bwd", C.ptr(dz), C.ptr(y), C.ptr(o), C.ptr(dy), C.ptr(d_o), C.ptr(scale), float(logs_coef), m,
           c, C.dt(dtype), C.stream())
    return dy, d_o


def loss(z, logs_partial, logdets, sigma, out=None):
    """WaveGlowLoss: (sum z^2 / (2 sigma^2) - sum logs_partial - M * sum logdets) / (M * 8) -> fp32 [1]."""
    C.require_cuda(z, logs_partial, logdets, out)
    m = _state(z, "loss")
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=z.device)
    ws = torch.empty(coupling_partials(2 * m), dtype=torch.float32, device=z.device)
    C.call("dle_wg_loss", C.ptr(z), C.ptr(logs_partial), logs_partial.numel(), C.ptr(logdets), logdets.numel(), float(sigma),
           m, C.ptr(out), C.ptr(ws), C.stream())
    return out


def dz_init(z, scale, coef):
    C.require_cuda(z, scale)
    m = _state(z, "dz_init")
    dz = torch.empty_like(z)
    C.call("dle_wg_dz_init", C.ptr(z), C.ptr(dz), C.ptr(scale), float(coef), m, C.stream())
    return dz


def weight_norm_fwd(v, g, w16, cip=None):
    """w16 (contiguous 16-bit rows of Kt*Cip) <- g * v / ||v|| re-laid out as [co, tap*Cip + ci]; g None: plain weight."""
    C.require_cuda(v, g, w16)
    co, ci, kt = v.shape
    cip = ci if cip is None else cip
    if not v.is_contiguous() or not w16.is_contiguous() or w16.numel() < co * kt * cip:
        raise ValueError("weight_norm_fwd: bad operand layout")
    C.call("dle_wg_weight_norm_fwd", C.ptr(v), C.ptr(g), C.ptr(w16), co, ci, kt, cip, C.dt(w16), C.stream())
    return w16


def weight_norm_bwd(dw, v, g, dv, dg, cip=None):
    """(dv, dg) from the fp32 GEMM-layout gradient dw [co, Kt*Cip]."""
    C.require_cuda(dw, v, g, dv, dg)
    co, ci, kt = v.shape
    cip = ci if cip is None else cip
    if not dw.is_contiguous() or dw.dtype != torch.float32 or dw.numel() < co * kt * cip:
        raise ValueError("weight_norm_bwd: dw must be contiguous fp32 [co, Kt*Cip]")
    C.call("dle_wg_weight_norm_bwd", C.ptr(dw), C.ptr(v), C.ptr(g), C.ptr(dv), C.ptr(dg), co, ci, kt, cip, C.stream())


def upsample_weight(w, bias, dtype, stride):
    """ConvTranspose1d weight [Cm, Cm, ksize] -> (b16 [stride*Cm, (ksize/stride)*Cm], bias repeated per phase [stride*Cm])."""
    C.require_cuda(w, bias)
    cm, _, ks = w.shape
    b16 = torch.empty((stride * cm, (ks // stride) * cm), dtype=dtype, device=w.device)
    rep = torch.empty(stride * cm, dtype=torch.float32, device=w.device)
    C.call("dle_wg_upsample_weight", C.ptr(w), C.ptr(bias), C.ptr(b16), C.ptr(rep), cm, ks, stride, C.dt(dtype), C.stream())
    return b16, rep


def upsample_weight_bwd(db, dw, stride):
    C.require_cuda(db, dw)
    cm, _, ks = dw.shape
    if not db.is_contiguous() or db.dtype != torch.float32:
        raise ValueError("upsample_weight_bwd: db must be contiguous fp32")
    C.call("dle_wg_upsample_weight_bwd", C.ptr(db), C.ptr(dw), cm, ks, stride, C.stream())


class WeightNormTable:
    """Device table for the one-launch forms (csrc/waveglow.hip WG_WN_FIELDS): entries = dicts with v [Co, Ci, Kt] fp32,
    g (or None), w16, dw (fp32 GEMM-layout gradient or None), dv, dg (or None), cip."""
    FIELDS = 11

    def __init__(self, entries, device):
        import numpy as np
        self.entries = entries
        rows, start = [], 0
        for e in entries:
            co, ci, kt = e["v"].shape
            if not e["v"].is_contiguous():
                raise ValueError("weight-norm table: v must be contiguous")
            ptr = [0 if e.get(k) is None else e[k].data_ptr() for k in ("v", "g", "w16", "dw", "dv", "dg")]
            rows.append([start, co, ci, kt, e.get("cip") or ci] + ptr)
            start += co
        self.n, self.total_rows = len(entries), start
        self.table = torch.from_numpy(np.asarray(rows, dtype=np.int64).reshape(-1)).to(device)


def weight_norm_fwd_batched(tab, dtype):
    C.call("dle_wg_weight_norm_fwd_batched", C.ptr(tab.table), tab.n, tab.total_rows, C.dt(dtype), C.stream())


def weight_norm_bwd_batched(tab):
    C.call("dle_wg_weight_norm_bwd_batched", C.ptr(tab.table), tab.n, tab.total_rows, C.stream())


class LogdetTable:
    """(element offset from the flat parameter buffer, c) of every flow's invertible 1x1 convolution."""

    def __init__(self, offsets_and_c, device):
        import numpy as np
        self.host = list(offsets_and_c)
        self.n = len(self.host)
        self.table = torch.from_numpy(np.asarray(self.host, dtype=np.int64).reshape(-1)).to(device)


def logdet_inv_batched(flat, tab, logdets, winv_t_all, signs):
    """logdets[f], signs[f], winv_t_all[f, :c*c] for every flow in one launch."""
    C.require_cuda(flat, logdets, winv_t_all, signs)
    C.call("dle_wg_logdet_inv_batched", C.ptr(flat), C.ptr(tab.table), C.ptr(logdets), C.ptr(winv_t_all), C.ptr(signs), tab.n,
           C.stream())
