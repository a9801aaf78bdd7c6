"""Tensor-level wrappers of the Tacotron2 entry points of the C ABI (include/dle_mi355x.h, csrc/tacotron2.hip): allocation and
argument marshalling only; no CPU path.  Reference pieces they stand in for (SpeechSynthesis/Tacotron2/tacotron2/): the pointwise
part of nn.LSTM / nn.LSTMCell with the dropout that follows it (model.py:205-214,425-444), the location-sensitive attention of one
decoder step (model.py:79-121), the mel terms of Tacotron2Loss (loss_function.py:42-44), torch.tanh of the postnet (model.py:170).
"""
import torch

from .. import _cabi as C


def inv_keep(p):
    """Scale of a kept element: the drop probability is quantised to 1/65536 (csrc/dropout.h make_drop)."""
    thr = min(max(int(p * 65536.0 + 0.5), 0), 65535)
    return 65536.0 / (65536 - thr)


def _ld(t, what):
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError("%s: expected a 2-D view with unit inner stride" % what)
    return t.stride(0) if t.shape[0] > 1 else t.shape[1]


def tanh_fwd(x):
    C.require_cuda(x)
    if not x.is_contiguous():
        raise ValueError("tanh_fwd: contiguous input")
    y = torch.empty_like(x)
    C.call("dle_t2_tanh_fwd", C.ptr(x), C.ptr(y), x.numel(), C.dt(x), C.stream())
    return y


def lstm_fwd(gates, c_prev, c_out, h_dsts, keep=None, keep_index=0, p=0.0, live=None, h_prev=None, out_dst=None):
    """Pointwise part of nn.LSTM / nn.LSTMCell + the dropout on the hidden state (statement: include/dle_mi355x.h,
    dle_t2_lstm_fwd; reference tacotron2/model.py:205-214,425-444).  gates: 16-bit [B, 4H] row-strided view, replaced by the gate
    activations; h_dsts: up to three 16-bit [B, H] row-strided views."""
    C.require_cuda(gates, c_prev, c_out, keep, live, h_prev, out_dst, *h_dsts)
    b, h4 = gates.shape
    hh = h4 // 4
    if len(h_dsts) > 3:
        raise ValueError("lstm_fwd: at most three destinations")
    d = list(h_dsts) + [None] * (3 - len(h_dsts))
    C.call("dle_t2_lstm_fwd", C.ptr(gates), _ld(gates, "gates"), C.ptr(c_prev), C.ptr(c_out), C.ptr(d[0]),
           _ld(d[0], "h dst") if d[0] is not None else 0, C.ptr(d[1]), _ld(d[1], "h dst") if d[1] is not None else 0, C.ptr(d[2]),
           _ld(d[2], "h dst") if d[2] is not None else 0, C.ptr(keep), int(keep_index), float(inv_keep(p) if keep is not None else 1.0),
           C.ptr(live), C.ptr(h_prev), _ld(h_prev, "h_prev") if h_prev is not None else 0, C.ptr(out_dst),
           _ld(out_dst, "out_dst") if out_dst is not None else 0, b, hh, C.dt(gates), C.stream())


def lstm_gemm_fwd(x, w, bias, addend, c_prev, c_out, gates, h_dsts, keep=None, keep_index=0, p=0.0):
    """One LSTMCell step in one launch: gates = x w^T (+ bias) (+ addend) -> lstm_fwd (statement: include/dle_mi355x.h,
    dle_t2_lstm_gemm_fwd).  x 16-bit [B, K] row-strided, w 16-bit [4H, K], gates 16-bit [B, 4H] row-strided (receives the gate
    activations), addend 16-bit [B, 4H] with the pitch of `gates`; h_dsts: up to three 16-bit [B, H] row-strided views."""
    C.require_cuda(x, w, bias, addend, c_prev, c_out, gates, keep, *h_dsts)
    b, k = x.shape
    h4 = w.shape[0]
    hh = h4 // 4
    if len(h_dsts) > 3:
        raise ValueError("lstm_gemm_fwd: at most three destinations")
    if addend is not None and _ld(addend, "addend") != _ld(gates, "gates"):
        raise ValueError("lstm_gemm_fwd: the addend shares the row pitch of the gates")
    d = list(h_dsts) + [None] * (3 - len(h_dsts))
    C.call("dle_t2_lstm_gemm_fwd", C.ptr(x), _ld(x, "x"), C.ptr(w), _ld(w, "w"), C.ptr(bias), C.ptr(addend), C.ptr(c_prev),
           C.ptr(c_out), C.ptr(gates), _ld(gates, "gates"), C.ptr(d[0]), _ld(d[0], "h dst") if d[0] is not None else 0, C.ptr(d[1]),
           _ld(d[1], "h dst") if d[1] is not None else 0, C.ptr(d[2]), _ld(d[2], "h dst") if d[2] is not None else 0, C.ptr(keep),
           int(keep_index), float(inv_keep(p) if keep is not None else 1.0), b, hh, k, C.dt(x), C.stream())


def lstm_bwd(dh, dc_next, act, c_prev, dgates, dc_prev, keep=None, keep_index=0, p=0.0, live=None, dh_prev=None, dh_add=()):
    """dh_add: up to two more fp32 [B, H] row-strided pieces of the hidden state's gradient, summed on load."""
    C.require_cuda(dh, dc_next, act, c_prev, dgates, dc_prev, keep, live, dh_prev, *dh_add)
    b, hh = dh.shape
    if len(dh_add) > 2:
        raise ValueError("lstm_bwd: at most two extra gradient pieces")
    x = list(dh_add) + [None] * (2 - len(dh_add))
    C.call("dle_t2_lstm_bwd", C.ptr(dh), _ld(dh, "dh"), C.ptr(x[0]), _ld(x[0], "dh") if x[0] is not None else 0, C.ptr(x[1]),
           _ld(x[1], "dh") if x[1] is not None else 0, C.ptr(dc_next), C.ptr(act), _ld(act, "act"), C.ptr(c_prev), C.ptr(dgates),
           _ld(dgates, "dgates"), C.ptr(dc_prev), C.ptr(keep), int(keep_index), float(inv_keep(p) if keep is not None else 1.0),
           C.ptr(live), C.ptr(dh_prev), b, hh, C.dt(act), C.stream())


def attention_fwd(q, pl, v, memory, lengths, awc_prev, tanh_out, aw_out, awc_next, ctx_dsts, wloc=None, kl=0):
    """wloc (16-bit [A, KK], k = tap * 2 + channel): `pl` is the processed memory alone and the location term is formed inside."""
    C.require_cuda(q, pl, v, memory, lengths, awc_prev, tanh_out, aw_out, awc_next, wloc, *ctx_dsts)
    b, a = q.shape
    ti = pl.shape[0] // b
    e = memory.shape[1]
    if len(ctx_dsts) > 3:
        raise ValueError("attention_fwd: at most three destinations")
    d = list(ctx_dsts) + [None] * (3 - len(ctx_dsts))
    C.call("dle_t2_attention_fwd", C.ptr(q), C.ptr(pl), C.ptr(v), C.ptr(memory), C.ptr(lengths), C.ptr(awc_prev), C.ptr(tanh_out),
           C.ptr(aw_out), C.ptr(awc_next), C.ptr(d[0]), _ld(d[0], "ctx dst") if d[0] is not None else 0, C.ptr(d[1]),
           _ld(d[1], "ctx dst") if d[1] is not None else 0, C.ptr(d[2]), _ld(d[2], "ctx dst") if d[2] is not None else 0,
           C.ptr(wloc), int(kl), wloc.shape[1] if wloc is not None else 0, b, ti, a, e, C.dt(pl), C.stream())


def attention_bwd(d_ctx, d_aw_in, aw, tanh_out, v, memory, d_memory, d_pl, dq, dv_acc, d_pm_acc, d_ctx_add=(), d_aw_add=None,
                  dq16=None, dctx16=None, wloc_t=None, kl=0, d_prev=None, d_cum=None):
    """dv_acc fp32 [B, A] (per-sample partials); d_ctx_add: up to two more fp32 [B, E] row-strided pieces of the context gradient,
    d_aw_add one more [B, Ti] piece of the weights' gradient (summed on load); dq16 / dctx16: 16-bit copies of dq / of the summed
    context gradient (operands of the products that consume them).  See include/dle_mi355x.h."""
    C.require_cuda(d_ctx, d_aw_in, aw, tanh_out, v, memory, d_memory, d_pl, dq, dv_acc, d_pm_acc, d_aw_add, dq16, dctx16, wloc_t, d_prev,
                   d_cum, *d_ctx_add)
    b, ti = aw.shape
    if len(d_ctx_add) > 2:
        raise ValueError("attention_bwd: at most two extra context-gradient pieces")
    x = list(d_ctx_add) + [None] * (2 - len(d_ctx_add))
    if dv_acc.shape != (b, v.numel()):
        raise ValueError("attention_bwd: dv_acc holds one row of partial sums per sample, [B, A]")
    for t in (d_aw_in, d_aw_add):
        if t is not None and not t.is_contiguous():
            raise ValueError("attention_bwd: contiguous [B, Ti] weight gradients")
    C.call("dle_t2_attention_bwd", C.ptr(d_ctx), _ld(d_ctx, "d_ctx"), C.ptr(x[0]), _ld(x[0], "d_ctx") if x[0] is not None else 0,
           C.ptr(x[1]), _ld(x[1], "d_ctx") if x[1] is not None else 0, C.ptr(d_aw_in), C.ptr(d_aw_add), C.ptr(aw), C.ptr(tanh_out),
           C.ptr(v), C.ptr(memory), C.ptr(d_memory), C.ptr(d_pl), C.ptr(dq), C.ptr(dq16), C.ptr(dctx16), C.ptr(dv_acc),
           C.ptr(d_pm_acc), C.ptr(wloc_t), int(kl), wloc_t.shape[0] if wloc_t is not None else 0, C.ptr(d_prev), C.ptr(d_cum),
           b, ti, v.numel(), memory.shape[1], C.dt(tanh_out), C.stream())


def sum_steps(x, out):
    """out (fp32, flat) += sum over the leading dimension of x (16-bit [n, ...], contiguous)."""
    C.require_cuda(x, out)
    n = x.shape[0]
    r = x.numel() // n
    if not x.is_contiguous() or not out.is_contiguous() or out.numel() != r or out.dtype != torch.float32:
        raise ValueError("sum_steps: contiguous x [n, R] and fp32 out [R]")
    C.call("dle_t2_sum_steps", C.ptr(x), C.ptr(out), n, r, C.dt(x), C.stream())


def location_bwd(dcol, d_prev, d_cum, b, ti, kl):
    """dcol 16-bit [B*Ti, KL*8] -> d_prev fp32 [B, Ti] (written), d_cum fp32 [B, Ti] (accumulated)."""
    C.require_cuda(dcol, d_prev, d_cum)
    if not (dcol.is_contiguous() and d_prev.is_contiguous() and d_cum.is_contiguous()) or dcol.shape[1] != kl * 8:
        raise ValueError("location_bwd: contiguous dcol [B*Ti, KL*8], d_prev / d_cum [B, Ti]")
    C.call("dle_t2_location_bwd", C.ptr(dcol), C.ptr(d_prev), C.ptr(d_cum), b, ti, kl, C.dt(dcol), C.stream())


def mel_loss(out_all, post, target, n_mel, scale, d_out, d_post):
    C.require_cuda(out_all, post, target, scale, d_out, d_post)
    r = out_all.shape[0]
    loss = torch.empty(1, dtype=torch.float32, device=out_all.device)
    ws = torch.empty(1024, dtype=torch.float32, device=out_all.device)
    C.call("dle_t2_mel_loss", C.ptr(out_all), _ld(out_all, "out_all"), C.ptr(post), C.ptr(target), C.ptr(scale), C.ptr(d_out),
           _ld(d_out, "d_out"), C.ptr(d_post), C.ptr(loss), C.ptr(ws), r, n_mel, C.dt(post), C.stream())
    return loss


def mask_rows(x, cols, lengths, b, to, value):
    """--mask-padding: x [B*To, >= cols] (fp32 or 16-bit, row pitch from its stride) rows (b, t) with t >= lengths[b], columns
    [0, cols) := value, in place."""
    C.require_cuda(x, lengths)
    if lengths.dtype != torch.int64 or lengths.numel() != b or x.shape[0] != b * to or x.shape[1] < cols or x.stride(1) != 1:
        raise ValueError("mask_rows: x [B*To, >= cols] with unit column stride, int64 lengths [B]")
    C.call("dle_t2_mask_rows", C.ptr(x), x.stride(0), cols, C.ptr(lengths), b, to, float(value), C.dt(x), C.stream())
