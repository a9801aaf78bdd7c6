"""TEST DOUBLES (tests/ only) for the C-ABI calls the Tacotron2 engine makes, on top of tests/_waveglow_doubles.py: plain-torch
statements of what each entry point computes.  `-m "not gpu"`: `install(monkeypatch)` swaps them in so that the HOST sequencing of
tacotron2/engine.py (layouts, strided views, BPTT bookkeeping) is checked against the reference-generated fixture on the CPU;
`-m gpu`: every new HIP kernel is compared with its double.  The product never imports this file and has no CPU path.
"""
import numpy as np
import torch

from tests import _waveglow_doubles as W

ACT_NONE, ACT_RELU, ACT_RELU_BWD, ACT_ADD, ACT_TANH, ACT_TANH_BWD = 0, 1, 3, 4, 6, 7


def gemm(a, b, m, n, k, a_kc, b_kc, out=None, out_dtype=None, bias=None, act=ACT_NONE, aux=None, mask_src=None, splitk=1,
         accumulate=False, alpha=1.0, lda=None, ldb=None):
    assert aux is None and lda is None and ldb is None
    am = a[:m, :k] if a_kc else a[:k, :m].t()
    bm = b[:n, :k] if b_kc else b[:k, :n].t()
    acc = (am.double() @ bm.double().t()).float() * alpha
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == n
        acc = acc + bias
    if act == ACT_ADD:
        acc = acc + mask_src.float()
    elif act == ACT_RELU:
        acc = torch.relu(acc)
    elif act == ACT_RELU_BWD:
        acc = acc * (mask_src.float() > 0)
    else:
        assert act == ACT_NONE and mask_src is None
    if out is None:
        out = torch.empty((m, n), dtype=out_dtype or a.dtype, device=a.device)
    assert out.shape == (m, n)
    if accumulate:
        acc = acc + out.float()
    out.copy_(acc)
    return out


def cast_rows(x, out_dtype, cols_out=None, out=None):
    r, c = x.shape
    co = c if cols_out is None else cols_out
    if out is None:
        out = torch.zeros((r, co), dtype=out_dtype, device=x.device)
    out[:, :c].copy_(x)
    if co > c:
        out[:, c:].zero_()
    return out


def cast(x, out_dtype, out=None):
    if out is None:
        return x.to(out_dtype)
    out.copy_(x)
    return out


def bn_fwd(x, gamma, beta, running_mean=None, running_var=None, eps=1e-5, momentum=0.1, residual=None, relu=True, out=None,
           want_mask=False):
    assert residual is None and not want_mask
    c = x.shape[-1]
    x2 = x.reshape(-1, c).float()
    mean = x2.mean(0)
    var = x2.var(0, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + eps)
    if running_mean is not None:
        n = x2.shape[0]
        running_mean.mul_(1 - momentum).add_(momentum * mean)
        running_var.mul_(1 - momentum).add_(momentum * var * n / max(n - 1, 1))
    y = (x2 - mean) * rstd * gamma + beta
    if relu:
        y = torch.relu(y)
    y = y.to(x.dtype).view_as(x)
    if out is not None:
        out.copy_(y)
        y = out
    return y, mean, rstd


def bn_fwd_apply(x, mean, rstd, gamma, beta, residual=None, relu=True, want_mask=False):
    """dle_bn_fwd_apply with given statistics (the eval-mode BatchNorm): -> (y, None)."""
    assert residual is None and not want_mask
    y = (x.float() - mean) * rstd * gamma + beta
    return (torch.relu(y) if relu else y).to(x.dtype), None


def mask_rows(x, cols, lengths, b, to, value):
    """dle_t2_mask_rows: rows (b, t) with t >= lengths[b] of x [B*To, >= cols], columns [0, cols) := value (parse_output,
    model.py:648-655)."""
    past = (torch.arange(to)[None, :] >= lengths[:, None]).reshape(b * to)
    x[:, :cols][past] = value


def bn_bwd(dy, y, x, mean, rstd, gamma, dgamma, dbeta, want_skip_grad=False, dx_out=None, relu_mask=None):
    assert relu_mask is None and not want_skip_grad
    c = x.shape[-1]
    g = dy.reshape(-1, c).float()
    if y is not None:
        g = g * (y.reshape(-1, c).float() > 0)
    xh = (x.reshape(-1, c).float() - mean) * rstd
    dbeta.copy_(g.sum(0))
    dgamma.copy_((g * xh).sum(0))
    n = g.shape[0]
    dx = (gamma * rstd) * (g - dbeta / n - xh * dgamma / n)
    dx = dx.to(x.dtype).view_as(x)
    if dx_out is not None:
        dx_out.copy_(dx)
        dx = dx_out
    return dx, None


class Masks:
    """Dropout keep masks of a doubles run, by call order, so that the oracle can replay them site by site."""
    rng = np.random.default_rng(0)
    log = []
    replay = None            # list of keep masks to serve in call order instead of drawing (the masks a HIP run drew)

    @classmethod
    def reset(cls, seed, replay=None):
        cls.rng, cls.log = np.random.default_rng(seed), []
        cls.replay = list(replay) if replay is not None else None


def _pack(keep):
    flat = keep.reshape(-1).to(torch.uint8)
    assert flat.numel() % 8 == 0
    w = (1 << torch.arange(8, dtype=torch.int32)).to(torch.int32)
    return (flat.view(-1, 8).to(torch.int32) * w).sum(1).to(torch.uint8)


def unpack_dropout_mask(mask, shape):
    bits = (mask.to(torch.int32).unsqueeze(1) >> torch.arange(8, dtype=torch.int32)) & 1
    return bits.reshape(shape).bool()


def inv_keep(p):
    thr = min(max(int(p * 65536.0 + 0.5), 0), 65535)                 # csrc/dropout.h make_drop: p quantised to 1/65536
    return 65536.0 / (65536 - thr)


def dropout_fwd(x, p, seed, offset, offset_base=None):
    if Masks.replay is not None:
        keep = Masks.replay.pop(0).reshape(x.shape).bool()
    else:
        keep = torch.from_numpy(Masks.rng.random(tuple(x.shape)) >= p)
    Masks.log.append(keep)
    return (x.float() * keep * inv_keep(p)).to(x.dtype), _pack(keep)


def dropout_bwd(dy, mask, p):
    keep = unpack_dropout_mask(mask, dy.shape)
    return (dy.float() * keep * inv_keep(p)).to(dy.dtype)


def rows_gather(src, idx):
    return src[idx]


def embed_scatter_add_(grad_word, dz, ids):
    grad_word.index_add_(0, ids, dz.float())


def act_bwd(g, src, act):
    assert act == ACT_TANH_BWD
    return (g.float() * (1 - src.float() ** 2)).to(g.dtype)


def relu_bwd(g, y, out=None):
    r = (g.float() * (y.float() > 0)).to(g.dtype)
    if out is None:
        return r
    out.copy_(r)
    return out


def axpby_(x, y, out, a=1.0, b=1.0):
    out.copy_(a * x + (b * y if y is not None else 0))
    return out


def bce_with_logits(logits, target, grad_scale=None, want_grad=True, ld_logits=1):
    x = (logits[:, 0] if logits.dim() == 2 else logits.reshape(-1)).float()             # element i at logits[i * ld_logits]
    loss = torch.nn.functional.binary_cross_entropy_with_logits(x, target).reshape(1)
    s = 1.0 if grad_scale is None else float(grad_scale)
    dl = ((torch.sigmoid(x) - target) * s / target.numel()).to(logits.dtype) if want_grad else None
    return loss, dl


# ---------------------------------------------------------------- csrc/tacotron2.hip
def tanh_fwd(x):
    return torch.tanh(x.float()).to(x.dtype)


def lstm_fwd(gates, c_prev, c_out, h_dsts, keep=None, keep_index=0, p=0.0, live=None, h_prev=None, out_dst=None):
    """torch.nn.LSTMCell pointwise part on gates [B, 4H] (i, f, g, o; biases already added).  The gate activations replace
    `gates` in place (saved for backward).  c_out fp32; h (after dropout: keep bits `keep` starting at element keep_index,
    scale inv_keep(p)) goes to every view of h_dsts.  live (fp32 [B], optional): rows with live == 0 keep their state
    (h_prev, c_prev) and write 0 to out_dst -- the packed-sequence semantics of the encoder."""
    b, h4 = gates.shape
    hh = h4 // 4
    g = gates.float()
    i, f, gg, o = torch.sigmoid(g[:, :hh]), torch.sigmoid(g[:, hh:2 * hh]), torch.tanh(g[:, 2 * hh:3 * hh]), torch.sigmoid(g[:, 3 * hh:])
    gates.copy_(torch.cat([i, f, gg, o], dim=1))
    c = f * c_prev + i * gg
    h = o * torch.tanh(c)
    if keep is not None:
        bits = unpack_dropout_mask(keep, (-1,))[keep_index:keep_index + b * hh].view(b, hh)
        h = h * bits * inv_keep(p)
    if live is not None:
        lv = live.view(b, 1)
        if out_dst is not None:
            out_dst.copy_(h * lv)
        h = lv * h + (1 - lv) * h_prev.float()
        c = lv * c + (1 - lv) * c_prev
    c_out.copy_(c)
    for d in h_dsts:
        d.copy_(h)


def lstm_bwd(dh, dc_next, act, c_prev, dgates, dc_prev, keep=None, keep_index=0, p=0.0, live=None, dh_prev=None, dh_add=()):
    """Backward of lstm_fwd: dh fp32 [B, H] (+ the pieces in dh_add) = gradient wrt the (dropped) h; act = saved activations
    (i, f, g, o); writes dgates (16-bit, may be `act` itself), dc_prev fp32.  live: dead rows pass dh / dc through to dh_prev /
    dc_prev and get zero dgates."""
    b, hh = dh.shape
    for x in dh_add:
        dh = dh + x
    a = act.float()
    i, f, gg, o = a[:, :hh], a[:, hh:2 * hh], a[:, 2 * hh:3 * hh], a[:, 3 * hh:]
    g = dh.clone()
    if keep is not None:
        bits = unpack_dropout_mask(keep, (-1,))[keep_index:keep_index + b * hh].view(b, hh)
        g = g * bits * inv_keep(p)
    tc = torch.tanh(f * c_prev + i * gg)                            # the candidate cell, recomputed (dead rows carried c_prev)
    do = g * tc
    dc = dc_next + g * o * (1 - tc * tc)
    di, df, dg = dc * gg, dc * c_prev, dc * i
    dcp = dc * f
    dga = torch.cat([di * i * (1 - i), df * f * (1 - f), dg * (1 - gg * gg), do * o * (1 - o)], dim=1)
    if live is not None:
        lv = live.view(b, 1)
        dga = dga * lv
        dcp = lv * dcp + (1 - lv) * dc_next
        dh_prev.copy_((1 - lv) * dh)                                # the carried state's gradient (the GEMM adds the live part)
    dgates.copy_(dga)
    dc_prev.copy_(dcp)


def _loc_cols(awc, b, ti, kl):
    """[B, Ti, kl * 2]: cols[b, t, j*2 + c] = awc[b, t + j - kl//2, c] (zero outside the sample), the compact row gather."""
    x = awc.float().view(b, ti, 8)[:, :, :2]
    pad = kl // 2
    xp = torch.zeros(b, ti + 2 * pad, 2)
    xp[:, pad:pad + ti] = x
    return torch.stack([xp[:, j:j + ti] for j in range(kl)], dim=2).reshape(b, ti, kl * 2)


def attention_fwd(q, pl, v, memory, lengths, awc_prev, tanh_out, aw_out, awc_next, ctx_dsts, wloc=None, kl=0):
    """Location-sensitive attention of one decoder step (tacotron2/model.py:79-121): e = v . tanh(q + pl), masked softmax over the
    text positions, context = weights x memory.  q fp32 [B, A]; pl 16-bit [B*Ti, A] (processed memory + location term); v fp32 [A];
    memory 16-bit [B*Ti, E]; lengths int64 [B]; awc_prev / awc_next 16-bit [B*Ti, 8] = (previous weights, cumulative weights, 0 ...):
    the next step's location-convolution input; tanh_out 16-bit [B*Ti, A] saved; aw_out fp32 [B, Ti]; context (16-bit) to ctx_dsts."""
    b, a = q.shape
    ti = pl.shape[0] // b
    plf = pl.float().view(b, ti, a)
    if wloc is not None:                                             # pl = processed memory alone; + the location term
        if awc_prev is not None:
            plf = plf + (_loc_cols(awc_prev, b, ti, kl).double() @ wloc.double()[:, :kl * 2].t()).float()
    th = torch.tanh(q.view(b, 1, a) + plf)
    tanh_out.copy_(th.view(b * ti, a))
    e = (tanh_out.float().view(b, ti, a) * v.view(1, 1, a)).sum(2)
    pad = torch.arange(ti)[None, :] >= lengths[:, None]
    aw = torch.softmax(e.masked_fill(pad, -float("inf")), dim=1)
    aw_out.copy_(aw)
    ctx = torch.bmm(aw.unsqueeze(1), memory.float().view(b, ti, -1)).squeeze(1)
    for d in ctx_dsts:
        d.copy_(ctx)
    nxt = torch.zeros((b, ti, 8), dtype=torch.float32)
    nxt[:, :, 0] = aw
    nxt[:, :, 1] = awc_prev.float().view(b, ti, 8)[:, :, 1] + aw if awc_prev is not None else aw
    awc_next.copy_(nxt.view(b * ti, 8))


def attention_bwd(d_ctx, d_aw_in, aw, tanh_out, v, memory, d_memory, d_pl, dq, dv_acc, d_pm_acc, d_ctx_add=(), d_aw_add=None,
                  dq16=None, dctx16=None, wloc_t=None, kl=0, d_prev=None, d_cum=None):
    """Backward of attention_fwd.  d_ctx fp32 [B, E] (+ the pieces in d_ctx_add); d_aw_in (+ d_aw_add) fp32 [B, Ti] (gradient
    reaching the weights through the location input / cumulative weights of later steps); accumulates d_memory fp32 [B*Ti, E],
    dv_acc fp32 [B, A] (per-sample partial sums), d_pm_acc fp32 [B*Ti, A]; writes d_pl 16-bit [B*Ti, A] (gradient of q + pl inside
    the tanh), dq fp32 [B, A] (its sum over the text positions) and / or its 16-bit copy dq16, and the summed context gradient in
    16 bits (dctx16) when asked."""
    b, ti = aw.shape
    a = v.numel()
    for x in d_ctx_add:
        d_ctx = d_ctx + x
    if d_aw_add is not None:
        d_aw_in = d_aw_in + d_aw_add
    if dctx16 is not None:
        dctx16.copy_(d_ctx)
    mem = memory.float().view(b, ti, -1)
    if d_memory is not None:
        d_memory.add_((aw.unsqueeze(2) * d_ctx.unsqueeze(1)).reshape(b * ti, -1))
    d_aw = (mem * d_ctx.unsqueeze(1)).sum(2) + d_aw_in
    d_e = aw * (d_aw - (aw * d_aw).sum(1, keepdim=True))
    th = tanh_out.float().view(b, ti, a)
    dv_acc.add_((d_e.unsqueeze(2) * th).sum(1))
    d_pre = d_e.unsqueeze(2) * v.view(1, 1, a) * (1 - th * th)
    d_pl.copy_(d_pre.view(b * ti, a))
    if dq is not None:
        dq.copy_(d_pre.sum(1))
    if dq16 is not None:
        dq16.copy_(d_pre.sum(1))
    if d_pm_acc is not None:
        d_pm_acc.add_(d_pre.view(b * ti, a))                           # the fp32 value, not its 16-bit rounding
    if wloc_t is not None:
        # transposed location convolution on the ROUNDED d_pl (what the kernel keeps in LDS): dcol = d_pl x W_loc, then the
        # anti-diagonal sums d weights[s][c] = sum_j dcol[s - j + pad][j*2 + c]
        dcol = (d_pl.double().view(b, ti, a) @ wloc_t.double()[:kl * 2].t()).float().view(b, ti, kl, 2)
        pad = kl // 2
        out = torch.zeros(b, ti, 2)
        for j in range(kl):
            lo, hi = max(0, pad - j), min(ti, ti + pad - j)
            if lo < hi:
                out[:, lo + j - pad:hi + j - pad] += dcol[:, lo:hi, j]
        d_prev.copy_(out[:, :, 0])
        d_cum.add_(out[:, :, 1])


def sum_steps(x, out):
    out.add_(x.double().sum(0).float().view(out.shape))


def location_bwd(dcol, d_prev, d_cum, b, ti, kl):
    """Transpose of the row gather col[(b, t), j*8 + c] = awc[b, t + j - kl//2, c] for the two live channels: d_prev (c = 0) is
    written, d_cum (c = 1) accumulated; fp32 [B, Ti]."""
    d = dcol.float().view(b, ti, kl, 8)
    pad = kl // 2
    out = torch.zeros(b, ti, 2)
    for j in range(kl):
        lo, hi = max(0, pad - j), min(ti, ti + pad - j)               # rows t with 0 <= t + j - pad < Ti
        if lo < hi:
            out[:, lo + j - pad:hi + j - pad] += d[:, lo:hi, j, :2]
    d_prev.copy_(out[:, :, 0])
    d_cum.add_(out[:, :, 1])


def transpose_cast(x, out_dtype, out=None):
    y = x.t().to(out_dtype)
    if out is None:
        return y.contiguous()
    out.copy_(y)
    return out


def mel_loss(out_all, post, target, n_mel, scale, d_out, d_post):
    """Tacotron2Loss mel terms (loss_function.py:42-44): MSE(mel_out, target) + MSE(mel_out + post, target), mean over all
    elements.  out_all fp32 [R, ld] (columns < n_mel = mel_out); post 16-bit [R, n_mel]; target fp32 [R, n_mel].  Writes
    d_out (16-bit [R, ld]: columns < n_mel = scale * d loss / d mel_out, the rest untouched) and d_post (16-bit [R, n_mel])."""
    mo = out_all[:, :n_mel]
    mp = mo + post.float()
    n = target.numel()
    loss = ((mo - target) ** 2).sum() / n + ((mp - target) ** 2).sum() / n
    s = float(scale)
    g2 = 2.0 * (mp - target) / n * s
    d_post.copy_(g2)
    d_out[:, :n_mel].copy_(2.0 * (mo - target) / n * s + g2)
    return loss.reshape(1)


def lstm_gemm_fwd(x, w, bias, addend, c_prev, c_out, gates, h_dsts, keep=None, keep_index=0, p=0.0):
    """dle_t2_lstm_gemm_fwd = the gates product (rounded to the storage type like a dle_gemm output) followed by lstm_fwd."""
    pre = x.double() @ w.double().t()
    if bias is not None:
        pre = pre + bias.double()
    if addend is not None:
        pre = pre + addend.double()
    gates.copy_(pre.to(gates.dtype))
    lstm_fwd(gates, c_prev, c_out, h_dsts, keep=keep, keep_index=keep_index, p=p)


def install(monkeypatch):
    from deeplearningexamples_amd import _cabi as C
    from deeplearningexamples_amd import functional as F
    from deeplearningexamples_amd.tacotron2 import engine, ops
    from deeplearningexamples_amd.waveglow import ops as wops
    import types
    me = globals()
    monkeypatch.setattr(C, "require_cuda", lambda *a: None)
    for name in ("gemm", "cast_rows", "cast", "bn_fwd", "bn_bwd", "dropout_fwd", "dropout_bwd", "rows_gather", "embed_scatter_add_",
                 "act_bwd", "bce_with_logits", "relu_bwd", "axpby_", "transpose_cast", "bn_fwd_apply"):
        monkeypatch.setattr(F, name, me[name])
    for name in ("colsum", "copy_rows", "check_nonfinite_", "amp_update_scale_", "gemm_batched"):
        monkeypatch.setattr(F, name, getattr(W, name))
    for name in ("taps", "taps_bwd", "weight_norm_fwd", "weight_norm_bwd"):
        monkeypatch.setattr(wops, name, getattr(W, name))
    for name in ("tanh_fwd", "lstm_fwd", "lstm_gemm_fwd", "lstm_bwd", "attention_fwd", "attention_bwd", "location_bwd", "sum_steps", "mel_loss", "inv_keep", "mask_rows"):
        monkeypatch.setattr(ops, name, me[name])
    fake_mt = types.SimpleNamespace(TableCache=W.TableCache, streaming_chunk=W.streaming_chunk, l2norm=W.l2norm, adam=W.adam)
    monkeypatch.setattr(engine, "mt", fake_mt)
