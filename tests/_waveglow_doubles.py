"""TEST DOUBLES (tests/ only) for the C-ABI calls the WaveGlow engine makes: plain-torch statements of what each entry point
computes, used two ways --
  * `-m "not gpu"`: `install(monkeypatch)` swaps them in for the wrappers so the HOST sequencing of waveglow/engine.py (layouts,
    slices, which gradient lands where) can be checked against the reference-generated fixture without a GPU;
  * `-m gpu`: each HIP kernel is compared with its double on the same inputs (tests/test_gpu_waveglow.py).
The product never imports this file and has no such path: deeplearningexamples_amd raises on CPU tensors / a missing library.
"""
import math

import torch

ACT_NONE, ACT_ADD = 0, 4


def gemm(a, b, m, n, k, a_kc, b_kc, out=None, out_dtype=None, bias=None, act=ACT_NONE, aux=None, mask_src=None, splitk=1,
         accumulate=False, alpha=1.0, lda=None, ldb=None):
    """dle_gemm: C[m, n] = act(alpha * A(m, k) B(n, k) + bias); a_kc / b_kc: operand stored [rows][k] or [k][rows]."""
    assert aux is None and lda is None and ldb is None
    am = a[:m, :k] if a_kc else a[:k, :m].t()
    bm = b[:n, :k] if b_kc else b[:k, :n].t()
    acc = (am.double() @ bm.double().t()).float() * alpha
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == n
        acc = acc + bias
    if act == ACT_ADD:
        assert mask_src.shape == (m, n)
        acc = acc + mask_src.float()
    else:
        assert act == ACT_NONE and mask_src is None
    if out is None:
        out = torch.empty((m, n), dtype=out_dtype or a.dtype, device=a.device)
    assert out.shape == (m, n)
    if accumulate:
        acc = acc + out.float()
    out.copy_(acc)
    return out


def colsum(x, out=None, accumulate=False):
    s = x.float().sum(0)
    if out is None:
        return s
    out.view(-1)[:s.numel()].copy_(s + (out.view(-1)[:s.numel()] if accumulate else 0))
    return out


def copy_rows(src, dst):
    dst.copy_(src)
    return dst


def nchw_to_nhwc(x, out_dtype, c_padded=None):
    n, c, h, w = x.shape
    assert c_padded in (None, c)
    return x.permute(0, 2, 3, 1).contiguous().to(out_dtype)


def check_nonfinite_(x, found_inf):
    if not bool(torch.isfinite(x).all()):
        found_inf.fill_(1.0)


def amp_update_scale_(scale, growth_tracker, found_inf, inv_scale=None, growth_factor=2.0, backoff_factor=0.5,
                      growth_interval=2000, clear_found_inf=True):
    if float(found_inf) != 0:
        scale.mul_(backoff_factor)
        growth_tracker.zero_()
    else:
        growth_tracker.add_(1)
        if int(growth_tracker) >= growth_interval:
            scale.mul_(growth_factor)
            growth_tracker.zero_()
    if inv_scale is not None:
        inv_scale.copy_(1.0 / scale)
    if clear_found_inf:
        found_inf.zero_()


# ---------------------------------------------------------------- csrc/waveglow.hip
def taps(x, batch, steps, ntaps, dilation, left, out=None):
    ch = x.shape[1]
    xb = x.reshape(batch, steps, ch)
    col = torch.zeros((batch, steps, ntaps, ch), dtype=x.dtype, device=x.device)
    for k in range(ntaps):
        sh = (k - left) * dilation                       # col[:, t, k] = x[:, t + sh]
        lo, hi = max(0, -sh), min(steps, steps - sh)
        if hi > lo:
            col[:, lo:hi, k] = xb[:, lo + sh:hi + sh]
    col = col.view(batch * steps, ntaps * ch)
    if out is not None:
        out.copy_(col)
        return out
    return col


def taps_bwd(dcol, batch, steps, ch, ntaps, dilation, left, out, addend=None):
    d = dcol.view(batch, steps, ntaps, ch).float()
    acc = torch.zeros((batch, steps, ch), dtype=torch.float32, device=dcol.device)
    if addend is not None:
        acc += addend.float().reshape(batch, steps, ch)
    for k in range(ntaps):
        sh = (k - left) * dilation                       # dx[:, t] += dcol[:, t - sh, k]
        lo, hi = max(0, sh), min(steps, steps + sh)
        if hi > lo:
            acc[:, lo:hi] += d[:, lo - sh:hi - sh, k]
    out.copy_(acc.view(batch * steps, ch))
    return out


def gate_fwd(s, nc, out=None):
    r = torch.tanh(s[:, :nc].float()) * torch.sigmoid(s[:, nc:2 * nc].float())
    if out is None:
        return r.to(s.dtype)
    out.copy_(r)
    return out


def gate_bwd(dacts, s, ds):
    nc = dacts.shape[1]
    th, sg, g = torch.tanh(s[:, :nc].float()), torch.sigmoid(s[:, nc:2 * nc].float()), dacts.float()
    ds[:, :nc].copy_(g * sg * (1 - th * th))
    ds[:, nc:2 * nc].copy_(g * th * sg * (1 - sg))
    return ds


def _w8(w, c):
    w8 = torch.eye(8, dtype=torch.float32, device=w.device)
    w8[8 - c:, 8 - c:] = w.reshape(c, c)
    return w8


def invconv_fwd(x, w, c, dtype):
    y = x @ _w8(w, c).t()
    a0 = torch.zeros((x.shape[0], 8), dtype=dtype, device=x.device)
    a0[:, :c // 2] = y[:, 8 - c:8 - c + c // 2].to(dtype)
    return y, a0


def logdet_inv(w, c, logdet_out, sign_out):
    wm = w.reshape(c, c).double()
    sign, ld = torch.linalg.slogdet(wm)
    logdet_out.fill_(float(ld))
    sign_out.fill_(float(sign))
    return torch.linalg.inv(wm).t().contiguous().float()


def invconv_bwd(dy, da0, x, w, winv_t, dw_out, scale, logdet_coef, c):
    off, nh = 8 - c, c // 2
    g = dy.clone()
    if da0 is not None:
        g[:, off:off + nh] += da0[:, :nh]
    dx = g @ _w8(w, c)
    dw = (g[:, off:].double().t() @ x[:, off:].double()).float() - float(scale) * logdet_coef * winv_t.reshape(c, c)
    dw_out.view(-1)[:c * c].copy_(dw.reshape(-1))
    return dx


def coupling_partials(m):
    return 4


def coupling_fwd(y, o, c, logs_partial):
    off, nh = 8 - c, c // 2
    z = y.clone()
    log_s, bb = o[:, nh:2 * nh], o[:, :nh]
    z[:, off + nh:] = torch.exp(log_s) * y[:, off + nh:] + bb
    logs_partial.zero_()
    logs_partial.view(-1)[0] = log_s.double().sum().float()
    return z


def coupling_bwd(dz, y, o, scale, logs_coef, c, dtype):
    off, nh = 8 - c, c // 2
    es = torch.exp(o[:, nh:2 * nh])
    g1 = dz[:, off + nh:]
    dy = dz.clone()
    dy[:, off + nh:] = g1 * es
    d_o = torch.zeros((dz.shape[0], 8), dtype=torch.float32, device=dz.device)
    d_o[:, :nh] = g1
    d_o[:, nh:2 * nh] = g1 * y[:, off + nh:] * es - float(scale) * logs_coef
    return dy, d_o.to(dtype)


def loss(z, logs_partial, logdets, sigma, out=None):
    m = z.shape[0]
    v = ((z.double() ** 2).sum() / (2 * sigma * sigma) - logs_partial.double().sum() - m * logdets.double().sum()) / (m * 8)
    r = torch.tensor([float(v)], dtype=torch.float32, device=z.device)
    if out is not None:
        out.copy_(r)
        return out
    return r


def dz_init(z, scale, coef):
    return z * (float(scale) * coef)


def weight_norm_fwd(v, g, w16, cip=None):
    co, ci, kt = v.shape
    cip = ci if cip is None else cip
    w = v if g is None else v * (g.reshape(co, 1, 1) / v.flatten(1).norm(dim=1).view(co, 1, 1))
    lay = torch.zeros((co, kt, cip), dtype=torch.float32, device=v.device)
    lay[:, :, :ci] = w.permute(0, 2, 1)
    w16.view(-1)[:co * kt * cip].copy_(lay.reshape(-1))
    return w16


def weight_norm_bwd(dw, v, g, dv, dg, cip=None):
    co, ci, kt = v.shape
    cip = ci if cip is None else cip
    d = dw.reshape(-1)[:co * kt * cip].view(co, kt, cip)[:, :, :ci].permute(0, 2, 1)      # -> [co, ci, kt]
    if g is None:
        dv.copy_(d)
        return
    nrm = v.flatten(1).norm(dim=1).view(co, 1, 1)
    dot = (d * v).flatten(1).sum(1).view(co, 1, 1)
    dg.copy_((dot / nrm).view_as(dg))
    dv.copy_(g.reshape(co, 1, 1) / nrm * (d - v * dot / (nrm * nrm)))


def upsample_weight(w, bias, dtype, stride):
    cm, _, ks = w.shape
    nt = ks // stride
    # b16[(r, co), (j, ci)] = w[ci, co, r + stride * j]
    b = w.view(cm, cm, nt, stride).permute(3, 1, 2, 0).reshape(stride * cm, nt * cm)
    return b.to(dtype).contiguous(), bias.repeat(stride).contiguous()


def upsample_weight_bwd(db, dw, stride):
    cm, _, ks = dw.shape
    nt = ks // stride
    dw.copy_(db.view(stride, cm, nt, cm).permute(3, 1, 2, 0).reshape(cm, cm, ks))


def weight_norm_fwd_batched(tab, dtype):
    for e in tab.entries:
        weight_norm_fwd(e["v"], e.get("g"), e["w16"], e.get("cip"))


def weight_norm_bwd_batched(tab):
    for e in tab.entries:
        weight_norm_bwd(e["dw"], e["v"], e.get("g"), e["dv"], e.get("dg"), e.get("cip"))


def logdet_inv_batched(flat, tab, logdets, winv_t_all, signs):
    for f, (off, c) in enumerate(tab.host):
        w = flat[off:off + c * c]
        winv_t_all[f, :c * c] = logdet_inv(w, c, logdets[f:f + 1], signs[f:f + 1]).reshape(-1)


def gemm_batched(a, b, c, m, n, k, lda, ldb, ldc, a_kc, b_kc, batch, batch_inner, sa, sb, sc, alpha=1.0):
    """dle_gemm_batched on flat storage: slice z = (zo, zi) of X starts at X + zo * sx[0] + zi * sx[1] (elements)."""

    def view(t, off, rows, cols, ld):
        return torch.as_strided(t, (rows, cols), (ld, 1), t.storage_offset() + off)
    for z in range(batch):
        zo, zi = divmod(z, batch_inner)
        am = view(a, zo * sa[0] + zi * sa[1], m if a_kc else k, k if a_kc else m, lda)
        bm = view(b, zo * sb[0] + zi * sb[1], n if b_kc else k, k if b_kc else n, ldb)
        cm = view(c, zo * sc[0] + zi * sc[1], m, n, ldc)
        am = am if a_kc else am.t()
        bm = bm if b_kc else bm.t()
        cm.copy_((am.double() @ bm.double().t()).float() * alpha)
    return c


def colsum_batched(table, m, n, ld, dtype):
    """dle_colsum_batched: fp32 column sums of every [m, n] matrix of the table into its destination."""
    for x, out in table.entries:
        assert x.shape == (m, n) and ld == n
        out.copy_(x.double().sum(0).float())


# ---------------------------------------------------------------- csrc/multi_tensor.hip (as the engine uses it)
class _Table:
    def __init__(self, lists, chunk):
        self.lists, self.chunk = lists, chunk


class TableCache:
    def get(self, tag, lists, chunk=65536):
        return _Table(lists, chunk)


def streaming_chunk(lists, want_blocks=1024, lo=2048):
    return 2048


def l2norm(table, noop_flag=None, per_tensor=False):
    tot = math.sqrt(sum(float((t.double() ** 2).sum()) for t in table.lists[0]))
    return torch.tensor([tot], dtype=torch.float32), torch.empty(0)


def adam(table, lr, beta1, beta2, eps, weight_decay, step, skip_flag=None, inv_scale=None, grad_norm=None, max_grad_norm=0.0):
    """GradScaler.unscale_ + clip_grad_norm_ + torch.optim.Adam.step on the flat lists (dle_mt_adam)."""
    if skip_flag is not None and float(skip_flag) != 0:
        return
    lr = float(lr)
    is_ = 1.0 if inv_scale is None else float(inv_scale)
    gs = is_
    if grad_norm is not None and max_grad_norm > 0:
        coef = max_grad_norm / (float(grad_norm) * is_ + 1e-6)
        if coef < 1:
            gs = is_ * coef
    t = int(step)
    bc1, bc2 = 1 - beta1 ** t, 1 - beta2 ** t
    for g, p, m, v in zip(*table.lists):
        gr = g * gs + weight_decay * p
        m.mul_(beta1).add_(gr, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(gr, gr, value=1 - beta2)
        p.sub_((lr / bc1) * m / (v.sqrt() / math.sqrt(bc2) + eps))


def install(monkeypatch):
    """Swap the doubles in for the wrappers the WaveGlow engine calls (CPU tests of the host sequencing only)."""
    import types
    from deeplearningexamples_amd import _cabi as C
    from deeplearningexamples_amd import functional as F
    from deeplearningexamples_amd.waveglow import engine, ops
    me = globals()
    monkeypatch.setattr(C, "require_cuda", lambda *a: None)
    for name in ("gemm", "gemm_batched", "colsum", "colsum_batched", "copy_rows", "nchw_to_nhwc", "check_nonfinite_", "amp_update_scale_"):
        monkeypatch.setattr(F, name, me[name])
    for name in ("taps", "taps_bwd", "gate_fwd", "gate_bwd", "invconv_fwd", "logdet_inv", "invconv_bwd", "coupling_partials",
                 "coupling_fwd", "coupling_bwd", "loss", "dz_init", "weight_norm_fwd", "weight_norm_bwd", "upsample_weight",
                 "upsample_weight_bwd", "weight_norm_fwd_batched", "weight_norm_bwd_batched", "logdet_inv_batched"):
        monkeypatch.setattr(ops, name, me[name])
    fake_mt = types.SimpleNamespace(TableCache=TableCache, streaming_chunk=streaming_chunk, l2norm=l2norm, adam=adam)
    monkeypatch.setattr(engine, "mt", fake_mt)
