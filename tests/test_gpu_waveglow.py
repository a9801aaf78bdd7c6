"""WaveGlow train step on the MI355X: every kernel of csrc/waveglow.hip and dle_mt_adam against the
plain-torch statement of the same entry point (tests/_waveglow_doubles.py, evaluated on the CPU), the GEMM shapes this path
adds (8-wide sides, fp32 outputs, column-sliced outputs with an addend), and the whole step against the fixture the
REFERENCE's WaveGlow + WaveGlowLoss produced (tests/golden/waveglow_loss.npz) and against torch.optim.Adam on the oracle.
Bars: loss 1e-3 relative (north_star); gradients within the 16-bit storage floor measured with the fp64-accumulating
doubles at the same storage dtype (fp16: 3e-4 on norms, bf16: 3.6e-3) times a margin."""
import os

import numpy as np
import pytest
import torch

from tests import _waveglow_doubles as D

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
DTYPES = [torch.float16, torch.bfloat16]


def _tol(dtype):
    return dict(rtol=2e-3, atol=2e-3) if dtype == torch.float16 else dict(rtol=1.6e-2, atol=1.6e-2)


def _ops():
    from deeplearningexamples_amd.waveglow import ops
    return ops


def _close(got, ref, **kw):
    np.testing.assert_allclose(got.detach().float().cpu().numpy(), ref.detach().float().cpu().numpy(), **kw)


# ------------------------------------------------------------------------------------------------- kernels
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("b,t,ch,nt,dil,left", [(2, 256, 64, 3, 1, 1), (3, 37, 24, 3, 4, 1), (2, 50, 8, 3, 64, 1),
                                                (2, 8, 80, 4, -1, 0), (1, 1000, 512, 3, 128, 1), (2, 33, 16, 5, 2, 2)])
def test_taps_and_transpose(cuda, dtype, b, t, ch, nt, dil, left):
    ops = _ops()
    g = torch.Generator().manual_seed(b * 1000 + t)
    x = torch.randn(b * t, ch, generator=g).to(dtype)
    col = ops.taps(x.to(cuda), b, t, nt, dil, left)
    assert torch.equal(col.cpu(), D.taps(x, b, t, nt, dil, left))
    wide = torch.randn(b * t, 2 * ch, generator=g).to(dtype)                 # x = the left half of a wider matrix (row stride 2 C)
    assert torch.equal(ops.taps(wide.to(cuda)[:, :ch], b, t, nt, dil, left).cpu(), D.taps(wide[:, :ch], b, t, nt, dil, left))
    dcol = torch.randn(b * t, nt * ch, generator=g).to(dtype)
    add = torch.randn(b * t, 2 * ch, generator=g).to(dtype)
    for with_add in (False, True):
        ref = torch.zeros(b * t, 2 * ch, dtype=dtype)
        D.taps_bwd(dcol, b, t, ch, nt, dil, left, out=ref[:, :ch], addend=add[:, ch:] if with_add else None)
        out = torch.zeros(b * t, 2 * ch, dtype=dtype, device=cuda)
        add_d = add.to(cuda)
        ops.taps_bwd(dcol.to(cuda), b, t, ch, nt, dil, left, out=out[:, :ch], addend=add_d[:, ch:] if with_add else None)
        _close(out, ref, **_tol(dtype))
    # in place: out IS the addend (how the engine accumulates the residual path)
    buf = add.clone().to(cuda)
    ops.taps_bwd(dcol.to(cuda), b, t, ch, nt, dil, left, out=buf[:, :ch], addend=buf[:, :ch])
    ref = add.clone()
    D.taps_bwd(dcol, b, t, ch, nt, dil, left, out=ref[:, :ch], addend=add[:, :ch].clone())
    _close(buf, ref, **_tol(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("m,nc", [(512, 64), (77, 8), (1000, 512)])
def test_gate_forward_backward(cuda, dtype, m, nc):
    ops = _ops()
    g = torch.Generator().manual_seed(m + nc)
    s_all = (torch.randn(m, 3 * 2 * nc, generator=g) * 2).to(dtype)         # a column slice of a wider matrix, like s_all
    s = s_all[:, 2 * nc:4 * nc]
    acts = ops.gate_fwd(s_all.to(cuda)[:, 2 * nc:4 * nc], nc)
    _close(acts, D.gate_fwd(s, nc), **_tol(dtype))
    da = torch.randn(m, nc, generator=g).to(dtype)
    ds_ref = torch.zeros(m, 6 * nc, dtype=dtype)
    D.gate_bwd(da, s, ds_ref[:, 2 * nc:4 * nc])
    ds = torch.zeros(m, 6 * nc, dtype=dtype, device=cuda)
    ops.gate_bwd(da.to(cuda), s_all.to(cuda)[:, 2 * nc:4 * nc], ds[:, 2 * nc:4 * nc])
    _close(ds, ds_ref, **_tol(dtype))


def _rot(c, g, noise=0.05):
    q, _ = torch.linalg.qr(torch.randn(c, c, generator=g))
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return (q + noise * torch.randn(c, c, generator=g)).contiguous()


@pytest.mark.parametrize("c", [8, 6, 4, 2])
@pytest.mark.parametrize("m", [512, 1000, 70001])
def test_invertible_conv_forward_backward_and_logdet(cuda, c, m):
    ops = _ops()
    g = torch.Generator().manual_seed(c * 7 + m)
    w = _rot(c, g)
    x = torch.randn(m, 8, generator=g)
    y, a0 = ops.invconv_fwd(x.to(cuda), w.to(cuda), c, torch.float16)
    yr, a0r = D.invconv_fwd(x, w, c, torch.float16)
    _close(y, yr, rtol=1e-5, atol=1e-5)
    _close(a0, a0r, rtol=1e-3, atol=1e-3)
    assert torch.equal(y[:, :8 - c].cpu(), x[:, :8 - c])                    # channels emitted early pass through
    ld, sg = torch.zeros(1, device=cuda), torch.zeros(1, device=cuda)
    winv_t = ops.logdet_inv(w.to(cuda), c, ld, sg)
    ldr, sgr = torch.zeros(1), torch.zeros(1)
    winv_tr = D.logdet_inv(w, c, ldr, sgr)
    _close(ld, ldr, rtol=1e-5, atol=1e-6)
    _close(winv_t, winv_tr, rtol=1e-4, atol=1e-5)
    assert float(sg) == float(sgr) == 1.0
    wneg = w.clone()
    wneg[0] = -wneg[0]
    ops.logdet_inv(wneg.to(cuda), c, ld, sg)
    assert float(sg) == -1.0 and abs(float(ld) - float(ldr)) < 1e-5          # log|det| and the sign torch.logdet would refuse
    dy, da0 = torch.randn(m, 8, generator=g), torch.randn(m, 8, generator=g)
    scale = torch.tensor([128.0])
    for d in (None, da0):
        dw, dwr = torch.zeros(64, device=cuda), torch.zeros(64)
        dx = ops.invconv_bwd(dy.to(cuda), None if d is None else d.to(cuda), x.to(cuda), w.to(cuda), winv_t, dw, scale.to(cuda),
                             0.125, c)
        dxr = D.invconv_bwd(dy, d, x, w, winv_tr, dwr, scale, 0.125, c)
        _close(dx, dxr, rtol=1e-5, atol=1e-5)
        _close(dw[:c * c], dwr[:c * c], rtol=2e-4, atol=2e-3 * (m ** 0.5) / 30)
        assert float(dw[c * c:].abs().max()) == 0 if c < 8 else True


@pytest.mark.parametrize("c", [8, 6, 4])
@pytest.mark.parametrize("dtype", DTYPES)
def test_affine_coupling_and_loss(cuda, c, dtype):
    ops = _ops()
    m = 1000
    g = torch.Generator().manual_seed(c)
    y = torch.randn(m, 8, generator=g)
    o = torch.randn(m, 8, generator=g) * 0.5
    parts = ops.coupling_partials(m)
    lp = torch.zeros(parts, device=cuda)
    z = ops.coupling_fwd(y.to(cuda), o.to(cuda), c, lp)
    lpr = torch.zeros(4)
    zr = D.coupling_fwd(y, o, c, lpr)
    _close(z, zr, rtol=2e-6, atol=2e-6)
    assert abs(float(lp.sum()) - float(lpr.sum())) <= 1e-4 * (1 + abs(float(lpr.sum())))
    dz = torch.randn(m, 8, generator=g)
    scale = torch.tensor([64.0])
    dy, d_o = ops.coupling_bwd(dz.to(cuda), y.to(cuda), o.to(cuda), scale.to(cuda), 1.0 / (m * 8), c, dtype)
    dyr, d_or = D.coupling_bwd(dz, y, o, scale, 1.0 / (m * 8), c, dtype)
    _close(dy, dyr, rtol=2e-6, atol=2e-6)
    _close(d_o, d_or, **_tol(dtype))
    assert float(d_o[:, c:].abs().max()) == 0 if c < 8 else True
    logdets = torch.tensor([0.3, -0.2, 0.05])
    loss = ops.loss(z, lp, logdets.to(cuda), 0.8)
    _close(loss, D.loss(zr, lpr, logdets, 0.8), rtol=2e-5)
    _close(ops.dz_init(z, scale.to(cuda), 0.37), D.dz_init(zr, scale, 0.37), rtol=1e-5, atol=1e-4)   # z itself differs by ulps


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("co,ci,kt,cip", [(128, 64, 3, 64), (64, 4, 1, 8), (64, 3, 1, 8), (128, 640, 1, 640), (16, 80, 8, 80),
                                          (1024, 512, 3, 512)])
def test_weight_norm_operand_and_gradient(cuda, dtype, co, ci, kt, cip):
    ops = _ops()
    g = torch.Generator().manual_seed(co + ci)
    v = torch.randn(co, ci, kt, generator=g) * 0.1
    gg = 1 + 0.1 * torch.randn(co, 1, 1, generator=g)
    for gain in (gg, None):
        w16 = torch.zeros(co, kt * cip, dtype=dtype, device=cuda)
        ops.weight_norm_fwd(v.to(cuda), None if gain is None else gain.to(cuda), w16, cip=cip)
        ref = D.weight_norm_fwd(v, gain, torch.zeros(co, kt * cip, dtype=dtype), cip=cip)
        _close(w16, ref, rtol=2e-3 if dtype == torch.float16 else 1e-2, atol=1e-6)
        dw = torch.randn(co, kt * cip, generator=g)
        dv, dg = torch.zeros(co, ci, kt, device=cuda), torch.zeros(co, 1, 1, device=cuda)
        ops.weight_norm_bwd(dw.to(cuda), v.to(cuda), None if gain is None else gain.to(cuda), dv, dg if gain is not None else None,
                            cip=cip)
        dvr, dgr = torch.zeros(co, ci, kt), torch.zeros(co, 1, 1)
        D.weight_norm_bwd(dw, v, gain, dvr, dgr, cip=cip)
        _close(dv, dvr, rtol=1e-4, atol=1e-4)
        if gain is not None:
            _close(dg, dgr, rtol=1e-4, atol=1e-4)
    # autograd of torch's own weight_norm formula (model.py:95-136 uses torch.nn.utils.weight_norm, dim 0)
    vv, g2 = v.clone().requires_grad_(True), gg.clone().requires_grad_(True)
    w = vv * (g2 / vv.flatten(1).norm(dim=1).view(co, 1, 1))
    dwt = dw.view(co, kt, cip)[:, :, :ci].permute(0, 2, 1)
    (w * dwt).sum().backward()
    dv, dg = torch.zeros(co, ci, kt, device=cuda), torch.zeros(co, 1, 1, device=cuda)
    ops.weight_norm_bwd(dw.to(cuda), v.to(cuda), gg.to(cuda), dv, dg, cip=cip)
    _close(dv, vv.grad, rtol=1e-4, atol=1e-4)
    _close(dg, g2.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dtype", DTYPES)
def test_upsampling_as_one_gemm_matches_conv_transpose1d(cuda, dtype):
    """ConvTranspose1d(80, 80, 1024, stride 256) (model.py:165-167,197-200) = taps(mel) x permuted weight, time-major output."""
    from deeplearningexamples_amd import functional as F
    ops = _ops()
    g = torch.Generator().manual_seed(2)
    b, fq, cm = 2, 8, 80
    w = torch.randn(cm, cm, 1024, generator=g) * 0.01
    bias = torch.randn(cm, generator=g) * 0.1
    mel = torch.randn(b, cm, fq, generator=g)
    b16, rep = ops.upsample_weight(w.to(cuda), bias.to(cuda), dtype, 256)
    b16r, repr_ = D.upsample_weight(w, bias, dtype, 256)
    assert torch.equal(b16.cpu(), b16r) and torch.equal(rep.cpu(), repr_)
    mel_cl = F.nchw_to_nhwc(mel.to(cuda).view(b, cm, fq, 1), dtype, c_padded=cm).view(b * fq, cm)
    col = ops.taps(mel_cl, b, fq, 4, -1, 0)
    up = F.gemm(col, b16, b * fq, 256 * cm, 4 * cm, True, True, bias=rep)
    ref = torch.nn.functional.conv_transpose1d(mel.to(dtype).float(), w.to(dtype).float(), bias, stride=256)[:, :, :fq * 256]
    _close(up.view(b, fq * 256, cm), ref.permute(0, 2, 1), **_tol(dtype))
    db = torch.randn(256 * cm, 4 * cm, generator=g)
    dw = torch.zeros(cm, cm, 1024, device=cuda)
    ops.upsample_weight_bwd(db.to(cuda), dw, 256)
    dwr = torch.zeros(cm, cm, 1024)
    D.upsample_weight_bwd(db, dwr, 256)
    assert torch.equal(dw.cpu(), dwr)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_shapes_this_path_adds(cuda, dtype):
    """8-wide sides (start / end), fp32 outputs, outputs that are column slices with an addend of the same leading dimension."""
    from deeplearningexamples_amd import _cabi as C
    from deeplearningexamples_amd import functional as F
    g = torch.Generator().manual_seed(4)
    m, nc = 1000, 64

    def r(*s, scale=0.3):
        return (torch.randn(*s, generator=g) * scale).to(dtype)

    def both(fn):
        return fn(F, lambda t: t.to(cuda)), fn(D, lambda t: t)

    tol = _tol(dtype)
    a0, ws, bs = r(m, 8), r(nc, 8), torch.randn(nc, generator=g)
    got, ref = both(lambda L, d: L.gemm(d(a0), d(ws), m, nc, 8, True, True, bias=d(bs)))                       # start
    _close(got, ref, **tol)
    out, we, be = r(m, nc), r(8, nc), torch.randn(8, generator=g)
    got, ref = both(lambda L, d: L.gemm(d(out), d(we), m, 8, nc, True, True, bias=d(be), out_dtype=torch.float32))   # end, fp32 out
    assert got.dtype == torch.float32
    _close(got, ref, rtol=1e-4, atol=1e-4)
    d_o = r(m, 8)
    got, ref = both(lambda L, d: L.gemm(d(d_o), d(we), m, nc, 8, True, False, out=d(torch.zeros(m, 2 * nc, dtype=dtype))[:, nc:]))
    _close(got, ref, **tol)                                                                                        # end dgrad into a slice
    got, ref = both(lambda L, d: L.gemm(d(d_o), d(out), 8, nc, m, False, False, out=d(torch.zeros(8, nc)),
                                        splitk=F.pick_splitk(8, nc, m)))                                           # end wgrad
    _close(got, ref, rtol=1e-3, atol=1e-3)
    dx0 = r(m, nc)
    got, ref = both(lambda L, d: L.gemm(d(dx0), d(ws), m, 8, nc, True, False, out_dtype=torch.float32))            # start dgrad
    _close(got, ref, rtol=1e-4, atol=1e-4)
    got, ref = both(lambda L, d: L.gemm(d(dx0), d(a0), nc, 8, m, False, False, out=d(torch.zeros(nc, 8)),
                                        splitk=F.pick_splitk(nc, 8, m)))                                           # start wgrad
    _close(got, ref, rtol=1e-3, atol=1e-3)
    # in_layer: output = a column slice of the all-flows matrix, addend = the same columns of the cond matrix
    wide = 6 * nc
    col, wi, bi, cond = r(m, 3 * nc), r(2 * nc, 3 * nc, scale=0.1), torch.randn(2 * nc, generator=g), r(m, wide)

    def in_layer(L, d):
        s_all = d(torch.zeros(m, wide, dtype=dtype))
        L.gemm(d(col), d(wi), m, 2 * nc, 3 * nc, True, True, out=s_all[:, 2 * nc:4 * nc], bias=d(bi), act=C.ACT_ADD,
               mask_src=d(cond)[:, 2 * nc:4 * nc])
        return s_all
    got, ref = both(in_layer)
    _close(got, ref, **tol)
    # res / skip halves of one weight, the running sums added
    acts, wr, br, x = r(m, nc), r(2 * nc, nc), torch.randn(2 * nc, generator=g), r(m, nc)
    got, ref = both(lambda L, d: L.gemm(d(acts), d(wr)[nc:], m, nc, nc, True, True, bias=d(br)[nc:], act=C.ACT_ADD, mask_src=d(x)))
    _close(got, ref, **tol)
    # one res_skip GEMM over the two-halves buffer [audio | skip sum]; last layer: the skip half only, in place in the right half
    xo = r(m, 2 * nc)
    got, ref = both(lambda L, d: L.gemm(d(acts), d(wr), m, 2 * nc, nc, True, True, bias=d(br), act=C.ACT_ADD, mask_src=d(xo),
                                        out=d(torch.zeros(m, 2 * nc, dtype=dtype))))
    _close(got, ref, **tol)
    got, ref = both(lambda L, d: L.gemm(d(acts), d(wr)[:nc], m, nc, nc, True, True, bias=d(br)[:nc], act=C.ACT_ADD,
                                        mask_src=d(xo)[:, nc:], out=d(torch.zeros(m, 2 * nc, dtype=dtype))[:, nc:]))
    _close(got, ref, **tol)
    got, ref = both(lambda L, d: L.gemm(d(xo)[:, nc:], d(we), m, 8, nc, True, True, bias=d(be), out_dtype=torch.float32))   # end from the half
    _close(got, ref, rtol=1e-4, atol=1e-4)
    got, ref = both(lambda L, d: L.gemm(d(d_o), d(xo)[:, nc:], 8, nc, m, False, False, out=d(torch.zeros(8, nc)),
                                        splitk=F.pick_splitk(8, nc, m)))                                           # end wgrad, strided B
    _close(got, ref, rtol=1e-3, atol=1e-3)
    # res_skip dgrad from the two-halves gradient buffer, last layer = right half only (row stride 2 nc)
    d_rs = r(m, 2 * nc)
    got, ref = both(lambda L, d: L.gemm(d(d_rs)[:, nc:], d(wr)[:nc], m, nc, nc, True, False))
    _close(got, ref, **tol)
    got, ref = both(lambda L, d: L.gemm(d(d_rs)[:, nc:], d(acts), nc, nc, m, False, False, out=d(torch.zeros(nc, nc)),
                                        splitk=F.pick_splitk(nc, nc, m)))
    _close(got, ref, rtol=1e-3, atol=2e-3)
    got, ref = both(lambda L, d: L.colsum(d(d_rs)[:, nc:], out=d(torch.zeros(nc))))
    _close(got, ref, rtol=1e-3, atol=2e-3)
    got, ref = both(lambda L, d: L.colsum(d(d_o), out=d(torch.zeros(8))))
    _close(got, ref, rtol=1e-3, atol=2e-3)


def test_adam_with_unscale_clip_and_skip(cuda):
    from deeplearningexamples_amd import multi_tensor as mt
    g = torch.Generator().manual_seed(9)
    n = 300007
    p0 = torch.randn(n, generator=g)
    dev = [torch.zeros(n, device=cuda) for _ in range(4)]
    dev[1].copy_(p0)
    cpu = [torch.zeros(n) for _ in range(4)]
    cpu[1].copy_(p0)
    tab = mt.TensorTable([[t] for t in dev], chunk=mt.streaming_chunk([[dev[0]]]))
    step_d = torch.zeros(1, dtype=torch.int32, device=cuda)
    ref_p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref_p], lr=1e-3, weight_decay=1e-6)
    scale = 1024.0
    for it in range(4):
        grad = torch.randn(n, generator=g) * (10.0 if it == 2 else 0.001)      # step 2 is clipped
        dev[0].copy_(grad * scale)
        cpu[0].copy_(grad * scale)
        skip = torch.tensor([1.0 if it == 1 else 0.0])                          # step 1 overflowed: skipped
        if it != 1:
            step_d += 1
            ref_p.grad = grad.clone()
            torch.nn.utils.clip_grad_norm_([ref_p], 0.5)
            opt.step()
        gn, _ = mt.l2norm(mt.TensorTable([[dev[0]]]))
        mt.adam(tab, torch.tensor([1e-3], device=cuda), 0.9, 0.999, 1e-8, 1e-6, step_d, skip_flag=skip.to(cuda),
                inv_scale=torch.tensor([1.0 / scale], device=cuda), grad_norm=gn, max_grad_norm=0.5)
        D.adam(D._Table([[t] for t in cpu], 0), 1e-3, 0.9, 0.999, 1e-8, 1e-6, step_d.cpu(), skip_flag=skip,
               inv_scale=torch.tensor([1.0 / scale]), grad_norm=gn.cpu(), max_grad_norm=0.5)
        for a, b in zip(dev[1:], cpu[1:]):
            _close(a, b, rtol=2e-5, atol=1e-7)
    _close(dev[1], ref_p, rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("dtype", DTYPES)
def test_table_driven_weight_norm_and_logdet(cuda, dtype):
    """One launch for every tensor (dle_wg_weight_norm_{fwd,bwd}_batched, dle_wg_logdet_inv_batched) == the per-tensor statements."""
    ops = _ops()
    g = torch.Generator().manual_seed(21)
    shapes = [(64, 4, 1, 8), (128, 64, 3, 64), (8, 64, 1, 64), (128, 80, 8, 80), (64, 64, 1, 64), (1024, 512, 3, 512), (6, 64, 1, 64)]
    host, dev = [], []
    for co, ci, kt, cip in shapes:
        plain = co in (8, 6)                                  # `end`: no gain, no gradient entry
        e = dict(v=torch.randn(co, ci, kt, generator=g) * 0.1, g=None if plain else 1 + 0.1 * torch.randn(co, 1, 1, generator=g),
                 w16=torch.zeros(co, kt * cip, dtype=dtype), dw=torch.randn(co, kt * cip, generator=g), dv=torch.zeros(co, ci, kt),
                 dg=None if plain else torch.zeros(co, 1, 1), cip=cip)
        host.append(e)
        dev.append({k: (v.to(cuda) if isinstance(v, torch.Tensor) else v) for k, v in e.items()})
    ops.weight_norm_fwd_batched(ops.WeightNormTable(dev, cuda), dtype)
    D.weight_norm_fwd_batched(ops.WeightNormTable(host, "cpu"), dtype)
    for h, d in zip(host, dev):
        _close(d["w16"], h["w16"], rtol=2e-3 if dtype == torch.float16 else 1e-2, atol=1e-6)
    hn, dn = [e for e in host if e["g"] is not None], [e for e in dev if e["g"] is not None]
    ops.weight_norm_bwd_batched(ops.WeightNormTable(dn, cuda))
    D.weight_norm_bwd_batched(ops.WeightNormTable(hn, "cpu"))
    for h, d in zip(hn, dn):
        _close(d["dv"], h["dv"], rtol=1e-4, atol=1e-4)
        _close(d["dg"], h["dg"], rtol=1e-4, atol=1e-4)
    flat = torch.zeros(400)
    entries = []
    for off, c in ((8, 8), (96, 8), (200, 6), (264, 4), (320, 2)):
        flat[off:off + c * c] = _rot(c, g).reshape(-1)
        entries.append((off, c))
    nf = len(entries)
    ld, sg, wi = torch.zeros(nf, device=cuda), torch.zeros(nf, device=cuda), torch.zeros(nf, 64, device=cuda)
    ops.logdet_inv_batched(flat.to(cuda), ops.LogdetTable(entries, cuda), ld, wi, sg)
    ldr, sgr, wir = torch.zeros(nf), torch.zeros(nf), torch.zeros(nf, 64)
    D.logdet_inv_batched(flat, ops.LogdetTable(entries, "cpu"), ldr, wir, sgr)
    _close(ld, ldr, rtol=1e-5, atol=1e-6)
    _close(wi, wir, rtol=1e-4, atol=1e-5)
    assert torch.equal(sg.cpu(), sgr)


@pytest.mark.parametrize("dtype", DTYPES)
def test_all_in_layer_weight_gradients_as_one_batched_gemm(cuda, dtype):
    """dw_in[z] = ds_all[:, z*2nc:(z+1)*2nc]^T x col_all[z]: column-sliced A (row stride = all columns), fp32 output."""
    from deeplearningexamples_amd import functional as F
    g = torch.Generator().manual_seed(31)
    for m, nc, nz in ((1000, 64, 6), (512, 128, 3)):
        ds = (torch.randn(m, nz * 2 * nc, generator=g) * 0.3).to(dtype)
        col = (torch.randn(nz, m, 3 * nc, generator=g) * 0.3).to(dtype)
        args = (2 * nc, 3 * nc, m, nz * 2 * nc, 3 * nc, 3 * nc, False, False, nz, 1, (2 * nc, 0), (m * 3 * nc, 0), (2 * nc * 3 * nc, 0))
        got = F.gemm_batched(ds.to(cuda), col.to(cuda), torch.zeros(nz, 2 * nc, 3 * nc, device=cuda), *args)
        ref = D.gemm_batched(ds, col, torch.zeros(nz, 2 * nc, 3 * nc), *args)
        _close(got, ref, rtol=2e-3, atol=2e-2)
        direct = torch.stack([ds[:, z * 2 * nc:(z + 1) * 2 * nc].float().t() @ col[z].float() for z in range(nz)])
        _close(got, direct, rtol=2e-3, atol=2e-2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_all_res_skip_weight_gradients_as_two_batched_gemms(cuda, dtype):
    """Slices (flow, layer < L-1): d_rs[k, i]^T x acts[k, i] (2nc x nc, outer / inner strides skip the last layer's slot); the last
    layers: the skip half of d_rs only (column-offset A with row stride 2nc), nc x nc."""
    from deeplearningexamples_amd import functional as F
    g = torch.Generator().manual_seed(41)
    nf, nl, m, nc = 3, 4, 512, 64
    d_rs = (torch.randn(nf, nl, m, 2 * nc, generator=g) * 0.3).to(dtype)
    acts = (torch.randn(nf * nl, m, nc, generator=g) * 0.3).to(dtype)

    def run(L, d):
        a, b, c = d(d_rs), d(acts), d(torch.zeros(nf, nl, 2 * nc, nc))
        L.gemm_batched(a, b, c, 2 * nc, nc, m, 2 * nc, nc, nc, False, False, nf * (nl - 1), nl - 1,
                       (nl * m * 2 * nc, m * 2 * nc), (nl * m * nc, m * nc), (nl * 2 * nc * nc, 2 * nc * nc))
        L.gemm_batched(a[0, nl - 1, :, nc:], b[nl - 1], c[0, nl - 1], nc, nc, m, 2 * nc, nc, nc, False, False, nf, 1,
                       (nl * m * 2 * nc, 0), (nl * m * nc, 0), (nl * 2 * nc * nc, 0))
        return c
    got, ref = run(F, lambda t: t.to(cuda)), run(D, lambda t: t)
    _close(got, ref, rtol=2e-3, atol=2e-2)
    for k in range(nf):
        for i in range(nl):
            a = d_rs[k, i].float() if i < nl - 1 else d_rs[k, i, :, nc:].float()
            want = a.t() @ acts[k * nl + i].float()
            _close(got[k, i, :want.shape[0]], want, rtol=2e-3, atol=2e-2)
    assert float(got[:, nl - 1, nc:].abs().max()) == 0                     # the unused half of the last layers' slots is untouched


# ------------------------------------------------------------------------------------------------- the step
def _trainer(cuda, dtype, cfg=None, seed=None, **kw):
    from oracle import waveglow_oracle as WO
    from deeplearningexamples_amd.waveglow.engine import WaveGlowTrainer
    from deeplearningexamples_amd.waveglow.model import WaveGlow
    c = WO.WAVEGLOW_CASE
    cfg = cfg or c["cfg"]
    state = WO.seeded_state(cfg, c["seed"] if seed is None else seed)
    model = WaveGlow(**cfg, device=cuda)
    model.load_reference_state(state)
    return WO, c, state, model, WaveGlowTrainer(model, compute_dtype=dtype, sigma=c["sigma"], **kw)


@pytest.mark.parametrize("dtype", DTYPES)
def test_step_loss_and_gradients_vs_reference_fixture(cuda, dtype):
    WO, c, state, model, tr = _trainer(cuda, dtype, init_loss_scale=65536.0)
    gold = np.load(os.path.join(HERE, "golden", "waveglow_loss.npz"))
    mel, audio = WO.seeded_inputs(c)
    loss = tr.forward(mel.to(cuda), audio.to(cuda))
    ref = float(gold["loss"][0])
    floor = 1.8e-4 if dtype == torch.float16 else 6.6e-4       # 16-bit STORAGE floor of this fixture (fp64-accumulating doubles)
    assert abs(float(loss) - ref) <= (1e-3 + floor) * abs(ref), (float(loss), ref)
    tr.backward()
    s = float(tr.scaler.scale)
    assert bool(torch.isfinite(tr.g.flat).all())
    nbar, sbar = (3e-3, 5e-3) if dtype == torch.float16 else (2e-2, 3e-2)
    for k in [f[len("gnorm."):] for f in gold.files if f.startswith("gnorm.")]:
        r = float(gold["gnorm." + k][0])
        assert abs(float(tr.g[k].norm()) / s - r) <= nbar * r + 1e-9, (k, float(tr.g[k].norm()) / s, r)
    for k in [f[len("grad."):] for f in gold.files if f.startswith("grad.")]:
        a, b = tr.g[k].cpu().numpy().reshape(-1)[:64] / s, gold["grad." + k]
        assert np.linalg.norm(a - b) <= sbar * np.linalg.norm(b), k


@pytest.mark.parametrize("dtype", DTYPES)
def test_three_steps_follow_torch_adam_on_the_oracle(cuda, dtype):
    """train.py:474-500: forward, scaled backward, unscale + clip_grad_norm_, Adam, scaler.update -- loss trajectory and weights."""
    WO, c, state, model, tr = _trainer(cuda, dtype, lr=1e-4, grad_clip_thresh=0.5, weight_decay=1e-6, init_loss_scale=4096.0)
    mel, audio = WO.seeded_inputs(c)
    p = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    opt = torch.optim.Adam(list(p.values()), lr=1e-4, weight_decay=1e-6)
    ref, got = [], []
    for _ in range(3):
        opt.zero_grad()
        lo = WO.waveglow_loss(p, c["cfg"], mel, audio, c["sigma"])
        lo.backward()
        torch.nn.utils.clip_grad_norm_(list(p.values()), 0.5)
        opt.step()
        ref.append(float(lo.detach()))
        got.append(float(tr.train_step(mel.to(cuda), audio.to(cuda))))
    np.testing.assert_allclose(got, ref, rtol=2e-3)
    assert int(tr.step_t) == 3 and float(tr.scaler.found_inf) == 0
    sd = model.state_dict()
    moved = sum(float((p[k].detach() - state[k]).norm()) ** 2 for k in p) ** 0.5
    dist = sum(float((sd[k].cpu() - p[k].detach()).norm()) ** 2 for k in p) ** 0.5
    assert dist <= 0.15 * moved, (dist, moved)         # Adam's sign-like first steps amplify 16-bit gradient noise near g = 0


def test_overflow_skips_and_recovers(cuda):
    WO, c, state, model, tr = _trainer(cuda, torch.float16, init_loss_scale=2.0 ** 40)      # fp16 gradients overflow at this scale
    mel, audio = WO.seeded_inputs(c)
    before = tr.p.flat.clone()
    tr.train_step(mel.to(cuda), audio.to(cuda))
    assert torch.equal(tr.p.flat, before) and int(tr.step_t) == 0 and float(tr.scaler.scale) == 2.0 ** 39
    tr.scaler.scale.fill_(1024.0)
    tr.scaler.inv_scale.fill_(1.0 / 1024.0)
    tr.train_step(mel.to(cuda), audio.to(cuda))
    assert int(tr.step_t) == 1 and not torch.equal(tr.p.flat, before)


def test_segment_ending_inside_a_frame_block(cuda):
    """--segment-length 8000 is 31.25 hops: the upsampled spectrogram is cut to the audio length (model.py:199-200)."""
    WO, c, state, model, tr = _trainer(cuda, torch.float16, init_loss_scale=1024.0)
    rng = np.random.default_rng(5)
    mel = torch.from_numpy(rng.standard_normal((2, 80, 5)).astype(np.float32))       # one frame more than the segment needs
    audio = torch.from_numpy((rng.standard_normal((2, 1000)) * 0.2).astype(np.float32))
    p = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    lo = WO.waveglow_loss(p, c["cfg"], mel, audio, c["sigma"])
    lo.backward()
    loss = tr.forward(mel.to(cuda), audio.to(cuda))
    assert abs(float(loss) - float(lo)) <= 1e-3 * abs(float(lo))
    tr.backward()
    for k in ("upsample.weight", "upsample.bias", "WN.0.cond_layers.1.weight_v", "convinv.3.conv.weight", "WN.2.start.weight_v"):
        a, b = tr.g[k].cpu() / 1024.0, p[k].grad
        assert float((a - b).norm()) <= 1e-2 * float(b.norm()), k


def test_reference_size_network_one_step_vs_oracle(cuda):
    """The reference's default network (12 flows, 8 layers, 512 channels: waveglow/arg_parser.py:38-64, 268 M parameters) at batch
    2 x 2048 samples against the oracle's autograd on the CPU."""
    from deeplearningexamples_amd.waveglow.model import DEFAULT_CONFIG
    WO, c, state, model, tr = _trainer(cuda, torch.float16, cfg=DEFAULT_CONFIG, seed=11, init_loss_scale=4096.0)
    case = dict(cfg=DEFAULT_CONFIG, seed=11, batch=2, segment=2048)
    mel, audio = WO.seeded_inputs(case)
    p = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    lo = WO.waveglow_loss(p, DEFAULT_CONFIG, mel, audio, c["sigma"])
    lo.backward()
    loss = tr.forward(mel.to(cuda), audio.to(cuda))
    # bar = north_star's 1e-3 + the fp16 storage floor of this network / input measured with the fp64-accumulating doubles
    # (6.5e-4 on the loss, <= 3.1e-2 on the worst parameter gradient; bf16 storage: 3.1e-3 / 1.6e-1 -- the reference trains
    # this model in fp16, BASELINE.json configs[4])
    assert abs(float(loss) - float(lo)) <= (1e-3 + 6.5e-4) * abs(float(lo)), (float(loss), float(lo))
    tr.backward()
    assert bool(torch.isfinite(tr.g.flat).all())
    for k in ("upsample.weight", "convinv.0.conv.weight", "convinv.11.conv.weight", "WN.0.in_layers.7.weight_v",
              "WN.5.cond_layers.3.weight_v", "WN.11.res_skip_layers.7.weight_g", "WN.8.end.weight", "WN.4.start.weight_v"):
        a, b = tr.g[k].cpu() / 4096.0, p[k].grad
        assert float((a - b).norm()) <= 5e-2 * float(b.norm()), (k, float((a - b).norm()) / float(b.norm()))
    tr.optimizer_step()
    assert int(tr.step_t) == 1


def test_graphed_step_matches_eager(cuda):
    """utils/graph.py GraphedStep (whole step in one HIP graph, inputs through static buffers) == the eager step."""
    from deeplearningexamples_amd.utils.graph import GraphedStep
    rng = np.random.default_rng(8)
    batches = [(torch.from_numpy(rng.standard_normal((2, 80, 8)).astype(np.float32)).to(cuda),
                torch.from_numpy((rng.standard_normal((2, 2048)) * 0.2).astype(np.float32)).to(cuda)) for _ in range(5)]
    WO, c, state, m1, t1 = _trainer(cuda, torch.float16, init_loss_scale=1024.0)
    eager = [float(t1.train_step(*b)) for b in batches]
    WO, c, state, m2, t2 = _trainer(cuda, torch.float16, init_loss_scale=1024.0)
    step = GraphedStep(t2.train_step, warmup_steps=2)
    graphed = [float(step(*b)) for b in batches]
    assert step.graph is not None
    np.testing.assert_allclose(graphed, eager, rtol=1e-6)
    assert torch.allclose(t1.p.flat, t2.p.flat, rtol=1e-5, atol=1e-7) and int(t2.step_t) == 5
