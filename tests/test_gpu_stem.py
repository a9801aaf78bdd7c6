"""The stem kernels (csrc/stem.hip: conv7x7 / 2 / pad 3 of a 3-channel image on a 4-channel NHWC input, forward + BatchNorm
partial sums, weight gradient, weight packing, the 4-channel layout conversion) against plain fp32 torch on the same 16-bit
rounded operands -- what cuDNN computes behind builder.conv7x7(3, 64, stride=2) (Classification/ConvNets/image_classification/
models/resnet.py:262-268, models/common.py:31-60).  GPU only.  Tolerances: outputs are rounded to 16 bits once (relative 2^-8 for
bf16, 2^-11 for fp16 of the value, plus fp32 accumulation-order noise); weight gradients are fp32 sums over N*P*Q pixels."""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu

SHAPES = [(2, 224, 224), (3, 64, 64), (2, 37, 53), (1, 9, 224)]


def _operands(cuda, dtype, n, h, w, seed=0):
    from deeplearningexamples_amd import functional as F
    g = torch.Generator().manual_seed(seed)
    img = torch.randn((n, 3, h, w), generator=g)
    wm = (torch.randn((64, 3, 7, 7), generator=g) * 0.1).contiguous(memory_format=torch.channels_last)
    x4 = F.nchw_to_nhwc(img.to(cuda), dtype, 4)
    w2 = F.stem_pack_weight(wm.to(cuda).contiguous(memory_format=torch.channels_last), dtype)
    xr = x4.float().cpu()[..., :3].permute(0, 3, 1, 2).contiguous()            # the rounded image, NCHW fp32
    wr = w2.float().cpu()[:, :, :7, :3].permute(0, 3, 1, 2).contiguous()       # the rounded weights, [64, 3, 7, 7]
    return img, wm, x4, w2, xr, wr


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_layout_and_weight_packing(cuda, dtype):
    img, wm, x4, w2, xr, wr = _operands(cuda, dtype, 2, 20, 24)
    assert x4.shape == (2, 20, 24, 4) and torch.all(x4[..., 3] == 0)
    assert torch.equal(xr, img.to(dtype).float())
    assert w2.shape == (64, 7, 8, 4) and torch.all(w2[:, :, 7, :] == 0) and torch.all(w2[..., 3] == 0)
    assert torch.equal(wr, wm.to(dtype).float())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n,h,w", SHAPES)
def test_stem_forward_and_partial_sums(cuda, dtype, n, h, w):
    from deeplearningexamples_amd import functional as F
    _, _, x4, w2, xr, wr = _operands(cuda, dtype, n, h, w, seed=h)
    y, part = F.stem_conv_fwd(x4, w2, want_stats=True)
    torch.cuda.synchronize()
    ref = TF.conv2d(xr, wr, stride=2, padding=3).permute(0, 2, 3, 1)            # [N, P, Q, 64] fp32
    assert y.shape == ref.shape
    got = y.float().cpu()
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    err = (got - ref).abs()
    assert float((err - (eps * ref.abs() + 2e-4)).max()) <= 0, float(err.max())
    # the statistics are those of the ROUNDED output, whatever the grouping
    from deeplearningexamples_amd import _cabi as C
    groups = C.lib().dle_stem_conv7_groups(n, h)
    ps = part[:groups * 2 * 64].view(groups, 2, 64).double().sum(0).cpu()
    rows = got.double().reshape(-1, 64)
    assert torch.allclose(ps[0], rows.sum(0), rtol=1e-5, atol=1e-2)
    assert torch.allclose(ps[1], (rows * rows).sum(0), rtol=1e-5, atol=1e-2)
    y2, _ = F.stem_conv_fwd(x4, w2, want_stats=False)
    assert torch.equal(y, y2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_stem_bn_statistics_match_the_generic_path(cuda, dtype):
    """conv + statistics through the stem kernel == through the implicit-GEMM convolution on the 8-channel image."""
    from deeplearningexamples_amd import functional as F
    img, wm, x4, w2, _, _ = _operands(cuda, dtype, 4, 96, 96, seed=5)
    x8 = F.nchw_to_nhwc(img.to(cuda), dtype, 8)
    w8 = torch.zeros((64, 7, 7, 8), dtype=dtype, device=cuda)
    w8[..., :3] = wm.to(cuda).permute(0, 2, 3, 1).to(dtype)
    rm = [torch.zeros(64, device=cuda) for _ in range(2)]
    rv = [torch.ones(64, device=cuda) for _ in range(2)]
    ya, ma, ra = F.stem_conv_fwd_bnstats(x4, w2, rm[0], rv[0])
    yb, mb, rb = F.conv2d_fwd_bnstats(x8, w8, 2, 3, rm[1], rv[1])
    assert float((ya.float() - yb.float()).abs().max()) <= (0.04 if dtype == torch.bfloat16 else 0.006)
    assert torch.allclose(ma, mb, atol=2e-3) and torch.allclose(ra, rb, rtol=2e-3)
    assert torch.allclose(rm[0], rm[1], atol=1e-3) and torch.allclose(rv[0], rv[1], rtol=1e-3)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n,h,w", SHAPES + [(40, 48, 32)])
def test_stem_weight_gradient(cuda, dtype, n, h, w):
    from deeplearningexamples_amd import functional as F
    _, _, x4, w2, xr, wr = _operands(cuda, dtype, n, h, w, seed=7 + h)
    p, q = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    g = torch.Generator().manual_seed(11)
    dy = torch.randn((n, p, q, 64), generator=g).to(dtype)
    out = torch.full((64 * 147,), 3.0, dtype=torch.float32, device=cuda)
    F.stem_conv_wgrad(dy.to(cuda), x4, out)
    torch.cuda.synchronize()
    wref = wr.clone().requires_grad_(True)
    TF.conv2d(xr, wref, stride=2, padding=3).backward(dy.float().permute(0, 3, 1, 2))
    ref = wref.grad.permute(0, 2, 3, 1).reshape(-1)                              # KRSC memory order
    got = out.cpu()
    assert float((got - ref).norm() / ref.norm()) < 2e-5, float((got - ref).abs().max())
    out2 = torch.zeros_like(out)
    F.stem_conv_wgrad(dy.to(cuda), x4, out2)
    assert torch.equal(out, out2)                                               # fixed summation order
    F.stem_conv_wgrad(dy.to(cuda), x4, out2, accumulate=True)
    assert torch.allclose(out2, 2 * out, rtol=1e-6)
