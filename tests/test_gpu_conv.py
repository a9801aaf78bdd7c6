"""Implicit-GEMM convolutions (fwd / dgrad / wgrad, NHWC + KRSC) vs torch's CPU conv2d on the same
16-bit-rounded operands, accumulated in float64.  GPU only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# n, h, w, c, ko, r, stride, pad   (the RN50 layer families + ragged sizes)
GEOMS = [
    (2, 8, 8, 64, 64, 1, 1, 0), (2, 8, 8, 64, 256, 1, 1, 0), (3, 9, 7, 64, 64, 3, 1, 1),
    (2, 14, 14, 128, 128, 3, 2, 1), (2, 14, 14, 256, 512, 1, 2, 0), (1, 15, 15, 64, 96, 3, 2, 1),
    (2, 32, 32, 8, 64, 7, 2, 3), (4, 7, 7, 512, 512, 3, 1, 1), (2, 7, 7, 2048, 512, 1, 1, 0),
    (5, 6, 10, 72, 40, 3, 1, 1), (1, 15, 15, 64, 96, 1, 2, 0), (3, 8, 12, 128, 64, 1, 2, 0),
    (3, 12, 20, 72, 128, 3, 2, 1), (1, 28, 28, 256, 256, 3, 2, 1),      # stride-2 3x3: the parity-class data gradient
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("geom", GEOMS)
def test_conv_fwd_dgrad_wgrad(cuda, geom, dtype):
    from deeplearningexamples_amd import functional as F
    n, h, w, c, ko, r, stride, pad = geom
    g = torch.Generator().manual_seed(sum(geom))
    x = (torch.randn(n, c, h, w, generator=g) * 0.5).to(dtype)              # NCHW logical
    wt = (torch.randn(ko, c, r, r, generator=g) * (1.0 / np.sqrt(c * r * r))).to(dtype)
    xd, wd = x.double().requires_grad_(), wt.double().requires_grad_()
    y = torch.nn.functional.conv2d(xd, wd, stride=stride, padding=pad)
    dy = (torch.randn(y.shape, generator=g) * 0.5).to(dtype)
    y.backward(dy.double())
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(cuda)
    w_krsc = wt.permute(0, 2, 3, 1).contiguous().to(cuda)
    dy_nhwc = dy.permute(0, 2, 3, 1).contiguous().to(cuda)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11

    def close(got, ref, what, scale):
        err = (got.double().cpu() - ref).abs().max().item()
        assert err <= 4 * eps * scale + 1e-6, "%s: err %g (scale %g) geom %s" % (what, err, scale, geom)

    yh = F.conv2d_fwd(x_nhwc, w_krsc, stride, pad)
    close(yh.permute(0, 3, 1, 2), y.detach(), "fwd", float(y.abs().max()))
    # conv + BatchNorm statistics from the convolution epilogue == conv, then the separate statistics pass
    rm1, rv1 = torch.zeros(ko, device=cuda), torch.ones(ko, device=cuda)
    rm2, rv2 = torch.zeros(ko, device=cuda), torch.ones(ko, device=cuda)
    yf, mean_f, rstd_f = F.conv2d_fwd_bnstats(x_nhwc, w_krsc, stride, pad, rm1, rv1)
    assert torch.equal(yf, yh)
    _, mean_s, rstd_s = F.bn_fwd(yh, torch.ones(ko, device=cuda), torch.zeros(ko, device=cuda), rm2, rv2, relu=False)
    np.testing.assert_allclose(mean_f.cpu().numpy(), mean_s.cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rstd_f.cpu().numpy(), rstd_s.cpu().numpy(), rtol=1e-4)
    np.testing.assert_allclose(rv1.cpu().numpy(), rv2.cpu().numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(rm1.cpu().numpy(), rm2.cpu().numpy(), rtol=1e-4, atol=1e-6)
    dxh = F.conv2d_dgrad(dy_nhwc, w_krsc, (h, w), stride, pad)
    close(dxh.permute(0, 3, 1, 2), xd.grad, "dgrad", float(xd.grad.abs().max()) + 1e-3)
    add = (torch.randn(n, h, w, c, generator=g)).to(dtype).to(cuda)
    dxa = F.conv2d_dgrad(dy_nhwc, w_krsc, (h, w), stride, pad, addend=add)
    close(dxa.permute(0, 3, 1, 2), xd.grad + add.double().cpu().permute(0, 3, 1, 2), "dgrad+add",
          float(xd.grad.abs().max()) + 4.0)
    for sk in (1, None, 3):
        dwh = F.conv2d_wgrad(dy_nhwc, x_nhwc, (r, r), stride, pad, splitk=sk)
        got = dwh.permute(0, 3, 1, 2).double().cpu()
        err = (got - wd.grad).abs().max().item()
        assert err <= 1e-3 * float(wd.grad.abs().max()) + 1e-4, ("wgrad", sk, err, geom)


def test_conv_relu_bias_epilogue_and_errors(cuda):
    from deeplearningexamples_amd import functional as F, _cabi as C
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 6, 6, 64, generator=g).bfloat16().to(cuda)
    w = (torch.randn(128, 3, 3, 64, generator=g) * 0.05).bfloat16().to(cuda)
    b = torch.randn(128, generator=g).to(cuda)
    y = F.conv2d_fwd(x, w, 1, 1, bias=b, act=C.ACT_RELU)
    ref = torch.nn.functional.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.float().cpu().permute(0, 3, 1, 2),
                                     bias=b.cpu(), padding=1).clamp(min=0).permute(0, 2, 3, 1)
    np.testing.assert_allclose(y.float().cpu().numpy(), ref.numpy(), rtol=2e-2, atol=2e-2)
    with pytest.raises(ValueError):
        F.conv2d_fwd(torch.zeros(1, 4, 4, 3, dtype=torch.bfloat16, device=cuda),
                     torch.zeros(8, 3, 3, 3, dtype=torch.bfloat16, device=cuda), 1, 1)      # C % 8 != 0
