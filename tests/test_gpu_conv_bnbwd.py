"""csrc/conv_bnbwd.hip: the BatchNorm backward (second pass) of a conv3 / bn3 bottleneck unit on the operand load of the unit's
1x1 data gradient (models/resnet.py:148-175 backward) -- bit-identical to dle_bn_bwd_apply + dle_gemm, and against a float64
restatement of BatchNorm's backward formula.  GPU only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("m,masked", [(50176, True), (5000, True), (4097, False), (200704, True)])
def test_fused_backward_is_the_two_launches(cuda, m, masked, dtype):
    from deeplearningexamples_amd import functional as F
    k, n = 256, 64
    gen = torch.Generator().manual_seed(m)
    t = torch.randn(m, k, generator=gen).to(dtype).to(cuda)
    dy = (torch.randn(m, k, generator=gen) * 0.01).to(dtype).to(cuda)
    w = (torch.randn(k, n, generator=gen) / 16).to(dtype).to(cuda)
    gamma = (torch.rand(k, generator=gen) + 0.5).to(cuda)
    mean = t.float().mean(0)
    rstd = 1.0 / torch.sqrt(t.float().var(0, unbiased=False) + 1e-5)
    bits = None
    if masked:
        keep = torch.rand(m, k, generator=gen) < 0.6
        bits = torch.from_numpy(np.packbits(keep.numpy().reshape(-1), bitorder="little")).to(cuda)
    # the two launches
    dg_s, db_s = torch.empty(k, device=cuda), torch.empty(k, device=cuda)
    dt_s, _ = F.bn_bwd(dy, None, t, mean, rstd, gamma, dg_s, db_s, relu_mask=bits)
    dx_s = F.gemm(dt_s, w, m, n, k, True, False)
    # the fused pass
    dg, db = torch.empty(k, device=cuda), torch.empty(k, device=cuda)
    out = F.bn_bwd_conv1x1_dgrad(dy, t, mean, rstd, gamma, dg, db, w, relu_mask=bits)
    assert out is not None
    dt, dx, _ = out
    assert torch.equal(dg, dg_s) and torch.equal(db, db_s)
    assert torch.equal(dt, dt_s)
    assert torch.equal(dx, dx_s)
    # float64 restatement of the formula on the same 16-bit inputs
    g = dy.double() * (keep.to(cuda).double() if masked else 1.0)
    xh = (t.double() - mean.double()) * rstd.double()
    ref = gamma.double() * rstd.double() * (g - g.mean(0) - xh * (g * xh).mean(0))
    step = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    assert (dt.double() - ref).abs().max().item() <= 1.5 * step * ref.abs().max().item() + 1e-6


def test_outside_the_envelope_nothing_is_launched(cuda):
    from deeplearningexamples_amd import functional as F
    t = torch.randn(8192, 128, device=cuda).half()
    out = F.bn_bwd_conv1x1_dgrad(t, t, torch.zeros(128, device=cuda), torch.ones(128, device=cuda), torch.ones(128, device=cuda),
                                 torch.empty(128, device=cuda), torch.empty(128, device=cuda), torch.randn(128, 32, device=cuda).half())
    assert out is None


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_second_batchnorm_reduction_rides_along(cuda, dtype):
    """bnred: dx [M, 64] is the gradient entering bn2 of the bottleneck -- its backward reduction (sum g, sum g xhat under bn2's
    ReLU keep bits) comes out of the same kernel; dt / dx stay bit-identical, the sums match dle_bn_bwd_reduce on dx."""
    from deeplearningexamples_amd import functional as F
    m, k, n = 50176 + 17, 256, 64
    gen = torch.Generator().manual_seed(5)
    t = torch.randn(m, k, generator=gen).to(dtype).to(cuda)
    dy = (torch.randn(m, k, generator=gen) * 0.01).to(dtype).to(cuda)
    w = (torch.randn(k, n, generator=gen) / 16).to(dtype).to(cuda)
    gamma = (torch.rand(k, generator=gen) + 0.5).to(cuda)
    mean, rstd = t.float().mean(0), 1.0 / torch.sqrt(t.float().var(0, unbiased=False) + 1e-5)
    keep = torch.rand(m, k, generator=gen) < 0.6
    bits = torch.from_numpy(np.packbits(keep.numpy().reshape(-1), bitorder="little")).to(cuda)
    t2 = (torch.randn(m, n, generator=gen) + 0.2).to(dtype).to(cuda)
    keep2 = torch.rand(m, n, generator=gen) < 0.5
    bits2 = torch.from_numpy(np.packbits(keep2.numpy().reshape(-1), bitorder="little")).to(cuda)
    mean2, rstd2 = t2.float().mean(0), 1.0 / torch.sqrt(t2.float().var(0, unbiased=False) + 1e-5)
    dg, db = torch.empty(k, device=cuda), torch.empty(k, device=cuda)
    plain = F.bn_bwd_conv1x1_dgrad(dy, t, mean, rstd, gamma, dg, db, w, relu_mask=bits)
    dg2, db2 = torch.full((n,), 9.0, device=cuda), torch.full((n,), 9.0, device=cuda)
    out = F.bn_bwd_conv1x1_dgrad(dy, t, mean, rstd, gamma, dg, db, w, relu_mask=bits, bnred=(t2, bits2, mean2, rstd2, dg2, db2))
    assert out[2] and not plain[2]
    assert torch.equal(out[0], plain[0]) and torch.equal(out[1], plain[1])
    dg_s, db_s = torch.empty(n, device=cuda), torch.empty(n, device=cuda)
    F.bn_bwd(out[1], None, t2, mean2, rstd2, torch.ones(n, device=cuda), dg_s, db_s, relu_mask=bits2)
    g = out[1].double() * keep2.to(cuda).double()
    xh = (t2.double() - mean2.double()) * rstd2.double()
    assert torch.all((db2.double() - g.sum(0)).abs() <= 3e-6 * g.abs().sum(0) + 1e-8)
    assert torch.all((dg2.double() - (g * xh).sum(0)).abs() <= 3e-6 * (g * xh).abs().sum(0) + 1e-8)
    assert torch.allclose(db2, db_s, rtol=1e-4, atol=1e-5 * float(g.abs().sum(0).max()))
    assert torch.allclose(dg2, dg_s, rtol=1e-4, atol=1e-5 * float((g * xh).abs().sum(0).max()))
