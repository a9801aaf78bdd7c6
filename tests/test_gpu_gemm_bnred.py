"""dle_gemm_expand_masked_bnred + dle_bn_bwd_finish: the backward reduction of a BatchNorm taken in the epilogue of the GEMM that
produces its input gradient (conv1's data gradient of the next ResNet bottleneck = the gradient of the previous block's output;
models/resnet.py:148-175 backward) -- dx bit-identical to dle_gemm(DLE_ACT_ADD_MASKED), the sums against dle_bn_bwd_reduce on
the same dx and a float64 restatement.  GPU only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("m,n,k", [(50176, 256, 64), (12544, 512, 128), (6272, 1024, 256), (4100, 256, 64)])
def test_reduction_in_the_producing_epilogue(cuda, m, n, k, dtype, monkeypatch):
    monkeypatch.setenv("DLE_GEMM_BNRED_K256", "1")        # (the K = 256 variant is off by default: slower than the two launches)
    from deeplearningexamples_amd import functional as F
    from deeplearningexamples_amd import _cabi as C
    gen = torch.Generator().manual_seed(m + n)
    g2 = (torch.randn(m, k, generator=gen) * 0.05).to(dtype).to(cuda)
    w = (torch.randn(k, n, generator=gen) / k ** 0.5).to(dtype).to(cuda)
    addend = (torch.randn(m, n, generator=gen) * 0.05).to(dtype).to(cuda)
    keep1 = torch.rand(m, n, generator=gen) < 0.5
    keep2 = torch.rand(m, n, generator=gen) < 0.6
    bits1 = torch.from_numpy(np.packbits(keep1.numpy().reshape(-1), bitorder="little")).to(cuda)
    bits2 = torch.from_numpy(np.packbits(keep2.numpy().reshape(-1), bitorder="little")).to(cuda)
    t2 = (torch.randn(m, n, generator=gen) * 1.5 + 0.3).to(dtype).to(cuda)
    mean2 = t2.float().mean(0)
    rstd2 = 1.0 / torch.sqrt(t2.float().var(0, unbiased=False) + 1e-5)
    ref = F.gemm(g2, w, m, n, k, True, False, act=C.ACT_ADD_MASKED, mask_src=addend, aux=bits1)
    dg, db = torch.full((n,), 5.0, device=cuda), torch.full((n,), 5.0, device=cuda)
    dx = F.gemm_masked_add_bnred(g2, w, m, n, k, addend, bits1, t2, bits2, mean2, rstd2, dg, db)
    assert dx is not None
    assert torch.equal(dx, ref)
    # the stand-alone reduction on the same gradient
    dg_s, db_s = torch.empty(n, device=cuda), torch.empty(n, device=cuda)
    gam = torch.ones(n, device=cuda)
    F.bn_bwd(dx, None, t2, mean2, rstd2, gam, dg_s, db_s, relu_mask=bits2)
    g = dx.double() * keep2.to(cuda).double()
    xh = (t2.double() - mean2.double()) * rstd2.double()
    ex_b, ex_g = g.sum(0), (g * xh).sum(0)
    mag_b, mag_g = g.abs().sum(0), (g * xh).abs().sum(0)
    assert torch.all((db.double() - ex_b).abs() <= 3e-6 * mag_b + 1e-7)
    assert torch.all((dg.double() - ex_g).abs() <= 3e-6 * mag_g + 1e-7)
    assert torch.allclose(db, db_s, rtol=1e-4, atol=1e-5 * float(mag_b.max()))
    assert torch.allclose(dg, dg_s, rtol=1e-4, atol=1e-5 * float(mag_g.max()))
    # fixed fold order: bit-identical on a second launch
    dg2, db2 = torch.empty_like(dg), torch.empty_like(db)
    dx2 = F.gemm_masked_add_bnred(g2, w, m, n, k, addend, bits1, t2, bits2, mean2, rstd2, dg2, db2)
    assert torch.equal(dx2, dx) and torch.equal(dg2, dg) and torch.equal(db2, db)


def test_declines_outside_the_envelope(cuda):
    from deeplearningexamples_amd import functional as F
    m, n, k = 4096, 128, 128                        # N < 2 K
    z = torch.zeros(m, n, device=cuda).half()
    b = torch.zeros(m * n // 8, dtype=torch.uint8, device=cuda)
    v = torch.zeros(n, device=cuda)
    assert F.gemm_masked_add_bnred(torch.zeros(m, k, device=cuda).half(), torch.zeros(k, n, device=cuda).half(), m, n, k, z, b, z, b,
                                   v, v + 1, v.clone(), v.clone()) is None
