"""dle_gemm_colsum: the data gradient of a linear layer through the activation derivative of the layer below (ReLU mask: dlrm/nn/
mlps.py:38-43 backward; stored GELU derivative: BERT/modeling.py:130-160) AND that layer's bias gradient (column sums of the
rounded output) from one launch -- against the separate launches
(dle_gemm with the same epilogue: bit-identical output; dle_colsum / a float64 sum for the column sums).  GPU only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("m,n,k", [(65536, 1024, 512), (16384, 512, 256), (4096, 256, 128), (1000, 264, 136), (130, 8, 8)])
def test_masked_dgrad_with_column_sums(cuda, m, n, k, dtype):
    from deeplearningexamples_amd import functional as F
    from deeplearningexamples_amd import _cabi as C
    gen = torch.Generator().manual_seed(m + n + k)
    g = (torch.randn(m, k, generator=gen) * 0.05).to(dtype).to(cuda)
    w = (torch.randn(k, n, generator=gen) / k ** 0.5).to(dtype).to(cuda)
    y = torch.relu(torch.randn(m, n, generator=gen)).to(dtype).to(cuda)          # forward activation of the layer below
    ref = F.gemm(g, w, m, n, k, True, False, out_dtype=dtype, act=C.ACT_RELU_BWD, mask_src=y)
    cs = torch.full((n,), 3.0, device=cuda)
    out = F.gemm_colsum(g, w, m, n, k, y, cs)
    assert out is not None
    assert torch.equal(out, ref)
    exact = out.to(torch.float64).sum(0)
    mag = out.to(torch.float64).abs().sum(0)
    assert torch.all((cs.to(torch.float64) - exact).abs() <= 2e-6 * mag + 1e-9)
    assert torch.allclose(cs, F.colsum(out), rtol=1e-4, atol=1e-5 * float(mag.max()))
    # masked columns really are zero where the activation was
    assert torch.all(out[y == 0] == 0)
    cs2 = torch.empty_like(cs)
    out2 = F.gemm_colsum(g, w, m, n, k, y, cs2)
    assert torch.equal(out2, out) and torch.equal(cs2, cs)                       # fixed fold order
    # the multiplicative epilogue (stored activation derivative), accumulating into the bias gradient
    refm = F.gemm(g, w, m, n, k, True, False, out_dtype=dtype, act=C.ACT_MUL, mask_src=y)
    cs3 = cs.clone()
    outm = F.gemm_colsum(g, w, m, n, k, y, cs3, act=C.ACT_MUL, accumulate=True)
    assert torch.equal(outm, refm)
    exm, magm = refm.to(torch.float64).sum(0), refm.to(torch.float64).abs().sum(0)
    assert torch.all((cs3.to(torch.float64) - cs.to(torch.float64) - exm).abs() <= 2e-6 * (magm + mag) + 1e-9)


def test_declines_outside_its_envelope(cuda):
    from deeplearningexamples_amd import functional as F
    g = torch.randn(64, 12, device=cuda).half()
    w = torch.randn(12, 20, device=cuda).half()
    y = torch.randn(64, 20, device=cuda).half()
    assert F.gemm_colsum(g, w, 64, 20, 12, y, torch.empty(20, device=cuda)) is None       # K, N not multiples of 8
    assert F.gemm_colsum(g.float(), w.float(), 64, 20, 12, y.float(), torch.empty(20, device=cuda)) is None
