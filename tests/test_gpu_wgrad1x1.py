"""The streaming 1x1 weight gradient (csrc/wgrad1x1.hip) against the split-K tile GEMM it replaces and fp64 torch: dw = dy^T x
(cuDNN's bwd-filter behind the 1x1 nn.Conv2d of models/resnet.py:148-175).  Same 16-bit products, fp32 accumulation: the two HIP
paths differ by summation order only.  GPU only."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(256, 64), (64, 256), (64, 64), (128, 256), (256, 128), (512, 128), (128, 512)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("ko,c", SHAPES)
@pytest.mark.parametrize("m", [8192, 50000])
def test_streaming_wgrad_matches_gemm_and_fp64(cuda, dtype, ko, c, m):
    from deeplearningexamples_amd import functional as F
    g = torch.Generator().manual_seed(ko + c + m)
    dy = torch.randn((m, ko), generator=g).to(dtype).to(cuda)
    x = torch.randn((m, c), generator=g).to(dtype).to(cuda)
    a = torch.full((ko, c), 5.0, dtype=torch.float32, device=cuda)
    assert F.wgrad1x1(dy, x, a), "inside the streaming kernel's envelope"
    a2 = torch.zeros_like(a)
    assert F.wgrad1x1(dy, x, a2) and torch.equal(a, a2)                       # fixed summation order
    assert F.wgrad1x1(dy, x, a2, accumulate=True) and torch.allclose(a2, 2 * a, rtol=1e-6, atol=1e-5)
    b = F.gemm(dy, x, ko, c, m, False, False, out=torch.empty_like(a), splitk=F.pick_splitk(ko, c, m, target_blocks=1024))
    ref = (dy.double().t() @ x.double())
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    assert float((a.double() - ref).abs().max()) <= 3e-5 * scale + 1e-3
    assert float((a - b).abs().max()) <= 6e-5 * scale + 2e-3


def test_outside_the_envelope_declines(cuda):
    from deeplearningexamples_amd import functional as F
    dy = torch.randn((8192, 1024), device=cuda).bfloat16()
    x = torch.randn((8192, 256), device=cuda).bfloat16()
    assert not F.wgrad1x1(dy, x, torch.empty((1024, 256), device=cuda))           # 1 MB of output: split-K GEMM
    assert not F.wgrad1x1(dy[:4096, :256].contiguous(), x[:4096, :64].contiguous(), torch.empty((256, 64), device=cuda))   # short contraction
