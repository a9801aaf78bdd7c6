"""The halo-tile 3x3 weight gradient (csrc/conv3x3_wgrad.hip) against the split-K implicit GEMM it replaces and against fp32
torch (cuDNN's bwd-filter behind the 3x3 nn.Conv2d of the bottleneck, models/resnet.py:126,148-175).  Both HIP paths multiply the
same 16-bit operands and accumulate in fp32: they differ by summation order only.  GPU only."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(2, 8, 8, 64, 64), (3, 14, 14, 128, 64), (1, 5, 9, 64, 128), (4, 56, 56, 64, 64), (2, 7, 7, 512, 512), (5, 28, 28, 128, 128),
          (1, 1, 1, 64, 64), (2, 3, 130, 64, 64)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n,h,w,c,ko", SHAPES)
def test_halo_wgrad_matches_gemm_path_and_torch(cuda, dtype, n, h, w, c, ko):
    from deeplearningexamples_amd import _cabi as C
    from deeplearningexamples_amd import functional as F
    g = torch.Generator().manual_seed(n * 1000 + h)
    x = torch.randn((n, h, w, c), generator=g).to(dtype).to(cuda)
    dy = torch.randn((n, h, w, ko), generator=g).to(dtype).to(cuda)
    old = C.lib().dle_conv3x3_wgrad_mode(1)
    try:
        a = torch.full((ko, 3, 3, c), 7.0, dtype=torch.float32, device=cuda)
        F.conv2d_wgrad(dy, x, (3, 3), 1, 1, out=a)
        a2 = torch.zeros_like(a)
        F.conv2d_wgrad(dy, x, (3, 3), 1, 1, out=a2)
        assert torch.equal(a, a2)                                         # fixed summation order
        F.conv2d_wgrad(dy, x, (3, 3), 1, 1, out=a2, accumulate=True)
        assert torch.allclose(a2, 2 * a, rtol=1e-6, atol=1e-6)
        C.lib().dle_conv3x3_wgrad_mode(0)
        b = torch.empty_like(a)
        F.conv2d_wgrad(dy, x, (3, 3), 1, 1, out=b)
    finally:
        C.lib().dle_conv3x3_wgrad_mode(old)
    torch.cuda.synchronize()
    scale = float(b.abs().max()) + 1e-6
    assert float((a - b).abs().max()) <= 2e-5 * scale + 1e-4, (float((a - b).abs().max()), scale)
    if n * h * w <= 4096:
        ref = torch.nn.grad.conv2d_weight(x.float().cpu().permute(0, 3, 1, 2), (ko, c, 3, 3), dy.float().cpu().permute(0, 3, 1, 2),
                                          stride=1, padding=1).permute(0, 2, 3, 1)
        assert float((a.cpu() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-4
