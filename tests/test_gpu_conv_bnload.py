"""conv + BatchNorm + ReLU as ONE unit (csrc/conv_bnload.hip): the producer's BatchNorm-apply (+ residual) + ReLU on the operand load
of the consuming 1x1 convolution, against the two-launch sequence it replaces (dle_bn_fwd_apply, then dle_conv2d_fwd_colstats) --
models/resnet.py:148-175 (relu(bn2(.)) -> conv3; relu(bn3(.) + residual) -> the next block's conv1), models/common.py:31-128.
The applied activation y and its keep bits must be BIT-IDENTICAL (same fp32 expression, same rounding point); the convolution
output is bit-identical where the two-launch path runs the same streaming kernel (channel-widening shapes) and within one 16-bit
rounding step of it elsewhere (a different fp32 summation order of the same products).  GPU only."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# (rows M as n x h x w, K, N, residual)
SHAPES = [((4, 32, 32), 64, 256, False), ((4, 32, 32), 128, 256, False), ((2, 56, 56), 256, 256, False),
          ((2, 56, 56), 256, 64, True), ((2, 56, 56), 256, 128, True), ((3, 40, 40), 64, 64, True), ((8, 28, 28), 128, 128, False),
          ((8, 28, 28), 512, 128, True)]          # K = 512: the 28 x 28 stage's conv1 consuming bn3 + identity (one 8-wave workgroup per CU)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("nhw,k,n,with_res", SHAPES)
def test_fused_unit_equals_apply_then_conv(cuda, dtype, nhw, k, n, with_res, monkeypatch):
    from deeplearningexamples_amd import functional as F
    if k == 512:
        monkeypatch.setenv("DLE_CONV_BNLOAD_K512", "1")          # (opt-in instantiation: no gain measured at batch 256)
    g = torch.Generator().manual_seed(k * 31 + n)
    shape = nhw + (k,)
    t = torch.randn(shape, generator=g).to(dtype).to(cuda)
    res = torch.randn(shape, generator=g).to(dtype).to(cuda) if with_res else None
    w = (torch.randn((n, 1, 1, k), generator=g) * 0.1).to(dtype).to(cuda)
    mean = (torch.randn(k, generator=g) * 0.1).to(cuda)
    rstd = (torch.rand(k, generator=g) + 0.5).to(cuda)
    gamma = (torch.rand(k, generator=g) + 0.5).to(cuda)
    beta = (torch.randn(k, generator=g) * 0.1).to(cuda)
    rm = [torch.zeros(n, device=cuda) for _ in range(2)]
    rv = [torch.ones(n, device=cuda) for _ in range(2)]
    fused = F.conv1x1_bnload_fwd(t, res, w, mean, rstd, gamma, beta, rm[0], rv[0])
    assert fused is not None, "inside the fused kernel's envelope"
    out, y, bits, mo, ro = fused
    y_ref, bits_ref = F.bn_fwd_apply(t, mean, rstd, gamma, beta, residual=res, relu=True, want_mask=True)
    out_ref, mr, rr = F.conv2d_fwd_bnstats(y_ref, w, 1, 0, rm[1], rv[1])
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref) and torch.equal(bits, bits_ref)
    if n >= 2 * k and n % 128 == 0:
        assert torch.equal(out, out_ref)
    else:
        d = (out.float() - out_ref.float()).abs()
        step = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
        assert float((d - (step * out_ref.float().abs() + 1e-3)).max()) <= 0
        assert float((d > 0).float().mean()) < 0.02
    assert torch.allclose(mo, mr, atol=2e-3, rtol=1e-3) and torch.allclose(ro, rr, rtol=2e-3)
    assert torch.allclose(rm[0], rm[1], atol=1e-3) and torch.allclose(rv[0], rv[1], rtol=1e-3)


def test_outside_the_envelope_declines(cuda):
    from deeplearningexamples_amd import functional as F
    t = torch.randn((2, 8, 8, 512), device=cuda).bfloat16()
    w = torch.randn((128, 1, 1, 512), device=cuda).bfloat16()
    v = torch.ones(512, device=cuda)
    assert F.conv1x1_bnload_fwd(t, None, w, v, v, v, v) is None                  # K = 512, M = 128
    t = torch.randn((2, 56, 56, 256), device=cuda).bfloat16()
    w = torch.randn((1024, 1, 1, 256), device=cuda).bfloat16()
    v = torch.ones(256, device=cuda)
    assert F.conv1x1_bnload_fwd(t, None, w, v, v, v, v) is None                  # 8 column tiles: the two-launch sequence wins


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_rn50_step_with_and_without_the_fused_units(cuda, dtype, monkeypatch):
    """The train step with conv + BN + ReLU fused where the consumer is a 1x1 convolution vs the step on stand-alone apply passes,
    on the damped-residual fixture network of tests/test_gpu_rn50_step.py (a random-init ResNet-50 is chaotic from step to step):
    the first loss is IDENTICAL (y / keep bits are bit-identical, the logits see only the summation order of the channel-narrowing
    convolutions), the following ones agree to the fixture's 16-bit noise level."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import resnet_oracle as RO
    import test_gpu_rn50_step as T
    c = RO.RN50_STEP_CONFIG
    state = RO.seeded_state(c["seed"])
    x, y = RO.seeded_batch(c["seed"] + 1, 16, 64)
    losses = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("DLE_RN50_FUSE_BN", mode)
        model, tr = T._build(cuda, dtype, c["lr"], state)
        assert tr.fuse_bn == (mode == "1")
        losses[mode] = [float(tr.train_step(x.to(cuda), y.to(cuda)).item()) for _ in range(3)]
    print(dtype, losses)
    assert abs(losses["1"][0] - losses["0"][0]) <= 2e-4 * abs(losses["0"][0])
    for a, b in zip(losses["1"], losses["0"]):
        assert abs(a - b) <= 3e-3 * abs(b)


def _bn_vectors(g, k, cuda):
    return ((torch.randn(k, generator=g) * 0.1).to(cuda), (torch.rand(k, generator=g) + 0.5).to(cuda),
            (torch.rand(k, generator=g) + 0.5).to(cuda), (torch.randn(k, generator=g) * 0.1).to(cuda))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("nhw,c", [((2, 28, 28), 512), ((3, 14, 14), 1024), ((5, 7, 7), 2048), ((2, 9, 9), 24)])
def test_dual_apply_equals_branch_apply_then_apply(cuda, dtype, nhw, c):
    """relu(bn3(t) + bn_ds(t_ds)) in one pass (bn_apply2_pf_kernel: the downsample branch's BatchNorm on the residual's load,
    models/resnet.py:166-173) == the branch's stand-alone apply followed by bn3's apply with a residual tensor, bit for bit."""
    from deeplearningexamples_amd import functional as F
    g = torch.Generator().manual_seed(c)
    t = torch.randn(nhw + (c,), generator=g).to(dtype).to(cuda)
    td = torch.randn(nhw + (c,), generator=g).to(dtype).to(cuda)
    bn, bnr = _bn_vectors(g, c, cuda), _bn_vectors(g, c, cuda)
    res, _ = F.bn_fwd_apply(td, *bnr, relu=False)
    y_ref, bits_ref = F.bn_fwd_apply(t, *bn, residual=res, relu=True, want_mask=True)
    y, bits = F.bn_fwd_apply(t, *bn, residual=td, relu=True, want_mask=True, residual_bn=bnr)
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref) and torch.equal(bits, bits_ref)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("nhw,k,n", [((2, 56, 56), 256, 64), ((2, 56, 56), 256, 128), ((3, 40, 40), 64, 64), ((4, 32, 32), 128, 128)])
def test_fused_unit_with_branch_batchnorm_on_the_residual_load(cuda, dtype, nhw, k, n):
    """conv_bnload_kernel<RES = 2> == the branch's stand-alone apply followed by the RES = 1 kernel: y, keep bits, out, statistics
    all bit-identical (the same kernel after the residual's rounding point)."""
    from deeplearningexamples_amd import functional as F
    g = torch.Generator().manual_seed(k * 17 + n)
    t = torch.randn(nhw + (k,), generator=g).to(dtype).to(cuda)
    td = torch.randn(nhw + (k,), generator=g).to(dtype).to(cuda)
    w = (torch.randn((n, 1, 1, k), generator=g) * 0.1).to(dtype).to(cuda)
    bn, bnr = _bn_vectors(g, k, cuda), _bn_vectors(g, k, cuda)
    res, _ = F.bn_fwd_apply(td, *bnr, relu=False)
    ref = F.conv1x1_bnload_fwd(t, res, w, *bn)
    got = F.conv1x1_bnload_fwd(t, td, w, *bn, res_bn=bnr)
    torch.cuda.synchronize()
    assert ref is not None and got is not None
    for a, b in zip(got, ref):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.bfloat16])
def test_rn50_step_with_and_without_the_downsample_batchnorm_on_load(cuda, dtype, monkeypatch):
    """DLE_RN50_FUSE_DSBN=1 (the downsample branch's BatchNorm applied where bn3's apply loads the residual) vs 0 (its own pass):
    every activation of the forward pass is bit-identical (the kernel-level tests above); the losses agree to the last fp32 digits
    (the step's fp32 loss / gradient reductions are not order-deterministic from run to run)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import resnet_oracle as RO
    import test_gpu_rn50_step as T
    c = RO.RN50_STEP_CONFIG
    state = RO.seeded_state(c["seed"])
    x, y = RO.seeded_batch(c["seed"] + 1, 16, 64)
    losses = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("DLE_RN50_FUSE_DSBN", mode)
        model, tr = T._build(cuda, dtype, c["lr"], state)
        assert tr.fuse_dsbn == (mode == "1")
        losses[mode] = [float(tr.train_step(x.to(cuda), y.to(cuda)).item()) for _ in range(3)]
    print(dtype, losses)
    for a, b in zip(losses["1"], losses["0"]):
        assert abs(a - b) <= 2e-6 * abs(b)
