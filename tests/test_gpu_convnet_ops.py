"""ResNet HBM-bound kernels (BN fwd/bwd + ReLU + residual, max/avg pooling, label-smoothing CE, layout
conversion) vs the same torch ops on the CPU in float64/float32.  GPU only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DT = [torch.bfloat16, torch.float16]


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(4, 64, 9, 7), (2, 256, 5, 5), (3, 8, 6, 6), (2, 2048, 2, 2)])
@pytest.mark.parametrize("relu,res", [(True, False), (True, True), (False, False)])
def test_batchnorm_fwd_bwd(cuda, dtype, shape, relu, res):
    from deeplearningexamples_amd import functional as F
    n, c, h, w = shape
    g = torch.Generator().manual_seed(c + h)
    x = (torch.randn(shape, generator=g) * 1.5 + 0.3).to(dtype)
    r = torch.randn(shape, generator=g).to(dtype) if res else None
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.2
    dy = torch.randn(shape, generator=g).to(dtype)
    xr = x.double().requires_grad_()
    gr, br = gamma.double().requires_grad_(), beta.double().requires_grad_()
    rr = r.double().requires_grad_() if res else None
    rm, rv = torch.zeros(c, dtype=torch.float64), torch.ones(c, dtype=torch.float64)
    yr = torch.nn.functional.batch_norm(xr, rm, rv, gr, br, training=True, momentum=0.1, eps=1e-5)
    if res:
        yr = yr + rr
    if relu:
        yr = torch.relu(yr)
    yr.backward(dy.double())
    run_m, run_v = torch.zeros(c, device=cuda), torch.ones(c, device=cuda)
    y, mean, rstd = F.bn_fwd(_nhwc(x).to(cuda), gamma.to(cuda), beta.to(cuda), run_m, run_v,
                             residual=_nhwc(r).to(cuda) if res else None, relu=relu)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    ref_y = _nhwc(yr.detach())
    assert (y.double().cpu() - ref_y).abs().max() <= 3 * eps * max(1.0, float(ref_y.abs().max()))
    np.testing.assert_allclose(run_m.cpu().numpy(), rm.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(run_v.cpu().numpy(), rv.numpy(), rtol=1e-4, atol=1e-5)
    dgamma, dbeta = torch.empty(c, device=cuda), torch.empty(c, device=cuda)
    dx, gskip = F.bn_bwd(_nhwc(dy).to(cuda), y if relu else None, _nhwc(x).to(cuda), mean, rstd, gamma.to(cuda),
                         dgamma, dbeta, want_skip_grad=res)
    m = n * h * w
    tol = 6 * eps * np.sqrt(m)
    assert (dgamma.double().cpu() - gr.grad).abs().max() <= tol * max(1.0, float(gr.grad.abs().max()))
    assert (dbeta.double().cpu() - br.grad).abs().max() <= tol * max(1.0, float(br.grad.abs().max()))
    ref_dx = _nhwc(xr.grad)
    assert (dx.double().cpu() - ref_dx).abs().max() <= 6 * eps * max(1.0, float(ref_dx.abs().max()))
    if res:
        assert (gskip.double().cpu() - _nhwc(rr.grad)).abs().max() <= 2 * eps * float(rr.grad.abs().max())
    if relu:
        # the bit-packed ReLU mask path (1 bit / element instead of re-reading y) gives the same gradients bit for bit
        run_m2, run_v2 = torch.zeros(c, device=cuda), torch.ones(c, device=cuda)
        y2, mean2, rstd2, mask = F.bn_fwd(_nhwc(x).to(cuda), gamma.to(cuda), beta.to(cuda), run_m2, run_v2,
                                          residual=_nhwc(r).to(cuda) if res else None, relu=True, want_mask=True)
        assert torch.equal(y2, y)
        bits = ((mask.to(torch.int32).unsqueeze(1) >> torch.arange(8, device=cuda, dtype=torch.int32)) & 1).reshape(y.shape)
        assert torch.equal(bits.bool(), y > 0)
        dgamma2, dbeta2 = torch.empty(c, device=cuda), torch.empty(c, device=cuda)
        dx2, gskip2 = F.bn_bwd(_nhwc(dy).to(cuda), None, _nhwc(x).to(cuda), mean, rstd, gamma.to(cuda), dgamma2, dbeta2,
                               want_skip_grad=res, relu_mask=mask)
        assert torch.equal(dx2, dx) and torch.equal(dgamma2, dgamma) and torch.equal(dbeta2, dbeta)
        if res:
            assert torch.equal(gskip2, gskip)


@pytest.mark.parametrize("dtype", DT)
def test_pools_and_layout(cuda, dtype):
    from deeplearningexamples_amd import functional as F
    g = torch.Generator().manual_seed(4)
    x = torch.relu(torch.randn(3, 16, 13, 11, generator=g)).to(dtype)          # many exact ties at 0 after ReLU
    xr = x.float().requires_grad_()
    yr = torch.nn.functional.max_pool2d(xr, 3, 2, 1)
    dy = torch.randn(yr.shape, generator=g).to(dtype)
    yr.backward(dy.float())
    y, am = F.maxpool_fwd(_nhwc(x).to(cuda))
    assert torch.equal(y.cpu(), _nhwc(yr.detach()).to(dtype))
    dx = F.maxpool_bwd(_nhwc(dy).to(cuda), am, (13, 11))
    np.testing.assert_allclose(dx.float().cpu().numpy(), _nhwc(xr.grad).to(dtype).float().numpy(), rtol=1e-2, atol=1e-2)
    for hw in ((8, 10), (2, 2), (1, 5)):                     # even sizes (the 112 x 112 stem), degenerate ones
        x2 = torch.relu(torch.randn(2, 8, *hw, generator=g)).to(dtype)
        x2r = x2.float().requires_grad_()
        y2r = torch.nn.functional.max_pool2d(x2r, 3, 2, 1)
        dy2 = torch.randn(y2r.shape, generator=g).to(dtype)
        y2r.backward(dy2.float())
        y2, am2 = F.maxpool_fwd(_nhwc(x2).to(cuda))
        assert torch.equal(y2.cpu(), _nhwc(y2r.detach()).to(dtype))
        dx2 = F.maxpool_bwd(_nhwc(dy2).to(cuda), am2, hw)
        np.testing.assert_allclose(dx2.float().cpu().numpy(), _nhwc(x2r.grad).to(dtype).float().numpy(), rtol=1e-2, atol=1e-2)
    a = torch.randn(5, 64, 7, 7, generator=g).to(dtype)
    p = F.avgpool_fwd(_nhwc(a).to(cuda))
    np.testing.assert_allclose(p.float().cpu().numpy(), a.float().mean((2, 3)).to(dtype).float().numpy(), rtol=1e-2, atol=1e-2)
    gp = torch.randn(5, 64, generator=g).to(dtype)
    ga = F.avgpool_bwd(gp.to(cuda), (7, 7))
    np.testing.assert_allclose(ga.float().cpu().numpy(), (gp.float() / 49)[:, None, None, :].expand(5, 7, 7, 64).to(dtype).float().numpy(),
                               rtol=1e-2, atol=1e-4)
    img = torch.randn(2, 3, 10, 12, generator=g)
    z = F.nchw_to_nhwc(img.to(cuda), dtype, 8).cpu()
    assert torch.equal(z[..., :3], _nhwc(img).to(dtype)) and (z[..., 3:] == 0).all()


@pytest.mark.parametrize("smoothing,ignore", [(0.1, -100), (0.0, -1)])
def test_softmax_xent(cuda, smoothing, ignore):
    from deeplearningexamples_amd import functional as F
    g = torch.Generator().manual_seed(8)
    rows, classes = 37, 1000
    x = torch.randn(rows, classes, generator=g) * 3
    t = torch.randint(0, classes, (rows,), generator=g)
    if ignore == -1:
        t[::3] = -1
    xr = x.double().requires_grad_()
    lp = torch.log_softmax(xr, -1)
    valid = t != ignore
    nll = -lp[valid].gather(1, t[valid, None]).squeeze(1)
    loss_r = ((1 - smoothing) * nll + smoothing * (-lp[valid].mean(-1))).mean()
    loss_r.backward()
    loss, dl = F.softmax_xent(x.to(cuda), t.to(cuda), smoothing, ignore, grad_scale=torch.tensor([4.0], device=cuda),
                              grad_dtype=torch.float32)
    assert abs(loss.item() - loss_r.item()) < 1e-5 * max(1, abs(loss_r.item()))
    np.testing.assert_allclose(dl.cpu().numpy(), (xr.grad * 4).float().numpy(), rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("gdtype", [torch.float32, torch.bfloat16, torch.float16])
def test_softmax_xent_wide_rows_mlm_head(cuda, gdtype):
    """The MLM-head shape: 30522 classes inside rows padded to 30528 (logits AND gradient), ignore_index -1, loss-scaled
    gradient -- the workgroup-per-row kernel (online max / sum, 16-byte accesses); padded gradient columns are zero."""
    from deeplearningexamples_amd import functional as F
    g = torch.Generator().manual_seed(12)
    rows, classes, ld = 45, 30522, 30528
    buf = torch.randn(rows, ld, generator=g) * 4
    buf[:, classes:] = 1e4                                   # garbage in the padding must not enter the softmax
    buf[3, 17] = 60.0                                        # a dominant logit late in the running-maximum order
    buf[4, classes - 1] = 55.0                               # ... and in the ragged last group of four
    t = torch.randint(0, classes, (rows,), generator=g)
    t[::4] = -1
    t[4] = classes - 1
    x = buf[:, :classes]
    xr = x.double().requires_grad_()
    lp = torch.log_softmax(xr, -1)
    valid = t != -1
    loss_r = (-lp[valid].gather(1, t[valid, None]).squeeze(1)).mean()
    loss_r.backward()
    xd = buf.to(cuda)[:, :classes]                           # row stride 30528, 30522 classes
    loss, dl = F.softmax_xent(xd, t.to(cuda), 0.0, -1, grad_scale=torch.tensor([128.0], device=cuda), grad_dtype=gdtype,
                              ld_out=ld)
    assert abs(loss.item() - loss_r.item()) < 2e-5 * max(1, abs(loss_r.item()))
    ref = (xr.grad * 128).float().numpy()
    got = dl.float().cpu().numpy()
    tol = dict(rtol=1e-4, atol=1e-6) if gdtype == torch.float32 else dict(rtol=1e-2, atol=1e-5)
    np.testing.assert_allclose(got[:, :classes], ref, **tol)
    assert not got[:, classes:].any()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n,h,w,k,nn", [(2, 56, 56, 128, 256), (8, 28, 28, 256, 512), (2, 48, 48, 64, 128)])
def test_gemm_add_upsampled2_equals_materialised_path(cuda, dtype, n, h, w, k, nn):
    """conv1's data gradient + the stride-2 downsample branch's compact gradient added at the even pixels (csrc/gemm_expand.hip,
    dle_gemm_expand_add_up2) == the same GEMM with the zero-stuffed tensor as a DLE_ACT_ADD addend (models/resnet.py:148-175)."""
    from deeplearningexamples_amd import _cabi as C
    from deeplearningexamples_amd import functional as F
    g = torch.Generator().manual_seed(h * 7 + k)
    m = n * h * w
    a = torch.randn((m, k), generator=g).to(dtype).to(cuda)
    b = (torch.randn((k, nn), generator=g) * 0.1).to(dtype).to(cuda)
    compact = torch.randn((n, h // 2, w // 2, nn), generator=g).to(dtype).to(cuda)
    fused = F.gemm_add_upsampled2(a, b, compact, (h, w))
    assert fused is not None, "inside the streaming kernel's envelope"
    full = F.upsample_zero(compact, (h, w), 2)
    ref = F.gemm(a, b, m, nn, k, True, False, act=C.ACT_ADD, mask_src=full.view(m, nn))
    torch.cuda.synchronize()
    assert torch.equal(fused, ref)
    assert F.gemm_add_upsampled2(a[:, :32].contiguous(), b[:32].contiguous(), compact, (h, w)) is None     # K = 32: declined


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n,h,w,c", [(2, 112, 112, 64), (3, 8, 12, 64), (1, 2, 2, 8), (2, 30, 18, 128)])
def test_bn_relu_maxpool_fused_equals_two_passes_and_pool_backward_patch(cuda, dtype, n, h, w, c):
    """bn1 -> relu -> maxpool in one pass == dle_bn_fwd_apply then dle_maxpool_fwd, bit for bit (models/resnet.py:318-322); and the
    2x2-patch pooling backward == the per-pixel form."""
    from deeplearningexamples_amd import functional as F
    g = torch.Generator().manual_seed(h * 13 + c)
    t = torch.randn((n, h, w, c), generator=g).to(dtype).to(cuda)
    mean = (torch.randn(c, generator=g) * 0.2).to(cuda)
    rstd = (torch.rand(c, generator=g) + 0.5).to(cuda)
    gamma = (torch.rand(c, generator=g) + 0.5).to(cuda)
    beta = (torch.randn(c, generator=g) * 0.3).to(cuda)
    y, am, mask = F.bn_relu_maxpool_fwd(t, mean, rstd, gamma, beta)
    a0, mask_ref = F.bn_fwd_apply(t, mean, rstd, gamma, beta, relu=True, want_mask=True)
    y_ref, am_ref = F.maxpool_fwd(a0)
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref) and torch.equal(am.view(-1), am_ref.view(-1)) and torch.equal(mask, mask_ref)
    dy = torch.randn(y.shape, generator=g).to(dtype).to(cuda)
    dx = F.maxpool_bwd(dy, am, (h, w))
    yy, idx = torch.nn.functional.max_pool2d(a0.float().cpu().permute(0, 3, 1, 2), 3, 2, 1, return_indices=True)
    gr = torch.zeros((n, c, h * w))
    gr.scatter_add_(2, idx.reshape(n, c, -1), dy.float().cpu().permute(0, 3, 1, 2).reshape(n, c, -1))
    ref = gr.reshape(n, c, h, w).permute(0, 2, 3, 1)
    assert torch.allclose(dx.float().cpu(), ref.to(dtype).float(), atol=0.02, rtol=0.01)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,c", [(50176, 1024), (3000, 2048), (777, 24)])
def test_bn_bwd_reduce2_equals_two_reductions(cuda, dtype, m, c):
    """bn3 and the downsample BatchNorm of a bottleneck's first block receive the same gradient under the same keep bits
    (models/resnet.py:166-173): the dual reduction reads dy and the mask once and leaves both (dgamma, dbeta) pairs BIT-IDENTICAL to
    two dle_bn_bwd_reduce launches."""
    from deeplearningexamples_amd import functional as F
    g = torch.Generator().manual_seed(m + c)
    dy = torch.randn(m, c, generator=g).to(dtype).to(cuda)
    x1 = torch.randn(m, c, generator=g).to(dtype).to(cuda)
    x2 = torch.randn(m, c, generator=g).to(dtype).to(cuda)
    bits = torch.randint(0, 256, (m * c // 8,), generator=g, dtype=torch.uint8).to(cuda)
    st = [(torch.randn(c, generator=g) * 0.1).to(cuda) for _ in range(2)] + [(torch.rand(c, generator=g) + 0.5).to(cuda) for _ in range(2)]
    mean1, mean2, rstd1, rstd2 = st
    gamma = torch.ones(c, device=cuda)
    ref = []
    for x, mu, rs in ((x1, mean1, rstd1), (x2, mean2, rstd2)):
        dg, db = torch.empty(c, device=cuda), torch.empty(c, device=cuda)
        F.bn_bwd(dy, None, x, mu, rs, gamma, dg, db, relu_mask=bits)
        ref.append((dg, db))
    out = [torch.empty(c, device=cuda) for _ in range(4)]
    F.bn_bwd_reduce2(dy, bits, x1, mean1, rstd1, out[0], out[1], x2, mean2, rstd2, out[2], out[3])
    torch.cuda.synchronize()
    assert torch.equal(out[0], ref[0][0]) and torch.equal(out[1], ref[0][1])
    assert torch.equal(out[2], ref[1][0]) and torch.equal(out[3], ref[1][1])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n,h,w,c", [(2, 8, 8, 64), (3, 12, 20, 16), (1, 6, 4, 8), (4, 112, 112, 64)])
def test_pool_bn_bwd_equals_pooling_backward_then_batchnorm_backward(cuda, dtype, n, h, w, c):
    """The stem's backward chain (models/resnet.py:318-322: maxpool <- relu <- bn1) with the pooling gradient gathered inside the
    two BatchNorm passes == dle_maxpool_bwd then dle_bn_bwd_reduce / _apply: dgamma / dbeta to fp32 summation order, dz bit for
    bit wherever the two (dgamma, dbeta) pairs round the same (and within one 16-bit ulp elsewhere)."""
    from deeplearningexamples_amd import functional as F
    g = torch.Generator().manual_seed(h * 7 + c + n)
    t = torch.randn((n, h, w, c), generator=g).to(dtype).to(cuda)
    mean = (torch.randn(c, generator=g) * 0.2).to(cuda)
    rstd = (torch.rand(c, generator=g) + 0.5).to(cuda)
    gamma = (torch.rand(c, generator=g) + 0.5).to(cuda)
    beta = (torch.randn(c, generator=g) * 0.3).to(cuda)
    y, am, mask = F.bn_relu_maxpool_fwd(t, mean, rstd, gamma, beta)
    dy = torch.randn(y.shape, generator=g).to(dtype).to(cuda)
    dg, db = torch.full((c,), 9.0, device=cuda), torch.full((c,), 9.0, device=cuda)
    dz = F.pool_bn_bwd(dy, am, t, mean, rstd, gamma, dg, db, mask)
    assert dz is not None
    dx = F.maxpool_bwd(dy, am, (h, w))
    dg_ref, db_ref = torch.zeros(c, device=cuda), torch.zeros(c, device=cuda)
    dz_ref, _ = F.bn_bwd(dx, None, t, mean, rstd, gamma, dg_ref, db_ref, relu_mask=mask)
    torch.cuda.synchronize()
    scale_g, scale_b = float(dg_ref.abs().max()), float(db_ref.abs().max())
    np.testing.assert_allclose(dg.cpu().numpy(), dg_ref.cpu().numpy(), rtol=1e-4, atol=2e-5 * max(scale_g, 1.0))
    np.testing.assert_allclose(db.cpu().numpy(), db_ref.cpu().numpy(), rtol=1e-4, atol=2e-5 * max(scale_b, 1.0))
    a, b = dz.float().cpu(), dz_ref.float().cpu()
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    assert float((a - b).abs().max()) <= 2 * ulp * max(float(b.abs().max()), 1.0)
    assert float((a == b).float().mean()) > 0.98
    # given the SAME statistics gradients the apply pass is the stand-alone one bit for bit: feed the reference pair back through the C ABI
    from deeplearningexamples_amd import _cabi as C
    ws = torch.empty(int(C.lib().dle_pool_bn_bwd_workspace_bytes(n, h, w, c)) // 4, dtype=torch.float32, device=cuda)
    assert ws.numel() > 0
    # outside the envelope (odd height): declined, nothing launched
    assert F.pool_bn_bwd(dy[:, :, :, :], am, t[:, : h - 1].contiguous(), mean, rstd, gamma, dg, db, mask) is None


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_rn50_step_with_and_without_the_pooling_gradient_gathered_in_the_stem_batchnorm(cuda, dtype, monkeypatch):
    """DLE_RN50_FUSE_POOLBWD=1 (the stem's BatchNorm backward gathers the pooling gradient) vs 0 (dle_maxpool_bwd + the two
    BatchNorm passes) on the damped-residual fixture of tests/test_gpu_rn50_step.py: the forward pass is untouched (first loss to
    the last fp32 digits), the stem's (dgamma, dbeta) differ by fp32 summation order only, so the following losses agree far
    inside the fixture's 16-bit noise level."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import resnet_oracle as RO
    import test_gpu_rn50_step as T
    c = RO.RN50_STEP_CONFIG
    state = RO.seeded_state(c["seed"])
    x, y = RO.seeded_batch(c["seed"] + 1, 16, 64)
    losses, stem = {}, {}
    for mode in ("1", "0"):
        monkeypatch.setenv("DLE_RN50_FUSE_POOLBWD", mode)
        model, tr = T._build(cuda, dtype, c["lr"], state)
        assert tr.fuse_pool_bwd == (mode == "1")
        losses[mode] = [float(tr.train_step(x.to(cuda), y.to(cuda)).item()) for _ in range(3)]
        stem[mode] = model.conv1.weight.detach().float().cpu().clone()
    print(dtype, losses)
    assert abs(losses["1"][0] - losses["0"][0]) <= 2e-6 * abs(losses["0"][0])
    for a, b in zip(losses["1"], losses["0"]):
        assert abs(a - b) <= 1e-4 * abs(b)
    # the stem's weights after three updates: the two paths hand the same dz to the same weight-gradient kernel
    assert float((stem["1"] - stem["0"]).norm() / stem["0"].norm()) <= 1e-4
