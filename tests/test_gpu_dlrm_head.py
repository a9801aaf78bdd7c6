"""The fused head of the DLRM top model (csrc/dlrm_head.hip: last linear layer with one output + BCEWithLogitsLoss(mean) + the
backward of both in one pass) vs (a) a float64 restatement of the reference's ops (torch.nn.Linear -> BCEWithLogitsLoss ->
autograd, dlrm/scripts/main.py:556,589-592) with the 16-bit rounding points of the separate launches and (b) the separate
launches themselves (dle_gemm x 3 + dle_bce_logits + dle_colsum x 2).  GPU only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NP = {torch.float16: np.float16}


def _r16(x, dtype):
    """Round a float64 / float32 numpy array through the 16-bit storage type."""
    return torch.from_numpy(np.asarray(x, np.float32)).to(dtype).to(torch.float64).numpy()


def _reference(h64, w64, z, y, scale, dtype):
    """Everything downstream of the (16-bit) logits z, in float64 with the 16-bit rounding points of the separate launches."""
    loss = np.mean(np.maximum(z, 0) - z * y + np.log1p(np.exp(-np.abs(z))))
    dz = _r16((1.0 / (1.0 + np.exp(-z)) - y) * scale / len(y), dtype)
    dh = np.where(h64 > 0, _r16(dz[:, None] * w64[None, :], dtype), 0.0)
    return loss, dz, dh, dz @ h64, dz.sum(), dh.sum(0)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("m,k", [(65536, 256), (4099, 512), (1000, 64), (37, 8), (1, 256), (8192, 136)])
def test_head_matches_float64_restatement(cuda, m, k, dtype):
    from deeplearningexamples_amd import functional as F
    g = torch.Generator().manual_seed(m * 7 + k)
    h = torch.relu(torch.randn(m, k, generator=g)).to(dtype).to(cuda)                    # post-ReLU activations: ~half zeros
    w = (torch.randn(k, generator=g) / k ** 0.5).to(dtype).to(cuda)
    bias = torch.tensor([0.1], device=cuda)
    y = (torch.rand(m, generator=g) < 0.3).float().to(cuda)
    scale = torch.tensor([1024.0], device=cuda)
    gw, gb, gp = (torch.full((k,), 7.0, device=cuda), torch.full((1,), 7.0, device=cuda), torch.full((k,), 7.0, device=cuda))
    ws = F.HeadWorkspace(m, k, cuda)
    loss, dh, z = F.head_bce_fwd_bwd(h, w, bias, y, scale, gw, gb, ws, gprev_bias=gp, want_logits=True)
    h64, w64 = h.to(torch.float64).cpu().numpy(), w.to(torch.float64).cpu().numpy()
    y64 = y.cpu().numpy().astype(np.float64)
    step = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7                           # relative 16-bit spacing
    # the logit: a K-term fp32 dot product + bias rounded to 16 bits -- within one 16-bit step of the float64 one
    zz = z.to(torch.float64).cpu().numpy()
    exact = h64 @ w64 + np.float64(np.float32(0.1))
    assert np.all(np.abs(zz - exact) <= step * np.maximum(np.abs(exact), 2.0 ** -14) + 1e-6)
    # everything downstream of the kernel's own logits
    rl, rdz, rdh, rgw, rgb, rgp = _reference(h64, w64, zz, y64, 1024.0, dtype)
    assert abs(loss.item() - rl) <= 1e-5 * abs(rl) + 1e-7
    got = dh.to(torch.float64).cpu().numpy()
    assert np.array_equal(got == 0, rdh == 0)
    tiny = 2.0 ** -24 if dtype == torch.float16 else 0.0                   # fp16 subnormal spacing
    assert np.all(np.abs(got - rdh) <= 2.5 * step * np.abs(rdh) + 2.5 * tiny)   # (fp32 exp vs float64 exp: dz may sit one step away)
    assert np.mean(got == rdh) > 0.98
    mag = np.abs(rdz) @ h64
    assert np.all(np.abs(gw.to(torch.float64).cpu().numpy() - rgw) <= 1e-4 * mag + 1e-9)
    assert abs(gb.item() - rgb) <= 1e-4 * np.abs(rdz).sum() + 1e-9
    assert np.all(np.abs(gp.to(torch.float64).cpu().numpy() - rgp) <= 1e-4 * np.abs(rdh).sum(0) + 1e-9)
    # the workspace counter is back at zero and a second launch is bit-identical (sums folded in a fixed order)
    gw2, gb2, gp2 = torch.empty_like(gw), torch.empty_like(gb), torch.empty_like(gp)
    loss2, dh2, z2 = F.head_bce_fwd_bwd(h, w, bias, y, scale, gw2, gb2, ws, gprev_bias=gp2, want_logits=True)
    for a, b in ((loss, loss2), (dh, dh2), (z, z2), (gw, gw2), (gb, gb2), (gp, gp2)):
        assert torch.equal(a, b)
    # no loss scale, no bias, no side outputs
    loss3, dh3, _ = F.head_bce_fwd_bwd(h, w, None, y, None, gw2, gb2, ws)
    z3 = _r16(np.float32(1) * (h64 @ w64), dtype)
    rl3 = np.mean(np.maximum(z3, 0) - z3 * y64 + np.log1p(np.exp(-np.abs(z3))))
    assert abs(loss3.item() - rl3) <= 2e-3 * abs(rl3)                       # (logits one step apart on a few rows)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_head_matches_separate_launches(cuda, dtype):
    """Same inputs through the five launches the fused kernel replaces (the DLE_DLRM_FUSE_HEAD=0 path of DlrmTop)."""
    from deeplearningexamples_amd import functional as F
    from deeplearningexamples_amd import _cabi as C
    m, k = 16384, 256
    g = torch.Generator().manual_seed(11)
    h = torch.relu(torch.randn(m, k, generator=g)).to(dtype).to(cuda)
    w = (torch.randn(1, k, generator=g) / 16).to(dtype).to(cuda)
    bias = torch.tensor([-0.2], device=cuda)
    y = (torch.rand(m, generator=g) < 0.5).float().to(cuda)
    scale = torch.tensor([4096.0], device=cuda)
    logits = F.gemm(h, w, m, 1, k, True, True, out_dtype=dtype, bias=bias)
    loss_s, dl = F.bce_with_logits(logits, y, grad_scale=scale)
    dl = dl.view(-1, 1)
    gw_s = torch.empty(1, k, device=cuda)
    F.gemm(dl, h, 1, k, m, False, False, out=gw_s, splitk=F.pick_splitk(1, k, m))
    gb_s = F.colsum(dl)
    gh_s = F.gemm(dl, w, m, k, 1, True, False, out_dtype=dtype, act=C.ACT_RELU_BWD, mask_src=h)
    gp_s = F.colsum(gh_s)
    gw, gb, gp = torch.empty(k, device=cuda), torch.empty(1, device=cuda), torch.empty(k, device=cuda)
    ws = F.HeadWorkspace(m, k, cuda)
    loss, gh, z = F.head_bce_fwd_bwd(h, w.view(-1), bias, y, scale, gw, gb, ws, gprev_bias=gp, want_logits=True)
    same = (z == logits.view(-1))
    assert same.float().mean().item() > 0.97                 # (fp32 summation order inside the dot product differs)
    assert abs(loss.item() - loss_s.item()) <= 2e-5 * abs(loss_s.item())
    assert torch.equal(gh[same], gh_s[same])
    denom = gw_s.abs().max().item()
    assert (gw - gw_s.view(-1)).abs().max().item() <= 3e-3 * denom
    assert abs(gb.item() - gb_s.item()) <= 3e-3 * dl.float().abs().sum().item() / 50 + 1e-6
    assert (gp - gp_s).abs().max().item() <= 3e-3 * gp_s.abs().max().item()


def test_step_with_and_without_the_fused_head(cuda, monkeypatch):
    """Three train steps of the Criteo-shape configuration, DLE_DLRM_FUSE_HEAD = 1 vs 0: same losses and weights up to fp32
    summation order."""
    from oracle import dlrm_step_oracle as SO
    from tests.test_gpu_dlrm_step import _build
    cfg = dict(SO.DLRM_STEP_CONFIGS["criteo_shape"])
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("DLE_DLRM_FUSE_HEAD", mode)
        model, trainer, _ = _build(cfg, cuda, torch.float16)
        assert trainer.fuse_head == (mode == "1")
        num, cat, click = SO.seeded_dlrm_batch(cfg["sizes"], cfg["num"], cfg["batch"], cfg["seed"] + 1000)
        num, cat, click = num.to(cuda), cat.to(cuda), click.to(cuda)
        losses = [float(trainer.train_step(num, cat, click).item()) for _ in range(3)]
        out[mode] = (losses, model.top_model.out.weight.detach().clone(), model.top_model.out.bias.detach().clone(),
                     model.top_model.mlp.linears[-1].bias.detach().clone(), model.bottom_model.mlp.linears[0].weight.detach().clone())
    la, lb = np.asarray(out["1"][0]), np.asarray(out["0"][0])
    assert np.all(np.abs(la - lb) <= 2e-4 * np.abs(lb)), (la, lb)
    for a, b in zip(out["1"][1:], out["0"][1:]):
        assert (a - b).abs().max().item() <= 2e-3 * b.abs().max().item() + 1e-6
