"""Fused self-attention kernels (dle_attention_fwd / dle_attention_bwd) against an fp64 restatement of
BertSelfAttention.forward (LanguageModeling/BERT/modeling.py:340-384) under the keep mask the kernel itself drew, the
mask bit-exact against the KAT-pinned Philox oracle, and against the unfused batched-GEMM path of the same library."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(qkv, mask_add, keep, b, s, nh, scale, inv_keep, dctx=None):
    """fp64 attention of the 16-bit inputs; keep: bool [b, nh, s, s] or None.  -> ctx [T, H] (, dqkv [T, 3H])."""
    t, h3 = qkv.shape
    h = h3 // 3
    d = h // nh
    x = qkv.double().cpu().view(b, s, 3, nh, d).permute(2, 0, 3, 1, 4).contiguous().requires_grad_(dctx is not None)
    q, k, v = x[0], x[1], x[2]
    sc = torch.matmul(q, k.transpose(-1, -2)) * scale
    if mask_add is not None:
        sc = sc + mask_add.double().cpu()[:, None, None, :]
    p = torch.softmax(sc, -1)
    if keep is not None:
        p = torch.where(keep.cpu(), p * inv_keep, torch.zeros((), dtype=torch.float64))
    ctx = torch.matmul(p, v).permute(0, 2, 1, 3).reshape(t, h)
    if dctx is None:
        return ctx.detach()
    ctx.backward(dctx.double().cpu())
    dqkv = x.grad.permute(1, 3, 0, 2, 4).reshape(t, h3)
    return ctx.detach(), dqkv


def _relerr(got, ref):
    got, ref = got.double().cpu(), ref.double()
    return float((got - ref).norm() / (ref.norm() + 1e-30))


@pytest.mark.parametrize("dtype,bar_f,bar_b", [(torch.float16, 1.5e-3, 4e-3), (torch.bfloat16, 1e-2, 2.5e-2)])
@pytest.mark.parametrize("p", [0.0, 0.1])
@pytest.mark.parametrize("s", [128, 256, 512])
def test_attention_fwd_bwd_vs_fp64(cuda, dtype, bar_f, bar_b, p, s):
    """s = 128: one kernel each way.  s = 256 / 512 (BERT phase 2): K / V streamed in 128-key blocks -- forward with online max /
    sum then a second pass under the final statistics, backward as a per-query-block kernel (delta, dQ) and a per-key-block kernel
    (dK, dV); same bars, same mask contract (bit-exact against the Philox oracle on the [B, heads, S, S] chunk index)."""
    from deeplearningexamples_amd import functional as F
    from oracle import philox_oracle as P
    b, nh, d = 3, 4, 64
    h = nh * d
    g = torch.Generator().manual_seed(5)
    qkv = (torch.randn(b * s, 3 * h, generator=g) * 0.8).to(dtype).to(cuda)
    dctx = (torch.randn(b * s, h, generator=g) * 0.5).to(dtype).to(cuda)
    # additive mask as the engine builds it (modeling.py:868-869): the tail of sequences 1, 2 is padding
    att = torch.ones(b, s)
    att[1, (100 * s) // 128:] = 0
    att[2, (37 * s) // 128:] = 0
    mask_add = ((1.0 - att) * -10000.0).to(cuda)
    scale = 1.0 / math.sqrt(d)
    seed, off = 0x1234567887654321, (1 << 34) + 9
    ctx, stats, mbits = F.attention_fwd(qkv, mask_add, b, s, nh, scale, p, seed, off, want_mask=True)
    keep = None
    if p > 0:
        keep = F.unpack_dropout_mask(mbits, (b, nh, s, s))
        ref_keep = P.keep_mask(b * nh * s * s, p, seed, off).reshape(b, nh, s, s)
        assert np.array_equal(keep.cpu().numpy(), ref_keep), "dropout mask differs from the Philox oracle"
        assert abs(float(keep.float().mean()) - (1 - p)) < 5e-3
    else:
        assert mbits is None
    dqkv = F.attention_bwd(qkv, dctx, mask_add, stats, b, s, nh, scale, p, seed, off)
    torch.cuda.synchronize()
    ref_ctx, ref_dqkv = _reference(qkv, mask_add, keep, b, s, nh, scale, float(P.inv_keep(p)) if p > 0 else 1.0, dctx)
    e = _relerr(ctx, ref_ctx)
    assert e < bar_f, ("ctx", e)
    assert float((ctx.double().cpu() - ref_ctx).abs().max()) < 6 * bar_f * float(ref_ctx.abs().max())
    for i, nm in enumerate(("dq", "dk", "dv")):
        e = _relerr(dqkv[:, i * h:(i + 1) * h], ref_dqkv[:, i * h:(i + 1) * h])
        assert e < bar_b, (nm, e)
    # row statistics: max of the scaled + masked scores and 1 / sum of exp
    x = qkv.double().cpu().view(b, s, 3, nh, d).permute(2, 0, 3, 1, 4)
    sc = torch.matmul(x[0], x[1].transpose(-1, -2)) * scale + mask_add.double().cpu()[:, None, None, :]
    mx = sc.max(-1).values
    st = stats.double().cpu().view(b, nh, s, -1)
    assert st.shape[-1] == (2 if s == 128 else 4)
    assert float((st[..., 0] - mx).abs().max()) < 1e-4 * (1 + float(mx.abs().max()))
    inv = 1.0 / torch.exp(sc - mx[..., None]).sum(-1)
    assert float(((st[..., 1] - inv) / inv).abs().max()) < 1e-4
    # per-sequence column sums of dqkv (the QKV bias gradient partials), from the rounded values the kernel stored
    cs = torch.empty((b * (s // 128), 3 * h), dtype=torch.float32, device=cuda)
    dqkv_c = F.attention_bwd(qkv, dctx, mask_add, stats, b, s, nh, scale, p, seed, off, colsum_partial=cs)
    assert torch.equal(dqkv_c, dqkv)
    ref_cs = dqkv.float().view(b * (s // 128), 128, 3 * h).sum(1)
    assert float((cs - ref_cs).abs().max()) <= 1e-4 * (1.0 + float(ref_cs.abs().max()))
    if p > 0:          # the keep mask READ by the backward kernel instead of re-drawn (dle_attention_bwd_keep): the same bits
        dqkv_k = F.attention_bwd(qkv, dctx, mask_add, stats, b, s, nh, scale, p, seed, off, keep_mask=mbits)
        assert torch.equal(dqkv_k, dqkv)
    # determinism: the same call gives the same bits
    ctx2, stats2, _ = F.attention_fwd(qkv, mask_add, b, s, nh, scale, p, seed, off)
    dqkv2 = F.attention_bwd(qkv, dctx, mask_add, stats2, b, s, nh, scale, p, seed, off)
    assert torch.equal(ctx, ctx2) and torch.equal(dqkv, dqkv2)


@pytest.mark.parametrize("s", [128, 512])
def test_attention_matches_unfused_path(cuda, s):
    """Fused kernels vs the batched-GEMM + softmax kernels of the same library (they round the scores and the
    probabilities to 16 bits; the fused path keeps them in fp32): same masks, close values.  (s = 512: the phase-2 shape.)"""
    from deeplearningexamples_amd import functional as F
    dtype = torch.bfloat16
    b, nh, d = 2, 16, 64
    h = nh * d
    t = b * s
    g = torch.Generator().manual_seed(8)
    qkv = (torch.randn(t, 3 * h, generator=g) * 0.7).to(dtype).to(cuda)
    mask_add = torch.zeros(b, s, device=cuda)
    scale, p, seed, off = 0.125, 0.1, 77, 3
    ctx, stats, mbits = F.attention_fwd(qkv, mask_add, b, s, nh, scale, p, seed, off, want_mask=True)
    probs = torch.empty((b * nh, s, s), dtype=dtype, device=cuda)
    F.gemm_batched(qkv, qkv[:, h:], probs, s, s, d, 3 * h, 3 * h, s, True, True, b * nh, nh,
                   (s * 3 * h, d), (s * 3 * h, d), (nh * s * s, s * s))
    pdrop, mask_u = F.softmax_dropout_fwd_(probs, mask_add, nh * s, scale, p, seed, off)
    assert torch.equal(mbits, mask_u), "fused and unfused paths draw different masks for the same (seed, offset)"
    ctx_u = torch.empty((t, h), dtype=dtype, device=cuda)
    F.gemm_batched(pdrop, qkv[:, 2 * h:], ctx_u, s, d, s, s, 3 * h, h, True, False, b * nh, nh,
                   (nh * s * s, s * s), (s * 3 * h, d), (s * h, d))
    assert _relerr(ctx, ctx_u.double().cpu()) < 1.5e-2


def test_attention_bert_large_batch_sampled(cuda):
    """BASELINE.json configs[2] slice: B = 256 x 16 heads (4096 workgroups).  Sampled (sequence, head) pairs against
    fp64; every output element finite; untouched neighbours of the output buffers stay untouched."""
    from deeplearningexamples_amd import functional as F
    from oracle import philox_oracle as P
    dtype = torch.bfloat16
    b, s, nh, d = 256, 128, 16, 64
    h = nh * d
    t = b * s
    g = torch.Generator(device=cuda).manual_seed(11)
    qkv = (torch.randn(t, 3 * h, generator=g, device=cuda) * 0.6).to(dtype)
    dctx = (torch.randn(t, h, generator=g, device=cuda) * 0.3).to(dtype)
    mask_add = torch.zeros(b, s, device=cuda)
    mask_add[5, 90:] = -10000.0
    scale, p, seed, off = 0.125, 0.1, 1234, 17
    ctx, stats, mbits = F.attention_fwd(qkv, mask_add, b, s, nh, scale, p, seed, off, want_mask=True)
    dqkv = F.attention_bwd(qkv, dctx, mask_add, stats, b, s, nh, scale, p, seed, off)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(ctx.float()).all()) and bool(torch.isfinite(dqkv.float()).all())
    keep_all = F.unpack_dropout_mask(mbits, (b, nh, s, s))
    inv_keep = float(P.inv_keep(p))
    for (bi, hi) in [(0, 0), (5, 3), (128, 15), (255, 15), (77, 8)]:
        rows = slice(bi * s, (bi + 1) * s)
        cols = [slice(i * h + hi * d, i * h + (hi + 1) * d) for i in range(3)]
        sub = torch.cat([qkv[rows, c] for c in cols], dim=1)                  # [S, 3 d]: a one-head problem
        ref_ctx, ref_d = _reference(sub, mask_add[bi:bi + 1], keep_all[bi:bi + 1, hi:hi + 1], 1, s, 1, scale, inv_keep,
                                    dctx[rows, hi * d:(hi + 1) * d])
        assert _relerr(ctx[rows, hi * d:(hi + 1) * d], ref_ctx) < 1e-2, (bi, hi)
        for i, nm in enumerate(("dq", "dk", "dv")):
            assert _relerr(dqkv[rows, cols[i]], ref_d[:, i * d:(i + 1) * d]) < 2.5e-2, (bi, hi, nm)


def test_attention_argument_errors(cuda):
    from deeplearningexamples_amd import functional as F
    qkv = torch.zeros(2 * 64, 3 * 128, dtype=torch.bfloat16, device=cuda)
    assert not F.attention_supported(64, 64) and not F.attention_supported(128, 32) and F.attention_supported(128, 64)
    assert F.attention_supported(512, 64) and F.attention_supported(1024, 64) and not F.attention_supported(192, 64) \
        and not F.attention_supported(2048, 64)
    with pytest.raises((ValueError, RuntimeError)):
        F.attention_fwd(qkv, None, 2, 64, 2, 0.125)                          # S = 64 is outside the envelope
