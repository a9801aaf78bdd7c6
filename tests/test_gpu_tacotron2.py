"""Tacotron2 train step on the MI355X (second half): every kernel of csrc/tacotron2.hip against the plain-torch
statement of the same entry point (tests/_tacotron2_doubles.py, evaluated on the CPU), and the whole step against the oracle
that the reference's own Tacotron2 + Tacotron2Loss pins (tests/golden/tacotron2_loss.npz), evaluated under the dropout masks
the HIP RNG drew.  Bars: loss 1e-3 relative (north_star); gradients inside the 16-bit STORAGE floor measured with the
fp64-accumulating doubles at the same storage dtype (BPTT through ~30 decoder steps + the encoder: fp16 worst tensor 4 %, median
0.3 %; bf16 worst 14 %, median 2 %) times a margin."""
import numpy as np
import pytest
import torch

from tests import _tacotron2_doubles as D

pytestmark = pytest.mark.gpu
DTYPES = [torch.float16, torch.bfloat16]


def _tol(dtype):
    return dict(rtol=2e-3, atol=2e-3) if dtype == torch.float16 else dict(rtol=1.6e-2, atol=1.6e-2)


def _ops():
    from deeplearningexamples_amd.tacotron2 import ops
    return ops


def _close(got, ref, **kw):
    np.testing.assert_allclose(got.detach().float().cpu().numpy(), ref.detach().float().cpu().numpy(), **kw)


def _keep_bits(n, p, g):
    keep = torch.rand(n, generator=g) >= p
    return keep, D._pack(keep)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("b,h,with_drop,with_live", [(3, 96, True, False), (48, 1024, True, False), (5, 32, False, True), (104, 256, False, True)])
def test_lstm_cell_forward_backward(cuda, dtype, b, h, with_drop, with_live):
    ops = _ops()
    g = torch.Generator().manual_seed(b * 7 + h)
    wide = (torch.randn(b, 3, 4 * h, generator=g) * 1.5).to(dtype)                 # gates = a [B, 4H] slice with row stride 12 H
    c_prev = torch.randn(b, h, generator=g)
    h_prev = torch.randn(b, h, generator=g).to(dtype)
    keep, bits = _keep_bits(4 * b * h, 0.1, g) if with_drop else (None, None)
    kidx = 2 * b * h if with_drop else 0
    live = (torch.rand(b, generator=g) > 0.4).float() if with_live else None

    def run(L, d):
        gates = d(wide.clone())
        c_out, dst0, dst1 = d(torch.zeros(b, h)), d(torch.zeros(b, 2 * h, dtype=dtype)), d(torch.zeros(b, h, dtype=dtype))
        out_dst = d(torch.zeros(b, 3 * h, dtype=dtype)) if with_live else None
        L.lstm_fwd(gates[:, 1], d(c_prev), c_out, [dst0[:, h:], dst1], keep=d(bits) if with_drop else None, keep_index=kidx, p=0.1,
                   live=d(live) if with_live else None, h_prev=d(h_prev) if with_live else None,
                   out_dst=out_dst[:, h:2 * h] if with_live else None)
        return gates, c_out, dst0, dst1, out_dst
    got, ref = run(ops, lambda t: t.to(cuda)), run(D, lambda t: t)
    for a, r in zip(got, ref):
        if a is not None:
            _close(a, r, **_tol(dtype))
    # backward from the activations the forward left behind
    act = ref[0]
    dh, dc_next = torch.randn(b, h, generator=g), torch.randn(b, h, generator=g)

    def runb(L, d):
        a = d(act.clone())
        dcp, dhp = d(torch.zeros(b, h)), d(torch.zeros(b, h)) if with_live else None
        L.lstm_bwd(d(dh), d(dc_next), a[:, 1], d(c_prev), a[:, 1], dcp, keep=d(bits) if with_drop else None, keep_index=kidx, p=0.1,
                   live=d(live) if with_live else None, dh_prev=dhp)
        return a, dcp, dhp
    got, ref = runb(ops, lambda t: t.to(cuda)), runb(D, lambda t: t)
    for a, r in zip(got, ref):
        if a is not None:
            _close(a, r, **_tol(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("b,h,k,bias,add,drop", [(128, 1024, 1536, False, True, True), (128, 1024, 2560, True, False, True),
                                                 (3, 96, 160, True, True, False), (70, 32, 72, False, False, True), (200, 64, 64, True, True, True)])
def test_lstm_cell_fused_into_the_gates_product(cuda, dtype, b, h, k, bias, add, drop):
    """dle_t2_lstm_gemm_fwd (the cell as the epilogue of the few-row GEMM) == dle_gemm followed by dle_t2_lstm_fwd BIT FOR BIT (the
    pre-activation is rounded to the storage type exactly as the unfused product's output is), on strided operand / destination
    views as the decoder uses them; both against the double."""
    ops = _ops()
    from deeplearningexamples_amd import functional as F, _cabi as C
    g = torch.Generator().manual_seed(b + h + k)
    xw = (torch.randn(b, k + 16, generator=g) * 0.5).to(dtype)                    # operand = a column slice of a wider buffer
    w = (torch.randn(4 * h, k, generator=g) * (1.0 / k ** 0.5)).to(dtype)
    bs = torch.randn(4 * h, generator=g) * 0.1 if bias else None
    ad = (torch.randn(b, 4 * h, generator=g) * 0.5).to(dtype) if add else None
    c_prev = torch.randn(b, h, generator=g)
    keep, bits = _keep_bits(3 * b * h, 0.1, g) if drop else (None, None)
    kidx = b * h if drop else 0

    def run(L, d, fused):
        x = d(xw.clone())[:, 8:8 + k]
        gates = d(torch.zeros(b, 4 * h, dtype=dtype))
        c_out, dst0, dst1 = d(torch.zeros(b, h)), d(torch.zeros(b, 2 * h + 8, dtype=dtype)), d(torch.zeros(b, h, dtype=dtype))
        dsts = [dst0[:, h:2 * h], dst1]
        kw = dict(keep=d(bits) if drop else None, keep_index=kidx, p=0.1)
        if fused:
            L.lstm_gemm_fwd(x, d(w), d(bs) if bias else None, d(ad) if add else None, d(c_prev), c_out, gates, dsts, **kw)
        else:
            F.gemm(x, d(w), b, 4 * h, k, True, True, out=gates, bias=d(bs) if bias else None, act=C.ACT_ADD if add else C.ACT_NONE,
                   mask_src=d(ad) if add else None)
            L.lstm_fwd(gates, d(c_prev), c_out, dsts, **kw)
        return gates, c_out, dst0, dst1
    dev = lambda t: t.to(cuda)
    got, two = run(ops, dev, True), run(ops, dev, False)
    assert torch.equal(got[0], two[0]), "fused and unfused gate activations differ"
    for a, r in zip(got[1:], two[1:]):                        # (the cell's fp32 arithmetic may contract its multiply-adds differently)
        assert float((a.float() - r.float()).abs().max()) <= 2.0 ** -9 * max(1.0, float(r.float().abs().max()))
    ref = run(D, lambda t: t, True)
    for a, r in zip(got, ref):
        _close(a, r, **_tol(dtype))
    assert float(got[2][:, :h].abs().max()) == 0 and float(got[2][:, 2 * h:].abs().max()) == 0      # neighbours of the strided view


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("b,ti,a,e", [(3, 23, 32, 64), (48, 150, 128, 512), (2, 300, 128, 512)])
def test_attention_step_forward_backward(cuda, dtype, b, ti, a, e):
    ops = _ops()
    g = torch.Generator().manual_seed(b + ti)
    q = torch.randn(b, a, generator=g)
    pl = torch.randn(b * ti, a, generator=g).to(dtype)
    v = torch.randn(a, generator=g) * 0.5
    mem = torch.randn(b * ti, e, generator=g).to(dtype)
    lengths = torch.randint(ti // 2, ti + 1, (b,), generator=g)
    lengths[0] = ti
    awc_prev = torch.zeros(b * ti, 8, dtype=dtype)
    awc_prev[:, :2] = torch.rand(b * ti, 2, generator=g).to(dtype)

    def run(L, d):
        th, aw, nxt = d(torch.zeros(b * ti, a, dtype=dtype)), d(torch.zeros(b, ti)), d(torch.ones(b * ti, 8, dtype=dtype))
        c0, c1 = d(torch.zeros(b, 2 * e, dtype=dtype)), d(torch.zeros(b, e, dtype=dtype))
        L.attention_fwd(d(q), d(pl), d(v), d(mem), d(lengths), d(awc_prev), th, aw, nxt, [c0[:, e:], c1])
        return th, aw, nxt, c0, c1
    got, ref = run(ops, lambda t: t.to(cuda)), run(D, lambda t: t)
    _close(got[0], ref[0], **_tol(dtype))
    _close(got[1], ref[1], rtol=2e-2, atol=2e-4)                                  # the saved tanh is rounded to 16 bits in both
    for k in (2, 3, 4):
        _close(got[k], ref[k], **_tol(dtype))
    pad = torch.arange(ti)[None, :] >= lengths[:, None]
    assert float(got[1].cpu()[pad].abs().max() if pad.any() else 0.0) == 0.0      # no weight on padded text positions
    # backward, from the CPU forward's saved tensors
    th, aw = ref[0], ref[1]
    d_ctx, d_aw_in = torch.randn(b, e, generator=g), torch.randn(b, ti, generator=g) * 0.1
    # the context gradient arrives in three row-strided pieces, the weights' gradient in two (summed on load)
    wide = torch.randn(b, 3 * e + 8, generator=g)
    pieces = (wide[:, 4:4 + e], wide[:, 8 + e:8 + 2 * e])
    d_aw2 = torch.randn(b, ti, generator=g) * 0.05
    base_mem, base_pm, base_dv = torch.randn(b * ti, e, generator=g), torch.randn(b * ti, a, generator=g), torch.randn(b, a, generator=g)

    def runb(L, d, extended):
        dmem, dpm, dv = d(base_mem.clone()), d(base_pm.clone()), d(base_dv.clone())
        dpl, dq = d(torch.zeros(b * ti, a, dtype=dtype)), d(torch.zeros(b, a))
        dq16, dc16 = d(torch.zeros(b, a, dtype=dtype)), d(torch.zeros(b, e, dtype=dtype))
        if extended:
            w = d(wide)
            L.attention_bwd(d(d_ctx), d(d_aw_in), d(aw), d(th), d(v), d(mem), None, dpl, None, dv, dpm,
                            d_ctx_add=(w[:, 4:4 + e], w[:, 8 + e:8 + 2 * e]), d_aw_add=d(d_aw2), dq16=dq16, dctx16=dc16)
        else:
            L.attention_bwd(d(d_ctx), d(d_aw_in), d(aw), d(th), d(v), d(mem), dmem, dpl, dq, dv, dpm)
        return dmem, dpl, dq, dv, dpm, dq16, dc16
    for extended in (False, True):
        got, ref = runb(ops, lambda t: t.to(cuda), extended), runb(D, lambda t: t, extended)
        _close(got[0], ref[0], rtol=1e-4, atol=1e-4)
        _close(got[1], ref[1], **_tol(dtype))
        _close(got[3], ref[3], rtol=2e-3, atol=5e-3)
        _close(got[4], ref[4], rtol=2e-3, atol=2e-3)
        if extended:
            _close(got[5], ref[5], **_tol(dtype))
            _close(got[6], ref[6], **_tol(dtype))
        else:
            _close(got[2], ref[2], rtol=2e-3, atol=2e-3)
    # the deferred fold of the kept per-step gradients
    xs = torch.randn(7, b * ti * a, generator=g).to(dtype)
    acc0 = torch.randn(b * ti * a, generator=g)
    o1, o2 = acc0.clone().to(cuda), acc0.clone()
    ops.sum_steps(xs.to(cuda), o1)
    D.sum_steps(xs, o2)
    _close(o1, o2, rtol=1e-5, atol=1e-5)
    # deterministic: per-sample partial sums, no atomics
    a1, a2 = runb(ops, lambda t: t.to(cuda), True), runb(ops, lambda t: t.to(cuda), True)
    assert torch.equal(a1[3], a2[3]) and torch.equal(a1[4], a2[4])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("b,ti,kl", [(3, 23, 31), (16, 160, 31), (2, 9, 5)])
def test_location_backward_and_transposed_copies(cuda, dtype, b, ti, kl):
    ops = _ops()
    from deeplearningexamples_amd import functional as F
    g = torch.Generator().manual_seed(b + ti)
    dcol = torch.randn(b * ti, kl * 8, generator=g).to(dtype)
    prev0, cum0 = torch.randn(b, ti, generator=g), torch.randn(b, ti, generator=g)

    def run(L, d):
        p, c = d(prev0.clone()), d(cum0.clone())
        L.location_bwd(d(dcol), p, c, b, ti, kl)
        return p, c
    got, ref = run(ops, lambda t: t.to(cuda)), run(D, lambda t: t)
    _close(got[0], ref[0], rtol=1e-5, atol=1e-5)
    _close(got[1], ref[1], rtol=1e-5, atol=1e-5)
    # transposed 16-bit working copies: from the fp32 master and from a 16-bit row-strided view
    w = torch.randn(70, 200, generator=g)
    _close(F.transpose_cast(w.to(cuda), dtype), w.t().to(dtype), rtol=0, atol=0)
    w16 = torch.randn(130, 96, generator=g).to(dtype)
    _close(F.transpose_cast(w16.to(cuda)[:, 8:72], dtype), w16[:, 8:72].t(), rtol=0, atol=0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_lstm_backward_sums_its_gradient_pieces(cuda, dtype):
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    b, h = 48, 256
    act = torch.rand(b, 4 * h, generator=g).to(dtype)
    act[:, 2 * h:3 * h] = (torch.rand(b, h, generator=g) * 2 - 1).to(dtype)
    c_prev, dc_next = torch.randn(b, h, generator=g), torch.randn(b, h, generator=g)
    wide = torch.randn(b, 3 * h + 4, generator=g)

    def run(L, d):
        w = d(wide)
        dg, dcp = d(torch.zeros(b, 4 * h, dtype=dtype)), d(torch.zeros(b, h))
        L.lstm_bwd(w[:, :h], d(dc_next), d(act), d(c_prev), dg, dcp, dh_add=(w[:, h + 4:2 * h + 4], w[:, 2 * h + 4:]))
        return dg, dcp
    got, ref = run(ops, lambda t: t.to(cuda)), run(D, lambda t: t)
    _close(got[0], ref[0], **_tol(dtype))
    _close(got[1], ref[1], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dtype", DTYPES)
def test_mel_loss_and_tanh(cuda, dtype):
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    r, nm, ld = 93, 80, 88
    out_all = torch.randn(r, ld, generator=g)
    post = torch.randn(r, nm, generator=g).to(dtype)
    target = torch.randn(r, nm, generator=g) - 2
    scale = torch.tensor([128.0])

    def run(L, d):
        d_out, d_post = d(torch.zeros(r, ld, dtype=dtype)), d(torch.zeros(r, nm, dtype=dtype))
        loss = L.mel_loss(d(out_all), d(post), d(target), nm, d(scale), d_out, d_post)
        return loss, d_out, d_post
    got, ref = run(ops, lambda t: t.to(cuda)), run(D, lambda t: t)
    _close(got[0], ref[0], rtol=1e-5)
    _close(got[1], ref[1], **_tol(dtype))
    _close(got[2], ref[2], **_tol(dtype))
    x = (torch.randn(4097, generator=g) * 2).to(dtype)
    _close(ops.tanh_fwd(x.to(cuda)), D.tanh_fwd(x), **_tol(dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_mask_rows(cuda, dtype):
    """dle_t2_mask_rows == masked_fill over the frames past each sample's length (parse_output, model.py:648-655), on a strided view."""
    from deeplearningexamples_amd.tacotron2 import ops
    g = torch.Generator().manual_seed(5)
    b, to, cols, ld = 5, 37, 80, 88
    lengths = torch.tensor([37, 30, 18, 1, 0], dtype=torch.int64)
    x = torch.randn(b * to, ld, generator=g).to(dtype)
    want = x.clone().view(b, to, ld)
    past = torch.arange(to)[None, :] >= lengths[:, None]
    want[:, :, :cols] = want[:, :, :cols].masked_fill(past[:, :, None], 0.0)
    want[:, :, cols] = want[:, :, cols].masked_fill(past, 1e3)
    xd = x.to(cuda)
    ops.mask_rows(xd, cols, lengths.to(cuda), b, to, 0.0)
    ops.mask_rows(xd[:, cols:], 1, lengths.to(cuda), b, to, 1e3)
    assert torch.equal(xd.cpu(), want.view(b * to, ld))
    with pytest.raises(ValueError):
        ops.mask_rows(xd, cols, lengths.to(cuda).int(), b, to, 0.0)


def _engine_masks(tr, F):
    """The keep masks the engine drew (HIP counter-based RNG), in the order tests/test_tacotron2_host.py _Replay expects."""
    sv = tr.sv
    b, ti, to = sv["b"], sv["ti"], sv["to"]
    un = lambda m, shape: F.unpack_dropout_mask(m, shape).cpu()
    log = [un(s["mask"], s["y"].shape) for s in sv["enc"]]
    log += [un(sv["m1"], sv["l1"].shape), un(sv["m2"], sv["l2"].shape)]
    log += [un(sv["keep_a"], (to * b * tr.Ha,)), un(sv["keep_d"], (to * b * tr.Hd,))]
    log += [un(s["mask"], s["y"].shape) for s in sv["post"]]
    return log


@pytest.mark.parametrize("dtype,case,masked", [(torch.float16, None, False), (torch.bfloat16, None, False),
                                               # 24 text positions: the memory gradient of the context goes through the batched GEMM
                                               (torch.float16, dict(text_lengths=[24, 17, 9], mel_lengths=[20, 29, 13]), False),
                                               # --mask-padding (model.py:648-655)
                                               (torch.float16, None, True), (torch.bfloat16, None, True)])
def test_step_loss_and_gradients_vs_oracle_under_the_hip_masks(cuda, dtype, case, masked):
    from oracle import tacotron2_oracle as TO
    from deeplearningexamples_amd import functional as F
    from deeplearningexamples_amd.tacotron2.engine import Tacotron2Trainer
    from deeplearningexamples_amd.tacotron2.model import Tacotron2
    from tests.test_tacotron2_host import _Replay
    c = dict(TO.TACOTRON2_CASE, **(case or {}))
    cfg = c["cfg"]
    state = TO.seeded_state(cfg, c["seed"])
    model = Tacotron2(device=cuda, **cfg)
    model.load_reference_state(state)
    scale = 65536.0                                            # GradScaler's default: small fp16 gradients stay out of the subnormals
    tr = Tacotron2Trainer(model, compute_dtype=dtype, init_loss_scale=scale, mask_padding=masked)
    text, tl, mel, gate, ml = TO.seeded_batch(c)
    loss = tr.forward(text.to(cuda), tl.to(cuda), mel.to(cuda), gate.to(cuda), ml.to(cuda))
    tr.backward()
    assert bool(torch.isfinite(tr.g.flat).all())
    replay = _Replay(_engine_masks(tr, F), mel.shape[2], text.shape[0], cfg["attention_rnn_dim"], cfg["decoder_rnn_dim"])
    p = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    lo, (_, _, _, align) = TO.tacotron2_loss(p, cfg, text, tl, mel, gate, replay, output_lengths=ml if masked else None)
    lo.backward()
    assert replay.calls == len(replay.sites)
    if masked:                                                 # the flag changes the loss: the unmasked value is not within the bar
        lo_un = TO.tacotron2_loss({k: v.detach() for k, v in p.items()}, cfg, text, tl, mel, gate,
                                  _Replay(_engine_masks(tr, F), mel.shape[2], text.shape[0], cfg["attention_rnn_dim"],
                                          cfg["decoder_rnn_dim"]))[0]
        assert abs(float(lo_un) - float(lo.detach())) > 5e-3 * abs(float(lo.detach()))
    assert abs(float(loss) - float(lo.detach())) <= 1e-3 * abs(float(lo.detach())), (float(loss), float(lo.detach()))
    _close(tr.sv["aw"].permute(1, 0, 2), align, rtol=5e-2, atol=2e-3 if dtype == torch.float16 else 1e-2)
    # floor of the worst tensor (always an encoder one: its gradient crosses the decoder sweep, the attention, the bi-LSTM sweep and
    # three conv + BatchNorm layers) over six mask draws with the doubles: fp16 1 - 6 %, bf16 ~14 %; medians 0.3 % / 2 %
    worst_bar, med_bar = (0.15, 1.5e-2) if dtype == torch.float16 else (0.35, 6e-2)
    errs = {}
    for k, v in p.items():
        if float(v.grad.norm()) > 1e-5:                        # (conv biases in front of a BatchNorm have zero gradient)
            errs[k] = float((tr.g[k].cpu() / scale - v.grad).norm() / v.grad.norm())
    bad = {k: e for k, e in errs.items() if e > worst_bar}
    assert not bad, bad
    assert float(np.median(list(errs.values()))) <= med_bar
    # keep rates of the drawn masks
    log = _engine_masks(tr, F)
    assert abs(float(log[0].float().mean()) - 0.5) < 0.03 and abs(float(log[5].float().mean()) - 0.9) < 0.03


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_validation_pass_vs_oracle_in_eval_mode(cuda, dtype):
    """Tacotron2Trainer.eval_loss == the oracle's eval-mode loss (train.py:273-318: BatchNorm on its running buffers, only the
    prenet's dropout on) under the two prenet masks the engine drew; the pass leaves buffers and training state untouched."""
    from oracle import tacotron2_oracle as TO
    from deeplearningexamples_amd import functional as F
    from deeplearningexamples_amd.tacotron2.engine import Tacotron2Trainer
    from deeplearningexamples_amd.tacotron2.model import Tacotron2
    c = TO.TACOTRON2_CASE
    cfg = c["cfg"]
    state = dict(TO.seeded_state(cfg, c["seed"]))
    state.update(TO.seeded_running_stats(cfg, c["seed"]))
    model = Tacotron2(device=cuda, **cfg)
    model.load_reference_state(state)
    tr = Tacotron2Trainer(model, compute_dtype=dtype)
    text, tl, mel, gate, ml = TO.seeded_batch(c)
    dev = [t.to(cuda) for t in (text, tl, mel, gate, ml)]
    bufs = {k: v.clone() for k, v in model.named_buffers()}
    # the engine clears what it saved after the pass: catch the prenet masks while it runs
    masks = []
    real = tr._drop

    def spy(x, p):
        y, m = real(x, p)
        masks.append(F.unpack_dropout_mask(m, x.shape).cpu())
        return y, m

    tr._drop = spy
    loss = tr.eval_loss(*dev)
    tr._drop = real
    assert len(masks) == 2 and tr.training and tr.sv is None
    for k, v in model.named_buffers():
        assert torch.equal(v, bufs[k]), k

    class Replay:
        calls = 0

        def __call__(self, x, p):
            keep = masks[self.calls].reshape(x.shape)
            self.calls += 1
            return x * keep / (1.0 - p)

    with torch.no_grad():
        lo, _ = TO.tacotron2_loss(state, cfg, text, tl, mel, gate, Replay(), training=False)
    tol = 2e-3 if dtype == torch.float16 else 1e-2
    assert abs(float(loss) - float(lo)) <= tol * abs(float(lo)), (float(loss), float(lo))
    # and differs from the training-mode loss on the same batch (batch statistics, dropout everywhere)
    assert abs(float(tr.forward(*dev[:4])) - float(lo)) > 5e-3 * abs(float(lo))


def _oracle_under_engine_masks(TO, F, tr, p, cfg, batch):
    """Oracle loss + autograd under the keep masks the HIP RNG drew for the engine's last forward."""
    from tests.test_tacotron2_host import _Replay
    text, tl, mel, gate = batch
    replay = _Replay(_engine_masks(tr, F), mel.shape[2], text.shape[0], cfg["attention_rnn_dim"], cfg["decoder_rnn_dim"])
    lo, outs = TO.tacotron2_loss(p, cfg, text, tl, mel, gate, replay)
    assert replay.calls == len(replay.sites)
    return lo, outs


@pytest.mark.parametrize("dtype", DTYPES)
def test_default_widths_step_vs_oracle_under_the_hip_masks(cuda, dtype, monkeypatch):
    """The network bench.py times (tacotron2/arg_parser.py:40-107: 512-wide encoder, 1024-unit LSTM cells, attention 128 / 32 x 31,
    prenet 256; 28.2 M parameters) on 4 utterances, 40 text positions, 60 decoder steps: loss 1e-3, alignments, every parameter
    gradient (tacotron2/model.py:405-519, loss_function.py:31-46).  Gradient bars = 1.3 x the 16-bit STORAGE floor of this case
    UNDER THE SAME MASKS: the product engine re-run on the CPU over the fp64-accumulating doubles with 16-bit storage
    (tools/storage_floor_f1.py's measurement; over mask draws the worst-tensor floor moves between 1.9 and 3 % in fp16, 8 - 10 %
    in bf16 -- profiles/old/r03_t2_default_storage_floor.txt -- so it is taken for the draw at hand, not from a table)."""
    from oracle import tacotron2_oracle as TO
    from deeplearningexamples_amd import functional as F
    from deeplearningexamples_amd.tacotron2.engine import Tacotron2Trainer
    from deeplearningexamples_amd.tacotron2.model import Tacotron2
    c = TO.TACOTRON2_DEFAULT_CASE
    cfg = c["cfg"]
    state = TO.seeded_state(cfg, c["seed"])
    model = Tacotron2(device=cuda, **cfg)
    model.load_reference_state(state)
    scale = 65536.0
    tr = Tacotron2Trainer(model, compute_dtype=dtype, init_loss_scale=scale)
    text, tl, mel, gate, ml = TO.seeded_batch(c)
    loss = tr.forward(text.to(cuda), tl.to(cuda), mel.to(cuda), gate.to(cuda))
    tr.backward()
    assert bool(torch.isfinite(tr.g.flat).all())
    masks = _engine_masks(tr, F)
    p = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    lo, (_, _, _, align) = _oracle_under_engine_masks(TO, F, tr, p, cfg, (text, tl, mel, gate))
    lo.backward()
    assert abs(float(loss) - float(lo.detach())) <= 1e-3 * abs(float(lo.detach())), (float(loss), float(lo.detach()))
    _close(tr.sv["aw"].permute(1, 0, 2), align, rtol=5e-2, atol=2e-3 if dtype == torch.float16 else 1e-2)
    live = [k for k, v in p.items() if float(v.grad.norm()) > 1e-5]
    errs = {k: float((tr.g[k].cpu() / scale - p[k].grad).norm() / p[k].grad.norm()) for k in live}
    # the floor: same engine, same masks, same storage dtype, fp64-accumulating plain-torch doubles on the CPU
    D.install(monkeypatch)
    D.Masks.reset(0, replay=masks)
    cpu_model = Tacotron2(**cfg)
    cpu_model.load_reference_state(state)
    ftr = Tacotron2Trainer(cpu_model, compute_dtype=dtype, amp=True, init_loss_scale=scale)
    ftr.forward(text, tl, mel, gate)
    ftr.backward()
    floor = {k: float((ftr.g[k] / scale - p[k].grad).norm() / p[k].grad.norm()) for k in live}
    worst_floor, med_floor = max(floor.values()), float(np.median(list(floor.values())))
    print(dtype, "worst / median relative L2 gradient error: HIP %.3e / %.3e, 16-bit storage floor under the same masks %.3e / %.3e"
          % (max(errs.values()), float(np.median(list(errs.values()))), worst_floor, med_floor))
    assert worst_floor < (0.05 if dtype == torch.float16 else 0.2)                     # the floor itself is a usable bar
    bad = {k: (e, floor[k]) for k, e in errs.items() if e > 1.3 * worst_floor}
    assert not bad, bad
    assert float(np.median(list(errs.values()))) <= 1.3 * med_floor


@pytest.mark.parametrize("case_name", ["small", "default"])
def test_three_steps_follow_torch_adam_on_the_oracle(cuda, case_name):
    """train.py:474-500 for -m Tacotron2: forward, scaled backward, unscale + clip_grad_norm_(1.0), Adam(weight decay 1e-6),
    scaler.update -- three iterations; the oracle + torch.optim.Adam + clip_grad_norm_ take the same three steps under the masks the
    HIP RNG drew at each step: loss trajectory and the weights after the third step."""
    from oracle import tacotron2_oracle as TO
    from deeplearningexamples_amd import functional as F
    from deeplearningexamples_amd.tacotron2.engine import Tacotron2Trainer
    from deeplearningexamples_amd.tacotron2.model import Tacotron2
    c = TO.TACOTRON2_CASE if case_name == "small" else TO.TACOTRON2_DEFAULT_CASE
    cfg = c["cfg"]
    state = TO.seeded_state(cfg, c["seed"])
    model = Tacotron2(device=cuda, **cfg)
    model.load_reference_state(state)
    tr = Tacotron2Trainer(model, compute_dtype=torch.float16, lr=1e-3, weight_decay=1e-6, grad_clip_thresh=1.0, init_loss_scale=65536.0)
    text, tl, mel, gate, ml = TO.seeded_batch(c)
    dev_batch = [t.to(cuda) for t in (text, tl, mel, gate)]
    p = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    opt = torch.optim.Adam(list(p.values()), lr=1e-3, weight_decay=1e-6)
    ref, got = [], []
    for _ in range(3):
        loss = tr.forward(*dev_batch)
        tr.backward()
        opt.zero_grad()
        lo, _ = _oracle_under_engine_masks(TO, F, tr, p, cfg, (text, tl, mel, gate))
        lo.backward()
        torch.nn.utils.clip_grad_norm_(list(p.values()), 1.0)
        opt.step()
        tr.optimizer_step()
        ref.append(float(lo.detach()))
        got.append(float(loss))
    np.testing.assert_allclose(got, ref, rtol=2e-3)
    assert int(tr.step_t) == 3 and float(tr.scaler.found_inf) == 0
    assert ref[-1] < ref[0]
    sd = model.state_dict()
    moved = sum(float((p[k].detach() - state[k]).norm()) ** 2 for k in p) ** 0.5
    dist = sum(float((sd[k].cpu() - p[k].detach()).norm()) ** 2 for k in p) ** 0.5
    assert dist <= 0.15 * moved, (dist, moved)         # Adam's sign-like first steps amplify 16-bit gradient noise near g = 0
    assert int(sd["postnet.convolutions.0.1.num_batches_tracked"]) == 3


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("b,ti,a,e,kl", [(3, 23, 32, 64, 31), (24, 160, 128, 512, 31), (2, 40, 64, 128, 5)])
def test_attention_step_with_the_location_term_fused(cuda, dtype, b, ti, a, e, kl):
    """wloc / wloc_t given: the kernels form the location term (2-channel convolution over the previous / cumulative weights +
    dense, one pre-multiplied [A, KK] operand, model.py:40-76) and its transposed convolution themselves -- against the plain-torch
    statement, and against the unfused HIP path (row gather + GEMM + kernel) on the same inputs."""
    ops = _ops()
    g = torch.Generator().manual_seed(b * 3 + ti)
    kk = (2 * kl + 31) // 32 * 32
    q = torch.randn(b, a, generator=g)
    pm = torch.randn(b * ti, a, generator=g).to(dtype)
    v = torch.randn(a, generator=g) * 0.5
    mem = torch.randn(b * ti, e, generator=g).to(dtype)
    lengths = torch.randint(ti // 2, ti + 1, (b,), generator=g)
    lengths[0] = ti
    awc_prev = torch.zeros(b * ti, 8, dtype=dtype)
    awc_prev[:, :2] = torch.rand(b * ti, 2, generator=g).to(dtype)
    wloc = torch.zeros(a, kk, dtype=dtype)
    wloc[:, :2 * kl] = (torch.randn(a, 2 * kl, generator=g) * 0.3).to(dtype)
    wloc_t = wloc.t().contiguous()

    def run(L, d):
        th, aw, nxt = d(torch.zeros(b * ti, a, dtype=dtype)), d(torch.zeros(b, ti)), d(torch.ones(b * ti, 8, dtype=dtype))
        c0 = d(torch.zeros(b, e, dtype=dtype))
        L.attention_fwd(d(q), d(pm), d(v), d(mem), d(lengths), d(awc_prev), th, aw, nxt, [c0], wloc=d(wloc), kl=kl)
        return th, aw, nxt, c0
    got, ref = run(ops, lambda t: t.to(cuda)), run(D, lambda t: t)
    _close(got[0], ref[0], **_tol(dtype))
    _close(got[1], ref[1], rtol=3e-2, atol=3e-4)
    _close(got[2], ref[2], **_tol(dtype))
    _close(got[3], ref[3], **_tol(dtype))
    th, aw = ref[0], ref[1]
    d_ctx, d_aw_in = torch.randn(b, e, generator=g), torch.randn(b, ti, generator=g) * 0.1
    cum0 = torch.randn(b, ti, generator=g) * 0.1

    def runb(L, d):
        dpl, dq16 = d(torch.zeros(b * ti, a, dtype=dtype)), d(torch.zeros(b, a, dtype=dtype))
        dv = d(torch.zeros(b, a))
        prev, cum = d(d_aw_in.clone()), d(cum0.clone())           # in place: d_aw0 -> d_prev, d_aw1 -> d_cum
        L.attention_bwd(d(d_ctx), prev, d(aw), d(th), d(v), d(mem), None, dpl, None, dv, None, d_aw_add=cum, dq16=dq16,
                        wloc_t=d(wloc_t), kl=kl, d_prev=prev, d_cum=cum)
        return dpl, dq16, dv, prev, cum
    got, ref = runb(ops, lambda t: t.to(cuda)), runb(D, lambda t: t)
    _close(got[0], ref[0], **_tol(dtype))
    _close(got[1], ref[1], **_tol(dtype))
    _close(got[2], ref[2], rtol=2e-3, atol=5e-3)
    scale = float(ref[3].abs().max())
    _close(got[3], ref[3], rtol=2e-3, atol=2e-3 * scale)
    _close(got[4], ref[4], rtol=2e-3, atol=2e-3 * max(scale, float(ref[4].abs().max())))
