"""Host sequencing of the WaveGlow engine on the CPU: the C-ABI calls are replaced by the plain-torch
test doubles of tests/_waveglow_doubles.py (fp32), everything else -- layouts, column / row slices, which gradient lands in
which slot, the optimizer sequence -- is the product's code, checked against the fixture the REFERENCE's own WaveGlow +
WaveGlowLoss produced (tests/golden/waveglow_loss.npz) and against torch.optim.Adam.  The kernels themselves are checked on
the GPU (tests/test_gpu_waveglow.py)."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _trainer(monkeypatch, amp, **kw):
    from oracle import waveglow_oracle as WO
    from tests import _waveglow_doubles as D
    from deeplearningexamples_amd.waveglow.engine import WaveGlowTrainer
    from deeplearningexamples_amd.waveglow.model import WaveGlow
    D.install(monkeypatch)
    c = WO.WAVEGLOW_CASE
    torch.manual_seed(0)
    model = WaveGlow(**c["cfg"])
    state = WO.seeded_state(c["cfg"], c["seed"])
    model.load_reference_state(state)
    tr = WaveGlowTrainer(model, compute_dtype=torch.float32, amp=amp, sigma=c["sigma"], **kw)
    return WO, c, model, state, tr


def test_state_dict_names_and_shapes_match_the_reference():
    from oracle import waveglow_oracle as WO
    from deeplearningexamples_amd.waveglow.model import DEFAULT_CONFIG, WaveGlow
    for cfg in (WO.WAVEGLOW_SMALL, DEFAULT_CONFIG):
        if cfg is DEFAULT_CONFIG:
            from deeplearningexamples_amd.waveglow.model import param_layout
            got = {n: s for n, s, _ in param_layout(cfg)}
        else:
            got = {k: tuple(v.shape) for k, v in WaveGlow(**cfg).state_dict().items()}
        assert got == WO.param_shapes(cfg)


def test_initial_distributions_follow_the_reference():
    from oracle import waveglow_oracle as WO
    from deeplearningexamples_amd.waveglow.model import WaveGlow
    torch.manual_seed(3)
    m = WaveGlow(**WO.WAVEGLOW_SMALL)
    sd = m.state_dict()
    assert float(sd["WN.1.end.weight"].abs().max()) == 0 and float(sd["WN.1.end.bias"].abs().max()) == 0
    w = sd["convinv.2.conv.weight"].squeeze(-1)
    assert torch.allclose(w @ w.t(), torch.eye(w.shape[0]), atol=1e-5) and float(torch.det(w)) > 0
    v, g = sd["WN.0.in_layers.1.weight_v"], sd["WN.0.in_layers.1.weight_g"]
    assert torch.allclose(g.view(-1), v.flatten(1).norm(dim=1))
    assert float(v.abs().max()) <= 1.0 / np.sqrt(v.shape[1] * v.shape[2]) + 1e-7


@pytest.mark.parametrize("amp", [False, True])
def test_engine_sequence_reproduces_reference_loss_and_gradients(monkeypatch, amp):
    WO, c, model, state, tr = _trainer(monkeypatch, amp, init_loss_scale=1024.0)
    gold = np.load(os.path.join(HERE, "golden", "waveglow_loss.npz"))
    mel, audio = WO.seeded_inputs(c)
    loss = tr.forward(mel, audio)
    assert abs(float(loss) - float(gold["loss"][0])) <= 2e-6 * abs(float(gold["loss"][0]))
    tr.backward()
    s = float(tr.scaler.scale)
    assert s == (1024.0 if amp else 1.0)
    for k in [f[len("gnorm."):] for f in gold.files if f.startswith("gnorm.")]:
        ref = float(gold["gnorm." + k][0])
        assert abs(float(tr.g[k].norm()) / s - ref) <= 3e-4 * ref + 1e-9, k
    for k in [f[len("grad."):] for f in gold.files if f.startswith("grad.")]:
        np.testing.assert_allclose(tr.g[k].numpy().reshape(-1)[:64] / s, gold["grad." + k], rtol=3e-4, atol=2e-7)
    # padded slots stay zero (the 8-wide `end` GEMM writes whole slots)
    off, n = tr.g.offsets["WN.2.end.weight"]
    used = tr.g["WN.2.end.weight"].numel()
    assert float(tr.g.flat[off + used:off + n].abs().max()) == 0


def test_optimizer_sequence_matches_torch_adam_with_clipping(monkeypatch):
    WO, c, model, state, tr = _trainer(monkeypatch, True, init_loss_scale=256.0, lr=1e-3, grad_clip_thresh=0.5, weight_decay=1e-6)
    mel, audio = WO.seeded_inputs(c)
    # reference sequence (train.py:487-497) on the oracle's autograd
    p = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    opt = torch.optim.Adam(list(p.values()), lr=1e-3, weight_decay=1e-6)
    ref_losses, got = [], []
    for _ in range(3):
        opt.zero_grad()
        lo = WO.waveglow_loss(p, c["cfg"], mel, audio, c["sigma"])
        lo.backward()
        torch.nn.utils.clip_grad_norm_(list(p.values()), 0.5)
        opt.step()
        ref_losses.append(float(lo.detach()))
        got.append(float(tr.train_step(mel, audio)))
    np.testing.assert_allclose(got, ref_losses, rtol=2e-5)
    # Adam's first steps move every element by ~lr * sign(g): where g is at rounding-noise level the two fp32 evaluations may
    # disagree on the direction, so the bar is "a small fraction of the distance travelled" per tensor, not element-wise
    for k, v in model.state_dict().items():
        moved = float((p[k].detach() - state[k]).norm())
        assert float((v - p[k].detach()).norm()) <= 2e-3 * moved + 1e-9, k
    assert int(tr.step_t) == 3


def test_overflow_skips_the_step_and_backs_the_scale_off(monkeypatch):
    WO, c, model, state, tr = _trainer(monkeypatch, True, init_loss_scale=1024.0)
    mel, audio = WO.seeded_inputs(c)
    before = tr.p.flat.clone()
    tr.forward(mel, audio)
    tr.backward()
    tr.g.flat[5] = float("inf")
    tr.optimizer_step()
    assert torch.equal(tr.p.flat, before) and int(tr.step_t) == 0
    assert float(tr.scaler.scale) == 512.0 and float(tr.scaler.found_inf) == 0


def test_segment_that_is_not_a_multiple_of_the_hop(monkeypatch):
    """8000-sample segments (the reference's --segment-length 8000) end inside a 256-sample frame block: model.py:199-200."""
    WO, c, model, state, tr = _trainer(monkeypatch, False)
    rng = np.random.default_rng(5)
    t = 1000
    mel = torch.from_numpy(rng.standard_normal((2, 80, 4)).astype(np.float32))
    audio = torch.from_numpy((rng.standard_normal((2, t)) * 0.2).astype(np.float32))
    p = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    lo = WO.waveglow_loss(p, c["cfg"], mel, audio, c["sigma"])
    lo.backward()
    loss = tr.forward(mel, audio)
    assert abs(float(loss) - float(lo)) <= 2e-6 * abs(float(lo))
    tr.backward()
    for k in ("upsample.weight", "upsample.bias", "WN.0.cond_layers.1.weight_v", "convinv.3.conv.weight"):
        assert torch.allclose(tr.g[k], p[k].grad, rtol=3e-4, atol=2e-7), k


def _dp_worker(rank, world, port, ret):
    """One rank of a world-size-2 gloo run of the engine's data-parallel path (doubles for the kernels)."""
    import pytest as _pytest
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import waveglow_oracle as WO
        from tests import _waveglow_doubles as D
        from deeplearningexamples_amd.waveglow.engine import WaveGlowTrainer
        from deeplearningexamples_amd.waveglow.model import WaveGlow
        D.install(_pytest.MonkeyPatch())
        c = WO.WAVEGLOW_CASE
        torch.manual_seed(50 + rank)                                  # different initial replicas: the trainer must broadcast
        model = WaveGlow(**c["cfg"])
        if rank == 0:
            model.load_reference_state(WO.seeded_state(c["cfg"], c["seed"]))
        tr = WaveGlowTrainer(model, compute_dtype=torch.float32, amp=True, init_loss_scale=256.0, world_size=world, bucket_mb=1)
        mel, audio = WO.seeded_inputs(dict(c, batch=4))
        per = 4 // world
        tr.forward(mel[rank * per:(rank + 1) * per].contiguous(), audio[rank * per:(rank + 1) * per].contiguous())
        tr.backward()
        tr.buckets.wait()
        ret[rank] = (tr.g.flat.clone(), tr.p.flat.clone(), len(tr.buckets.buckets))
    finally:
        dist.destroy_process_group()


def test_data_parallel_gradients_two_ranks_gloo(monkeypatch):
    """Two ranks with the two halves of a batch: the bucketed mean all-reduce leaves the full-batch gradient on both; replicas
    built from different seeds start from rank 0's weights (DDP semantics, train.py:402-407)."""
    import torch.multiprocessing as mp
    port = 30100 + os.getpid() % 2000
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_dp_worker, args=(2, port, ret), nprocs=2, join=True)
        (g0, p0, nb), (g1, p1, _) = ret[0], ret[1]
    assert nb > 1 and torch.equal(g0, g1) and torch.equal(p0, p1)
    WO, c, model, state, tr = _trainer(monkeypatch, True, init_loss_scale=256.0)
    mel, audio = WO.seeded_inputs(dict(c, batch=4))
    tr.forward(mel, audio)
    tr.backward()
    assert torch.equal(p0, tr.p.flat)
    assert float((g0 - tr.g.flat).norm()) <= 1e-4 * float(tr.g.flat.norm())


def test_reference_size_network_host_sequence(monkeypatch):
    """The reference's default network (12 flows x 8 layers x 512 channels, 268 M parameters; waveglow/arg_parser.py:38-64) on a
    2 x 2048-sample batch: the engine's sequence (98,304-column cond / pre-activation matrices, 96-slice batched weight gradients,
    table-driven weight norm) against the oracle's autograd."""
    from oracle import waveglow_oracle as WO
    from tests import _waveglow_doubles as D
    from deeplearningexamples_amd.waveglow.engine import WaveGlowTrainer
    from deeplearningexamples_amd.waveglow.model import DEFAULT_CONFIG, WaveGlow
    D.install(monkeypatch)
    state = WO.seeded_state(DEFAULT_CONFIG, 11)
    model = WaveGlow(**DEFAULT_CONFIG)
    model.load_reference_state(state)
    tr = WaveGlowTrainer(model, compute_dtype=torch.float32, amp=True, init_loss_scale=64.0)
    mel, audio = WO.seeded_inputs(dict(cfg=DEFAULT_CONFIG, seed=11, batch=2, segment=2048))
    loss = tr.forward(mel, audio)
    tr.backward()
    p = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    lo = WO.waveglow_loss(p, DEFAULT_CONFIG, mel, audio, 1.0)
    lo.backward()
    assert abs(float(loss) - float(lo.detach())) <= 5e-6 * abs(float(lo.detach()))
    for k, v in p.items():
        assert float((tr.g[k] / 64.0 - v.grad).norm()) <= 2e-3 * float(v.grad.norm()) + 1e-7, k
    assert tr.p.flat.numel() >= 268e6
