"""WaveGlow checkpoint files (SpeechSynthesis/Tacotron2/train.py:185-255) on the CPU: the file this
port writes loads into the REFERENCE's own WaveGlow + torch.optim.Adam + GradScaler-state reader and continues identically;
the file the reference's layout produces loads here.  The engine runs on the test doubles of the C-ABI calls (fp32), as in
tests/test_waveglow_host.py; the reference modules are imported from /root/reference when it is mounted (build container)."""
import os

import numpy as np
import pytest
import torch

from oracle import _ref_import as R


def _trainer(monkeypatch, **kw):
    from oracle import waveglow_oracle as WO
    from tests import _waveglow_doubles as D
    from deeplearningexamples_amd.waveglow.engine import WaveGlowTrainer
    from deeplearningexamples_amd.waveglow.model import WaveGlow
    D.install(monkeypatch)
    c = WO.WAVEGLOW_CASE
    model = WaveGlow(**c["cfg"])
    model.load_reference_state(WO.seeded_state(c["cfg"], c["seed"]))
    return WO, c, WaveGlowTrainer(model, compute_dtype=torch.float32, amp=True, init_loss_scale=512.0, lr=1e-4, **kw)


def test_checkpoint_round_trip_resumes_identically(monkeypatch, tmp_path):
    from deeplearningexamples_amd.waveglow import train as T
    WO, c, tr = _trainer(monkeypatch)
    mel, audio = WO.seeded_inputs(c)
    for _ in range(2):
        tr.train_step(mel, audio)
    path = T.save_checkpoint(tr, 3, c["cfg"], str(tmp_path), "WaveGlow", 0, 1)
    assert os.path.basename(path) == "checkpoint_WaveGlow_3.pt"
    assert T.get_last_checkpoint_filename(str(tmp_path), "WaveGlow") == path
    cont = [float(tr.train_step(mel, audio)) for _ in range(2)]
    WO, c, tr2 = _trainer(monkeypatch)
    cfg, epoch = T.load_checkpoint(tr2, path, 0)
    assert epoch == 4 and cfg == c["cfg"] and int(tr2.step_t) == 2 and float(tr2.scaler.scale) == 512.0
    resumed = [float(tr2.train_step(mel, audio)) for _ in range(2)]
    np.testing.assert_allclose(resumed, cont, rtol=1e-6)
    assert torch.equal(tr.p.flat, tr2.p.flat)


@pytest.mark.skipif(not R.have_reference(), reason="reference tree not mounted (GPU box): checked in the build container")
def test_checkpoints_interchange_with_the_reference_classes(monkeypatch, tmp_path):
    from deeplearningexamples_amd.waveglow import train as T
    ref = R.import_waveglow()
    WO, c, tr = _trainer(monkeypatch)
    mel, audio = WO.seeded_inputs(c)
    for _ in range(2):
        tr.train_step(mel, audio)
    path = T.save_checkpoint(tr, 0, c["cfg"], str(tmp_path), "WaveGlow", 0, 1)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    # ---- ours -> the reference's model / optimizer (train.py:252-254)
    rm = ref.model.WaveGlow(**ck["config"])
    assert [n for n, _ in rm.named_parameters()] == T.reference_parameter_order(c["cfg"])
    rm.load_state_dict(ck["state_dict"])
    opt = torch.optim.Adam(rm.parameters(), lr=1e-4, weight_decay=0.0)
    opt.load_state_dict(ck["optimizer"])
    assert set(ck["scaler"]) == {"scale", "growth_factor", "backoff_factor", "growth_interval", "_growth_tracker"}
    crit = ref.loss_function.WaveGlowLoss(sigma=c["sigma"])
    rm.train()
    opt.zero_grad()
    lo = crit(rm((mel, audio)), audio)
    lo.backward()
    torch.nn.utils.clip_grad_norm_(rm.parameters(), 65504.0)
    opt.step()
    ours = float(tr.train_step(mel, audio))
    assert abs(ours - float(lo.detach())) <= 2e-6 * abs(ours)
    rsd = rm.state_dict()
    for k, v in tr.model.state_dict().items():
        moved = float((v - ck["state_dict"][k]).norm())
        assert float((v - rsd[k]).norm()) <= 5e-3 * moved + 1e-9, k
    # ---- the reference's file -> ours: state_dict + optimizer.state_dict() + scaler.state_dict() as train.py:205-211 writes them
    ref_ck = {"epoch": 7, "cuda_rng_state_all": torch.zeros(1, 8, dtype=torch.uint8),
              "random_rng_states_all": torch.random.get_rng_state()[None], "config": ck["config"],
              "state_dict": rm.state_dict(), "optimizer": opt.state_dict(),
              "scaler": {"scale": 256.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000, "_growth_tracker": 5}}
    p2 = str(tmp_path / "checkpoint_WaveGlow_7.pt")
    torch.save(ref_ck, p2)
    WO, c, tr3 = _trainer(monkeypatch)
    cfg, epoch = T.load_checkpoint(tr3, p2, 0)
    assert epoch == 8 and int(tr3.step_t) == 3 and float(tr3.scaler.scale) == 256.0 and int(tr3.scaler.growth_tracker) == 5
    opt.zero_grad()
    lo = crit(rm((mel, audio)), audio)
    assert abs(float(tr3.forward(mel, audio)) - float(lo.detach())) <= 2e-6 * abs(float(lo.detach()))
    assert torch.allclose(tr3.m["WN.1.in_layers.0.weight_v"], opt.state_dict()["state"][T.reference_parameter_order(c["cfg"]).index(
        "WN.1.in_layers.0.weight_v")]["exp_avg"])


def test_cli_flags_and_lr_annealing_follow_train_py():
    from deeplearningexamples_amd.waveglow import train as T
    a = T.parse_args("-m WaveGlow -o out --amp -lr 1e-4 --epochs 1001 -bs 10 --segment-length 8000 --weight-decay 0 "
                     "--grad-clip-thresh 65504.0 --cudnn-benchmark --cudnn-enabled --log-file nvlog.json".split())
    assert (a.batch_size, a.segment_length, a.grad_clip_thresh, a.amp, a.wn_channels, a.flows) == (10, 8000, 65504.0, True, 512, 12)
    assert T.get_model_config(a) == dict(n_mel_channels=80, n_flows=12, n_group=8, n_early_every=4, n_early_size=2,
                                         WN_config=dict(n_layers=8, kernel_size=3, n_channels=512))
    # train.py:324-342
    assert T.adjust_learning_rate(10, 1e-3, None, 0.1) == 1e-3
    assert abs(T.adjust_learning_rate(600, 1e-3, ["500", "1000", "1500"], 0.1) - 1e-4) < 1e-12
    assert abs(T.adjust_learning_rate(1200, 1e-3, ["500", "1000", "1500"], 0.3) - 1e-3 * 0.1) < 1e-12
    assert abs(T.adjust_learning_rate(600, 1e-3, ["500", "1000", "1500"], 0.3) - 1e-3 * 0.3) < 1e-12
