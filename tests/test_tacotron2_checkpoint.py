"""Tacotron2 checkpoint files and CLI (SpeechSynthesis/Tacotron2/train.py:185-255) on the CPU, the engine
on the test doubles of the C-ABI calls: round trip, and interchange with the REFERENCE's own Tacotron2 + torch.optim.Adam when the
reference tree is mounted (build container)."""
import os

import numpy as np
import pytest
import torch

from oracle import _ref_import as R


def _trainer(monkeypatch):
    from oracle import tacotron2_oracle as TO
    from tests import _tacotron2_doubles as D
    from deeplearningexamples_amd.tacotron2.engine import Tacotron2Trainer
    from deeplearningexamples_amd.tacotron2.model import Tacotron2
    D.install(monkeypatch)
    c = TO.TACOTRON2_CASE
    model = Tacotron2(**c["cfg"])
    model.load_reference_state(TO.seeded_state(c["cfg"], c["seed"]))
    return TO, D, c, Tacotron2Trainer(model, compute_dtype=torch.float32, amp=True, init_loss_scale=512.0, lr=1e-3)


def test_checkpoint_round_trip_resumes_identically(monkeypatch, tmp_path):
    from deeplearningexamples_amd.tacotron2 import train as T
    from deeplearningexamples_amd.waveglow import train as WT
    TO, D, c, tr = _trainer(monkeypatch)
    names = T.parameter_order(c["cfg"])
    batch = TO.seeded_batch(c)[:4]
    D.Masks.reset(1)
    for _ in range(2):
        tr.train_step(*batch)
    path = WT.save_checkpoint(tr, 5, tr.cfg, str(tmp_path), "Tacotron2", 0, 1, names)
    assert os.path.basename(path) == "checkpoint_Tacotron2_5.pt"
    D.Masks.reset(2)
    cont = [float(tr.train_step(*batch)) for _ in range(2)]
    TO, D, c, tr2 = _trainer(monkeypatch)
    cfg, epoch = WT.load_checkpoint(tr2, path, 0, names)
    assert epoch == 6 and int(tr2.step_t) == 2 and float(tr2.scaler.scale) == 512.0
    sd, sd2 = tr.model.state_dict(), tr2.model.state_dict()
    D.Masks.reset(2)
    resumed = [float(tr2.train_step(*batch)) for _ in range(2)]
    np.testing.assert_allclose(resumed, cont, rtol=1e-6)
    assert int(sd2["encoder.convolutions.1.1.num_batches_tracked"]) == 4 and torch.equal(tr.p.flat, tr2.p.flat)
    assert torch.equal(sd["postnet.convolutions.2.1.running_var"], tr2.model.state_dict()["postnet.convolutions.2.1.running_var"])


@pytest.mark.skipif(not R.have_reference(), reason="reference tree not mounted (GPU box): checked in the build container")
def test_checkpoint_loads_into_the_reference_classes(monkeypatch, tmp_path):
    from deeplearningexamples_amd.tacotron2 import train as T
    from deeplearningexamples_amd.waveglow import train as WT
    ref = R.import_tacotron2()
    TO, D, c, tr = _trainer(monkeypatch)
    names = T.parameter_order(c["cfg"])
    batch = TO.seeded_batch(c)[:4]
    D.Masks.reset(1)
    tr.train_step(*batch)
    path = WT.save_checkpoint(tr, 0, tr.cfg, str(tmp_path), "Tacotron2", 0, 1, names)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    rm = ref.model.Tacotron2(mask_padding=False, max_decoder_steps=2000, gate_threshold=0.5, decoder_no_early_stopping=False,
                             **c["cfg"])
    assert [n for n, _ in rm.named_parameters()] == names                 # Adam's state is indexed by this order
    assert set(rm.state_dict()) == set(ck["state_dict"])
    rm.load_state_dict(ck["state_dict"])
    opt = torch.optim.Adam(rm.parameters(), lr=1e-3, weight_decay=1e-6)
    opt.load_state_dict(ck["optimizer"])
    i = names.index("decoder.attention_rnn.weight_hh")
    assert torch.allclose(opt.state_dict()["state"][i]["exp_avg"], tr.m["decoder.attention_rnn.weight_hh"])
    assert float(opt.state_dict()["state"][i]["step"]) == 1.0
    assert float(rm.state_dict()["encoder.convolutions.0.1.running_mean"].abs().max()) > 0      # BatchNorm buffers travel too


def test_cli_flags_follow_train_py():
    from deeplearningexamples_amd.tacotron2 import train as T
    a = T.parse_args("-m Tacotron2 -o output/ --amp -lr 1e-3 --epochs 1501 -bs 128 --weight-decay 1e-6 --grad-clip-thresh 1.0 "
                     "--cudnn-enabled --log-file nvlog.json --anneal-steps 500 1000 1500 --anneal-factor 0.3".split())
    assert (a.batch_size, a.learning_rate, a.anneal_steps, a.anneal_factor, a.prenet_dim) == (128, 1e-3, ["500", "1000", "1500"], 0.3, 256)
    cfg = T.get_model_config(a)
    assert cfg["attention_rnn_dim"] == 1024 and cfg["postnet_n_convolutions"] == 5 and cfg["n_symbols"] == 148
