"""The WaveGlow loss oracle (checker of deeplearningexamples_amd/waveglow) against the fixture the REFERENCE's
own WaveGlow + WaveGlowLoss produced on CPU (tests/golden/waveglow_loss.npz, oracle/make_golden.py gen_waveglow: loss, the
gradient norm of every parameter, gradient slices).  CPU only."""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def test_waveglow_oracle_reproduces_reference_loss_and_gradients():
    from oracle import waveglow_oracle as WO
    c = WO.WAVEGLOW_CASE
    gold = np.load(os.path.join(HERE, "golden", "waveglow_loss.npz"))
    p = {k: v.clone().requires_grad_(True) for k, v in WO.seeded_state(c["cfg"], c["seed"]).items()}
    mel, audio = WO.seeded_inputs(c)
    loss = WO.waveglow_loss(p, c["cfg"], mel, audio, c["sigma"])
    loss.backward()
    assert abs(float(loss.detach()) - float(gold["loss"][0])) <= 1e-6 * abs(float(gold["loss"][0]))
    names = [k[len("gnorm."):] for k in gold.files if k.startswith("gnorm.")]
    assert sorted(names) == sorted(p), "fixture and oracle disagree on the parameter set"
    for k in names:
        ref = float(gold["gnorm." + k][0])
        assert abs(float(p[k].grad.norm()) - ref) <= 2e-4 * ref + 1e-9, k
    for k in [f[len("grad."):] for f in gold.files if f.startswith("grad.")]:
        np.testing.assert_allclose(p[k].grad.numpy().reshape(-1)[:64], gold["grad." + k], rtol=2e-4, atol=1e-7)
    # 12 flows / 8 layers / 512 channels (the reference's default arg_parser values): the shape table covers them too
    full = dict(n_mel_channels=80, n_flows=12, n_group=8, n_early_every=4, n_early_size=2,
                WN_config=dict(n_layers=8, n_channels=512, kernel_size=3))
    sh = WO.param_shapes(full)
    assert sh["convinv.11.conv.weight"] == (4, 4, 1) and sh["WN.0.cond_layers.7.weight_v"] == (1024, 640, 1)
    assert sh["WN.11.end.weight"] == (4, 512, 1) and sh["WN.5.res_skip_layers.7.weight_v"] == (512, 512, 1)
