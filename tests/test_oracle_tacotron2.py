"""The Tacotron2 loss oracle (checker of deeplearningexamples_amd/tacotron2) against the
fixture the REFERENCE's own Tacotron2 + Tacotron2Loss produced on CPU in training mode with the same dropout masks
(tests/golden/tacotron2_loss.npz, oracle/make_golden.py gen_tacotron2: loss, the gradient norm of all 60 parameters, gradient
slices, the last alignment row).  CPU only."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("masked", [False, True])
def test_tacotron2_oracle_reproduces_reference_loss_and_gradients(masked):
    """masked: --mask-padding (model.py:648-655); that fixture's gradients come from the reference's parse_output run on clones,
    because the unmodified reference raises in backward with the flag on (asserted by the generator)."""
    from oracle import tacotron2_oracle as TO
    c = TO.TACOTRON2_CASE
    gold = np.load(os.path.join(HERE, "golden", "tacotron2_loss_masked.npz" if masked else "tacotron2_loss.npz"))
    p = {k: v.clone().requires_grad_(True) for k, v in TO.seeded_state(c["cfg"], c["seed"]).items()}
    text, tl, mel, gate, ml = TO.seeded_batch(c)
    stream = TO.MaskStream(c["seed"] + 2)
    loss, (mel_out, mel_post, gate_out, align) = TO.tacotron2_loss(p, c["cfg"], text, tl, mel, gate, stream,
                                                                    output_lengths=ml if masked else None)
    loss.backward()
    assert stream.calls == int(gold["dropout_calls"][0]) == 3 + 2 + 2 * int(ml.max()) + 5
    assert abs(float(loss.detach()) - float(gold["loss"][0])) <= 2e-6 * abs(float(gold["loss"][0]))
    np.testing.assert_allclose(align[:, -1].detach().numpy(), gold["alignment_last"], atol=1e-6)
    # attention never looks at padded text positions, rows sum to one
    pad = torch.arange(text.shape[1])[None, :] >= tl[:, None]
    assert float(align.detach()[pad[:, None, :].expand_as(align)].abs().max()) == 0
    np.testing.assert_allclose(align.detach().sum(2).numpy(), 1.0, atol=1e-5)
    if masked:
        past = torch.arange(mel.shape[2])[None, :] >= ml[:, None]
        assert float(mel_out.detach()[past[:, None, :].expand_as(mel_out)].abs().max()) == 0
        assert float(mel_post.detach()[past[:, None, :].expand_as(mel_post)].abs().max()) == 0
        assert bool((gate_out.detach()[past] == 1e3).all())
    names = [k[len("gnorm."):] for k in gold.files if k.startswith("gnorm.")]
    assert sorted(names) == sorted(p) and len(names) == 60
    for k in names:
        ref = float(gold["gnorm." + k][0])
        # (a convolution bias in front of a training-mode BatchNorm has a mathematically zero gradient: rounding noise ~1e-7)
        assert abs(float(p[k].grad.norm()) - ref) <= 5e-4 * ref + 1e-6, k
    for k in [f[len("grad."):] for f in gold.files if f.startswith("grad.")]:
        np.testing.assert_allclose(p[k].grad.numpy().reshape(-1)[:64], gold["grad." + k], rtol=5e-4, atol=1e-6)
    # the reference's default widths (tacotron2/arg_parser.py:40-107) are in the shape table too
    sh = TO.param_shapes(TO.TACOTRON2_DEFAULT)
    assert sh["decoder.attention_rnn.weight_ih"] == (4096, 768) and sh["decoder.decoder_rnn.weight_ih"] == (4096, 1536)
    assert sh["decoder.linear_projection.linear_layer.weight"] == (80, 1536) and sh["encoder.lstm.weight_hh_l0_reverse"] == (1024, 256)
    assert sum(int(np.prod(s)) for s in sh.values()) == 28193153                    # 28.2 M trainable parameters


def test_tacotron2_oracle_reproduces_the_reference_validation_pass():
    """model.eval() (train.py:273-318): BatchNorm on seeded running buffers, only the prenet's two dropouts drawn."""
    from oracle import tacotron2_oracle as TO
    c = TO.TACOTRON2_CASE
    gold = np.load(os.path.join(HERE, "golden", "tacotron2_loss_eval.npz"))
    p = dict(TO.seeded_state(c["cfg"], c["seed"]))
    p.update(TO.seeded_running_stats(c["cfg"], c["seed"]))
    text, tl, mel, gate, ml = TO.seeded_batch(c)
    stream = TO.MaskStream(c["seed"] + 2)
    with torch.no_grad():
        loss, (_, mel_post, _, align) = TO.tacotron2_loss(p, c["cfg"], text, tl, mel, gate, stream, training=False)
    assert stream.calls == int(gold["dropout_calls"][0]) == 2
    assert abs(float(loss) - float(gold["loss"][0])) <= 2e-6 * abs(float(gold["loss"][0]))
    np.testing.assert_allclose(align[:, -1].numpy(), gold["alignment_last"], atol=1e-6)
    np.testing.assert_allclose(mel_post[:, :4].numpy(), gold["mel_post_slice"], atol=5e-5)
