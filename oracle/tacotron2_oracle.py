"""CPU restatement of the Tacotron2 training loss (TEST INFRASTRUCTURE ONLY; the Tacotron2 half of SURVEY.md section 8 row f1 --
the checker of the HIP path in deeplearningexamples_amd/tacotron2; nothing in the product imports it).

Follows, in plain fp32 torch on the CPU (paths relative to /root/reference/PyTorch/SpeechSynthesis/Tacotron2/):
    tacotron2/model.py:604-620,667-681  Tacotron2.forward: embedding, encoder, teacher-forced decoder, postnet residual
    tacotron2/model.py:177-216          Encoder: 3 x (Conv1d k5 + BatchNorm1d(train) + ReLU + dropout 0.5), packed bi-LSTM
    tacotron2/model.py:40-121           LocationLayer / Attention: v . tanh(W q + U f(prev, cumulative weights) + V memory), mask -inf,
                                        softmax over time, context = weights x memory
    tacotron2/model.py:124-135          Prenet: 2 x (Linear no bias + ReLU + dropout 0.5, ALWAYS on)
    tacotron2/model.py:405-455,457-519  Decoder.decode / forward: attention LSTMCell, dropout 0.1, attention, decoder LSTMCell,
                                        dropout 0.1, linear projection + gate on [decoder_hidden | context]
    tacotron2/model.py:138-174          Postnet: 4 x (Conv1d k5 + BN + tanh + dropout 0.5) + (Conv1d + BN + dropout 0.5)
    tacotron2/loss_function.py:31-46    MSE(mel) + MSE(mel_postnet) + BCEWithLogits(gate)
    tacotron2/model.py:648-655          parse_output: --mask-padding (off by default) overwrites the frames past each sample's length
Dropout is EXTERNAL: every dropout site asks `drop(x, p)` for its mask in the reference's call order, so that the reference run
(F.dropout patched to the same stream), this oracle and the HIP path (counter-based masks, replayed site by site) see identical masks.
BatchNorm uses batch statistics (training mode) and does not update running buffers here.  Parameter names are the reference's
state_dict keys.  Pinned by tests/golden/tacotron2_loss.npz and tacotron2_loss_masked.npz (oracle/make_golden.py gen_tacotron2: loss, every parameter gradient
norm, gradient slices; the generator asserts oracle == reference).
"""
import numpy as np
import torch
import torch.nn.functional as TF

TACOTRON2_SMALL = dict(n_mel_channels=80, n_symbols=148, symbols_embedding_dim=64, encoder_kernel_size=5, encoder_n_convolutions=3,
                       encoder_embedding_dim=64, attention_rnn_dim=96, attention_dim=32, attention_location_n_filters=8,
                       attention_location_kernel_size=31, n_frames_per_step=1, decoder_rnn_dim=96, prenet_dim=48,
                       postnet_embedding_dim=64, postnet_kernel_size=5, postnet_n_convolutions=5,
                       p_attention_dropout=0.1, p_decoder_dropout=0.1)
TACOTRON2_DEFAULT = dict(TACOTRON2_SMALL, symbols_embedding_dim=512, encoder_embedding_dim=512, attention_rnn_dim=1024,
                         attention_dim=128, attention_location_n_filters=32, decoder_rnn_dim=1024, prenet_dim=256,
                         postnet_embedding_dim=512)                                   # tacotron2/arg_parser.py:40-107
TACOTRON2_CASE = dict(cfg=TACOTRON2_SMALL, seed=17, text_lengths=[23, 19, 12], mel_lengths=[31, 27, 16])
# the reference's default widths (the network bench.py times) on a batch the oracle's autograd finishes in seconds: 4 utterances,
# 40 text positions (a multiple of 8: the context's memory gradient takes the batched-GEMM path, as at bench size), 60 decoder steps
TACOTRON2_DEFAULT_CASE = dict(cfg=TACOTRON2_DEFAULT, seed=23, text_lengths=[40, 33, 26, 18], mel_lengths=[60, 52, 47, 31])


class MaskStream:
    """Bernoulli keep-masks from one numpy PCG64 stream, in call order; `drop(x, p)` = x * mask / (1 - p) (F.dropout, training)."""

    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.calls = 0

    def __call__(self, x, p):
        self.calls += 1
        if p <= 0:
            return x
        keep = torch.from_numpy((self.rng.random(tuple(x.shape)) >= p).astype(np.float32))
        return x * keep / (1.0 - p)


def _bn(x, p, name, eps=1e-5, training=True):
    """BatchNorm1d on [B, C, T].  Training mode: statistics over (B, T), biased variance (torch.nn.BatchNorm1d.forward); eval mode:
    the running buffers p[name + ".running_mean" / ".running_var"]."""
    if training:
        mean = x.mean(dim=(0, 2), keepdim=True)
        var = x.var(dim=(0, 2), unbiased=False, keepdim=True)
    else:
        mean, var = p[name + ".running_mean"].view(1, -1, 1), p[name + ".running_var"].view(1, -1, 1)
    return (x - mean) / torch.sqrt(var + eps) * p[name + ".weight"].view(1, -1, 1) + p[name + ".bias"].view(1, -1, 1)


def _lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.LSTMCell: gates i, f, g, o in that order."""
    g = x @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
    i, f, gg, o = g.chunk(4, dim=1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    return torch.sigmoid(o) * torch.tanh(c2), c2


def _lstm_dir(x, lengths, p, pre, reverse):
    """One direction of nn.LSTM over a packed batch: x [B, T, C]; steps at t >= length leave the state untouched and output 0
    (pack_padded_sequence / pad_packed_sequence semantics, model.py:205-214)."""
    sfx = "_reverse" if reverse else ""
    w_ih, w_hh = p[pre + "weight_ih_l0" + sfx], p[pre + "weight_hh_l0" + sfx]
    b_ih, b_hh = p[pre + "bias_ih_l0" + sfx], p[pre + "bias_hh_l0" + sfx]
    b, t, _ = x.shape
    hd = w_hh.shape[1]
    h, c = torch.zeros(b, hd), torch.zeros(b, hd)
    outs = [None] * t
    for step in (range(t - 1, -1, -1) if reverse else range(t)):
        h2, c2 = _lstm_cell(x[:, step], h, c, w_ih, w_hh, b_ih, b_hh)
        live = (step < lengths).float().unsqueeze(1)
        h, c = live * h2 + (1 - live) * h, live * c2 + (1 - live) * c
        outs[step] = live * h2
    return torch.stack(outs, dim=1)


def tacotron2_loss(p, cfg, text, text_lengths, mel, gate_target, drop, output_lengths=None, training=True):
    """Tacotron2.forward + Tacotron2Loss.  text int64 [B, T_in] (sorted by length, descending), mel [B, 80, T_out] zero padded,
    gate_target [B, T_out]; p: name -> tensor; drop: MaskStream-like; output_lengths (int64 [B]) given = --mask-padding
    (Tacotron2.parse_output, model.py:648-655: frames past a sample's length are overwritten -- mel outputs 0, gate energies 1e3 --
    before the loss, which cuts their gradient).  training = False: model.eval() -- the validation pass of train.py:273-318 --
    BatchNorm on its running buffers (p[<bn>.running_mean / .running_var]) and only the prenet's dropout drawn (model.py:133 passes
    training=True unconditionally).  -> (loss, (mel_out, mel_post, gate_out, alignments))."""
    nconv, pconv = cfg["encoder_n_convolutions"], cfg["postnet_n_convolutions"]
    pa, pdrop = cfg["p_attention_dropout"], cfg["p_decoder_dropout"]
    x = TF.embedding(text, p["embedding.weight"]).transpose(1, 2)                       # [B, E, T_in]
    for i in range(nconv):
        pre = "encoder.convolutions.%d." % i
        k = p[pre + "0.conv.weight"].shape[2]
        x = TF.conv1d(x, p[pre + "0.conv.weight"], p[pre + "0.conv.bias"], padding=(k - 1) // 2)
        x = torch.relu(_bn(x, p, pre + "1", training=training))
        x = drop(x, 0.5) if training else x
    x = x.transpose(1, 2)
    memory = torch.cat([_lstm_dir(x, text_lengths, p, "encoder.lstm.", False),
                        _lstm_dir(x, text_lengths, p, "encoder.lstm.", True)], dim=2)     # [B, T_in, E]
    b, t_in, _ = memory.shape
    t_out = mel.shape[2]
    # decoder (teacher forcing): go frame + the target frames through the prenet, all steps at once (model.py:473-476)
    dec_in = torch.cat([torch.zeros(1, b, mel.shape[1]), mel.permute(2, 0, 1)], dim=0)   # [T_out + 1, B, 80]
    for i in range(2):
        dec_in = drop(torch.relu(dec_in @ p["decoder.prenet.layers.%d.linear_layer.weight" % i].t()), 0.5)
    pad_mask = torch.arange(t_in)[None, :] >= text_lengths[:, None]
    ah = ac = torch.zeros(b, cfg["attention_rnn_dim"])
    dh = dc = torch.zeros(b, cfg["decoder_rnn_dim"])
    aw = aw_cum = torch.zeros(b, t_in)
    ctx = torch.zeros(b, memory.shape[2])
    att = "decoder.attention_layer."
    processed_memory = memory @ p[att + "memory_layer.linear_layer.weight"].t()
    kloc = p[att + "location_layer.location_conv.conv.weight"].shape[2]
    mel_outs, gate_outs, aligns = [], [], []
    for step in range(t_out):
        ah, ac = _lstm_cell(torch.cat([dec_in[step], ctx], dim=1), ah, ac, p["decoder.attention_rnn.weight_ih"],
                            p["decoder.attention_rnn.weight_hh"], p["decoder.attention_rnn.bias_ih"], p["decoder.attention_rnn.bias_hh"])
        ah = drop(ah, pa) if training else ah
        loc = TF.conv1d(torch.stack([aw, aw_cum], dim=1), p[att + "location_layer.location_conv.conv.weight"], padding=(kloc - 1) // 2)
        loc = loc.transpose(1, 2) @ p[att + "location_layer.location_dense.linear_layer.weight"].t()
        q = (ah @ p[att + "query_layer.linear_layer.weight"].t()).unsqueeze(1)
        e = (torch.tanh(q + loc + processed_memory) @ p[att + "v.linear_layer.weight"].t()).squeeze(2)
        aw = torch.softmax(e.masked_fill(pad_mask, -float("inf")), dim=1)
        ctx = torch.bmm(aw.unsqueeze(1), memory).squeeze(1)
        aw_cum = aw_cum + aw
        dh, dc = _lstm_cell(torch.cat([ah, ctx], dim=1), dh, dc, p["decoder.decoder_rnn.weight_ih"], p["decoder.decoder_rnn.weight_hh"],
                            p["decoder.decoder_rnn.bias_ih"], p["decoder.decoder_rnn.bias_hh"])
        dh = drop(dh, pdrop) if training else dh
        hc = torch.cat([dh, ctx], dim=1)
        mel_outs.append(hc @ p["decoder.linear_projection.linear_layer.weight"].t() + p["decoder.linear_projection.linear_layer.bias"])
        gate_outs.append((hc @ p["decoder.gate_layer.linear_layer.weight"].t() + p["decoder.gate_layer.linear_layer.bias"]).squeeze(1))
        aligns.append(aw)
    mel_out = torch.stack(mel_outs, dim=2)                                              # [B, 80, T_out]
    gate_out = torch.stack(gate_outs, dim=1)
    y = mel_out
    for i in range(pconv):
        pre = "postnet.convolutions.%d." % i
        k = p[pre + "0.conv.weight"].shape[2]
        y = _bn(TF.conv1d(y, p[pre + "0.conv.weight"], p[pre + "0.conv.bias"], padding=(k - 1) // 2), p, pre + "1", training=training)
        y = torch.tanh(y) if i < pconv - 1 else y
        y = drop(y, 0.5) if training else y
    mel_post = mel_out + y
    if output_lengths is not None:
        past = torch.arange(t_out)[None, :] >= output_lengths[:, None]                   # [B, T_out]
        mel_out = mel_out.masked_fill(past[:, None, :], 0.0)
        mel_post = mel_post.masked_fill(past[:, None, :], 0.0)
        gate_out = gate_out.masked_fill(past, 1e3)
    loss = TF.mse_loss(mel_out, mel) + TF.mse_loss(mel_post, mel) + TF.binary_cross_entropy_with_logits(gate_out, gate_target)
    return loss, (mel_out, mel_post, gate_out, torch.stack(aligns, dim=1))


def param_shapes(cfg):
    """name -> shape of the trainable parameters of Tacotron2(**cfg) (model.py:583-620 and the modules it builds)."""
    e, enc, mel = cfg["symbols_embedding_dim"], cfg["encoder_embedding_dim"], cfg["n_mel_channels"] * cfg["n_frames_per_step"]
    sh = {"embedding.weight": (cfg["n_symbols"], e)}
    for i in range(cfg["encoder_n_convolutions"]):
        pre = "encoder.convolutions.%d." % i
        sh[pre + "0.conv.weight"] = (enc, enc, cfg["encoder_kernel_size"])
        sh[pre + "0.conv.bias"] = sh[pre + "1.weight"] = sh[pre + "1.bias"] = (enc,)
    h = enc // 2
    for sfx in ("", "_reverse"):
        sh["encoder.lstm.weight_ih_l0" + sfx], sh["encoder.lstm.weight_hh_l0" + sfx] = (4 * h, enc), (4 * h, h)
        sh["encoder.lstm.bias_ih_l0" + sfx] = sh["encoder.lstm.bias_hh_l0" + sfx] = (4 * h,)
    pn, ar, dr, ad = cfg["prenet_dim"], cfg["attention_rnn_dim"], cfg["decoder_rnn_dim"], cfg["attention_dim"]
    sh["decoder.prenet.layers.0.linear_layer.weight"], sh["decoder.prenet.layers.1.linear_layer.weight"] = (pn, mel), (pn, pn)
    sh["decoder.attention_rnn.weight_ih"], sh["decoder.attention_rnn.weight_hh"] = (4 * ar, pn + enc), (4 * ar, ar)
    sh["decoder.attention_rnn.bias_ih"] = sh["decoder.attention_rnn.bias_hh"] = (4 * ar,)
    att = "decoder.attention_layer."
    sh[att + "query_layer.linear_layer.weight"], sh[att + "memory_layer.linear_layer.weight"] = (ad, ar), (ad, enc)
    sh[att + "v.linear_layer.weight"] = (1, ad)
    sh[att + "location_layer.location_conv.conv.weight"] = (cfg["attention_location_n_filters"], 2, cfg["attention_location_kernel_size"])
    sh[att + "location_layer.location_dense.linear_layer.weight"] = (ad, cfg["attention_location_n_filters"])
    sh["decoder.decoder_rnn.weight_ih"], sh["decoder.decoder_rnn.weight_hh"] = (4 * dr, ar + enc), (4 * dr, dr)
    sh["decoder.decoder_rnn.bias_ih"] = sh["decoder.decoder_rnn.bias_hh"] = (4 * dr,)
    sh["decoder.linear_projection.linear_layer.weight"], sh["decoder.linear_projection.linear_layer.bias"] = (mel, dr + enc), (mel,)
    sh["decoder.gate_layer.linear_layer.weight"], sh["decoder.gate_layer.linear_layer.bias"] = (1, dr + enc), (1,)
    pe, npc = cfg["postnet_embedding_dim"], cfg["postnet_n_convolutions"]
    for i in range(npc):
        pre = "postnet.convolutions.%d." % i
        cin, cout = (mel if i == 0 else pe), (mel if i == npc - 1 else pe)
        sh[pre + "0.conv.weight"] = (cout, cin, cfg["postnet_kernel_size"])
        sh[pre + "0.conv.bias"] = sh[pre + "1.weight"] = sh[pre + "1.bias"] = (cout,)
    return sh


def seeded_running_stats(cfg, seed):
    """BatchNorm running buffers for the eval-mode fixtures: means ~ N(0, 0.3), variances in [0.5, 1.5]."""
    rng = np.random.default_rng(seed + 101)
    out = {}
    for name, shape in sorted(param_shapes(cfg).items()):
        if name.endswith(".1.weight"):
            out[name[:-6] + "running_mean"] = torch.from_numpy((0.3 * rng.standard_normal(shape)).astype(np.float32))
            out[name[:-6] + "running_var"] = torch.from_numpy((0.5 + rng.random(shape)).astype(np.float32))
    return out


def seeded_state(cfg, seed):
    """Every trainable parameter from one numpy PCG64 stream (sorted names): Xavier-sized weights, BatchNorm weights near 1."""
    rng = np.random.default_rng(seed)
    st = {}
    for name, shape in sorted(param_shapes(cfg).items()):
        if name.endswith(".1.weight"):                                        # BatchNorm gamma
            st[name] = (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif name.endswith("bias") or "bias_" in name:
            st[name] = (0.05 * rng.standard_normal(shape)).astype(np.float32)
        elif name == "embedding.weight":
            st[name] = (rng.standard_normal(shape) * 0.3).astype(np.float32)
        else:
            fan_in = int(np.prod(shape[1:]))
            st[name] = (rng.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)
    return {k: torch.from_numpy(v) for k, v in st.items()}


def seeded_batch(case):
    """TextMelCollate's output for synthetic items (tacotron2/data_function.py:100-138): text ids padded with 0, mel zero padded,
    gate target 1 from the last real frame on; sorted by text length, descending."""
    rng = np.random.default_rng(case["seed"] + 1)
    tl, ml = np.asarray(case["text_lengths"]), np.asarray(case["mel_lengths"])
    assert (np.diff(tl) <= 0).all()
    b, nmel = len(tl), case["cfg"]["n_mel_channels"]
    text = np.zeros((b, tl.max()), np.int64)
    mel = np.zeros((b, nmel, ml.max()), np.float32)
    gate = np.zeros((b, ml.max()), np.float32)
    for i in range(b):
        text[i, :tl[i]] = rng.integers(1, case["cfg"]["n_symbols"], tl[i])
        mel[i, :, :ml[i]] = rng.standard_normal((nmel, ml[i])).astype(np.float32) * 1.5 - 4.0
        gate[i, ml[i] - 1:] = 1
    return (torch.from_numpy(text), torch.from_numpy(tl.astype(np.int64)), torch.from_numpy(mel), torch.from_numpy(gate),
            torch.from_numpy(ml.astype(np.int64)))
