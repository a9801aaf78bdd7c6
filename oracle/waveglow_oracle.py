"""CPU restatement of the WaveGlow training loss (TEST INFRASTRUCTURE ONLY; SURVEY.md section 8 row f1 -- the checker of the HIP
path in deeplearningexamples_amd/waveglow; nothing in the product imports it).

Follows, in plain fp32 torch on the CPU (paths relative to /root/reference/PyTorch/SpeechSynthesis/Tacotron2/):
    waveglow/model.py:34-41     fused_add_tanh_sigmoid_multiply
    waveglow/model.py:44-85     Invertible1x1Conv: z = W a (per time step), log det W x batch x groups
    waveglow/model.py:87-157    WN: weight-normalised start / dilated in_layers / cond_layers / res_skip 1x1, plain `end`
    waveglow/model.py:160-231   WaveGlow.forward: ConvTranspose1d upsampling of the mel, grouping by n_group, the flows with
                                early outputs, affine coupling audio_1 = exp(log_s) audio_1 + b
    waveglow/loss_function.py:30-48  sum z^2 / (2 sigma^2) - sum log_s - sum log det W, / (B x n_group x T / n_group)
Weight normalisation is written out (w = g * v / ||v||, norm over all but the first axis: torch.nn.utils.weight_norm, dim 0),
parameters keep the reference's state_dict names (`WN.3.in_layers.2.weight_g`, `convinv.1.conv.weight`, `upsample.weight` ...).
Pinned by tests/golden/waveglow_loss.npz, produced by oracle/make_golden.py gen_waveglow from the reference's own WaveGlow +
WaveGlowLoss (loss and every parameter gradient; the generator asserts oracle == reference).
"""
import numpy as np
import torch
import torch.nn.functional as TF

WAVEGLOW_SMALL = dict(n_mel_channels=80, n_flows=4, n_group=8, n_early_every=2, n_early_size=2,
                      WN_config=dict(n_layers=3, n_channels=64, kernel_size=3))
WAVEGLOW_CASE = dict(cfg=WAVEGLOW_SMALL, seed=7, batch=2, segment=2048, sigma=1.0)     # 8 mel frames of hop 256 per segment


def _wn(p, name):
    """torch.nn.utils.weight_norm(name='weight', dim=0): w = g * v / ||v||_{all axes but 0}."""
    v, g = p[name + ".weight_v"], p[name + ".weight_g"]
    return v * (g / v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1))))


def wn_forward(p, pre, audio, spect, n_layers, n_channels, kernel_size):
    """WN.forward (model.py:138-157)."""
    audio = TF.conv1d(audio, _wn(p, pre + "start"), p[pre + "start.bias"])
    output = 0
    for i in range(n_layers):
        d = 2 ** i
        pad = (kernel_size * d - d) // 2
        a = TF.conv1d(audio, _wn(p, pre + "in_layers.%d" % i), p[pre + "in_layers.%d.bias" % i], dilation=d, padding=pad)
        c = TF.conv1d(spect, _wn(p, pre + "cond_layers.%d" % i), p[pre + "cond_layers.%d.bias" % i])
        s = a + c
        acts = torch.tanh(s[:, :n_channels]) * torch.sigmoid(s[:, n_channels:])
        rs = TF.conv1d(acts, _wn(p, pre + "res_skip_layers.%d" % i), p[pre + "res_skip_layers.%d.bias" % i])
        if i < n_layers - 1:
            audio = rs[:, :n_channels] + audio
            skip = rs[:, n_channels:]
        else:
            skip = rs
        output = output + skip
    return TF.conv1d(output, p[pre + "end.weight"], p[pre + "end.bias"])


def waveglow_loss(p, cfg, mel, audio, sigma=1.0):
    """WaveGlow.forward + WaveGlowLoss.forward.  mel [B, 80, frames], audio [B, T]; p: name -> tensor."""
    ng, wn = cfg["n_group"], cfg["WN_config"]
    spect = TF.conv_transpose1d(mel, p["upsample.weight"], p["upsample.bias"], stride=256)
    assert spect.size(2) >= audio.size(1)
    spect = spect[:, :, :audio.size(1)]
    spect = spect.unfold(2, ng, ng).permute(0, 2, 1, 3)
    spect = spect.contiguous().view(spect.size(0), spect.size(1), -1).permute(0, 2, 1)
    a = audio.unfold(1, ng, ng).permute(0, 2, 1)
    outs, log_s_total, log_det_total = [], 0.0, 0.0
    for k in range(cfg["n_flows"]):
        if k % cfg["n_early_every"] == 0 and k > 0:
            outs.append(a[:, :cfg["n_early_size"]])
            a = a[:, cfg["n_early_size"]:]
        w = p["convinv.%d.conv.weight" % k].squeeze(-1)
        log_det_total = log_det_total + a.size(0) * a.size(2) * torch.logdet(w.float())
        a = TF.conv1d(a, w.unsqueeze(-1))
        nh = a.size(1) // 2
        a0, a1 = a[:, :nh], a[:, nh:]
        o = wn_forward(p, "WN.%d." % k, a0, spect, wn["n_layers"], wn["n_channels"], wn["kernel_size"])
        log_s, b = o[:, nh:], o[:, :nh]
        a1 = torch.exp(log_s) * a1 + b
        log_s_total = log_s_total + log_s.sum()
        a = torch.cat([a0, a1], 1)
    outs.append(a)
    z = torch.cat(outs, 1)
    loss = (z * z).sum() / (2 * sigma * sigma) - log_s_total - log_det_total
    return loss / (z.size(0) * z.size(1) * z.size(2))


def seeded_inputs(case):
    """Synthetic LJSpeech-shaped pair: a mel of segment / 256 frames and an audio segment in [-1, 1]."""
    rng = np.random.default_rng(case["seed"] + 1)
    frames = case["segment"] // 256
    mel = rng.standard_normal((case["batch"], case["cfg"]["n_mel_channels"], frames)).astype(np.float32) * 2.0 - 5.0
    audio = (rng.standard_normal((case["batch"], case["segment"])) * 0.2).clip(-1, 1).astype(np.float32)
    return torch.from_numpy(mel), torch.from_numpy(audio)


def param_shapes(cfg):
    """name -> shape of WaveGlow(**cfg).state_dict() (model.py:50-63, 95-136, 165-186)."""
    wn, ng, mel = cfg["WN_config"], cfg["n_group"], cfg["n_mel_channels"]
    nc, ks = wn["n_channels"], wn["kernel_size"]
    sh = {"upsample.weight": (mel, mel, 1024), "upsample.bias": (mel,)}
    n_half, n_rem = ng // 2, ng
    for k in range(cfg["n_flows"]):
        if k % cfg["n_early_every"] == 0 and k > 0:
            n_half -= cfg["n_early_size"] // 2
            n_rem -= cfg["n_early_size"]
        sh["convinv.%d.conv.weight" % k] = (n_rem, n_rem, 1)
        pre = "WN.%d." % k

        def conv(name, cout, cin, kk, normed=True):
            sh[pre + name + ".bias"] = (cout,)
            if normed:
                sh[pre + name + ".weight_g"] = (cout, 1, 1)
                sh[pre + name + ".weight_v"] = (cout, cin, kk)
            else:
                sh[pre + name + ".weight"] = (cout, cin, kk)
        conv("start", nc, n_half, 1)
        conv("end", 2 * n_half, nc, 1, normed=False)
        for i in range(wn["n_layers"]):
            conv("in_layers.%d" % i, 2 * nc, nc, ks)
            conv("cond_layers.%d" % i, 2 * nc, mel * ng, 1)
            conv("res_skip_layers.%d" % i, 2 * nc if i < wn["n_layers"] - 1 else nc, nc, 1)
    return sh


def seeded_state(cfg, seed):
    """Every parameter from one numpy PCG64 stream (sorted names; stable across machines): small normal weights, weight_g near 1,
    invertible 1x1 matrices = a random rotation plus noise (well-conditioned, determinant > 0).  The reference's own
    initialisation has `end` = 0, which would leave log_s, b and most gradients identically zero."""
    rng = np.random.default_rng(seed)
    st = {}
    for name, shape in sorted(param_shapes(cfg).items()):
        if name.startswith("convinv."):
            q, _ = np.linalg.qr(rng.standard_normal((shape[0], shape[0])))
            if np.linalg.det(q) < 0:
                q[:, 0] = -q[:, 0]
            st[name] = (q + 0.05 * rng.standard_normal(q.shape)).astype(np.float32).reshape(shape)
        elif name.endswith("weight_g"):
            st[name] = (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif name.endswith("bias"):
            st[name] = (0.05 * rng.standard_normal(shape)).astype(np.float32)
        else:
            fan_in = int(np.prod(shape[1:]))
            st[name] = (rng.standard_normal(shape) * (0.5 / np.sqrt(fan_in))).astype(np.float32)
    return {k: torch.from_numpy(v) for k, v in st.items()}
