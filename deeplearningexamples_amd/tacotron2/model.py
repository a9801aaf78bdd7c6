"""Tacotron2 parameters with the reference's module tree and state_dict keys (Tacotron2 half).

Mirrors SpeechSynthesis/Tacotron2/tacotron2/model.py `Tacotron2` (:583-620) and the modules it builds (Encoder :177-216, Decoder
:255-300, Attention / LocationLayer :40-77, Prenet :124-130, Postnet :138-174): the same names, shapes, initial distributions
(xavier_uniform with the reference's gains, torch defaults for LSTM / BatchNorm / biases) and BatchNorm buffers, so a checkpoint
of one loads into the other.  This class only HOLDS parameters (views of ONE flat fp32 buffer, like waveglow/model.py); the
arithmetic is the kernel sequence of tacotron2/engine.py.
"""
import math

import torch
from torch import nn

from ..waveglow.model import FlatViews, _child

DEFAULT_CONFIG = dict(n_mel_channels=80, n_symbols=148, symbols_embedding_dim=512, encoder_kernel_size=5, encoder_n_convolutions=3,
                      encoder_embedding_dim=512, attention_rnn_dim=1024, attention_dim=128, attention_location_n_filters=32,
                      attention_location_kernel_size=31, n_frames_per_step=1, decoder_rnn_dim=1024, prenet_dim=256,
                      postnet_embedding_dim=512, postnet_kernel_size=5, postnet_n_convolutions=5,
                      p_attention_dropout=0.1, p_decoder_dropout=0.1)                  # tacotron2/arg_parser.py:40-107


def param_shapes(cfg):
    """[(name, shape)] in the reference's named_parameters() order (embedding, encoder, decoder, postnet)."""
    e, enc, mel = cfg["symbols_embedding_dim"], cfg["encoder_embedding_dim"], cfg["n_mel_channels"] * cfg["n_frames_per_step"]
    if e != enc or cfg["n_frames_per_step"] != 1:
        raise ValueError("symbols_embedding_dim must equal encoder_embedding_dim; n_frames_per_step = 1 (the reference's defaults)")
    out = [("embedding.weight", (cfg["n_symbols"], e))]
    for i in range(cfg["encoder_n_convolutions"]):
        pre = "encoder.convolutions.%d." % i
        out += [(pre + "0.conv.weight", (enc, enc, cfg["encoder_kernel_size"])), (pre + "0.conv.bias", (enc,)),
                (pre + "1.weight", (enc,)), (pre + "1.bias", (enc,))]
    h = enc // 2
    for sfx in ("", "_reverse"):
        out += [("encoder.lstm.weight_ih_l0" + sfx, (4 * h, enc)), ("encoder.lstm.weight_hh_l0" + sfx, (4 * h, h)),
                ("encoder.lstm.bias_ih_l0" + sfx, (4 * h,)), ("encoder.lstm.bias_hh_l0" + sfx, (4 * h,))]
    pn, ar, dr, ad = cfg["prenet_dim"], cfg["attention_rnn_dim"], cfg["decoder_rnn_dim"], cfg["attention_dim"]
    att = "decoder.attention_layer."
    out += [("decoder.prenet.layers.0.linear_layer.weight", (pn, mel)), ("decoder.prenet.layers.1.linear_layer.weight", (pn, pn)),
            ("decoder.attention_rnn.weight_ih", (4 * ar, pn + enc)), ("decoder.attention_rnn.weight_hh", (4 * ar, ar)),
            ("decoder.attention_rnn.bias_ih", (4 * ar,)), ("decoder.attention_rnn.bias_hh", (4 * ar,)),
            (att + "query_layer.linear_layer.weight", (ad, ar)), (att + "memory_layer.linear_layer.weight", (ad, enc)),
            (att + "v.linear_layer.weight", (1, ad)),
            (att + "location_layer.location_conv.conv.weight", (cfg["attention_location_n_filters"], 2, cfg["attention_location_kernel_size"])),
            (att + "location_layer.location_dense.linear_layer.weight", (ad, cfg["attention_location_n_filters"])),
            ("decoder.decoder_rnn.weight_ih", (4 * dr, ar + enc)), ("decoder.decoder_rnn.weight_hh", (4 * dr, dr)),
            ("decoder.decoder_rnn.bias_ih", (4 * dr,)), ("decoder.decoder_rnn.bias_hh", (4 * dr,)),
            ("decoder.linear_projection.linear_layer.weight", (mel, dr + enc)), ("decoder.linear_projection.linear_layer.bias", (mel,)),
            ("decoder.gate_layer.linear_layer.weight", (1, dr + enc)), ("decoder.gate_layer.linear_layer.bias", (1,))]
    pe, npc = cfg["postnet_embedding_dim"], cfg["postnet_n_convolutions"]
    for i in range(npc):
        pre = "postnet.convolutions.%d." % i
        cin, cout = (mel if i == 0 else pe), (mel if i == npc - 1 else pe)
        out += [(pre + "0.conv.weight", (cout, cin, cfg["postnet_kernel_size"])), (pre + "0.conv.bias", (cout,)),
                (pre + "1.weight", (cout,)), (pre + "1.bias", (cout,))]
    return out


def bn_names(cfg):
    return (["encoder.convolutions.%d.1" % i for i in range(cfg["encoder_n_convolutions"])] +
            ["postnet.convolutions.%d.1" % i for i in range(cfg["postnet_n_convolutions"])])


class Tacotron2(nn.Module):
    def __init__(self, device="cpu", uniform_initialize_bn_weight=False, **cfg):
        """cfg: the reference's model config keys (unknown / inference-only ones are ignored).  uniform_initialize_bn_weight:
        models.get_model's init_bn (models.py:53-62, the training entry point's default): BatchNorm weights ~ U[0, 1)."""
        super().__init__()
        self.cfg = dict(DEFAULT_CONFIG)
        self.cfg.update({k: v for k, v in cfg.items() if k in DEFAULT_CONFIG})
        shapes = param_shapes(self.cfg)
        self.layout = [(n, s, (int(math.prod(s)) + 7) // 8 * 8) for n, s in shapes]
        self.store = FlatViews(self.layout, device)
        for name, _, _ in self.layout:
            parts = name.split(".")
            _child(self, parts[:-1]).register_parameter(parts[-1], nn.Parameter(self.store[name]))
        for bn in bn_names(self.cfg):
            node = _child(self, bn.split("."))
            c = self.store[bn + ".weight"].numel()
            node.register_buffer("running_mean", torch.zeros(c, device=device))
            node.register_buffer("running_var", torch.ones(c, device=device))
            node.register_buffer("num_batches_tracked", torch.zeros((), dtype=torch.long, device=device))
        self.reset_parameters()
        if uniform_initialize_bn_weight:
            with torch.no_grad():
                for bn in bn_names(self.cfg):
                    self.store[bn + ".weight"].uniform_()

    @torch.no_grad()
    def reset_parameters(self):
        """ConvNorm / LinearNorm: xavier_uniform with gain('relu' | 'tanh' | 'sigmoid' | 'linear') (tacotron2_common/layers.py:
        35-66), biases torch's default U(+-1/sqrt(fan_in)); nn.LSTM / LSTMCell: U(+-1/sqrt(hidden)); BatchNorm 1 / 0; the
        embedding U(+-sqrt(3) sqrt(2 / (n_symbols + dim))) (model.py:594-596)."""
        gain = {"relu": math.sqrt(2.0), "tanh": 5.0 / 3, "sigmoid": 1.0, "linear": 1.0}
        npc = self.cfg["postnet_n_convolutions"]
        params = dict(self.named_parameters())
        for name, p in params.items():
            if name == "embedding.weight":
                val = math.sqrt(3.0) * math.sqrt(2.0 / sum(p.shape))
                p.uniform_(-val, val)
            elif ".lstm." in name or "_rnn." in name:
                hid = params[name.rsplit(".", 1)[0] + (".weight_hh_l0" if ".lstm." in name else ".weight_hh")].shape[1]
                p.uniform_(-1.0 / math.sqrt(hid), 1.0 / math.sqrt(hid))
            elif name.endswith(".1.weight"):
                p.fill_(1.0)
            elif name.endswith(".1.bias"):
                p.zero_()
            elif name.endswith("weight"):
                if "encoder.convolutions" in name:
                    g = gain["relu"]
                elif "postnet.convolutions" in name:
                    g = gain["linear"] if name.startswith("postnet.convolutions.%d." % (npc - 1)) else gain["tanh"]
                elif "query_layer" in name or "memory_layer" in name or "location_dense" in name:
                    g = gain["tanh"]
                elif "gate_layer" in name:
                    g = gain["sigmoid"]
                else:
                    g = gain["linear"]
                rf = int(math.prod(p.shape[2:])) if p.dim() > 2 else 1
                bound = g * math.sqrt(6.0 / ((p.shape[0] + p.shape[1]) * rf))
                p.uniform_(-bound, bound)
            else:                                                     # Conv1d / Linear bias: U(+-1/sqrt(fan_in))
                w = params[name[:-4] + "weight"]
                fan_in = int(math.prod(w.shape[1:]))
                p.uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))

    def load_reference_state(self, state):
        own = dict(self.named_parameters())
        own.update(dict(self.named_buffers()))
        missing = sorted(set(dict(self.named_parameters())) - set(state))
        if missing:
            raise KeyError("state_dict is missing %s" % missing[:4])
        with torch.no_grad():
            for k, v in state.items():
                if k not in own:
                    raise KeyError("unexpected key %s" % k)
                own[k].copy_(torch.as_tensor(v).to(own[k].device, own[k].dtype).reshape(own[k].shape))
