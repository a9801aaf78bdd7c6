"""WaveGlow parameters with the reference's module tree and state_dict keys.

Mirrors SpeechSynthesis/Tacotron2/waveglow/model.py: `WaveGlow` (:160-186), `WN` (:95-136), `Invertible1x1Conv` (:51-63) --
the same names (`upsample.weight`, `convinv.3.conv.weight`, `WN.3.in_layers.2.weight_g`, `WN.3.end.bias` ...), shapes and
initial distributions, so a checkpoint of one loads into the other.  This class only HOLDS parameters; the arithmetic is the
kernel sequence of waveglow/engine.py.

Storage: every fp32 master parameter is a view into ONE flat buffer (`flat`), laid out so that the engine's GEMM-shaped reads
are contiguous (the cond-layer biases of all flows follow each other, so do the in-layer biases; `end.weight` / `end.bias` own 8-row / 8-float slots whose
tail stays zero, because the `end` GEMM runs 8 output channels wide).  Gradients and the Adam moments use the same offsets
in their own flat buffers, so the data-parallel all-reduce, the norm and the optimizer see one tensor each.
"""
import math

import torch
from torch import nn

DEFAULT_CONFIG = dict(n_mel_channels=80, n_flows=12, n_group=8, n_early_every=4, n_early_size=2,
                      WN_config=dict(n_layers=8, n_channels=512, kernel_size=3))     # waveglow/arg_parser.py:38-64
UPSAMPLE_KERNEL, UPSAMPLE_STRIDE = 1024, 256                                         # model.py:165-167


def flow_channels(cfg):
    """[(n_remaining_channels, n_half)] per flow (model.py:178-186)."""
    n_half, n_rem, out = cfg["n_group"] // 2, cfg["n_group"], []
    for k in range(cfg["n_flows"]):
        if k % cfg["n_early_every"] == 0 and k > 0:
            n_half -= cfg["n_early_size"] // 2
            n_rem -= cfg["n_early_size"]
        out.append((n_rem, n_half))
    return out


def param_layout(cfg):
    """[(name, shape, slot_elems)] in flat-buffer order."""
    wn, mel, ng = cfg["WN_config"], cfg["n_mel_channels"], cfg["n_group"]
    nc, ks, nl = wn["n_channels"], wn["kernel_size"], wn["n_layers"]
    if ng != 8:
        raise ValueError("the flow-state kernels are built for n_group = 8 (the reference's value)")
    out = []

    def add(name, shape, slot=None):
        n = int(math.prod(shape))
        slot = n if slot is None else slot
        out.append((name, tuple(shape), (slot + 7) // 8 * 8))         # 32-byte slots: 16-byte aligned views, zero tails

    add("upsample.weight", (mel, mel, UPSAMPLE_KERNEL))
    add("upsample.bias", (mel,))
    for k in range(cfg["n_flows"]):                                    # contiguous: ONE bias vector for the all-flows cond GEMM
        for i in range(nl):
            add("WN.%d.cond_layers.%d.bias" % (k, i), (2 * nc,))
    for k in range(cfg["n_flows"]):                                    # contiguous, same order: their gradients are one copy of
        for i in range(nl):                                            # the cond-bias gradients (s = in_layer + cond_layer)
            add("WN.%d.in_layers.%d.bias" % (k, i), (2 * nc,))
    for k, (n_rem, n_half) in enumerate(flow_channels(cfg)):
        add("convinv.%d.conv.weight" % k, (n_rem, n_rem, 1))
        pre = "WN.%d." % k
        add(pre + "start.bias", (nc,))
        add(pre + "start.weight_g", (nc, 1, 1))
        add(pre + "start.weight_v", (nc, n_half, 1))
        add(pre + "end.weight", (2 * n_half, nc, 1), slot=8 * nc)
        add(pre + "end.bias", (2 * n_half,), slot=8)
        for i in range(nl):
            add(pre + "in_layers.%d.weight_g" % i, (2 * nc, 1, 1))
            add(pre + "in_layers.%d.weight_v" % i, (2 * nc, nc, ks))
        for i in range(nl):
            add(pre + "cond_layers.%d.weight_g" % i, (2 * nc, 1, 1))
            add(pre + "cond_layers.%d.weight_v" % i, (2 * nc, mel * ng, 1))
        for i in range(nl):
            rs = 2 * nc if i < nl - 1 else nc
            add(pre + "res_skip_layers.%d.bias" % i, (rs,))
            add(pre + "res_skip_layers.%d.weight_g" % i, (rs, 1, 1))
            add(pre + "res_skip_layers.%d.weight_v" % i, (rs, nc, 1))
    return out


class _Node(nn.Module):
    """Plain container: gives parameters the dotted names of the reference's module tree."""


def _child(root, path):
    node = root
    for part in path:
        if not hasattr(node, part):
            node.add_module(part, _Node())
        node = getattr(node, part)
    return node


class FlatViews:
    """name -> view of one flat fp32 buffer, same offsets as the parameters."""

    def __init__(self, layout, device, flat=None):
        self.total = sum(s for _, _, s in layout)
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=device) if flat is None else flat
        self.views, self.offsets = {}, {}
        off = 0
        for name, shape, slot in layout:
            self.views[name] = self.flat[off:off + int(math.prod(shape))].view(shape)
            self.offsets[name] = (off, slot)
            off += slot

    def __getitem__(self, name):
        return self.views[name]

    def slot(self, name):
        off, n = self.offsets[name]
        return self.flat[off:off + n]


class WaveGlow(nn.Module):
    """Parameter holder with the reference's state_dict (see the module docstring)."""

    def __init__(self, n_mel_channels=80, n_flows=12, n_group=8, n_early_every=4, n_early_size=2, WN_config=None,
                 device="cpu"):
        super().__init__()
        self.cfg = dict(n_mel_channels=n_mel_channels, n_flows=n_flows, n_group=n_group, n_early_every=n_early_every,
                        n_early_size=n_early_size, WN_config=dict(WN_config or DEFAULT_CONFIG["WN_config"]))
        self.layout = param_layout(self.cfg)
        self.store = FlatViews(self.layout, device)
        for name, _, _ in self.layout:
            parts = name.split(".")
            _child(self, parts[:-1]).register_parameter(parts[-1], nn.Parameter(self.store[name]))
        self.reset_parameters()

    @torch.no_grad()
    def reset_parameters(self):
        """The reference's initial distributions: Conv1d / ConvTranspose1d defaults (kaiming_uniform(a = sqrt 5) = U(+-1/sqrt
        fan_in) for weight and bias), weight_norm's g = ||v||, `end` zero (model.py:110-115), convinv = a random rotation with
        det +1 (model.py:56-63)."""
        for name, p in self.named_parameters():
            if name.startswith("convinv."):
                c = p.shape[0]
                q = torch.linalg.qr(torch.empty(c, c).normal_())[0]
                if torch.det(q) < 0:
                    q[:, 0] = -q[:, 0]
                p.copy_(q.view(c, c, 1))
            elif ".end." in name:
                p.zero_()
            elif name.endswith("weight_g"):
                continue
            elif name.endswith(("weight_v", "weight")):
                fan_in = p.shape[1] * p.shape[2]             # torch's fan_in = size(1) * receptive field, also for ConvTranspose1d
                p.uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))
            elif name.endswith("bias"):
                w = name[:-4] + ("weight" if name.startswith("upsample") else "weight_v")
                wt = dict(self.named_parameters())[w]
                fan_in = wt.shape[1] * wt.shape[2]
                p.uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))
        params = dict(self.named_parameters())
        for name, p in params.items():
            if name.endswith("weight_g"):
                v = params[name[:-1] + "v"]
                p.copy_(v.flatten(1).norm(dim=1).view_as(p))

    def load_reference_state(self, state):
        """Copy a reference-named state dict (tensors or arrays) into the flat storage."""
        own = dict(self.named_parameters())
        missing = sorted(set(own) - set(state))
        extra = sorted(set(state) - set(own))
        if missing or extra:
            raise KeyError("state_dict mismatch: missing %s, unexpected %s" % (missing[:4], extra[:4]))
        with torch.no_grad():
            for k, p in own.items():
                p.copy_(torch.as_tensor(state[k]).to(p.device, torch.float32).reshape(p.shape))
