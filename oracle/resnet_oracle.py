"""CPU restatement of the reference's ResNet-50 v1.5 train step (TEST INFRASTRUCTURE ONLY).

Follows, in fp32 torch functional ops on the CPU (paths relative to
/root/reference/PyTorch/Classification/ConvNets/image_classification/):
    models/resnet.py:107-175    Bottleneck (stride on the 3x3 = "v1.5"), downsample = conv1x1(stride) + BN
    models/resnet.py:261-322    stem 7x7/2 + BN + ReLU + maxpool 3/2/1, [3,4,6,3] blocks, avgpool, fc
    models/common.py:31-128     Conv2d(bias=False, padding=k//2), BatchNorm2d (eps 1e-5, momentum 0.1)
    smoothing.py:33-40          LabelSmoothing
    optimizers.py:34-56         SGD(momentum, weight_decay) with weight decay skipped for parameters whose
                                NAME contains "bn" (so the downsample BN, named downsample.1, IS decayed)
    training.py:86-96,167-186   loss -> backward -> optimizer.step
Pinned by tests/golden/rn50_step.npz (per-step losses of the reference's own resnet50 module + optimizer,
oracle/make_golden.py gen_rn50).
"""
import numpy as np
import torch
import torch.nn.functional as TF

from oracle.storage import q as q_store

LAYERS, WIDTHS, EXPANSION = [3, 4, 6, 3], [64, 128, 256, 512], 4


def param_shapes(num_classes=1000):
    """Ordered (name, shape) list with the reference's state_dict names (parameters only)."""
    out = [("conv1.weight", (64, 3, 7, 7)), ("bn1.weight", (64,)), ("bn1.bias", (64,))]
    inpl = 64
    for li, (w, n) in enumerate(zip(WIDTHS, LAYERS)):
        for bi in range(n):
            pre = "layers.%d.%d." % (li, bi)
            out += [(pre + "conv1.weight", (w, inpl, 1, 1)), (pre + "bn1.weight", (w,)), (pre + "bn1.bias", (w,)),
                    (pre + "conv2.weight", (w, w, 3, 3)), (pre + "bn2.weight", (w,)), (pre + "bn2.bias", (w,)),
                    (pre + "conv3.weight", (w * EXPANSION, w, 1, 1)), (pre + "bn3.weight", (w * EXPANSION,)),
                    (pre + "bn3.bias", (w * EXPANSION,))]
            if bi == 0:
                out += [(pre + "downsample.0.weight", (w * EXPANSION, inpl, 1, 1)),
                        (pre + "downsample.1.weight", (w * EXPANSION,)), (pre + "downsample.1.bias", (w * EXPANSION,))]
            inpl = w * EXPANSION
    out += [("fc.weight", (num_classes, 2048)), ("fc.bias", (num_classes,))]
    return out


def seeded_state(seed, num_classes=1000):
    """Deterministic init from a numpy PCG64 stream: kaiming-normal(fan_in, relu) convs, BN gamma ~ U(0.5,1.5)
    and small beta (non-trivial affine so that parity tests see them), small fc."""
    rng = np.random.default_rng(seed)
    st = {}
    for name, shape in param_shapes(num_classes):
        if name.endswith("conv1.weight") or "conv" in name.split(".")[-2] or name.endswith("downsample.0.weight"):
            fan_in = shape[1] * shape[2] * shape[3]
            st[name] = (rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        elif name == "fc.weight":
            st[name] = (rng.standard_normal(shape) * 0.02).astype(np.float32)
        elif name == "fc.bias":
            st[name] = (rng.standard_normal(shape) * 0.01).astype(np.float32)
        elif name.endswith("bn3.weight"):
            # damped residual branches (the reference's own knob is --last_bn_0_init, resnet.py:143): with
            # gamma3 ~ 1 a random-init ResNet-50 amplifies 16-bit rounding noise ~20x (6e-2 of the max logit in
            # fp16) and per-step losses become chaotic; at 0.1..0.3 fp32 and 16-bit storage agree to 3e-3.
            st[name] = rng.uniform(0.1, 0.3, shape).astype(np.float32)
        elif name.endswith(".weight"):
            st[name] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        else:
            st[name] = (rng.standard_normal(shape) * 0.1).astype(np.float32)
    return {k: torch.from_numpy(v) for k, v in st.items()}


def seeded_batch(seed, batch, size, num_classes=1000):
    """SynteticDataLoader (dataloaders.py:531-542): randn images, randint labels."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((batch, 3, size, size)).astype(np.float32)
    y = rng.integers(0, num_classes, batch).astype(np.int64)
    return torch.from_numpy(x), torch.from_numpy(y)


class ResNet50Oracle:
    """storage_dtype=None: the reference's fp32 CPU path.  storage_dtype=torch.float16/bfloat16: the same math
    with activations, conv/fc weights and activation gradients ROUNDED to that type wherever the AMP path keeps a
    tensor in 16 bits (fp32 accumulation, fp32 BN statistics, fp32 master weights) -- used to separate the
    precision floor of 16-bit storage from kernel errors on this ill-conditioned random-init network."""

    def __init__(self, state, lr, momentum=0.875, weight_decay=3.0517578125e-05, smoothing=0.1, eps=1e-5,
                 bn_momentum=0.1, storage_dtype=None):
        self.sd = storage_dtype
        self.p = {k: v.clone().float().requires_grad_(True) for k, v in state.items()}
        self.buf = {k: None for k in self.p}
        self.run = {}
        self.lr, self.mom, self.wd, self.smoothing, self.eps, self.bnm = lr, momentum, weight_decay, smoothing, eps, bn_momentum

    def _q(self, t):
        return q_store(t, self.sd)

    def _bn(self, x, name):
        rm = self.run.setdefault(name + ".running_mean", torch.zeros(x.shape[1]))
        rv = self.run.setdefault(name + ".running_var", torch.ones(x.shape[1]))
        return TF.batch_norm(x, rm, rv, self.p[name + ".weight"], self.p[name + ".bias"], training=True,
                             momentum=self.bnm, eps=self.eps)

    def _conv(self, x, name, **kw):
        return self._q(TF.conv2d(x, self._q(self.p[name]), **kw))

    def forward(self, x):
        if self.sd is not None:
            return self._forward_16(x)
        p = self.p
        x = TF.conv2d(x, p["conv1.weight"], stride=2, padding=3)
        x = torch.relu(self._bn(x, "bn1"))
        x = TF.max_pool2d(x, 3, 2, 1)
        for li, n in enumerate(LAYERS):
            for bi in range(n):
                pre = "layers.%d.%d." % (li, bi)
                stride = 2 if (bi == 0 and li > 0) else 1
                res = x
                o = torch.relu(self._bn(TF.conv2d(x, p[pre + "conv1.weight"]), pre + "bn1"))
                o = torch.relu(self._bn(TF.conv2d(o, p[pre + "conv2.weight"], stride=stride, padding=1), pre + "bn2"))
                o = self._bn(TF.conv2d(o, p[pre + "conv3.weight"]), pre + "bn3")
                if bi == 0:
                    res = self._bn(TF.conv2d(x, p[pre + "downsample.0.weight"], stride=stride), pre + "downsample.1")
                x = torch.relu(o + res)
        x = TF.adaptive_avg_pool2d(x, 1).flatten(1)
        return TF.linear(x, p["fc.weight"], p["fc.bias"])

    def _forward_16(self, x):
        """Same graph; every tensor the AMP path stores in 16 bits is rounded where it is produced."""
        q = self._q
        x = q(x)
        x = q(torch.relu(self._bn(self._conv(x, "conv1.weight", stride=2, padding=3), "bn1")))
        x = TF.max_pool2d(x, 3, 2, 1)
        for li, n in enumerate(LAYERS):
            for bi in range(n):
                pre = "layers.%d.%d." % (li, bi)
                stride = 2 if (bi == 0 and li > 0) else 1
                res = x
                o = q(torch.relu(self._bn(self._conv(x, pre + "conv1.weight"), pre + "bn1")))
                o = q(torch.relu(self._bn(self._conv(o, pre + "conv2.weight", stride=stride, padding=1), pre + "bn2")))
                o = self._bn(self._conv(o, pre + "conv3.weight"), pre + "bn3")
                if bi == 0:
                    res = q(self._bn(self._conv(x, pre + "downsample.0.weight", stride=stride), pre + "downsample.1"))
                x = q(torch.relu(o + res))
        x = q(TF.adaptive_avg_pool2d(x, 1).flatten(1))
        return TF.linear(x, q(self.p["fc.weight"]), self.p["fc.bias"])

    def loss(self, logits, target):
        lp = torch.log_softmax(logits, dim=-1)
        nll = -lp.gather(-1, target.unsqueeze(1)).squeeze(1)
        return ((1.0 - self.smoothing) * nll + self.smoothing * (-lp.mean(-1))).mean()

    def step(self, x, target, lr=None):
        lr = self.lr if lr is None else lr
        for v in self.p.values():
            v.grad = None
        loss = self.loss(self.forward(x), target)
        loss.backward()
        with torch.no_grad():
            for k, v in self.p.items():
                d = v.grad
                wd = 0.0 if "bn" in k else self.wd
                if wd:
                    d = d + wd * v
                if self.mom:
                    self.buf[k] = d.clone() if self.buf[k] is None else self.buf[k] * self.mom + d
                    d = self.buf[k]
                v -= lr * d
        return float(loss.detach())


# make_golden.gen_rn50 also stores `sensitivity`: the relative loss change under a 1e-6 input perturbation in
# fp32 (a chaos check of the configuration itself; ~1e-7 with the damped init above).
RN50_STEP_CONFIG = dict(seed=5, batch=32, size=64, lr=1e-3, steps=4, num_classes=1000)

# BASELINE.json configs[0]: the reference's own CPU-runnable case -- synthetic 224x224, batch 32 (2 steps are pinned;
# the full 100 iterations are ~1 h of host time and add nothing to parity).  Same seeded weights as above.
RN50_STEP_CONFIG_224 = dict(seed=5, batch=32, size=224, lr=1e-3, steps=2, num_classes=1000)
