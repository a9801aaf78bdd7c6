"""CPU restatement of the reference's BERT pre-training step (TEST INFRASTRUCTURE ONLY).

Follows, in fp32 torch functional ops on the CPU (paths relative to /root/reference/PyTorch/LanguageModeling/BERT/):
    modeling.py:285-301    BertEmbeddings (word + position + token-type, LayerNorm eps 1e-12, dropout)
    modeling.py:340-384    BertSelfAttention (QK^T / sqrt(d) + (1-mask)*-10000, softmax, PV)
    modeling.py:394-434    BertSelfOutput / BertIntermediate (tanh-GELU) / BertOutput
    modeling.py:518-524    BertPooler (token 0, dense + tanh)
    modeling.py:545-595    MLM head on the masked rows only (dense+GELU+LayerNorm, tied decoder + bias), NSP head
    run_pretraining.py:75-95   BertPretrainingCriterion = CE(ignore_index=-1) MLM + CE NSP
    lamb_amp_opt/fused_lamb/fused_lamb.py:131-258 + csrc/multi_tensor_lamb.cu  LAMB step (oracle/lamb_oracle.py)
    schedulers.py:123-136  PolyWarmUpScheduler
Dropout: the golden fixture runs with probabilities 0 (the reference's CUDA Philox stream cannot be reproduced on the
CPU).  The dropout sites themselves (modeling.py:296 embeddings, :369 attention probabilities,
:396 / :432 before the residual LayerNorms) are restated here with EXTERNALLY supplied keep masks, so the HIP step in
training mode is checked against this oracle under the masks the HIP RNG produced.
Pinned by tests/golden/bert_step.npz: per-step losses of the reference's own BertForPreTraining module with the
oracle's LAMB (the reference has no CPU LAMB), oracle/make_golden.py gen_bert.
"""
import math

import numpy as np
import torch
import torch.nn.functional as TF

from oracle.storage import q as q_store

from . import lamb_oracle as L


def param_shapes(cfg):
    h, i, v, p, t = cfg["hidden"], cfg["intermediate"], cfg["vocab"], cfg["max_pos"], cfg["type_vocab"]
    out = [("bert.embeddings.word_embeddings.weight", (v, h)), ("bert.embeddings.position_embeddings.weight", (p, h)),
           ("bert.embeddings.token_type_embeddings.weight", (t, h)),
           ("bert.embeddings.LayerNorm.weight", (h,)), ("bert.embeddings.LayerNorm.bias", (h,))]
    for l in range(cfg["layers"]):
        pre = "bert.encoder.layer.%d." % l
        for nm in ("query", "key", "value"):
            out += [(pre + "attention.self.%s.weight" % nm, (h, h)), (pre + "attention.self.%s.bias" % nm, (h,))]
        out += [(pre + "attention.output.dense.weight", (h, h)), (pre + "attention.output.dense.bias", (h,)),
                (pre + "attention.output.LayerNorm.weight", (h,)), (pre + "attention.output.LayerNorm.bias", (h,)),
                (pre + "intermediate.dense_act.weight", (i, h)), (pre + "intermediate.dense_act.bias", (i,)),
                (pre + "output.dense.weight", (h, i)), (pre + "output.dense.bias", (h,)),
                (pre + "output.LayerNorm.weight", (h,)), (pre + "output.LayerNorm.bias", (h,))]
    out += [("bert.pooler.dense_act.weight", (h, h)), ("bert.pooler.dense_act.bias", (h,)),
            ("cls.predictions.bias", (v,)),
            ("cls.predictions.transform.dense_act.weight", (h, h)), ("cls.predictions.transform.dense_act.bias", (h,)),
            ("cls.predictions.transform.LayerNorm.weight", (h,)), ("cls.predictions.transform.LayerNorm.bias", (h,)),
            ("cls.seq_relationship.weight", (2, h)), ("cls.seq_relationship.bias", (2,))]
    return out


def seeded_state(cfg, seed):
    """normal(0, 0.02) weights like init_bert_weights (modeling.py:705-720), non-trivial LayerNorm affine and
    small biases so that parity tests exercise them."""
    rng = np.random.default_rng(seed)
    st = {}
    for name, shape in param_shapes(cfg):
        if "LayerNorm.weight" in name:
            a = rng.uniform(0.8, 1.2, shape)
        elif name.endswith(".bias"):
            a = rng.standard_normal(shape) * 0.02
        else:
            a = rng.standard_normal(shape) * 0.02
        st[name] = torch.from_numpy(a.astype(np.float32))
    return st


def seeded_batch(cfg, seed, batch, masked_per_seq=None):
    """lddl-shaped synthetic batch (run_pretraining.py:603-609): int64 ids/types/mask/labels, NSP labels."""
    rng = np.random.default_rng(seed)
    s, v = cfg["seq"], cfg["real_vocab"]
    ids = rng.integers(0, v, (batch, s)).astype(np.int64)
    split = rng.integers(s // 4, 3 * s // 4, batch)
    tt = (np.arange(s)[None, :] >= split[:, None]).astype(np.int64)
    lens = rng.integers(3 * s // 4, s + 1, batch)
    mask = (np.arange(s)[None, :] < lens[:, None]).astype(np.int64)
    labels = -np.ones((batch, s), np.int64)
    k = masked_per_seq or max(1, (20 * s) // 128)
    for b in range(batch):
        pos = rng.choice(int(lens[b]), k, replace=False)
        labels[b, pos] = rng.integers(0, v, k)
    nsp = rng.integers(0, 2, batch).astype(np.int64)
    return tuple(torch.from_numpy(a) for a in (ids, tt, mask, labels, nsp))


def gelu(x):
    return TF.gelu(x, approximate="tanh")


def poly_warmup_lr(step_after, base_lr, warmup, total_steps, degree=0.5):
    """schedulers.py:123-136 with last_epoch = step + 1."""
    progress = step_after / total_steps
    return base_lr * progress / warmup if progress < warmup else base_lr * ((1.0 - progress) ** degree)


class BertOracle:
    """storage_dtype=None: the reference's fp32 CPU path.  storage_dtype=torch.float16 / bfloat16: the same fp32 math with
    the tensors the AMP path keeps in 16 bits (GEMM weights, every activation between two kernels, their gradients)
    rounded where they are produced (oracle/storage.py) -- the measured precision floor of the loss-parity bars.  LayerNorm
    parameters, biases, softmax internals, the MLM / NSP logits, the losses and the optimizer stay fp32, as in the engine."""

    def __init__(self, cfg, state, lr=6e-3, warmup=0.2843, total_steps=7038, weight_decay=0.01, max_grad_norm=1.0,
                 storage_dtype=None):
        self.cfg = cfg
        self.sd = storage_dtype
        self.p = {k: v.clone().float().requires_grad_(True) for k, v in state.items()}
        self.m = {k: np.zeros(tuple(v.shape), np.float32) for k, v in state.items()}
        self.v = {k: np.zeros(tuple(v.shape), np.float32) for k, v in state.items()}
        self.step_count = 0
        self.lr, self.warmup, self.total, self.wd, self.max_norm = lr, warmup, total_steps, weight_decay, max_grad_norm

    def forward(self, ids, tt, mask, labels, masks=None, p_hidden=0.0, p_attn=0.0):
        """masks (optional): {"emb": bool [b,s,h], "attn<l>": bool [b,nh,s,s], "out1_<l>" / "out2_<l>": bool [b,s,h]}
        keep masks of the nn.Dropout sites; kept values are scaled by 1/(1-p) with p quantised like the C ABI."""
        p, c = self.p, self.cfg
        Q = lambda t: q_store(t, self.sd)
        q16 = lambda pr: round(pr * 65536.0) / 65536.0
        drop = lambda t, key, pr: t if masks is None or pr <= 0 else t * masks[key].to(t.dtype) / (1.0 - q16(pr))
        b, s = ids.shape
        h, nh = c["hidden"], c["heads"]
        d = h // nh
        e = (p["bert.embeddings.word_embeddings.weight"][ids] + p["bert.embeddings.position_embeddings.weight"][:s][None]
             + p["bert.embeddings.token_type_embeddings.weight"][tt])
        x = Q(TF.layer_norm(e, (h,), p["bert.embeddings.LayerNorm.weight"], p["bert.embeddings.LayerNorm.bias"], 1e-12))
        x = Q(drop(x, "emb", p_hidden))
        ext = (1.0 - mask.float())[:, None, None, :] * -10000.0
        for l in range(c["layers"]):
            pre = "bert.encoder.layer.%d." % l
            lin = lambda t, n: Q(TF.linear(t, Q(p[pre + n + ".weight"]), p[pre + n + ".bias"]))
            q = lin(x, "attention.self.query").view(b, s, nh, d).transpose(1, 2)
            k = lin(x, "attention.self.key").view(b, s, nh, d).transpose(1, 2)
            v = lin(x, "attention.self.value").view(b, s, nh, d).transpose(1, 2)
            sc = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(d) + ext
            probs = drop(torch.softmax(sc, -1), "attn%d" % l, p_attn)
            ctx = Q(torch.matmul(probs, v).transpose(1, 2).reshape(b, s, h))
            a = Q(TF.layer_norm(drop(lin(ctx, "attention.output.dense"), "out1_%d" % l, p_hidden) + x, (h,),
                                p[pre + "attention.output.LayerNorm.weight"], p[pre + "attention.output.LayerNorm.bias"], 1e-12))
            # (the engine's FFN-1 GEMM applies bias + GELU to the fp32 accumulator and stores the result once)
            it = Q(gelu(TF.linear(a, Q(p[pre + "intermediate.dense_act.weight"]), p[pre + "intermediate.dense_act.bias"])))
            x = Q(TF.layer_norm(drop(lin(it, "output.dense"), "out2_%d" % l, p_hidden) + a, (h,),
                                p[pre + "output.LayerNorm.weight"], p[pre + "output.LayerNorm.bias"], 1e-12))
        pooled = Q(torch.tanh(TF.linear(x[:, 0], Q(p["bert.pooler.dense_act.weight"]), p["bert.pooler.dense_act.bias"])))
        flat = x.reshape(-1, h)
        sel = torch.nonzero(labels.reshape(-1) != -1).squeeze(1)
        t = Q(gelu(TF.linear(flat[sel], Q(p["cls.predictions.transform.dense_act.weight"]),
                             p["cls.predictions.transform.dense_act.bias"])))
        t = Q(TF.layer_norm(t, (h,), p["cls.predictions.transform.LayerNorm.weight"],
                            p["cls.predictions.transform.LayerNorm.bias"], 1e-12))
        scores = TF.linear(t, Q(p["bert.embeddings.word_embeddings.weight"])) + p["cls.predictions.bias"]     # fp32 logits
        nsp = TF.linear(pooled, Q(p["cls.seq_relationship.weight"]), p["cls.seq_relationship.bias"])
        return scores, nsp, sel

    def loss(self, ids, tt, mask, labels, nsp_labels, masks=None, p_hidden=0.0, p_attn=0.0):
        scores, nsp, sel = self.forward(ids, tt, mask, labels, masks, p_hidden, p_attn)
        mlm = TF.cross_entropy(scores, labels.reshape(-1)[sel])
        return mlm + TF.cross_entropy(nsp, nsp_labels)

    def step(self, ids, tt, mask, labels, nsp_labels):
        for v in self.p.values():
            v.grad = None
        loss = self.loss(ids, tt, mask, labels, nsp_labels)
        loss.backward()
        self.lamb_update({k: v.grad.numpy() for k, v in self.p.items()})
        return float(loss.detach())

    def lamb_update(self, grads):
        """FusedLAMBAMP.step (fused_lamb.py:131-258): global grad norm over ALL params, two groups (decay / no decay
        for bias, LayerNorm: run_pretraining.py:422-427), adam_w_mode, grad averaging, bias correction."""
        names = list(self.p)
        gn, _ = L.l2norm([grads[k] for k in names])
        self.step_count += 1
        lr = poly_warmup_lr(self.step_count, self.lr, self.warmup, self.total)
        no_decay = ("bias", "gamma", "beta", "LayerNorm")
        for group_wd, sel in ((self.wd, [k for k in names if not any(nd in k for nd in no_decay)]),
                              (0.0, [k for k in names if any(nd in k for nd in no_decay)])):
            g = [grads[k] for k in sel]
            pp = [self.p[k].detach().numpy() for k in sel]
            _, p2, m2, v2, _ = L.lamb_step(g, pp, [self.m[k] for k in sel], [self.v[k] for k in sel], lr, 0.9, 0.999,
                                           1e-6, self.step_count, True, group_wd, True, 1, gn, np.float32(self.max_norm))
            for k, a, b, c in zip(sel, p2, m2, v2):
                with torch.no_grad():
                    self.p[k].copy_(torch.from_numpy(a))
                self.m[k], self.v[k] = b, c


BERT_TINY = dict(hidden=256, heads=4, layers=2, intermediate=1024, vocab=1024, real_vocab=1000, max_pos=512,
                 type_vocab=2, seq=128)
BERT_STEP_CONFIG = dict(cfg=BERT_TINY, seed=21, batch=4, steps=5, lr=6e-3, warmup=0.2843, total_steps=20)

# One encoder layer at BERT-LARGE width (hidden 1024, 16 heads of 64, FFN 4096, the padded 30528 vocabulary, S = 128):
# the GEMM / attention / LayerNorm shapes of BASELINE.json configs[2], small enough for the CPU reference in seconds.
BERT_LARGE_1L = dict(hidden=1024, heads=16, layers=1, intermediate=4096, vocab=30528, real_vocab=30522, max_pos=512,
                     type_vocab=2, seq=128)
BERT_STEP_CONFIG_LARGE = dict(cfg=BERT_LARGE_1L, seed=33, batch=4, steps=2, lr=6e-3, warmup=0.2843, total_steps=20)

# The full 24-layer BERT-Large of BASELINE.json configs[2] (the model bench.py times), batch 4, S = 128, 20 masked positions per
# sequence, 2 LAMB steps: ~40 s per forward + backward on the CPU reference.  Fixture: tests/golden/bert_step_large24.npz.
BERT_LARGE_24L = dict(BERT_LARGE_1L, layers=24)
BERT_STEP_CONFIG_LARGE24 = dict(cfg=BERT_LARGE_24L, seed=44, batch=4, steps=2, lr=6e-3, warmup=0.2843, total_steps=20)


def large24_probe_names(cfg=None):
    """The parameters whose first-step gradients the 24-layer fixture keeps: embeddings, encoder layers 0 / 12 / 23 (all 16
    tensors each), pooler and the heads."""
    cfg = cfg or BERT_LARGE_24L
    keep = []
    for name, _ in param_shapes(cfg):
        if name.startswith("bert.encoder.layer."):
            if int(name.split(".")[3]) in (0, cfg["layers"] // 2, cfg["layers"] - 1):
                keep.append(name)
        else:
            keep.append(name)
    return keep


def grad_sample_index(n, cap=4096):
    """Strided sample of a flattened gradient (<= cap elements): what the fixture stores of each probed tensor."""
    step = max(1, n // cap)
    return np.arange(0, n, step)[:cap]
