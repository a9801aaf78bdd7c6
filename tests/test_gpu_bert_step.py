"""BERT pre-training step on the HIP kernels vs the reference's own BertForPreTraining run on CPU
(tests/golden/bert_step.npz, oracle/make_golden.py gen_bert; dropout 0) and vs the CPU oracle live.  GPU only.
Tolerance: per-step loss within 1e-3 relative (BASELINE.json north_star) + the 16-bit STORAGE floor of the network, which
the oracle measures itself (oracle/storage.py: the fp32 restatement with every tensor the AMP path keeps in fp16 / bf16
rounded where it is produced; committed next to the reference losses in the fixtures by oracle/make_golden.py floors)."""
import os

import numpy as np
import pytest
import torch

from oracle import bert_oracle as BO

pytestmark = pytest.mark.gpu


def _build(cuda, dtype, c, state, p_hidden=0.0, p_attn=0.0):
    from deeplearningexamples_amd.bert.model import BertForPreTraining
    from deeplearningexamples_amd.bert.engine import BertTrainer
    model = BertForPreTraining(c["cfg"], device=cuda)
    res = model.load_state_dict({k: v.clone() for k, v in state.items()}, strict=False)
    assert not res.unexpected_keys and res.missing_keys == ["cls.predictions.decoder.weight"] or not res.missing_keys
    tr = BertTrainer(model, lr=c["lr"], warmup=c["warmup"], total_steps=c["total_steps"], compute_dtype=dtype,
                     init_loss_scale=1024.0, hidden_dropout=p_hidden, attention_dropout=p_attn, seed=1234)
    return model, tr


CONFIGS = {"tiny": ("BERT_STEP_CONFIG", "bert_step.npz"),
           # one encoder layer at BERT-LARGE width: hidden 1024, 16 heads, FFN 4096, vocabulary 30528, S = 128 -- the
           # GEMM / attention / LayerNorm shapes of BASELINE.json configs[2]
           "large1l": ("BERT_STEP_CONFIG_LARGE", "bert_step_large1l.npz")}


@pytest.mark.parametrize("which", ["tiny", "large1l"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_bert_losses_match_reference(cuda, golden_dir, dtype, which):
    c = getattr(BO, CONFIGS[which][0])
    gold = np.load(os.path.join(golden_dir, CONFIGS[which][1]))
    state = BO.seeded_state(c["cfg"], c["seed"])
    model, tr = _build(cuda, dtype, c, state)
    batch = [t.to(cuda) for t in BO.seeded_batch(c["cfg"], c["seed"] + 1, c["batch"])]
    losses = [float(tr.train_step(*batch).item()) for _ in range(c["steps"])]
    ref = gold["losses"]
    rel = np.abs(np.asarray(losses) - ref) / ref
    floor = np.abs(gold["losses_%s_storage" % ("fp16" if dtype == torch.float16 else "bf16")] - ref) / ref
    print(dtype, "losses", losses, "reference", ref.tolist(), "rel err / 1e-3", (rel / 1e-3).tolist(), "storage floor", floor.tolist())
    assert np.all(rel <= 1e-3), (rel, floor)      # the BARE 1e-3 of north_star; the storage floor is context only (printed)
    assert losses[-1] < losses[0] - (0.2 if which == "tiny" else 0.0)
    named = dict(model.named_parameters())
    ref = gold["final_pooler_bias"]
    assert np.abs(named["bert.pooler.dense_act.bias"].detach().cpu().numpy() - ref).max() <= 0.05 * np.abs(ref).max() + 1e-4
    ref = gold["final_query_row"]
    got = named["bert.encoder.layer.0.attention.self.query.weight"].detach().cpu().numpy()[:4]
    assert np.abs(got - ref).max() <= 0.05 * np.abs(ref).max()


@pytest.mark.parametrize("which", ["tiny", "large1l"])
@pytest.mark.parametrize("dtype,bar", [(torch.float16, 0.03), (torch.bfloat16, 0.12)])
def test_bert_first_step_gradients_vs_oracle(cuda, dtype, bar, which):
    c = getattr(BO, CONFIGS[which][0])
    state = BO.seeded_state(c["cfg"], c["seed"])
    model, tr = _build(cuda, dtype, c, state)
    cpu_batch = BO.seeded_batch(c["cfg"], 99, 3)
    orc = BO.BertOracle(c["cfg"], state)
    lo = orc.loss(*cpu_batch)
    lo.backward()
    floor = abs(float(BO.BertOracle(c["cfg"], state, storage_dtype=dtype).loss(*cpu_batch)) - float(lo))
    loss, dlog, dnsp = tr.forward(*[t.to(cuda) for t in cpu_batch])
    print(dtype, which, "loss", loss.item(), "oracle", float(lo), "storage floor", floor)
    assert abs(loss.item() - float(lo)) <= 1e-3 * float(lo) + floor
    tr.backward(dlog, dnsp)
    torch.cuda.synchronize()
    scale = float(tr.scaler.scale.item()) if tr.scaler.enabled else 1.0
    bad, report = [], []
    for n, p in orc.p.items():
        g = tr.gview[n].reshape(-1).cpu().double() / scale
        r = p.grad.reshape(-1).double()
        if float(r.norm()) < 1e-6:
            # key biases: softmax is invariant to a per-row constant, the exact gradient is 0 (the reference's is
            # rounding noise); 16-bit rounding leaves a small non-zero residue here
            assert float(g.norm()) < 2e-2 * float(orc.p[n.replace("key.bias", "query.bias")].grad.norm()) + 1e-6, n
            continue
        rel = float((g - r).norm() / (r.norm() + 1e-12))
        report.append((n.replace("bert.encoder.layer.", "L"), round(rel, 4)))
        if rel > bar:
            bad.append(report[-1])
    print("relative L2 gradient errors (every 5th):", report[::5])
    assert not bad, bad[:12]


def _masks_from(tr, cfg, b, s):
    """Unpack the keep masks of the last forward/backward (engine keeps them when keep_activations is set)."""
    from deeplearningexamples_amd import functional as F
    sv = tr._last_sv
    h, nh = cfg["hidden"], cfg["heads"]
    masks = {"emb": F.unpack_dropout_mask(sv["mask0"], (b, s, h)).cpu()}
    for l, a in enumerate(sv["layers"]):
        masks["attn%d" % l] = F.unpack_dropout_mask(a["mask_a"], (b, nh, s, s)).cpu()
        masks["out1_%d" % l] = F.unpack_dropout_mask(a["mask_1"], (b, s, h)).cpu()
        masks["out2_%d" % l] = F.unpack_dropout_mask(a["mask_2"], (b, s, h)).cpu()
    return masks


@pytest.mark.parametrize("dtype,bar", [(torch.float16, 0.03), (torch.bfloat16, 0.12)])
def test_bert_training_mode_dropout_vs_oracle(cuda, dtype, bar):
    """Training mode (hidden / attention dropout 0.1, the reference's bert_config.json): loss and every gradient
    against the oracle under the SAME keep masks; keep rate, determinism and mask freshness of the RNG."""
    c = BO.BERT_STEP_CONFIG
    cfg = c["cfg"]
    state = BO.seeded_state(cfg, c["seed"])
    model, tr = _build(cuda, dtype, c, state, 0.1, 0.1)
    tr.keep_activations = True
    cpu_batch = BO.seeded_batch(cfg, 99, 3)
    b, s = cpu_batch[0].shape
    loss, dlog, dnsp = tr.forward(*[t.to(cuda) for t in cpu_batch])
    tr.backward(dlog, dnsp)
    torch.cuda.synchronize()
    masks = _masks_from(tr, cfg, b, s)
    keep = np.mean([float(m.float().mean()) for m in masks.values()])
    assert abs(keep - 0.9) < 5e-3, keep
    assert not torch.equal(masks["out1_0"], masks["out2_0"]) and not torch.equal(masks["out1_0"], masks["out1_1"])
    orc = BO.BertOracle(cfg, state)
    lo = orc.loss(*cpu_batch, masks=masks, p_hidden=0.1, p_attn=0.1)
    lo.backward()
    floor = abs(float(BO.BertOracle(cfg, state, storage_dtype=dtype).loss(*cpu_batch, masks=masks, p_hidden=0.1, p_attn=0.1)) - float(lo))
    assert abs(loss.item() - float(lo)) <= 1e-3 * float(lo) + floor, (loss.item(), float(lo), floor)
    scale = float(tr.scaler.scale.item()) if tr.scaler.enabled else 1.0
    bad = []
    for n, p in orc.p.items():
        g = tr.gview[n].reshape(-1).cpu().double() / scale
        r = p.grad.reshape(-1).double()
        if float(r.norm()) < 1e-6:
            continue
        rel = float((g - r).norm() / (r.norm() + 1e-12))
        if rel > bar:
            bad.append((n, round(rel, 4)))
    assert not bad, bad[:12]
    # same seed, same call sequence -> same masks; the next step draws new ones
    model2, tr2 = _build(cuda, dtype, c, state, 0.1, 0.1)
    tr2.keep_activations = True
    l2, d2, n2 = tr2.forward(*[t.to(cuda) for t in cpu_batch])
    tr2.backward(d2, n2)
    m2 = _masks_from(tr2, cfg, b, s)
    assert all(torch.equal(masks[k], m2[k]) for k in masks)
    assert abs(float(l2.item()) - float(loss.item())) <= 1e-6 * float(loss.item())      # (loss reduction uses fp32 atomics)
    l3, d3, n3 = tr2.forward(*[t.to(cuda) for t in cpu_batch])
    tr2.backward(d3, n3)
    m3 = _masks_from(tr2, cfg, b, s)
    assert not torch.equal(m3["emb"], m2["emb"])


def test_dropout_entry_points(cuda):
    """dle_dropout_fwd / bwd: scaling, bit-packed mask layout, p quantisation, offsets decorrelate."""
    from deeplearningexamples_amd import functional as F
    x = torch.randn(4096, 64).to(torch.bfloat16).to(cuda)
    y, m = F.dropout_fwd(x, 0.1, 7, 1)
    keep = F.unpack_dropout_mask(m, x.shape)
    inv = 65536.0 / (65536 - round(0.1 * 65536))
    ref = torch.where(keep, x.float() * inv, torch.zeros_like(x, dtype=torch.float32)).to(torch.bfloat16)
    assert torch.equal(y, ref)
    assert abs(float(keep.float().mean()) - 0.9) < 3e-3
    dx = F.dropout_bwd(x, m, 0.1)
    assert torch.equal(dx, ref)
    y2, m2 = F.dropout_fwd(x, 0.1, 7, 2)
    agree = float((F.unpack_dropout_mask(m2, x.shape) == keep).float().mean())
    assert abs(agree - (0.81 + 0.01)) < 5e-3, agree          # independent masks agree with prob p^2 + (1-p)^2
    y3, m3 = F.dropout_fwd(x, 0.1, 7, 1)
    assert torch.equal(m3, m)
    y0, m0 = F.dropout_fwd(x, 0.0, 7, 1)
    assert torch.equal(y0, x) and int(m0.min()) == 255


def test_dropout_masks_bit_exact_vs_philox_oracle(cuda):
    """The keep masks of all three dropout entry points equal the KAT-pinned CPU Philox restatement bit for bit
    (integer work: bit-exact), including 64-bit seeds / offsets and the chunk numbering of the fused kernels."""
    from deeplearningexamples_amd import functional as F
    from oracle import philox_oracle as P
    seed, off = 0x9E3779B97F4A7C15, (1 << 33) + 5
    g = torch.Generator().manual_seed(3)
    x = torch.randn(512, 256, generator=g).to(torch.bfloat16).to(cuda)
    y, m = F.dropout_fwd(x, 0.1, seed, off)
    keep = F.unpack_dropout_mask(m, x.shape).cpu().numpy()
    ref = P.keep_mask(x.numel(), 0.1, seed, off).reshape(x.shape)
    assert np.array_equal(keep, ref)
    exp = torch.where(torch.from_numpy(ref).to(cuda), x.float() * float(P.inv_keep(0.1)), torch.zeros((), device=cuda))
    assert torch.equal(y, exp.to(torch.bfloat16))
    res = torch.randn(512, 256, generator=g).to(torch.bfloat16).to(cuda)
    gamma, beta = torch.ones(256, device=cuda), torch.zeros(256, device=cuda)
    _, z, _, _, m2 = F.dropout_add_layernorm_fwd(x, gamma, beta, res, 0.25, seed + 1, off + 1)
    ref2 = P.keep_mask(x.numel(), 0.25, seed + 1, off + 1).reshape(x.shape)
    assert np.array_equal(F.unpack_dropout_mask(m2, x.shape).cpu().numpy(), ref2)
    zd = torch.where(torch.from_numpy(ref2).to(cuda), x.float() * float(P.inv_keep(0.25)), torch.zeros((), device=cuda))
    assert torch.equal(z, (zd.to(torch.bfloat16).float() + res.float()).to(torch.bfloat16))
    scores = torch.randn(6, 128, 128, generator=g).to(torch.bfloat16).to(cuda)
    madd = torch.zeros(2, 128, device=cuda)
    dropped, m3 = F.softmax_dropout_fwd_(scores, madd, 3 * 128, 0.125, 0.1, 11, 12)
    ref3 = P.keep_mask(scores.numel(), 0.1, 11, 12).reshape(scores.shape)
    assert np.array_equal(F.unpack_dropout_mask(m3, scores.shape).cpu().numpy(), ref3)
    assert torch.equal(dropped != 0, torch.from_numpy(ref3).to(cuda) & (scores != 0))


def test_bert_phase2_seq512_vs_oracle(cuda):
    """Phase 2 of the reference's recipe (run_pretraining.py --phase2: sequence length 512, 80 predictions) on the FUSED attention
    kernels (K / V streamed in 128-key blocks, csrc/attention.hip: the [B, 16, 512, 512] scores never reach HBM).  Loss and every
    gradient of the first step vs the CPU oracle."""
    c = BO.BERT_STEP_CONFIG
    cfg = dict(c["cfg"], seq=512)
    state = BO.seeded_state(cfg, c["seed"])
    model, tr = _build(cuda, torch.float16, dict(c, cfg=cfg), state)
    cpu_batch = BO.seeded_batch(cfg, 7, 2)
    assert cpu_batch[0].shape == (2, 512)
    orc = BO.BertOracle(cfg, state)
    lo = orc.loss(*cpu_batch)
    lo.backward()
    loss, dlog, dnsp = tr.forward(*[t.to(cuda) for t in cpu_batch])
    assert tr._sv["fused_attn"]
    assert abs(loss.item() - float(lo)) <= 1e-3 * float(lo)
    tr.backward(dlog, dnsp)
    torch.cuda.synchronize()
    scale = float(tr.scaler.scale.item())
    bad = []
    for n, p in orc.p.items():
        g = tr.gview[n].reshape(-1).cpu().double() / scale
        r = p.grad.reshape(-1).double()
        if float(r.norm()) < 1e-6:
            continue
        rel = float((g - r).norm() / (r.norm() + 1e-12))
        if rel > 0.03:
            bad.append((n, round(rel, 4)))
    assert not bad, bad[:12]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_dropout_add_layernorm_bwd_equals_the_three_passes(cuda, dtype):
    """dle_dropout_add_layernorm_bwd = LayerNorm backward + dropout backward + bias column sums in one pass: the same bits
    for dz / dx / dgamma / dbeta as the separate kernels, the same bias gradient up to fp32 summation order."""
    from deeplearningexamples_amd import functional as F
    g = torch.Generator().manual_seed(2)
    rows, h, p = 1000, 1024, 0.1
    x = torch.randn(rows, h, generator=g).to(dtype).to(cuda)
    res = torch.randn(rows, h, generator=g).to(dtype).to(cuda)
    dy = torch.randn(rows, h, generator=g).to(dtype).to(cuda)
    gamma = (1 + 0.1 * torch.randn(h, generator=g)).to(cuda)
    beta = torch.zeros(h, device=cuda)
    y, z, mean, rstd, mask = F.dropout_add_layernorm_fwd(x, gamma, beta, res, p, 5, 6)
    dg1, db1 = torch.zeros(h, device=cuda), torch.zeros(h, device=cuda)
    dz1 = F.layernorm_bwd(dy, z, mean, rstd, gamma, dg1, db1)
    dx1 = F.dropout_bwd(dz1, mask, p)
    bias1 = F.colsum(dx1)
    dg2, db2, bias2 = torch.zeros(h, device=cuda), torch.zeros(h, device=cuda), torch.full((h,), 3.0, device=cuda)
    dz2, dx2 = F.dropout_add_layernorm_bwd(dy, z, mean, rstd, gamma, mask, p, dg2, db2, dbias=bias2, accumulate=False)
    assert torch.equal(dz1, dz2) and torch.equal(dx1, dx2)
    assert torch.allclose(dg1, dg2, rtol=1e-5, atol=1e-4) and torch.allclose(db1, db2, rtol=1e-5, atol=1e-4)
    assert torch.allclose(bias1, bias2, rtol=1e-5, atol=1e-3)
    F.dropout_add_layernorm_bwd(dy, z, mean, rstd, gamma, mask, p, dg2, db2, dbias=bias2, accumulate=True)
    assert torch.allclose(bias2, 2 * bias1, rtol=1e-5, atol=2e-3)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_bert_large_24_layers_vs_reference(cuda, golden_dir, dtype):
    """The model bench.py times -- 24-layer BERT-Large, S = 128, 20 masked positions per sequence (BASELINE.json configs[2]) --
    against the reference's own BertForPreTraining + criterion run on CPU at batch 4 (tests/golden/bert_step_large24.npz,
    oracle/make_golden.py gen_bert_large24; run_pretraining.py:518-536, modeling.py:788-958): the losses of 2 LAMB steps within
    1e-3 + the measured 16-bit storage floor, and the first-step gradients of the embeddings, encoder layers 0 / 12 / 23 and the
    heads (strided samples of every probed tensor) within 1.5 x the storage floor the oracle measured for the same tensor
    (+ 0.5 % / 1 % absolute: the floor is one realisation of the rounding noise, not a bound)."""
    c = BO.BERT_STEP_CONFIG_LARGE24
    gold = np.load(os.path.join(golden_dir, "bert_step_large24.npz"))
    tag = "fp16" if dtype == torch.float16 else "bf16"
    state = BO.seeded_state(c["cfg"], c["seed"])
    model, tr = _build(cuda, dtype, c, state)
    del state
    batch = [t.to(cuda) for t in BO.seeded_batch(c["cfg"], c["seed"] + 1, c["batch"])]
    # ---- step 1 split into its parts: forward + backward (gradients), then the optimizer
    loss, dlog, dnsp = tr.forward(*batch)
    tr._reduce_now = True
    tr.backward(dlog, dnsp)
    torch.cuda.synchronize()
    scale = float(tr.scaler.scale.item()) if tr.scaler.enabled else 1.0
    floors = gold["grad_floor_" + tag]
    slack = 0.005 if dtype == torch.float16 else 0.01
    bad, report = [], []
    for i, n in enumerate(BO.large24_probe_names(c["cfg"])):
        ref, ref_norm = gold["g%03d" % i].astype(np.float64), float(gold["gn%03d" % i])
        g = tr.gview[n].reshape(-1)
        idx = torch.from_numpy(BO.grad_sample_index(g.numel())).to(cuda)
        got = g[idx].double().cpu().numpy() / scale
        if ref_norm < 1e-6:
            # key biases: softmax is invariant to a per-row constant, the exact gradient is 0 and the reference's is rounding noise
            qn = float(gold["gn%03d" % (i - 2)])                       # the query bias of the same layer
            assert float(tr.gview[n].double().norm()) / scale < 2e-2 * qn + 1e-6, n
            continue
        rel = float(np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-30))
        nrm = float(tr.gview[n].double().norm()) / scale
        report.append((n.replace("bert.encoder.layer.", "L"), round(rel, 4), round(float(floors[i]), 4)))
        if rel > 1.5 * floors[i] + slack or abs(nrm - ref_norm) > (1.5 * floors[i] + slack) * ref_norm:
            bad.append(report[-1] + (nrm, ref_norm))
    print(tag, "gradient error / storage floor (every 6th):", report[::6])
    assert not bad, bad[:12]
    tr.optimizer_step()
    losses = [float(loss.item()), float(tr.train_step(*batch).item())]
    ref = gold["losses"]
    rel = np.abs(np.asarray(losses) - ref) / ref
    floor = np.abs(gold["losses_%s_storage" % tag] - ref) / ref
    print(tag, "24-layer losses", losses, "reference", ref.tolist(), "rel err / 1e-3", (rel / 1e-3).tolist(), "storage floor", floor.tolist())
    assert np.all(rel <= 1e-3), (rel, floor)      # the BARE 1e-3 of north_star; the storage floor is context only (printed)
    named = dict(model.named_parameters())
    r = gold["final_pooler_bias"]
    assert np.abs(named["bert.pooler.dense_act.bias"].detach().cpu().numpy() - r).max() <= 0.05 * np.abs(r).max() + 1e-4
