"""ResNet-50 v1.5 train step on the HIP kernels vs the reference's own module run on CPU in fp32
(tests/golden/rn50_step.npz, oracle/make_golden.py gen_rn50) and vs the CPU oracle run live.  GPU only.

Configuration: batch 32, 64x64 synthetic images, seeded weights with damped residual branches (bn3 gamma in
0.1..0.3 -- with gamma ~ 1 a random-init ResNet-50 amplifies 16-bit rounding ~20x and is chaotic from step to
step, measured in oracle/resnet_oracle.py; the damped network is well conditioned: fixture `sensitivity` ~1e-7).
Tolerance (relative, per-step loss): 1e-3 (BASELINE north_star) + the 16-bit STORAGE floor of this network for the
dtype at that step, as the oracle measured it when the fixture was made (golden `losses_<dtype>_storage`: the same
fp32 math with every tensor the AMP path keeps in 16 bits rounded where it is produced; fp16 <= 4e-5, bf16 <= 7.5e-4)."""
import os

import numpy as np
import pytest
import torch

from oracle import resnet_oracle as RO

pytestmark = pytest.mark.gpu


def _build(cuda, dtype, lr, state):
    from deeplearningexamples_amd.convnets.resnet import ResNet50
    from deeplearningexamples_amd.convnets.engine import ResNetTrainer
    model = ResNet50(device=cuda)
    missing = model.load_state_dict({k: v.clone() for k, v in state.items()}, strict=False)
    assert not missing.unexpected_keys and all("running" in k or "num_batches" in k for k in missing.missing_keys)
    tr = ResNetTrainer(model, lr=lr, compute_dtype=dtype, static_loss_scale=128.0)
    return model, tr


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_rn50_losses_match_reference(cuda, golden_dir, dtype):
    c = RO.RN50_STEP_CONFIG
    gold = np.load(os.path.join(golden_dir, "rn50_step.npz"))
    assert gold["sensitivity"].max() < 1e-5            # the configuration itself is not chaotic
    state = RO.seeded_state(c["seed"])
    model, tr = _build(cuda, dtype, c["lr"], state)
    x, y = RO.seeded_batch(c["seed"] + 100, c["batch"], c["size"])
    x, y = x.to(cuda), y.to(cuda)
    losses = [float(tr.train_step(x, y).item()) for _ in range(c["steps"])]
    ref = gold["losses"]
    floor = np.abs(gold["losses_%s_storage" % ("fp16" if dtype == torch.float16 else "bf16")] - ref) / ref
    rel = np.abs(np.asarray(losses) - ref) / ref
    print(dtype, "losses", losses, "reference", ref.tolist(), "rel err / 1e-3", (rel / 1e-3).tolist(), "storage floor", floor.tolist())
    # the BARE 1e-3 of north_star (measured: fp16 <= 3e-5, bf16 <= 3.2e-4); the 16-bit storage floor the oracle measured for this
    # network is printed for context and no longer part of the bar
    assert np.all(rel <= 1e-3), (rel, floor)
    assert losses[-1] < losses[0]
    tr.sync_counters()
    assert int(model.bn1.num_batches_tracked.item()) == c["steps"]
    np.testing.assert_allclose(model.bn1.running_mean.cpu().numpy(), gold["final_bn1_running_mean"], rtol=2e-2, atol=2e-3)
    np.testing.assert_allclose(model.bn1.running_var.cpu().numpy(), gold["final_bn1_running_var"], rtol=2e-2, atol=2e-3)
    # parameters after 4 steps (memory order of conv1 is KRSC)
    w = model.conv1.weight.detach().cpu().numpy()
    assert np.abs(w - gold["final_conv1_weight"]).max() <= 2e-2 * np.abs(gold["final_conv1_weight"]).max()
    np.testing.assert_allclose(model.fc.bias.detach().cpu().numpy(), gold["final_fc_bias"], rtol=5e-2, atol=2e-4)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_rn50_first_step_gradients_vs_oracle(cuda, dtype):
    """Logits, loss and EVERY parameter gradient of the first step vs torch autograd on the CPU oracle.

    Gradient bar: at batch 8 the gradient of this network in 16-bit storage differs from the fp32 gradient by
    5-15 % (fp16) in relative L2 -- measured on the CPU by the SAME oracle with activations / weights / activation
    gradients rounded where the AMP path stores 16 bits (storage_dtype).  The HIP path must be no worse than
    that floor: err(hip, fp32) <= 1.3 * err(16-bit-storage oracle, fp32) + 0.01 for every parameter."""
    from deeplearningexamples_amd import functional as F
    c = dict(RO.RN50_STEP_CONFIG, batch=8)
    state = RO.seeded_state(c["seed"])
    model, tr = _build(cuda, dtype, 0.0, state)
    x, y = RO.seeded_batch(77, c["batch"], c["size"])
    orc = RO.ResNet50Oracle(state, lr=0.0)
    lo = orc.step(x, y)
    emu = RO.ResNet50Oracle(state, lr=0.0, storage_dtype=dtype)
    l16 = emu.step(x, y)
    with torch.no_grad():
        ref_logits = orc.forward(x)
        emu_logits = RO.ResNet50Oracle(state, lr=0.0, storage_dtype=dtype).forward(x)
    # bars: 1e-3 (north_star) + what 16-bit storage alone costs on this network, measured by the oracle
    lfloor = abs(l16 - lo) / lo
    efloor = float((emu_logits - ref_logits).abs().max() / ref_logits.abs().max())
    logits = tr.forward(x.to(cuda))
    err = float((logits.cpu() - ref_logits).abs().max() / ref_logits.abs().max())
    loss, dl = F.softmax_xent(logits, y.to(cuda), smoothing=0.1, grad_dtype=dtype, grad_scale=tr.scaler.scale)
    print("max logit error / max logit %.3e (storage floor %.3e), loss hip %.6f oracle %.6f (storage floor %.3e)"
          % (err, efloor, loss.item(), lo, lfloor))
    assert err <= 2e-2 + 1.5 * efloor
    assert abs(loss.item() - lo) <= 1e-3 * lo                  # bare north_star bar; lfloor printed above for context
    scale = float(tr.scaler.scale.item())
    tr.backward(dl)
    torch.cuda.synchronize()
    report, bad = [], []
    for n, p in model.named_parameters():
        ph = lambda t: (t.permute(0, 2, 3, 1) if t.dim() == 4 else t).reshape(-1).double()      # memory (KRSC) order
        g = tr.gview[n].view(-1).cpu().double() / scale
        r32, r16 = ph(orc.p[n].grad), ph(emu.p[n].grad)
        e_hip = float((g - r32).norm() / (r32.norm() + 1e-12))
        e_floor = float((r16 - r32).norm() / (r32.norm() + 1e-12))
        report.append((n, round(e_hip, 4), round(e_floor, 4)))
        if e_hip > 1.3 * e_floor + 0.01:
            bad.append(report[-1])
    print("(name, hip-vs-fp32, 16-bit-storage-floor) every 9th:", report[::9])
    assert not bad, bad[:10]
    assert report[-2][1] < (0.01 if dtype == torch.float16 else 0.04)    # fc.weight: one GEMM away from the loss


def test_rn50_gradient_accumulation(cuda):
    """--optimizer-batch-size = 2 x batch (main.py:405-416, training.py:86-96,167-186): two train_step calls, ONE optimizer
    step on (g(mb1) + g(mb2)) / 2, the loss of each call divided by 2; BatchNorm statistics per micro-batch.  Checked against
    the same engine run without accumulation: its two micro-batch gradients, averaged on the host, and its own SGD step."""
    from deeplearningexamples_amd import functional as F
    c = RO.RN50_STEP_CONFIG
    state = RO.seeded_state(c["seed"])
    xa, ya = RO.seeded_batch(5, 8, c["size"])
    xb, yb = RO.seeded_batch(6, 8, c["size"])
    xa, ya, xb, yb = xa.to(cuda), ya.to(cuda), xb.to(cuda), yb.to(cuda)
    from deeplearningexamples_amd.convnets.resnet import ResNet50
    from deeplearningexamples_amd.convnets.engine import ResNetTrainer

    def make(acc):
        model = ResNet50(device=cuda)
        model.load_state_dict({k: v.clone() for k, v in state.items()}, strict=False)
        return model, ResNetTrainer(model, lr=0.05, compute_dtype=torch.bfloat16, grad_acc_steps=acc)

    m1, t1 = make(1)
    grads, losses = [], []
    for x, y in ((xa, ya), (xb, yb)):
        logits = t1.forward(x)
        loss, dl = F.softmax_xent(logits, y, smoothing=0.1, grad_dtype=torch.bfloat16)
        t1.backward(dl)
        grads.append(t1.flat_grad.clone())
        losses.append(loss.item())
    w_before = m1.fc.weight.detach().clone()
    t1.flat_grad.copy_((grads[0] + grads[1]) / 2)
    t1.optimizer_step()
    m2, t2 = make(2)
    la = t2.train_step(xa, ya)
    assert torch.equal(m2.fc.weight, w_before), "the optimizer must not step on the first micro-batch"
    lb = t2.train_step(xb, yb)
    assert abs(la.item() - losses[0] / 2) < 1e-6 and abs(lb.item() - losses[1] / 2) < 1e-6
    ref = (grads[0] + grads[1]) / 2
    err = float((t2.flat_grad - ref).abs().max() / ref.abs().max())
    assert err < 1e-6, err
    for (n, p), (_, q) in zip(m2.named_parameters(), m1.named_parameters()):
        assert torch.allclose(p, q, rtol=1e-6, atol=1e-7), n
    assert not torch.equal(m2.fc.weight, w_before)
    np.testing.assert_allclose(m2.bn1.running_mean.cpu().numpy(), m1.bn1.running_mean.cpu().numpy(), rtol=1e-6, atol=1e-7)
    # the third call starts the next accumulation window
    t2.train_step(xa, ya)
    assert t2.steps_since_update == 1


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_rn50_unit_backward_teacher_forced_vs_fp32(cuda, dtype):
    """The discriminating gradient bar (the end-to-end floor above is 14 % / 40 % of the fp32 gradient: BatchNorm's backward
    cancels a common mode, so 16-bit rounding is amplified ~14x per layer and accumulates over 53 layers).  Here every conv +
    BN (+ ReLU) (+ residual) unit of the step is checked ON ITS OWN: the gradient the HIP step handed to the unit (dy, the ReLU
    bits, the residual-branch addend) and the tensors it saved in forward go through plain fp32 torch on the CPU
    (models/resnet.py:148-175, models/common.py:31-128 backward = torch autograd's conv / batch_norm formulas), and the unit's
    d gamma, d beta, dW and dx must agree.  One unit deep, the 16-bit floor is the rounding of ONE stored tensor (the
    normalised-gradient operand: 2^-9 bf16, 2^-12 fp16 relative L2 per element, averaged down in the reductions):
    bars 2 % (bf16) / 0.4 % (fp16) relative L2 -- < 5 % as a parity bar should be."""
    import torch.nn.functional as TF
    from deeplearningexamples_amd import functional as F
    from deeplearningexamples_amd.convnets import resnet as R
    c = dict(RO.RN50_STEP_CONFIG, batch=8)
    state = RO.seeded_state(c["seed"])
    model, tr = _build(cuda, dtype, 0.0, state)
    x, y = RO.seeded_batch(77, c["batch"], c["size"])
    records = []
    orig = R.ConvBN.backward

    def stuffed(t3):
        """("up2", compact [N,P,Q,C], (H, W)) -> the zero-stuffed [N,H,W,C] tensor it stands for."""
        _, compact, (hh, ww) = t3
        full = torch.zeros((compact.shape[0], hh, ww, compact.shape[3]), dtype=compact.dtype, device=compact.device)
        full[:, ::2, ::2] = compact
        return full

    def spy(self, dy, need_dx=True, dx_addend=None, dy_mask=None, compact_dx=False, bnred=None, pooled=None):
        saved = self.saved
        dx = orig(self, dy, need_dx=need_dx, dx_addend=dx_addend, dy_mask=dy_mask, compact_dx=compact_dx, bnred=bnred, pooled=pooled)
        if pooled is not None:              # the stem: its BatchNorm backward gathered the pooling gradient itself -- this tensor
            dy = F.maxpool_bwd(pooled[0], pooled[1], saved[1].shape[1:3])
        is_up2 = lambda v: isinstance(v, tuple) and v[0] == "up2"
        records.append((self, dy, stuffed(dx_addend) if is_up2(dx_addend) else dx_addend, dy_mask, saved,
                        stuffed(dx) if is_up2(dx) else dx))
        return dx
    R.ConvBN.backward = spy
    try:
        logits = tr.forward(x.to(cuda))
        loss, dl = F.softmax_xent(logits, y.to(cuda), smoothing=0.1, grad_dtype=dtype, grad_scale=tr.scaler.scale)
        tr.backward(dl)
        torch.cuda.synchronize()
    finally:
        R.ConvBN.backward = orig
    assert len(records) == 53
    bar = 4e-3 if dtype == torch.float16 else 2e-2
    nchw = lambda t: t.detach().float().cpu().permute(0, 3, 1, 2).contiguous()
    bits = lambda m, ref: F.unpack_dropout_mask(m, ref.shape).cpu().permute(0, 3, 1, 2).float()
    rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    worst = {}
    for u, dy, addend, dy_mask, (xs, t, mask, mean, rstd), dx in records:
        rmask = mask if u.relu else dy_mask
        g = nchw(dy)
        if rmask is not None:
            g = g * bits(rmask, dy)
        tt, mu, rs = nchw(t), mean.cpu().view(1, -1, 1, 1), rstd.cpu().view(1, -1, 1, 1)
        gamma = u.bn.weight.detach().cpu().view(1, -1, 1, 1)
        xhat = (tt - mu) * rs
        m = float(tt.numel() // tt.shape[1])
        dbeta, dgamma = g.sum(dim=(0, 2, 3)), (g * xhat).sum(dim=(0, 2, 3))
        gt = gamma * rs * (g - dbeta.view(1, -1, 1, 1) / m - xhat * dgamma.view(1, -1, 1, 1) / m)
        xin = nchw(xs)[:, :u.cin]
        w16 = u.w16.detach().float().cpu().permute(0, 3, 1, 2)[:, :u.cin].contiguous()           # KRSC -> KCRS
        dw = torch.nn.grad.conv2d_weight(xin, w16.shape, gt, stride=u.stride, padding=u.pad)
        # (the stem on its own kernels writes straight into the flat gradient of the channels_last master: [64][7][7][3])
        got_dw = u.gw_flat.detach().cpu().view(u.cout, u.k, u.k, u.cin).permute(0, 3, 1, 2) if xs.shape[-1] == 4 else \
            u.gw.detach().cpu().permute(0, 3, 1, 2)[:, :u.cin]
        errs = {"dgamma": rel(u.ggamma.cpu(), dgamma), "dbeta": rel(u.gbeta.cpu(), dbeta), "dW": rel(got_dw, dw)}
        if dx is not None:
            ref_dx = torch.nn.grad.conv2d_input(xin.shape, w16, gt, stride=u.stride, padding=u.pad)
            if isinstance(addend, tuple):
                ref_dx = ref_dx + nchw(addend[0]) * bits(addend[1], addend[0])
            elif addend is not None:
                ref_dx = ref_dx + nchw(addend)
            errs["dx"] = rel(nchw(dx)[:, :u.cin], ref_dx)
        for k, e in errs.items():
            if e > worst.get(k, (0.0, ""))[0]:
                worst[k] = (e, u.name_conv)
        bad = {k: e for k, e in errs.items() if e > bar}
        assert not bad, (u.name_conv, bad)
    print(dtype, "worst per-unit relative L2 errors:", {k: (round(v[0], 5), v[1]) for k, v in worst.items()})
