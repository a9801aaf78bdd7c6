"""Full DLRM train step on the HIP kernels vs (a) losses produced by the REFERENCE's DistributedDlrm on CPU
(tests/golden/dlrm_step_*.npz, oracle/make_golden.py) and (b) the CPU oracle run live.  GPU only.

Tolerance: per-step loss within 1e-3 relative (BASELINE.json north_star) -- the HIP path computes in
fp16/bf16 with fp32 accumulation and fp32 master weights, the reference path is fp32."""
import os

import numpy as np
import pytest
import torch

from oracle import dlrm_step_oracle as SO

pytestmark = pytest.mark.gpu


def _build(cfg, device, dtype):
    from deeplearningexamples_amd.dlrm.model import DistributedDlrm
    from deeplearningexamples_amd.dlrm.engine import DlrmTrainer
    model = DistributedDlrm(num_numerical_features=cfg["num"], categorical_feature_sizes=cfg["sizes"],
                            bottom_mlp_sizes=cfg["bottom"], top_mlp_sizes=cfg["top"], embedding_dim=cfg["dim"],
                            device=device, compute_dtype=dtype)
    state = SO.seeded_dlrm_state(cfg["sizes"], cfg["dim"], cfg["bottom"], cfg["top"], cfg["num"], cfg["seed"])
    SO.load_into_hip_model(model, state)
    trainer = DlrmTrainer(model, lr=cfg["lr"], batch_sizes_per_gpu=[cfg["batch"]], amp=True)
    return model, trainer, state


@pytest.mark.parametrize("name", ["tiny", "criteo_shape", "mixed_paths"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_step_losses_match_reference(cuda, golden_dir, name, dtype):
    cfg = SO.DLRM_STEP_CONFIGS[name]
    gold = np.load(os.path.join(golden_dir, "dlrm_step_%s.npz" % name))
    model, trainer, state = _build(cfg, cuda, dtype)
    num, cat, click = SO.seeded_dlrm_batch(cfg["sizes"], cfg["num"], cfg["batch"], cfg["seed"] + 1000)
    num, cat, click = num.to(cuda), cat.to(cuda), click.to(cuda)
    losses = [float(trainer.train_step(num, cat, click).item()) for _ in range(cfg["steps"])]
    ref = gold["losses"]
    rel = np.abs(np.asarray(losses) - ref) / ref
    # 1e-3 (north_star) + the 16-bit storage floor the oracle measures on this network (oracle/storage.py, fixture arrays)
    floor = np.abs(gold["losses_%s_storage" % ("fp16" if dtype == torch.float16 else "bf16")] - ref) / ref
    print(dtype, name, "rel err / 1e-3", (rel / 1e-3).tolist(), "storage floor", floor.tolist())
    assert np.all(rel <= 1e-3), (rel, floor)      # the BARE 1e-3 of north_star; the storage floor is context only (printed)
    assert trainer.scaler.found_inf.item() == 0
    # weights after the last step vs the reference's (fp32) weights
    got = {"out.weight": model.top_model.out.weight, "bottom_mlp.0.weight": model.bottom_model.mlp.linears[0].weight}
    for k, v in got.items():
        ref = gold["final." + k]
        err = np.abs(v.detach().cpu().numpy() - ref).max()
        assert err <= (2e-2 if dtype == torch.float16 else 6e-2) * np.abs(ref).max(), (k, err)
    if name == "tiny":
        ref = gold["final.embedding"]
        emb = model.bottom_model.embeddings.weight.detach().cpu().numpy()
        assert np.abs(emb - ref).max() <= 3e-2 * np.abs(ref).max()
        # the update must have touched exactly the looked-up rows
        init = state["embedding"].numpy()
        touched = np.zeros(init.shape[0], bool)
        off = np.concatenate([[0], np.cumsum(cfg["sizes"])])
        touched[(cat.cpu().numpy() + off[:-1]).reshape(-1)] = True
        assert np.array_equal(np.any(emb != init, axis=1) | ~touched, np.ones_like(touched))
        assert np.array_equal(emb[~touched], init[~touched])
    if name == "mixed_paths":
        # all three sparse-update paths inside the step (one-hot MFMA <= 128 rows, eight lists <= 4096, one list above):
        # the rows the reference's DistributedDlrm ends with, per probed table, as UPDATES (final - initial: the tables are
        # fp32 on both sides, the gradients that moved them are 16-bit on this side)
        rows = torch.from_numpy(gold["probe_rows"]).to(cuda)
        emb = model.bottom_model.embeddings.weight.detach()[rows].cpu().numpy()
        init = gold["init.embedding_probe"]
        np.testing.assert_array_equal(state["embedding"].numpy()[gold["probe_rows"]], init)
        d_ref, d_hip = gold["final.embedding_probe"] - init, emb - init
        off = np.concatenate([[0], np.cumsum(cfg["sizes"])])
        for t in SO.MIXED_PATHS_PROBE_TABLES:
            m = (gold["probe_rows"] >= off[t]) & (gold["probe_rows"] < off[t + 1])
            scale = np.abs(d_ref[m]).max()
            assert m.sum() > 0 and scale > 0
            err = np.abs(d_hip[m] - d_ref[m]).max()
            print("table", t, "rows", cfg["sizes"][t], "max |update|", scale, "max error", err)
            # (the gradients that produce these updates went through 16-bit activations: 1-3 % of the largest update in fp16,
            #  up to ~6 % in bf16, measured; a wrong path -- a lost duplicate, a row updated twice -- is off by >= 50 %)
            assert err <= (0.06 if dtype == torch.float16 else 0.12) * scale, (t, cfg["sizes"][t], err, scale)
    # the workspace invariant of the duplicate-free update
    assert int((model.bottom_model.embeddings.workspace().head != -1).sum().item()) == 0


def test_step_matches_live_oracle_and_skips_on_overflow(cuda):
    cfg = dict(SO.DLRM_STEP_CONFIGS["tiny"])
    model, trainer, state = _build(cfg, cuda, torch.float16)
    orc = SO.DlrmOracle(state, cfg["sizes"], cfg["lr"])
    num, cat, click = SO.seeded_dlrm_batch(cfg["sizes"], cfg["num"], cfg["batch"], 5)
    for _ in range(3):
        lo = orc.step(num, cat, click)
        lh = float(trainer.train_step(num.to(cuda), cat.to(cuda), click.to(cuda)).item())
        assert abs(lh - lo) <= 1e-3 * abs(lo)
    # force an overflow: absurd loss scale -> found_inf, the step is skipped, the scale is halved
    w_before = model.top_model.out.weight.detach().clone()
    e_before = model.bottom_model.embeddings.weight.detach().clone()
    trainer.scaler.scale.fill_(3.0e38)
    trainer.train_step(num.to(cuda), cat.to(cuda), click.to(cuda))
    assert torch.equal(model.top_model.out.weight.detach(), w_before)
    assert torch.equal(model.bottom_model.embeddings.weight.detach(), e_before)
    assert trainer.scaler.scale.item() == pytest.approx(1.5e38)
    assert int((model.bottom_model.embeddings.workspace().head != -1).sum().item()) == 0


def test_sparse_sgd_dedup_matches_oracle(cuda):
    """Heavy duplication (tables of 4..100 rows) + big tables, strided 16-bit gradient, vs float64 accumulate."""
    from deeplearningexamples_amd import functional as F
    from oracle import dlrm_oracle as O
    rng = np.random.default_rng(3)
    sizes = [4, 11, 5000, 97, 300, 20000, 129, 1]
    dim, b = 128, 4096
    off = O.table_offsets(sizes)
    w = rng.standard_normal((int(off[-1]), dim)).astype(np.float32)
    idx = np.stack([rng.integers(0, s, b) for s in sizes], 1).astype(np.int64)
    rows = O.offset_indices(idx, off)
    g = rng.standard_normal((b, len(sizes) + 1, dim)).astype(np.float16)     # slot 0 = bottom-MLP grad (unused)
    wd = torch.from_numpy(w).to(cuda)
    gd = torch.from_numpy(g).to(cuda)
    ws = F.EmbUpdateWorkspace(off, dim, cuda)
    lr, inv = 0.3, torch.tensor([0.5], device=cuda)
    F.emb_sgd_dedup_(wd, torch.from_numpy(rows).to(cuda), gd[:, 1:, :], ws, lr, scale=inv,
                     grad_batch_stride=(len(sizes) + 1) * dim)
    exp = O.sparse_sgd(w, rows, g[:, 1:, :].astype(np.float32) * 0.5, lr)
    np.testing.assert_allclose(wd.cpu().numpy(), exp, rtol=2e-5, atol=2e-5)
    assert int((ws.head != -1).sum().item()) == 0
    # atomic variant (reference gather_gpu_bwd_fuse_sgd semantics) agrees too
    wd2 = torch.from_numpy(w).to(cuda)
    F.emb_sparse_sgd_(wd2, torch.from_numpy(rows).to(cuda), gd[:, 1:, :].contiguous(), lr, scale=inv)
    np.testing.assert_allclose(wd2.cpu().numpy(), exp, rtol=2e-4, atol=2e-4)


def test_small_kernels(cuda):
    from deeplearningexamples_amd import functional as F
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1000, generator=g) * 4
    y = torch.randint(0, 2, (1000,), generator=g).float()
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        xd = x.to(dt)
        loss, dl = F.bce_with_logits(xd.to(cuda), y.to(cuda), grad_scale=torch.tensor([8.0], device=cuda))
        xr = xd.float().detach().clone().requires_grad_()
        ref = torch.nn.functional.binary_cross_entropy_with_logits(xr, y)
        ref.backward()
        assert abs(loss.item() - ref.item()) < 1e-5
        tol = 1e-6 if dt == torch.float32 else 1e-2
        np.testing.assert_allclose(dl.float().cpu().numpy(), (xr.grad * 8).to(dt).float().numpy(), rtol=tol, atol=1e-7)
    a = torch.randn(37, 13, generator=g)
    p = F.cast_rows(a.to(cuda), torch.float16, cols_out=16).cpu()
    assert torch.equal(p[:, :13], a.half()) and (p[:, 13:] == 0).all()
    assert torch.equal(F.cast(a.to(cuda), torch.bfloat16).cpu(), a.bfloat16())
    gg, yy = torch.randn(9, 64, generator=g).half(), torch.randn(9, 64, generator=g).half()
    r = F.relu_bwd(gg.to(cuda), yy.to(cuda)).cpu()
    assert torch.equal(r, torch.where(yy > 0, gg, torch.zeros_like(gg)))
    fi = torch.zeros(1, device=cuda)
    F.check_nonfinite_(torch.ones(4096, device=cuda).half(), fi)
    assert fi.item() == 0
    bad = torch.ones(4096, device=cuda).half()
    bad[777] = float("nan")
    F.check_nonfinite_(bad, fi)
    assert fi.item() == 1
    sc, tr, inv = torch.tensor([1024.0], device=cuda), torch.zeros(1, dtype=torch.int32, device=cuda), torch.zeros(1, device=cuda)
    F.amp_update_scale_(sc, tr, fi, inv, growth_interval=2)
    assert sc.item() == 512 and fi.item() == 0 and inv.item() == 1 / 512
    F.amp_update_scale_(sc, tr, fi, inv, growth_interval=2)
    F.amp_update_scale_(sc, tr, fi, inv, growth_interval=2)
    assert sc.item() == 1024 and tr.item() == 0


def test_evaluate_auc_and_loss(cuda):
    """dlrm/utils.evaluate (dist_evaluate, main.py:733-835): the validation loss of a batch equals the training loss of the
    step that sees the same weights; AUC of the HIP forward equals the AUC of the CPU oracle's logits."""
    from deeplearningexamples_amd.dlrm.utils import evaluate, roc_auc_score
    cfg = SO.DLRM_STEP_CONFIGS["tiny"]
    model, trainer, state = _build(cfg, cuda, torch.float16)
    batches = [[t.to(cuda) for t in SO.seeded_dlrm_batch(cfg["sizes"], cfg["num"], cfg["batch"], 50 + i)] for i in range(3)]
    auc, loss = evaluate(model, batches)
    assert 0.0 < auc < 1.0 and np.isfinite(loss)
    _, loss0 = evaluate(model, batches[:1])
    assert abs(loss0 - float(trainer.train_step(*batches[0]).item())) <= 1e-3 * loss0
    # the module's own forward (model/distributed.py:160-180) is the validation pass's forward
    lg = model(batches[1][0], batches[1][1], [cfg["batch"]]).reshape(-1).float()
    want = model.top_model(model.bottom_model(batches[1][0], batches[1][1])).reshape(-1).float()
    assert torch.equal(lg, want) and lg.shape[0] == cfg["batch"]
    orc = SO.DlrmOracle(state, cfg["sizes"], cfg["lr"])
    if hasattr(orc, "forward"):
        logits = torch.cat([orc.forward(*[t.cpu() for t in b[:2]]).reshape(-1) for b in batches])
        ref_auc = roc_auc_score(torch.cat([b[2].cpu() for b in batches]), logits.detach())
        model2, _, _ = _build(cfg, cuda, torch.float16)
        auc2, _ = evaluate(model2, batches)
        assert abs(auc2 - ref_auc) < 5e-3, (auc2, ref_auc)
