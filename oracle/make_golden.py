"""Generate tests/golden/* from the reference's own code (run in the build container).

    python oracle/make_golden.py [dlrm] [bert] [rn50]

TEST INFRASTRUCTURE.  Imports /root/reference (read-only) through oracle/_ref_import.py,
runs the reference's eager CPU path on seeded inputs and stores inputs + outputs as small
fixtures.  The fixtures travel to the GPU box; the reference does not.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import _ref_import as R  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
CRITEO_F15 = [7912889, 33823, 582469, 245828, 11, 2209, 10667, 104, 4, 968, 15, 8165896, 17139,
              2675940, 7156453, 302516, 12022, 97, 35, 7339, 20046, 4, 7105, 1382, 63, 5554114]


def gen_dlrm():
    ref = R.import_dlrm()
    du = ref.dist_utils
    out = {"device_mapping": [], "gpu_batch_sizes": [], "argsort": [], "tril": {}, "padding": {}}
    size_sets = {
        "criteo_f15": CRITEO_F15,
        "default_26x100000": [100000] * 26,
        "ties": [5, 5, 5, 3, 3, 9, 1, 1, 1, 1, 7],
        "thirty": [(i * 7919) % 1000 + 1 for i in range(30)],
        "ten": [10, 20, 30, 40, 50, 60, 70, 80, 90, 100],
    }
    for name, sizes in size_sets.items():
        for n in (1, 2, 3, 4, 5, 8, 16):
            if n - 1 > len(sizes):
                continue
            out["device_mapping"].append({"name": name, "sizes": sizes, "num_gpus": n,
                                          "result": du.get_device_mapping(sizes, n)})
    for gb, n in [(65536, 8), (65536, 1), (65536, 2), (65536, 4), (32768, 8), (2048, 2), (16384, 4),
                  (65536 - 64, 8), (4096 + 128, 3)]:
        try:
            res = list(du.get_gpu_batch_sizes(gb, n))
        except RuntimeError:
            res = None
        out["gpu_batch_sizes"].append({"global_batch": gb, "num_gpus": n, "result": res})
    for seq in ([3, 1, 2], [5, 5, 1, 5], [1, 1, 1], list(range(10, 0, -1))):
        out["argsort"].append({"seq": seq, "asc": du.argsort(seq), "desc": du.argsort(seq, True)})
    for nv in (2, 3, 8, 27, 31, 32):
        di = ref.interactions.DotInteraction(nv - 1, 128)
        out["tril"][str(nv)] = di._tril_indices.tolist()
        out["padding"][str(nv)] = {"num_interactions": di.num_interactions,
                                   "raw": di._raw_num_interactions}
    with open(os.path.join(GOLD, "dlrm_placement.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)

    # dot interaction fwd/bwd through the reference's DotInteraction + autograd (fp32 CPU)
    arrs = {}
    g = torch.Generator().manual_seed(1234)
    for (b, r, c) in [(16, 32, 32), (17, 31, 37), (15, 31, 37), (8, 27, 128), (3, 2, 8), (4, 27, 16)]:
        di = ref.interactions.DotInteraction(r - 1, c)
        x = torch.rand(b, r, c, generator=g, dtype=torch.float32).requires_grad_()
        y = di.interact(x, x[:, 0, :])
        ug = torch.rand(y.shape, generator=g, dtype=torch.float32)
        y.backward(ug)
        key = f"{b}x{r}x{c}"
        arrs[key + "_x"] = x.detach().numpy()
        arrs[key + "_y"] = y.detach().numpy()
        arrs[key + "_ug"] = ug.numpy()
        # total grad wrt x (includes the bottom-mlp slice landing on row 0)
        arrs[key + "_gx_total"] = x.grad.numpy()
    np.savez_compressed(os.path.join(GOLD, "dlrm_dot_interact.npz"), **arrs)

    # joint embedding: index + offset (+ hash) arithmetic and the gathered rows
    sizes = [11, 4, 968, 15, 97, 35, 63, 104]
    emb = ref.embeddings.JointEmbedding(sizes, 16, device="cpu", hash_indices=True)
    torch.manual_seed(7)
    torch.nn.init.uniform_(emb.embedding.weight.data, -1, 1)
    idx = torch.stack([torch.randint(0, 3 * s, (33,), generator=g) for s in sizes], dim=1)
    idx_in = idx.clone()
    out_e = emb(idx)[0]
    lr = 0.5
    opt = torch.optim.SGD(emb.parameters(), lr=lr)
    ug = torch.rand(out_e.shape, generator=g)
    w0 = emb.embedding.weight.detach().clone()
    out_e.backward(ug)
    opt.step()
    np.savez_compressed(
        os.path.join(GOLD, "dlrm_embedding.npz"),
        sizes=np.asarray(sizes), offsets=emb.offsets.numpy(), idx_in=idx_in.numpy(),
        idx_hashed=idx.numpy(), w0=w0.numpy(), out=out_e.detach().numpy(), ug=ug.numpy(),
        lr=np.float32(lr), w1=emb.embedding.weight.detach().numpy())
    print("dlrm golden written")


def gen_dlrm_step_mixed():
    """Only the mixed_paths case (the others are pinned already; it allocates a 1.2 M-row table three times over)."""
    gen_dlrm_step(only=("mixed_paths",))
    gen_floors(bert=False, dlrm_only=("mixed_paths",))


def gen_dlrm_step(only=None):
    """Per-step losses (+ final weights for the tiny case) of the REFERENCE's DistributedDlrm on CPU, fp32."""
    from oracle import dlrm_step_oracle as SO
    from oracle.dlrm_step_oracle import seeded_dlrm_state, seeded_dlrm_batch, DLRM_STEP_CONFIGS
    ref = R.import_dlrm()
    from dlrm.utils import distributed as du
    du.get_world_size = lambda: 1          # no process group in the build container
    for name, c in DLRM_STEP_CONFIGS.items():
        if only is not None and name not in only:
            continue
        if only is None and name == "mixed_paths":
            continue                       # (gen_dlrm_step_mixed)
        model = ref.model.DistributedDlrm(
            num_numerical_features=c["num"], categorical_feature_sizes=c["sizes"], bottom_mlp_sizes=c["bottom"],
            top_mlp_sizes=c["top"], embedding_type="joint", embedding_dim=c["dim"], interaction_op="dot",
            hash_indices=False, use_cpp_mlp=False, fp16=False, device="cpu")
        state0 = seeded_dlrm_state(c["sizes"], c["dim"], c["bottom"], c["top"], c["num"], c["seed"])
        with torch.no_grad():
            for i, l in enumerate([m for m in model.bottom_model.mlp.layers if isinstance(m, torch.nn.Linear)]):
                l.weight.copy_(state0[f"bottom_mlp.{i}.weight"]); l.bias.copy_(state0[f"bottom_mlp.{i}.bias"])
            for i, l in enumerate([m for m in model.top_model.mlp.layers if isinstance(m, torch.nn.Linear)]):
                l.weight.copy_(state0[f"top_mlp.{i}.weight"]); l.bias.copy_(state0[f"top_mlp.{i}.bias"])
            model.top_model.out.weight.copy_(state0["out.weight"]); model.top_model.out.bias.copy_(state0["out.bias"])
            model.bottom_model.embeddings.embedding.weight.copy_(state0["embedding"])
        num, cat, click = seeded_dlrm_batch(c["sizes"], c["num"], c["batch"], c["seed"] + 1000)
        mlp_params = list(model.top_model.parameters()) + list(model.bottom_model.mlp.parameters())
        opt_mlp = torch.optim.SGD(mlp_params, lr=c["lr"])
        opt_emb = torch.optim.SGD(model.bottom_model.embeddings.parameters(), lr=c["lr"])
        loss_fn = torch.nn.BCEWithLogitsLoss(reduction="mean")
        losses = []
        for _ in range(c["steps"]):
            for p_ in model.parameters():
                p_.grad = None
            out = model(num, cat.clone()).squeeze()
            loss = loss_fn(out, click)
            loss.backward()
            opt_mlp.step()
            opt_emb.step()
            losses.append(float(loss.detach()))
        # the restatement must reproduce the reference (same fp32 ops)
        orc = SO.DlrmOracle(state0, c["sizes"], c["lr"])
        ol = [orc.step(num, cat, click) for _ in range(c["steps"])]
        assert np.allclose(ol, losses, rtol=5e-5, atol=1e-6), (name, ol, losses)
        final = SO.state_from_reference(model)
        for k in final:
            fa, oa = final[k].numpy(), orc.p[k].detach().numpy()
            assert np.abs(fa - oa).max() <= 2e-3 * max(np.abs(fa).max(), 1e-3), (name, k, np.abs(fa - oa).max())
        arrs = {"losses": np.asarray(losses, np.float64)}
        if name == "tiny":
            for k, v in SO.state_to_numpy(final).items():
                arrs["final." + k] = v
        elif name == "mixed_paths":
            arrs["final.out.weight"] = final["out.weight"].numpy()
            arrs["final.bottom_mlp.0.weight"] = final["bottom_mlp.0.weight"].numpy()
            rows = SO.probe_rows(c, cat)
            arrs["probe_rows"] = rows
            arrs["final.embedding_probe"] = final["embedding"][torch.from_numpy(rows)].numpy()
            arrs["init.embedding_probe"] = state0["embedding"][torch.from_numpy(rows)].numpy()
        else:
            arrs["final.out.weight"] = final["out.weight"].numpy()
            arrs["final.bottom_mlp.0.weight"] = final["bottom_mlp.0.weight"].numpy()
            arrs["final.embedding_head"] = final["embedding"][:64].numpy()
        np.savez_compressed(os.path.join(GOLD, "dlrm_step_%s.npz" % name), **arrs)
        print("dlrm_step", name, "losses", arrs["losses"])


def gen_rn50_224():
    """BASELINE.json configs[0] shape (224x224, batch 32): the reference's module for 2 steps (minutes on 8 cores)."""
    gen_rn50(cfg_name="RN50_STEP_CONFIG_224", out_name="rn50_step_224.npz")


def gen_rn50(cfg_name="RN50_STEP_CONFIG", out_name="rn50_step.npz"):
    """Per-step losses of the REFERENCE's resnet50 + LabelSmoothing + get_sgd_optimizer on CPU (fp32)."""
    from oracle import resnet_oracle as RO
    ref = R.import_convnets()
    c = getattr(RO, cfg_name)
    model = ref.models.resnet50(pretrained=False)
    state0 = RO.seeded_state(c["seed"])
    sd = model.state_dict()
    assert [n for n, _ in model.named_parameters()] == [n for n, _ in RO.param_shapes()], "parameter names/order differ"
    for k, v in state0.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    model.load_state_dict({k: v.clone() for k, v in state0.items()}, strict=False)
    model.train()
    x, y = RO.seeded_batch(c["seed"] + 100, c["batch"], c["size"])
    loss_fn = ref.smoothing.LabelSmoothing(0.1)
    opt = ref.optimizers.get_sgd_optimizer(list(model.named_parameters()), c["lr"], momentum=0.875,
                                           weight_decay=3.0517578125e-05)
    losses = []
    for _ in range(c["steps"]):
        opt.zero_grad()
        loss = loss_fn(model(x), y)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    orc = RO.ResNet50Oracle(state0, c["lr"])
    ol = [orc.step(x, y) for _ in range(c["steps"])]
    assert np.allclose(ol[:2], losses[:2], rtol=1e-4), (ol, losses)
    orc2 = RO.ResNet50Oracle(state0, c["lr"])
    lp = [orc2.step(x * (1 + 1e-6), y) for _ in range(c["steps"])]
    sens = [abs(u - v) / abs(u) for u, v in zip(ol, lp)]
    fin = {k: v.detach().numpy() for k, v in model.state_dict().items()}
    # the precision floor of 16-bit STORAGE on this network, measured by the oracle itself: the same fp32 math with
    # every tensor the AMP path keeps in 16 bits rounded where it is produced (oracle/resnet_oracle.py storage_dtype)
    floors = {}
    for nm, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
        o16 = RO.ResNet50Oracle(state0, c["lr"], storage_dtype=dt)
        floors[nm] = [o16.step(x, y) for _ in range(c["steps"])]
    np.savez_compressed(os.path.join(GOLD, out_name), losses=np.asarray(losses, np.float64),
                        losses_fp16_storage=np.asarray(floors["fp16"], np.float64),
                        losses_bf16_storage=np.asarray(floors["bf16"], np.float64),
                        oracle_losses=np.asarray(ol, np.float64), sensitivity=np.asarray(sens, np.float64),
                        final_fc_bias=fin["fc.bias"], final_bn1_weight=fin["bn1.weight"],
                        final_bn1_running_mean=fin["bn1.running_mean"], final_bn1_running_var=fin["bn1.running_var"],
                        final_conv1_weight=fin["conv1.weight"])
    print(out_name, "losses", losses, "oracle", ol, "sensitivity to 1e-6 input noise", sens, "16-bit storage", floors)


def gen_lamb():
    """The reference's UNMODIFIED FusedLAMBAMP + PolyWarmUpScheduler + torch GradScaler stepping on CPU
    (fused_lamb_CUDA = oracle/lamb_cpu_ext.py): pins the optimizer's host sequence (run_pretraining.py:527-536)."""
    from oracle import lamb_oracle as L
    FusedLAMBAMP = R.import_fused_lamb()
    sched = R.import_bert().schedulers
    case = L.LAMB_GOLDEN_CASE
    params0, grads = L.lamb_golden_inputs(case)
    tp = {k: torch.nn.Parameter(torch.from_numpy(a.copy()).to(torch.float16 if half else torch.float32))
          for k, (a, half) in params0.items()}
    opt = FusedLAMBAMP([{"params": [tp[k] for k in names], "weight_decay": wd} for wd, names in case["groups"]],
                       lr=case["lr"])
    opt.setup_fp32_params()                                               # run_pretraining.py:477
    lr_sched = sched.PolyWarmUpScheduler(opt, warmup=case["warmup"], total_steps=case["total_steps"],
                                         base_lr=case["lr"], device="cpu")
    scaler = torch.amp.GradScaler("cpu", init_scale=case["init_scale"], growth_interval=case["growth_interval"])
    host = L.FusedLambHost(params0, case["groups"], case["lr"], case["warmup"], case["total_steps"],
                           init_scale=case["init_scale"], growth_interval=case["growth_interval"])
    rec = {"scale": [], "found_inf": [], "step": [], "lr": []}
    for it, g in enumerate(grads):
        scaler.scale(torch.zeros(1))        # what grad_scaler.scale(loss) does first: lazily create the scale tensor
        scale = float(scaler.get_scale())
        scaled = {}
        for k, p in tp.items():
            gs = (g[k] * np.float32(scale)).astype(np.float16 if p.dtype == torch.float16 else np.float32)
            if it in case["overflow_at"] and k == "w_c":
                gs = gs.copy(); gs.reshape(-1)[5] = np.inf
            scaled[k] = gs
            p.grad = torch.from_numpy(gs.copy())
        lr_sched.step()                                                   # run_pretraining.py:528
        scaler.step(opt)                                                  # :529
        found = float(sum(v.item() for v in scaler._found_inf_per_device(opt).values()))
        scaler.update()                                                   # :535
        opt.zero_grad(set_to_none=True)
        fo = host.optimizer_step(scaled)
        assert bool(found) == bool(fo), (it, found, fo)
        rec["scale"].append(scale); rec["found_inf"].append(found)
        rec["step"].append(int(opt.param_groups[0]["step"].item())); rec["lr"].append(float(opt.param_groups[0]["lr"]))
        assert host.step == rec["step"][-1] and np.float32(host.lr) == np.float32(rec["lr"][-1]), (it, host.step, host.lr, rec)
    out = {k: np.asarray(v, np.float64) for k, v in rec.items()}
    out["final_scale"] = np.float64(scaler.get_scale())
    flat = [p for grp in opt.param_groups for p in grp["params"]]
    flat32 = [p for grp in opt.param_groups_fp32 for p in grp["params"]]
    names = [k for _, ns in case["groups"] for k in ns]
    for k, p, p32 in zip(names, flat, flat32):
        master = (p32 if p32 is not None else p).detach().numpy()
        out["p_" + k] = master
        out["m_" + k] = opt.state[p]["exp_avg"].numpy()
        out["v_" + k] = opt.state[p]["exp_avg_sq"].numpy()
        if p.dtype == torch.float16:
            out["p16_" + k] = p.detach().numpy()
        # the restated host sequence == the reference classes, bit for bit
        assert np.array_equal(master, host.p[k]), k
        assert np.array_equal(out["m_" + k], host.m[k]) and np.array_equal(out["v_" + k], host.v[k]), k
        if p.dtype == torch.float16:
            assert np.array_equal(out["p16_" + k], host.p16[k]), k
    assert np.float32(out["final_scale"]) == host.scale
    np.savez_compressed(os.path.join(GOLD, "lamb_ref_steps.npz"), **out)
    print("lamb: steps", rec["step"], "scales", rec["scale"], "lr", rec["lr"])


def gen_lamb_trace():
    """CALL TRACE of the reference's unmodified FusedLAMBAMP.step (fused_lamb.py:131-260): every call the class makes into
    `fused_lamb_CUDA` over the golden scenario -- entry point, the tensor lists and scalar arguments AS THE CLASS PASSED THEM, and
    what the call left behind (returned norms / the lists it mutates, from oracle/lamb_cpu_ext.py) -- as a fixture the GPU box can
    replay through shims/fused_lamb_CUDA.py without the reference tree (tests/test_gpu_lamb_reference.py).  Data only: argument
    values and results, no reference source."""
    from oracle import lamb_oracle as L
    from oracle import lamb_cpu_ext
    calls = []

    def arr(t):
        return t.detach().clone().numpy()

    orig_l2, orig_lamb = lamb_cpu_ext.multi_tensor_l2norm, lamb_cpu_ext.multi_tensor_lamb

    def rec_l2(chunk_size, noop_flag, tensor_lists, per_tensor=False):
        c = {"fn": "l2norm", "chunk": int(chunk_size), "noop_in": arr(noop_flag), "lists": [[arr(t) for t in l] for l in tensor_lists],
             "per_tensor": per_tensor}
        tot, per = orig_l2(chunk_size, noop_flag, tensor_lists, per_tensor)
        c.update(noop_out=arr(noop_flag), total=arr(tot), per=arr(per))
        calls.append(c)
        return tot, per

    def rec_lamb(chunk_size, noop_flag, tensor_lists, lr, beta1, beta2, epsilon, step, bias_correction, weight_decay, grad_averaging,
                 mode, global_grad_norm, max_grad_norm, use_nvlamb, found_inf, inv_scale):
        c = {"fn": "lamb", "chunk": int(chunk_size), "noop_in": arr(noop_flag), "lists": [[arr(t) for t in l] for l in tensor_lists],
             "lr": arr(lr), "beta1": float(beta1), "beta2": float(beta2), "eps": float(epsilon), "step": arr(step),
             "bias_correction": int(bias_correction), "weight_decay": float(weight_decay), "grad_averaging": int(grad_averaging),
             "mode": int(mode), "global_grad_norm": arr(global_grad_norm), "max_grad_norm": arr(max_grad_norm),
             "use_nvlamb": use_nvlamb, "found_inf": arr(found_inf), "inv_scale": arr(inv_scale)}
        orig_lamb(chunk_size, noop_flag, tensor_lists, lr, beta1, beta2, epsilon, step, bias_correction, weight_decay, grad_averaging,
                  mode, global_grad_norm, max_grad_norm, use_nvlamb, found_inf, inv_scale)
        c["out"] = [[arr(t) for t in l] for l in tensor_lists]
        calls.append(c)

    lamb_cpu_ext.multi_tensor_l2norm, lamb_cpu_ext.multi_tensor_lamb = rec_l2, rec_lamb
    try:
        FusedLAMBAMP = R.import_fused_lamb()                              # binds fused_lamb_CUDA to the (now recording) functions
        sched = R.import_bert().schedulers
        case = L.LAMB_GOLDEN_CASE
        params0, grads = L.lamb_golden_inputs(case)
        tp = {k: torch.nn.Parameter(torch.from_numpy(a.copy()).to(torch.float16 if half else torch.float32))
              for k, (a, half) in params0.items()}
        opt = FusedLAMBAMP([{"params": [tp[k] for k in names], "weight_decay": wd} for wd, names in case["groups"]], lr=case["lr"])
        opt.setup_fp32_params()
        lr_sched = sched.PolyWarmUpScheduler(opt, warmup=case["warmup"], total_steps=case["total_steps"], base_lr=case["lr"], device="cpu")
        scaler = torch.amp.GradScaler("cpu", init_scale=case["init_scale"], growth_interval=case["growth_interval"])
        marks = []
        for it, g in enumerate(grads):
            scaler.scale(torch.zeros(1))
            scale = float(scaler.get_scale())
            for k, p in tp.items():
                gs = (g[k] * np.float32(scale)).astype(np.float16 if p.dtype == torch.float16 else np.float32)
                if it in case["overflow_at"] and k == "w_c":
                    gs = gs.copy(); gs.reshape(-1)[5] = np.inf
                p.grad = torch.from_numpy(gs.copy())
            lr_sched.step()
            scaler.step(opt)
            scaler.update()
            opt.zero_grad(set_to_none=True)
            marks.append(len(calls))
    finally:
        lamb_cpu_ext.multi_tensor_l2norm, lamb_cpu_ext.multi_tensor_lamb = orig_l2, orig_lamb
    path = os.path.join(GOLD, "lamb_ref_trace.npz")
    L.save_call_trace(path, calls, marks)
    kinds = [c["fn"] for c in calls]
    print("lamb trace: %d calls (%d l2norm, %d lamb) over %d optimizer steps -> %s (%d bytes)"
          % (len(calls), kinds.count("l2norm"), kinds.count("lamb"), len(marks), path, os.path.getsize(path)))


def gen_bert_large():
    """One encoder layer at BERT-Large width (BASELINE configs[2] shapes) through the reference's module."""
    gen_bert(cfg_name="BERT_STEP_CONFIG_LARGE", out_name="bert_step_large1l.npz", last_layer=0)


def gen_bert_large24():
    """The 24-layer BERT-Large bench.py times (BASELINE configs[2]) through the reference's BertForPreTraining + criterion
    (run_pretraining.py:75-95,518-536, modeling.py:788-958), batch 4 x S 128, 20 masked positions, dropout 0, 2 LAMB steps:
    per-step losses, strided samples + norms of the first-step gradients of the embeddings / layers 0, 12, 23 / heads, and the
    16-bit STORAGE floors (loss and per-tensor gradient error of the storage-emulating oracle) next to them."""
    from oracle import bert_oracle as BO
    ref = R.import_bert()
    c = BO.BERT_STEP_CONFIG_LARGE24
    cfg = c["cfg"]
    conf = ref.modeling.BertConfig(cfg["vocab"], hidden_size=cfg["hidden"], num_hidden_layers=cfg["layers"],
                                   num_attention_heads=cfg["heads"], intermediate_size=cfg["intermediate"],
                                   hidden_act="gelu", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                                   max_position_embeddings=cfg["max_pos"], type_vocab_size=cfg["type_vocab"])
    conf.output_all_encoded_layers = False
    model = ref.modeling.BertForPreTraining(conf, sequence_output_is_dense=True)
    state0 = BO.seeded_state(cfg, c["seed"])
    assert sorted(n for n, _ in model.named_parameters()) == sorted(state0)
    model.train()
    ids, tt, mask, labels, nsp = BO.seeded_batch(cfg, c["seed"] + 1, c["batch"])
    assert int((labels != -1).sum()) == 20 * c["batch"]
    orc = BO.BertOracle(cfg, state0, c["lr"], c["warmup"], c["total_steps"])
    loss_fn = torch.nn.CrossEntropyLoss(ignore_index=-1)
    probes = BO.large24_probe_names(cfg)
    out, losses = {}, []
    for step in range(c["steps"]):
        model.load_state_dict({k: v.detach().clone() for k, v in orc.p.items()}, strict=False)
        model.zero_grad()
        scores, nsp_scores = model(ids, tt, mask, labels)
        flat = labels.view(-1)
        loss = loss_fn(scores.view(-1, cfg["vocab"]), flat[flat != -1]) + loss_fn(nsp_scores.view(-1, 2), nsp.view(-1))
        loss.backward()
        losses.append(float(loss.detach()))
        grads = {k: v.grad.detach().clone() for k, v in model.named_parameters()}
        if step == 0:
            # the restatement's forward / backward == the reference module's, on the model the bench times
            for v in orc.p.values():
                v.grad = None
            lo = orc.loss(ids, tt, mask, labels, nsp)
            lo.backward()
            assert abs(float(lo.detach()) - losses[0]) <= 1e-5 * losses[0], (float(lo), losses[0])
            for k in probes:
                assert torch.allclose(grads[k], orc.p[k].grad, rtol=2e-3, atol=2e-6), k
            for i, k in enumerate(probes):
                g = grads[k].reshape(-1).double().numpy()
                out["g%03d" % i] = g[BO.grad_sample_index(g.size)].astype(np.float32)
                out["gn%03d" % i] = np.float64(np.linalg.norm(g))
            ref_g = {k: grads[k].reshape(-1).double() for k in probes}
        orc.lamb_update({k: v.numpy() for k, v in grads.items()})
        print("large24 step", step, "loss", losses[-1], flush=True)
    out["losses"] = np.asarray(losses, np.float64)
    out["final_pooler_bias"] = orc.p["bert.pooler.dense_act.bias"].detach().numpy()
    out["final_query_row"] = orc.p["bert.encoder.layer.0.attention.self.query.weight"].detach().numpy()[:4]
    del model, grads
    for nm, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
        so = BO.BertOracle(cfg, state0, c["lr"], c["warmup"], c["total_steps"], storage_dtype=dt)
        ls, fl = [], []
        for step in range(c["steps"]):
            for v in so.p.values():
                v.grad = None
            lo = so.loss(ids, tt, mask, labels, nsp)
            lo.backward()
            ls.append(float(lo.detach()))
            if step == 0:
                for k in probes:
                    r = ref_g[k]
                    g = so.p[k].grad.reshape(-1).double()
                    idx = torch.from_numpy(BO.grad_sample_index(r.numel()))
                    fl.append(float((g[idx] - r[idx]).norm() / (r[idx].norm() + 1e-30)))
            so.lamb_update({k: v.grad.numpy() for k, v in so.p.items()})
        out["losses_%s_storage" % nm] = np.asarray(ls, np.float64)
        out["grad_floor_%s" % nm] = np.asarray(fl, np.float64)
        print("large24 storage", nm, ls, "worst gradient floor", max(fl), flush=True)
        del so
    np.savez_compressed(os.path.join(GOLD, "bert_step_large24.npz"), **out)
    print("bert_step_large24.npz losses", losses)


def gen_bert(cfg_name="BERT_STEP_CONFIG", out_name="bert_step.npz", last_layer=1):
    """Per-step losses of the REFERENCE's BertForPreTraining (eager CPU, dropout 0) + the oracle's LAMB."""
    from oracle import bert_oracle as BO
    ref = R.import_bert()
    c = getattr(BO, cfg_name)
    cfg = c["cfg"]
    conf = ref.modeling.BertConfig(cfg["vocab"], hidden_size=cfg["hidden"], num_hidden_layers=cfg["layers"],
                                   num_attention_heads=cfg["heads"], intermediate_size=cfg["intermediate"],
                                   hidden_act="gelu", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                                   max_position_embeddings=cfg["max_pos"], type_vocab_size=cfg["type_vocab"])
    conf.output_all_encoded_layers = False
    model = ref.modeling.BertForPreTraining(conf, sequence_output_is_dense=True)
    state0 = BO.seeded_state(cfg, c["seed"])
    names = [n for n, _ in model.named_parameters()]
    assert sorted(names) == sorted(state0), (set(names) ^ set(state0))
    model.load_state_dict({k: v.clone() for k, v in state0.items()}, strict=False)
    model.train()
    ids, tt, mask, labels, nsp = BO.seeded_batch(cfg, c["seed"] + 1, c["batch"])
    orc = BO.BertOracle(cfg, state0, c["lr"], c["warmup"], c["total_steps"])
    loss_fn = torch.nn.CrossEntropyLoss(ignore_index=-1)
    losses, ol = [], []
    for _ in range(c["steps"]):
        # reference forward/backward on the oracle's current weights (LAMB has no CPU reference implementation)
        model.load_state_dict({k: v.detach().clone() for k, v in orc.p.items()}, strict=False)
        model.zero_grad()
        scores, nsp_scores = model(ids, tt, mask, labels)
        flat = labels.view(-1)
        loss = loss_fn(scores.view(-1, cfg["vocab"]), flat[flat != -1]) + loss_fn(nsp_scores.view(-1, 2), nsp.view(-1))
        loss.backward()
        losses.append(float(loss.detach()))
        lo = orc.loss(ids, tt, mask, labels, nsp)
        ol.append(float(lo.detach()))
        for v in orc.p.values():
            v.grad = None
        lo.backward()
        # gradients of the restatement == gradients of the reference module
        for k, v in model.named_parameters():
            assert torch.allclose(v.grad, orc.p[k].grad, rtol=2e-3, atol=2e-6), k
        orc.lamb_update({k: v.grad.numpy() for k, v in model.named_parameters()})
    assert np.allclose(ol, losses, rtol=1e-5), (ol, losses)
    np.savez_compressed(os.path.join(GOLD, out_name), losses=np.asarray(losses, np.float64),
                        final_pooler_bias=orc.p["bert.pooler.dense_act.bias"].detach().numpy(),
                        final_ln_weight=orc.p["bert.encoder.layer.%d.output.LayerNorm.weight" % last_layer].detach().numpy(),
                        final_query_row=orc.p["bert.encoder.layer.0.attention.self.query.weight"].detach().numpy()[:4])
    print(out_name, "losses", losses)


def gen_floors(bert=True, dlrm_only=None):
    """Append the measured 16-bit STORAGE floors to the BERT / DLRM step fixtures: the step oracle re-run with every tensor
    the AMP path keeps in fp16 / bf16 rounded where it is produced (oracle/storage.py).  The fp32 oracle must first reproduce
    the fixture's reference losses (it is what was pinned against the reference module when the fixture was written)."""
    from oracle import bert_oracle as BO
    from oracle import dlrm_step_oracle as SO
    for cfg_name, out_name in ((("BERT_STEP_CONFIG", "bert_step.npz"), ("BERT_STEP_CONFIG_LARGE", "bert_step_large1l.npz")) if bert else ()):
        c = getattr(BO, cfg_name)
        path = os.path.join(GOLD, out_name)
        arrs = dict(np.load(path))
        state0 = BO.seeded_state(c["cfg"], c["seed"])
        batch = BO.seeded_batch(c["cfg"], c["seed"] + 1, c["batch"])
        for nm, dt in (("fp32", None), ("fp16", torch.float16), ("bf16", torch.bfloat16)):
            orc = BO.BertOracle(c["cfg"], state0, c["lr"], c["warmup"], c["total_steps"], storage_dtype=dt)
            ls = np.asarray([orc.step(*batch) for _ in range(c["steps"])], np.float64)
            if dt is None:
                assert np.allclose(ls, arrs["losses"], rtol=1e-5), (out_name, ls, arrs["losses"])
            else:
                arrs["losses_%s_storage" % nm] = ls
        np.savez_compressed(path, **arrs)
        print(out_name, "reference", arrs["losses"], "fp16 storage", arrs["losses_fp16_storage"], "bf16 storage", arrs["losses_bf16_storage"])
    for name, c in SO.DLRM_STEP_CONFIGS.items():
        if (dlrm_only is not None and name not in dlrm_only) or (dlrm_only is None and name == "mixed_paths"):
            continue
        path = os.path.join(GOLD, "dlrm_step_%s.npz" % name)
        arrs = dict(np.load(path))
        state0 = SO.seeded_dlrm_state(c["sizes"], c["dim"], c["bottom"], c["top"], c["num"], c["seed"])
        num, cat, click = SO.seeded_dlrm_batch(c["sizes"], c["num"], c["batch"], c["seed"] + 1000)
        for nm, dt in (("fp32", None), ("fp16", torch.float16), ("bf16", torch.bfloat16)):
            orc = SO.DlrmOracle(state0, c["sizes"], c["lr"], storage_dtype=dt)
            ls = np.asarray([orc.step(num, cat, click) for _ in range(c["steps"])], np.float64)
            if dt is None:
                assert np.allclose(ls, arrs["losses"], rtol=5e-5, atol=1e-6), (name, ls, arrs["losses"])
            else:
                arrs["losses_%s_storage" % nm] = ls
        np.savez_compressed(path, **arrs)
        print("dlrm_step", name, "reference", arrs["losses"], "fp16 storage", arrs["losses_fp16_storage"], "bf16 storage", arrs["losses_bf16_storage"])


def gen_waveglow():
    """Loss and every parameter gradient of the REFERENCE's WaveGlow + WaveGlowLoss on CPU (small flow / WN sizes, full mel and
    grouping geometry): pins oracle/waveglow_oracle.py."""
    from oracle import waveglow_oracle as WO
    ref = R.import_waveglow()
    c = WO.WAVEGLOW_CASE
    model = ref.model.WaveGlow(**c["cfg"])
    state = WO.seeded_state(c["cfg"], c["seed"])
    ref_shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert ref_shapes == WO.param_shapes(c["cfg"]), set(ref_shapes.items()) ^ set(WO.param_shapes(c["cfg"]).items())
    model.load_state_dict(state)
    model.train()
    mel, audio = WO.seeded_inputs(c)
    crit = ref.loss_function.WaveGlowLoss(sigma=c["sigma"])
    loss = crit(model((mel, audio)), audio)
    loss.backward()
    p = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    lo = WO.waveglow_loss(p, c["cfg"], mel, audio, c["sigma"])
    lo.backward()
    assert abs(float(lo) - float(loss)) <= 1e-6 * abs(float(loss)), (float(lo), float(loss))
    arrs = {"loss": np.asarray([float(loss)], np.float64)}
    for k, v in model.named_parameters():
        g, go = v.grad, p[k].grad
        assert torch.allclose(g, go, rtol=2e-4, atol=1e-7), (k, float((g - go).abs().max()))
        arrs["gnorm." + k] = np.asarray([float(g.norm())], np.float64)
    for k in ("upsample.weight", "convinv.1.conv.weight", "WN.2.in_layers.1.weight_g", "WN.3.end.bias"):
        arrs["grad." + k] = dict(model.named_parameters())[k].grad.numpy().reshape(-1)[:64]
    np.savez_compressed(os.path.join(GOLD, "waveglow_loss.npz"), **arrs)
    print("waveglow_loss.npz loss", float(loss), "params", len(list(model.named_parameters())))


def gen_tacotron2():
    """tacotron2_loss.npz (mask_padding = False, the default) and tacotron2_loss_masked.npz (--mask-padding, model.py:648-655)."""
    _gen_tacotron2(False, "tacotron2_loss.npz")
    _gen_tacotron2(True, "tacotron2_loss_masked.npz")
    _gen_tacotron2_eval()


def _gen_tacotron2_eval():
    """tacotron2_loss_eval.npz: the validation pass (train.py:273-318) -- the reference model in eval() mode with seeded BatchNorm
    running buffers; only the prenet's dropout is drawn."""
    import torch.nn.functional as TF
    from oracle import tacotron2_oracle as TO
    ref = R.import_tacotron2()
    c = TO.TACOTRON2_CASE
    cfg = c["cfg"]
    model = ref.model.Tacotron2(mask_padding=False, max_decoder_steps=2000, gate_threshold=0.5, decoder_no_early_stopping=False, **cfg)
    state = dict(TO.seeded_state(cfg, c["seed"]))
    state.update(TO.seeded_running_stats(cfg, c["seed"]))
    model.load_state_dict(state, strict=False)
    model.eval()
    text, tl, mel, gate, ml = TO.seeded_batch(c)
    stream = TO.MaskStream(c["seed"] + 2)
    real_dropout = TF.dropout
    TF.dropout = lambda x, p=0.5, training=True, inplace=False: stream(x, p) if training else x
    try:
        with torch.no_grad():
            out = model((text, tl, mel, int(tl.max()), ml))
            loss = ref.loss_function.Tacotron2Loss()(out, (mel, gate))
    finally:
        TF.dropout = real_dropout
    stream2 = TO.MaskStream(c["seed"] + 2)
    with torch.no_grad():
        lo, (mo, mp, go, al) = TO.tacotron2_loss(state, cfg, text, tl, mel, gate, stream2, training=False)
    assert stream.calls == stream2.calls == 2, (stream.calls, stream2.calls)
    assert abs(float(lo) - float(loss)) <= 2e-6 * abs(float(loss)), (float(lo), float(loss))
    assert torch.allclose(al, out[3], atol=1e-6) and torch.allclose(mp, out[1], atol=5e-5)
    np.savez_compressed(os.path.join(GOLD, "tacotron2_loss_eval.npz"), loss=np.asarray([float(loss)], np.float64),
                        dropout_calls=np.asarray([stream.calls], np.int64), alignment_last=out[3][:, -1].numpy(),
                        mel_post_slice=out[1][:, :4, :].numpy())
    print("tacotron2_loss_eval.npz loss", float(loss))


def _gen_tacotron2(mask_padding, fname):
    """Loss and every parameter gradient of the REFERENCE's Tacotron2 + Tacotron2Loss on CPU (training mode, small widths, the
    full structure), with F.dropout bound to oracle.tacotron2_oracle.MaskStream so that reference and oracle draw the same masks:
    pins oracle/tacotron2_oracle.py (the Tacotron2 half of SURVEY.md 8 row f1)."""
    import torch.nn.functional as TF
    from oracle import tacotron2_oracle as TO
    ref = R.import_tacotron2()
    c = TO.TACOTRON2_CASE
    cfg = c["cfg"]
    model = ref.model.Tacotron2(mask_padding=mask_padding, max_decoder_steps=2000, gate_threshold=0.5, decoder_no_early_stopping=False, **cfg)
    state = TO.seeded_state(cfg, c["seed"])
    trainable = {k: tuple(v.shape) for k, v in model.named_parameters()}
    assert trainable == TO.param_shapes(cfg), set(trainable.items()) ^ set(TO.param_shapes(cfg).items())
    model.load_state_dict(state, strict=False)                       # BatchNorm running buffers keep their defaults
    model.train()
    text, tl, mel, gate, ml = TO.seeded_batch(c)
    if mask_padding:
        # the reference's parse_output fills mel_outputs IN PLACE after the postnet's first convolution has saved it for its
        # weight gradient: loss.backward() of the unmodified reference raises ("modified by an inplace operation"); its forward
        # and loss are fine.  To pin the gradients of the masking it INTENDS, its own parse_output runs on clones.
        try:
            ref.loss_function.Tacotron2Loss()(model((text, tl, mel, int(tl.max()), ml)), (mel, gate)).backward()
            raise AssertionError("expected the reference's --mask-padding backward to raise")
        except RuntimeError as e:
            assert "inplace" in str(e), e
        model.zero_grad()
        parse = model.parse_output
        model.parse_output = lambda outputs, lengths: parse([o.clone() for o in outputs], lengths)
    stream = TO.MaskStream(c["seed"] + 2)
    real_dropout = TF.dropout
    TF.dropout = lambda x, p=0.5, training=True, inplace=False: stream(x, p) if training else x
    try:
        out = model((text, tl, mel, int(tl.max()), ml))
        loss = ref.loss_function.Tacotron2Loss()(out, (mel, gate))
        loss.backward()
    finally:
        TF.dropout = real_dropout
    p = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    stream2 = TO.MaskStream(c["seed"] + 2)
    lo, (mo, mp, go, al) = TO.tacotron2_loss(p, cfg, text, tl, mel, gate, stream2, output_lengths=ml if mask_padding else None)
    lo.backward()
    assert stream.calls == stream2.calls, (stream.calls, stream2.calls)
    assert abs(float(lo) - float(loss)) <= 2e-6 * abs(float(loss)), (float(lo), float(loss))
    assert torch.allclose(al, out[3], atol=1e-6) and torch.allclose(mo, out[0], atol=2e-5)
    arrs = {"loss": np.asarray([float(loss)], np.float64), "dropout_calls": np.asarray([stream.calls], np.int64),
            "alignment_last": out[3][:, -1].detach().numpy()}
    for k, v in model.named_parameters():
        g, go_ = v.grad, p[k].grad
        assert torch.allclose(g, go_, rtol=5e-4, atol=max(2e-7, 2e-6 * float(g.abs().max()))), (k, float((g - go_).abs().max()), float(g.abs().max()))
        arrs["gnorm." + k] = np.asarray([float(g.norm())], np.float64)
    for k in ("embedding.weight", "encoder.lstm.weight_hh_l0_reverse", "decoder.attention_rnn.weight_ih",
              "decoder.attention_layer.location_layer.location_conv.conv.weight", "decoder.prenet.layers.0.linear_layer.weight",
              "postnet.convolutions.2.0.conv.weight", "decoder.gate_layer.linear_layer.weight"):
        arrs["grad." + k] = dict(model.named_parameters())[k].grad.numpy().reshape(-1)[:64]
    np.savez_compressed(os.path.join(GOLD, fname), **arrs)
    print(fname, "loss", float(loss), "params", len(trainable), "dropout calls", stream.calls)


FRONTEND_SENTENCES = ["Printing, in the only sense with which we are at present concerned, differs from most if not from all the arts.",
                      "Turn left on {HH AW1 S S T AH0 N} Street; Mr. Smith and Dr. Jones agreed -- didn't they?",
                      "  Odd   spacing\tand (parentheses), UPPER case: ok!  ", "St. Col. Ltd. esq. capt. ~ _ @ # weird*chars"]


def gen_tacotron2_frontend():
    """tests/golden/tacotron2_frontend.npz: the reference's own text_to_sequence (english / basic cleaners) on ASCII, digit-free
    sentences, TextMelCollate on a seeded ragged batch (n_frames_per_step 1 and 3), and |STFT| of a seeded waveform through
    tacotron2_common.stft.STFT (filter 1024, hop 256, Hann 1024): pins deeplearningexamples_amd/tacotron2/{text,data_function,
    audio}.py."""
    ref = R.import_tacotron2_frontend()
    arrs = {}
    for ci, cleaners in enumerate((["english_cleaners"], ["basic_cleaners"])):
        for si, sent in enumerate(FRONTEND_SENTENCES):
            arrs["seq.%d.%d" % (ci, si)] = np.asarray(ref.text.text_to_sequence(sent, cleaners), np.int64)
    arrs["symbols"] = np.asarray(ref.text.symbols)
    rng = np.random.default_rng(41)
    lens, mels = [7, 12, 3, 12, 9], [19, 31, 8, 25, 31]
    batch = [(torch.from_numpy(rng.integers(1, 148, l).astype(np.int32)), torch.from_numpy(rng.standard_normal((5, m)).astype(np.float32)), l + 2)
             for l, m in zip(lens, mels)]
    for nf in (1, 3):
        out = ref.data_function.TextMelCollate(nf)(batch)
        for k, t in zip(("text", "input_lengths", "mel", "gate", "output_lengths", "len_x"), out):
            arrs["collate%d.%s" % (nf, k)] = t.numpy()
    wav = (rng.random(5000) * 2 - 1).astype(np.float32)
    mag, _ = ref.stft.STFT(1024, 256, 1024).transform(torch.from_numpy(wav)[None])
    arrs["wav"] = wav
    arrs["stft_mag"] = mag[0].numpy()[::8]                              # every 8th frequency bin of all 20 frames
    np.savez_compressed(os.path.join(GOLD, "tacotron2_frontend.npz"), **arrs)
    print("tacotron2_frontend.npz", {k: v.shape for k, v in arrs.items() if not k.startswith("seq")})


if __name__ == "__main__":
    which = sys.argv[1:] or ["dlrm", "dlrm_step", "rn50", "lamb", "bert", "floors", "waveglow", "tacotron2", "tacotron2_frontend"]
    os.makedirs(GOLD, exist_ok=True)
    if not R.have_reference():
        sys.exit("reference not mounted; fixtures are generated in the build container only")
    for w in which:
        fn = globals().get("gen_" + w)
        if fn is None:
            print("skip", w)
            continue
        fn()
