"""Checkpoint formats against the reference's OWN classes on CPU: a file written by
deeplearningexamples_amd.utils.checkpoint loads into the reference (model + optimizer), a reference-written one is read
back by name.  Needs the reference tree (build container); on a box without it only the self-contained checks run."""
import os
import sys
import types

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import _ref_import as R  # noqa: E402
from deeplearningexamples_amd.utils import checkpoint as CK  # noqa: E402

needs_ref = pytest.mark.skipif(not R.have_reference(), reason="reference tree not mounted")


@pytest.fixture(autouse=True)
def _restore_import_state():
    """oracle/_ref_import stubs third-party modules (apex, dllogger, absl, dlrm.cuda_ext ...) and patches three torch
    attributes to run the reference on CPU: undo all of it so that later test modules see the real shims."""
    mods, path = dict(sys.modules), list(sys.path)
    saved = (torch.cuda.current_device, torch.Tensor.cuda, torch.cuda.synchronize)
    yield
    torch.cuda.current_device, torch.Tensor.cuda, torch.cuda.synchronize = saved
    for k in list(sys.modules):
        if k not in mods and not k.startswith(("torch", "numpy", "scipy", "_pytest", "pytest")):
            del sys.modules[k]          # stubs and reference modules; torch's own lazily imported parts stay
    for k, v in mods.items():
        if sys.modules.get(k) is not v:
            sys.modules[k] = v
    sys.path[:] = path


@needs_ref
def test_rn50_checkpoint_loads_into_reference_and_back(tmp_path):
    from deeplearningexamples_amd.convnets.resnet import ResNet50
    ref = R.import_convnets()
    torch.manual_seed(0)
    ours = ResNet50(device="cpu")
    rmodel = ref.models.resnet50(pretrained=False)
    assert [n for n, _ in ours.named_parameters()] == [n for n, _ in rmodel.named_parameters()]
    lr, mom, wd = 0.256, 0.875, 3.0517578125e-05
    # (1) the reference takes a step; its optimizer state is read back BY NAME
    ropt = ref.optimizers.get_sgd_optimizer(list(rmodel.named_parameters()), lr, mom, wd)
    g = torch.Generator().manual_seed(1)
    for p in rmodel.parameters():
        p.grad = torch.randn(p.shape, generator=g) * 0.01
    ropt.step()
    got = CK.rn50_momentum_from_optimizer_state(ours.named_parameters(), ropt.state_dict())
    rnamed = dict(rmodel.named_parameters())
    assert set(got) == set(rnamed)
    for n, buf in got.items():
        assert torch.equal(buf, ropt.state[rnamed[n]]["momentum_buffer"]), n
    # (2) a checkpoint written here: the reference's resume path (main.py:424-431, training.py:194-202) accepts it
    buffers = {n: torch.randn(p.shape, generator=g).contiguous(memory_format=torch.channels_last) if p.dim() == 4
               else torch.randn(p.shape, generator=g) for n, p in ours.named_parameters()}
    state = {"epoch": 3, "best_prec1": 12.5, "state_dict": ours.state_dict(),
             "optimizer": CK.rn50_optimizer_state(ours.named_parameters(), buffers, lr, mom, wd)}
    ck = CK.Checkpointer("checkpoint.pth.tar", checkpoint_dir=str(tmp_path), keep_last_n=1)
    ck.save_checkpoint(state, is_best=True, filename="checkpoint_0002.pth.tar")
    ck.save_checkpoint(state, is_best=False, filename="checkpoint_0003.pth.tar")
    assert sorted(os.listdir(tmp_path)) == ["checkpoint.pth.tar", "checkpoint_0003.pth.tar", "model_best.pth.tar"]
    loaded = torch.load(tmp_path / "checkpoint.pth.tar", weights_only=False)
    assert loaded["epoch"] == 3 and loaded["best_prec1"] == 12.5
    res = rmodel.load_state_dict(loaded["state_dict"], strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    for (n, a), (_, b) in zip(ours.state_dict().items(), rmodel.state_dict().items()):
        assert torch.equal(a, b), n
    ropt2 = ref.optimizers.get_sgd_optimizer(list(rmodel.named_parameters()), lr, mom, wd)
    ropt2.load_state_dict(loaded["optimizer"])
    for n, p in rmodel.named_parameters():
        assert torch.equal(ropt2.state[p]["momentum_buffer"], buffers[n]), n
    for ga, gb in zip(ropt2.param_groups, ropt.param_groups):
        assert {k: v for k, v in ga.items() if k != "params"} == {k: v for k, v in gb.items() if k != "params"}
        assert len(ga["params"]) == len(gb["params"])
    # before the first step torch.optim.SGD has no state at all
    assert CK.rn50_optimizer_state(ours.named_parameters(), None, lr, mom, wd)["state"] == {}


@needs_ref
def test_bert_checkpoint_loads_into_reference_lamb_and_back(tmp_path):
    from oracle import bert_oracle as BO
    from deeplearningexamples_amd.bert.model import BertForPreTraining
    FusedLAMBAMP = R.import_fused_lamb()
    refb = R.import_bert()
    cfg = BO.BERT_TINY
    torch.manual_seed(0)
    ours = BertForPreTraining(cfg, device="cpu")
    rcfg = refb.modeling.BertConfig(vocab_size_or_config_json_file=cfg["vocab"], hidden_size=cfg["hidden"],
                                    num_hidden_layers=cfg["layers"], num_attention_heads=cfg["heads"],
                                    intermediate_size=cfg["intermediate"], max_position_embeddings=cfg["max_pos"],
                                    type_vocab_size=cfg["type_vocab"])
    rmodel = refb.modeling.BertForPreTraining(rcfg, sequence_output_is_dense=True)
    names = [n for n, _ in ours.named_parameters()]
    assert names == [n for n, _ in rmodel.named_parameters()]
    # model: the tied decoder weight appears under both names in the reference's state_dict
    res = rmodel.load_state_dict(CK.bert_model_state(ours), strict=True)
    assert not res.missing_keys and not res.unexpected_keys

    def ref_opt():
        named = list(rmodel.named_parameters())
        nd = ["bias", "gamma", "beta", "LayerNorm"]                            # run_pretraining.py:341-349
        return FusedLAMBAMP([{"params": [p for n, p in named if not any(k in n for k in nd)], "weight_decay": 0.01},
                             {"params": [p for n, p in named if any(k in n for k in nd)], "weight_decay": 0.0}], lr=6e-3)
    g = torch.Generator().manual_seed(2)
    m = {n: torch.randn(p.shape, generator=g) for n, p in ours.named_parameters()}
    v = {n: torch.rand(p.shape, generator=g) for n, p in ours.named_parameters()}
    ours_state = CK.lamb_optimizer_state(names, m, v, torch.tensor(0.004), torch.tensor([17], dtype=torch.int32))
    torch.save({"model": CK.bert_model_state(ours), "optimizer": ours_state,
                "grad_scaler": CK.grad_scaler_state(1024.0, 5), "epoch": 1}, tmp_path / "ckpt_17.pt")
    ck = torch.load(tmp_path / "ckpt_17.pt", weights_only=False)
    opt = ref_opt()
    opt.load_state_dict(ck["optimizer"])                                       # run_pretraining.py:352-366
    rnamed = dict(rmodel.named_parameters())
    for n in names:
        assert torch.equal(opt.state[rnamed[n]]["exp_avg"], m[n]) and torch.equal(opt.state[rnamed[n]]["exp_avg_sq"], v[n]), n
    assert int(opt.param_groups[0]["step"].item()) == 17 and abs(float(opt.param_groups[1]["lr"]) - 0.004) < 1e-9
    assert [grp["weight_decay"] for grp in opt.param_groups] == [0.01, 0.0]
    sc = torch.amp.GradScaler("cpu")
    sc.load_state_dict(ck["grad_scaler"])
    assert sc.get_scale() == 1024.0
    # reverse: the reference's own state_dict is read back by name; hyper-parameter keys agree with ours
    m2, v2, step, lr = CK.lamb_moments_from_state(names, opt.state_dict())
    assert step == 17 and abs(lr - 0.004) < 1e-9 and all(torch.equal(m2[n], m[n]) and torch.equal(v2[n], v[n]) for n in names)
    ref_keys = set(ref_opt().state_dict()["param_groups"][0])
    assert set(ours_state["param_groups"][0]) == ref_keys, (set(ours_state["param_groups"][0]) ^ ref_keys)


def _fake_dlrm(sizes, dim, gen):
    def mlp(dims):
        return types.SimpleNamespace(
            weights=[torch.nn.Parameter(torch.randn(o, i, generator=gen)) for i, o in zip(dims[:-1], dims[1:])],
            biases=[torch.nn.Parameter(torch.randn(o, generator=gen)) for o in dims[1:]])
    emb = types.SimpleNamespace(weights=[torch.randn(n, dim, generator=gen) for n in sizes])
    emb.load_weights = lambda ws: [d.copy_(s) for d, s in zip(emb.weights, ws)]
    for m_ in ():
        pass
    bottom = types.SimpleNamespace(embeddings=emb, mlp=mlp([13, 32, dim]))
    top = types.SimpleNamespace(mlp=mlp([40, 24, 8]), out=torch.nn.Linear(8, 1))
    for part in (bottom.mlp, top.mlp):
        part.load_state = (lambda p: lambda ws, bs: [a.data.copy_(b) for a, b in zip(p.weights + p.biases, list(ws) + list(bs))])(part)
    return types.SimpleNamespace(bottom_model=bottom, top_model=top)


def test_dlrm_checkpoint_directory_round_trip(tmp_path):
    gen = torch.Generator().manual_seed(4)
    sizes, dim = [7, 3, 11], 8
    a, b = _fake_dlrm(sizes, dim, gen), _fake_dlrm(sizes, dim, gen)
    mapping = {"bottom_mlp": 0, "embedding": [[2, 0, 1]], "vectors_per_gpu": [4]}
    w = CK.make_distributed_checkpoint_writer(mapping, 0, True, {"embedding_dim": dim})
    w.save_checkpoint(a, str(tmp_path), epoch=1, step=99)
    names = sorted(os.listdir(tmp_path))
    assert names == sorted(["bottom_model.embeddings.%d.bin" % i for i in range(3)] + ["embeddings.%d.meta.pt" % i for i in range(3)] +
                           ["bottom_model.mlp.pt", "top_model.mlp.pt", "top_model.out.pt", "metadata.pt"])
    # table i of the file set is the i-th ORIGINAL feature: the rank's tables are listed in its mapping order
    raw = np.frombuffer(open(tmp_path / "bottom_model.embeddings.2.bin", "rb").read(), dtype=np.float32)
    assert np.array_equal(raw, a.bottom_model.embeddings.weights[0].numpy().reshape(-1))
    CK.make_distributed_checkpoint_loader(mapping, 0).load_checkpoint(b, str(tmp_path))
    for x, y in zip(a.bottom_model.embeddings.weights, b.bottom_model.embeddings.weights):
        assert torch.equal(x, y)
    for pa, pb in ((a.bottom_model.mlp, b.bottom_model.mlp), (a.top_model.mlp, b.top_model.mlp)):
        assert all(torch.equal(x, y) for x, y in zip(pa.weights + pa.biases, pb.weights + pb.biases))
    assert torch.equal(a.top_model.out.weight, b.top_model.out.weight)
    meta = torch.load(tmp_path / "metadata.pt", weights_only=False)
    assert meta["data"] == {"device_mapping": mapping, "epoch": 1, "step": 99} and meta["config"] == {"embedding_dim": dim}


@needs_ref
def test_dlrm_checkpoint_interchangeable_with_reference_writer(tmp_path):
    R.import_dlrm()
    from dlrm.utils.checkpointing import distributed as refck
    gen = torch.Generator().manual_seed(5)
    sizes, dim = [5, 9], 4
    a, b, c = _fake_dlrm(sizes, dim, gen), _fake_dlrm(sizes, dim, gen), _fake_dlrm(sizes, dim, gen)
    mapping = {"bottom_mlp": 0, "embedding": [[1, 0]], "vectors_per_gpu": [3]}
    d_ref, d_own = tmp_path / "ref", tmp_path / "own"
    refck.make_distributed_checkpoint_writer(mapping, 0, True, {"k": 1}).save_checkpoint(a, str(d_ref), 2, 7)
    CK.make_distributed_checkpoint_writer(mapping, 0, True, {"k": 1}).save_checkpoint(a, str(d_own), 2, 7)
    assert sorted(os.listdir(d_ref)) == sorted(os.listdir(d_own))
    for f in os.listdir(d_ref):
        if f.endswith(".bin"):
            assert open(d_ref / f, "rb").read() == open(d_own / f, "rb").read(), f
    CK.make_distributed_checkpoint_loader(mapping, 0).load_checkpoint(b, str(d_ref))        # reference-written -> here
    refck.make_distributed_checkpoint_loader(mapping, 0).load_checkpoint(c, str(d_own))     # written here -> reference
    for other in (b, c):
        assert all(torch.equal(x, y) for x, y in zip(a.bottom_model.embeddings.weights, other.bottom_model.embeddings.weights))
        assert all(torch.equal(x, y) for x, y in zip(a.top_model.mlp.weights, other.top_model.mlp.weights))
        assert torch.equal(a.top_model.out.bias, other.top_model.out.bias)
