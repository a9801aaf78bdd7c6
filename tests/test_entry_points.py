"""Entry points: CLI parsing / schedules / logger on CPU, and a short run of each CLI on the GPU."""
import json
import os

import numpy as np
import pytest


def test_dllogger_format(tmp_path):
    from deeplearningexamples_amd.utils import dllogger
    f = tmp_path / "log.json"
    dllogger.init([dllogger.JSONStreamBackend(dllogger.Verbosity.VERBOSE, str(f))])
    dllogger.metadata("train.loss", {"unit": None})
    dllogger.log(step=(0, 3), data={"train.loss": 1.5, "train.total_ips": 100.0})
    dllogger.flush()
    lines = open(f).read().strip().splitlines()
    assert all(l.startswith("DLLL ") for l in lines)
    rec = json.loads(lines[-1][5:])
    assert rec["type"] == "LOG" and rec["step"] == [0, 3] and rec["data"]["train.loss"] == 1.5


def test_rn50_lr_policies_match_reference_formulas():
    # optimizers.py:82-130 evaluated by hand
    from deeplearningexamples_amd.convnets.engine import lr_cosine_policy, lr_linear_policy, lr_step_policy
    cos = lr_cosine_policy(2.048, 8, 250)
    assert cos(0, 0) == pytest.approx(2.048 / 8) and cos(0, 7) == pytest.approx(2.048)
    assert cos(0, 129) == pytest.approx(0.5 * (1 + np.cos(np.pi * 121 / 242)) * 2.048)
    step = lr_step_policy(0.1, [30, 60, 80], 0.1, 5)
    assert step(0, 2) == pytest.approx(0.06) and step(0, 65) == pytest.approx(0.001)
    lin = lr_linear_policy(1.0, 2, 12)
    assert lin(0, 7) == pytest.approx(0.5)


def test_cli_parsers():
    from deeplearningexamples_amd.convnets.main import add_parser_arguments
    import argparse
    a = add_parser_arguments(argparse.ArgumentParser()).parse_args(
        ["--arch", "resnet50", "-b", "64", "--amp", "--label-smoothing", "0.1", "--lr-schedule", "cosine", "--warmup", "8"])
    assert a.batch_size == 64 and a.amp and a.lr_schedule == "cosine"
    from deeplearningexamples_amd.bert.run_pretraining import parse_arguments
    b = parse_arguments(["--train_batch_size", "64", "--gradient_accumulation_steps", "4", "--max_steps", "10", "--bf16"])
    assert b.steps_this_run == 10 and b.train_batch_size // b.gradient_accumulation_steps == 16
    with pytest.raises(ValueError):
        parse_arguments(["--train_batch_size", "10", "--gradient_accumulation_steps", "4"])
    from deeplearningexamples_amd.dlrm.main import parse_flags
    f = parse_flags(["--top_mlp_sizes", "64,32,1", "--synthetic_dataset_table_sizes", "100,200", "--amp"])
    assert f.top_mlp_sizes == [64, 32, 1] and f.synthetic_dataset_table_sizes == [100, 200] and f.amp


@pytest.mark.gpu
def test_entry_points_run_on_gpu(cuda, tmp_path):
    from deeplearningexamples_amd.convnets import main as rn
    from deeplearningexamples_amd.bert import run_pretraining as bp
    from deeplearningexamples_amd.dlrm import main as dl
    rn.main(["--batch-size", "8", "--image-size", "64", "--epochs", "1", "--prof", "3", "--amp", "--label-smoothing", "0.1",
             "--lr", "0.01", "--lr-schedule", "cosine", "--warmup", "1", "--workspace", str(tmp_path), "--print-freq", "1"])
    recs = [json.loads(l[5:]) for l in open(tmp_path / "experiment_raport.json")]
    assert any("train.loss" in r.get("data", {}) for r in recs) and "train.total_ips" in recs[-1]["data"]
    cfg = tmp_path / "tiny.json"
    cfg.write_text(json.dumps(dict(vocab_size=1000, hidden_size=256, num_attention_heads=4, num_hidden_layers=2,
                                   intermediate_size=1024, max_position_embeddings=512, type_vocab_size=2)))
    t = bp.main(["--config_file", str(cfg), "--train_batch_size", "8", "--gradient_accumulation_steps", "2", "--max_steps", "3",
                 "--json-summary", str(tmp_path / "bert.json"), "--bf16"])
    recs = [json.loads(l[5:]) for l in open(tmp_path / "bert.json")]
    assert "training_sequences_per_second" in recs[-1]["data"] and t.opt_steps == 3
    dl.main(["--batch_size", "2048", "--synthetic_dataset_table_sizes", "1000,50,7,20000", "--max_steps", "5", "--amp",
             "--log_path", str(tmp_path / "dlrm.json"), "--print_freq", "2", "--synthetic_dataset_num_entries", "16384"])
    recs = [json.loads(l[5:]) for l in open(tmp_path / "dlrm.json")]
    assert "average_train_throughput" in recs[-1]["data"]


def test_bert_lr_schedule_matches_reference_scheduler():
    """bert.engine.poly_warmup_lr == the oracle's restatement == (when the reference tree is present) the reference's
    own PolyWarmUpScheduler stepped on a dummy optimizer (LanguageModeling/BERT/schedulers.py:109-136)."""
    import os
    import sys
    import torch
    from deeplearningexamples_amd.bert.engine import poly_warmup_lr
    from oracle import bert_oracle as BO
    base, warm, total = 6e-3, 0.2843, 40
    ours = [poly_warmup_lr(s, base, warm, total) for s in range(1, total)]
    orc = [BO.poly_warmup_lr(s, base, warm, total) for s in range(1, total)]
    assert max(abs(a - b) for a, b in zip(ours, orc)) < 1e-12
    assert poly_warmup_lr(total + 5, base, warm, total) == 0.0
    ref_dir = "/root/reference/PyTorch/LanguageModeling/BERT"
    if os.path.isdir(ref_dir):
        sys.path.insert(0, ref_dir)
        try:
            import schedulers as RS
        finally:
            sys.path.remove(ref_dir)
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=base)
        sch = RS.PolyWarmUpScheduler(opt, warmup=warm, total_steps=total, base_lr=base, device="cpu")
        got = []
        for k in range(0, 11):            # FusedLAMBAMP keeps the completed-step count in param_group["step"]
            opt.param_groups[0]["step"] = torch.tensor(float(k))
            sch.step()                    # run_pretraining.py:529 steps the scheduler BEFORE the optimizer
            got.append(float(opt.param_groups[0]["lr"]))
        assert max(abs(a - b) for a, b in zip(got, ours[:11])) < 1e-9
