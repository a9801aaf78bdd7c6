"""The RN50 and BERT step oracles against the golden fixtures the REFERENCE's own modules produced on CPU
(tests/golden/{rn50_step,bert_step}.npz, oracle/make_golden.py gen_rn50 / gen_bert).  CPU only."""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def test_bert_oracle_reproduces_reference_losses():
    from oracle import bert_oracle as BO
    c = BO.BERT_STEP_CONFIG
    gold = np.load(os.path.join(HERE, "golden", "bert_step.npz"))
    orc = BO.BertOracle(c["cfg"], BO.seeded_state(c["cfg"], c["seed"]), c["lr"], c["warmup"], c["total_steps"])
    batch = BO.seeded_batch(c["cfg"], c["seed"] + 1, c["batch"])
    losses = [orc.step(*batch) for _ in range(c["steps"])]
    np.testing.assert_allclose(losses, gold["losses"], rtol=2e-5)
    np.testing.assert_allclose(orc.p["bert.pooler.dense_act.bias"].detach().numpy(), gold["final_pooler_bias"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(orc.p["bert.encoder.layer.0.attention.self.query.weight"].detach().numpy()[:4],
                               gold["final_query_row"], rtol=1e-3, atol=1e-5)
    # dropout hooks: all-keep masks with p = 0 leave the forward unchanged
    h, nh, s, b = c["cfg"]["hidden"], c["cfg"]["heads"], c["cfg"]["seq"], c["batch"]
    masks = {"emb": torch.ones(b, s, h, dtype=torch.bool)}
    for l in range(c["cfg"]["layers"]):
        masks.update({"attn%d" % l: torch.ones(b, nh, s, s, dtype=torch.bool),
                      "out1_%d" % l: torch.ones(b, s, h, dtype=torch.bool), "out2_%d" % l: torch.ones(b, s, h, dtype=torch.bool)})
    l0 = float(orc.loss(*batch))
    l1 = float(orc.loss(*batch, masks=masks, p_hidden=0.0, p_attn=0.0))
    assert l0 == l1


def test_rn50_oracle_reproduces_reference_losses():
    from oracle import resnet_oracle as RO
    c = RO.RN50_STEP_CONFIG
    gold = np.load(os.path.join(HERE, "golden", "rn50_step.npz"))
    torch.set_num_threads(max(torch.get_num_threads(), 4))
    orc = RO.ResNet50Oracle(RO.seeded_state(c["seed"]), c["lr"])
    x, y = RO.seeded_batch(c["seed"] + 100, c["batch"], c["size"])
    losses = [orc.step(x, y) for _ in range(2)]             # the fixture's own bar: first two steps to 1e-4
    np.testing.assert_allclose(losses, gold["losses"][:2], rtol=1e-4)
    np.testing.assert_allclose(losses, gold["oracle_losses"][:2], rtol=1e-5)


def test_storage_floors_in_fixtures_are_what_the_oracles_measure():
    """The 16-bit storage floors the GPU loss-parity bars add to 1e-3 (fixture arrays losses_<dtype>_storage): the BERT and
    DLRM tiny cases recomputed here; all floors are far below 1e-3 (the bars are dominated by north_star's figure)."""
    from oracle import bert_oracle as BO
    from oracle import dlrm_step_oracle as SO
    c = BO.BERT_STEP_CONFIG
    gold = np.load(os.path.join(HERE, "golden", "bert_step.npz"))
    batch = BO.seeded_batch(c["cfg"], c["seed"] + 1, c["batch"])
    orc = BO.BertOracle(c["cfg"], BO.seeded_state(c["cfg"], c["seed"]), c["lr"], c["warmup"], c["total_steps"],
                        storage_dtype=torch.bfloat16)
    losses = [orc.step(*batch) for _ in range(2)]
    np.testing.assert_allclose(losses, gold["losses_bf16_storage"][:2], rtol=2e-5)
    d = SO.DLRM_STEP_CONFIGS["tiny"]
    gd = np.load(os.path.join(HERE, "golden", "dlrm_step_tiny.npz"))
    st = SO.seeded_dlrm_state(d["sizes"], d["dim"], d["bottom"], d["top"], d["num"], d["seed"])
    num, cat, click = SO.seeded_dlrm_batch(d["sizes"], d["num"], d["batch"], d["seed"] + 1000)
    for nm, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
        o = SO.DlrmOracle(st, d["sizes"], d["lr"], storage_dtype=dt)
        ls = [o.step(num, cat, click) for _ in range(d["steps"])]
        np.testing.assert_allclose(ls, gd["losses_%s_storage" % nm], rtol=2e-5)
    for f, keys in (("bert_step.npz", 1), ("bert_step_large1l.npz", 1), ("dlrm_step_tiny.npz", 1), ("dlrm_step_criteo_shape.npz", 1), ("dlrm_step_mixed_paths.npz", 1)):
        g = np.load(os.path.join(HERE, "golden", f))
        for nm in ("fp16", "bf16"):
            floor = np.abs(g["losses_%s_storage" % nm] - g["losses"]) / g["losses"]
            assert floor.max() < 2e-4, (f, nm, floor)
