"""The reference's own DLRM test strategy (Recommendation/DLRM/tests/test_all_configs.sh, test_fspecs.sh,
test_custom_dot.sh, test_with_opts.sh) against the drop-in entry point: for every feature spec fixture, a synthetic dataset is
written in its layout (prepare_synthetic_dataset.py's job, tests/test_dlrm_data.synth_from_fixture) and `main --mode train
--dataset DIR <options>` runs over the option matrix of the scripts, WRITTEN AS THE SCRIPTS WRITE IT (absl syntax:
--optimized_mlp=True --cuda_graphs=True --interaction_op=dot --embedding_type=joint_sparse --amp=False).  The scripts' pass
criterion is exit code 0; here the logged losses must also be finite and near ln 2 (random labels), and the validation pass must
produce an AUC."""
import json
import math
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_dlrm_data import FSPECS, synth_from_fixture  # noqa: E402

pytestmark = pytest.mark.gpu
BATCH, STEPS = 2048, 6


def _train(tmp_path, data_dir, tag, opts):
    from deeplearningexamples_amd.dlrm import main as dl
    log = str(tmp_path / ("%s.json" % tag))
    dl.main(["--mode", "train", "--dataset", data_dir, "--batch_size=%d" % BATCH, "--test_batch_size=%d" % BATCH,
             "--print_freq=2", "--test_freq=4", "--log_path", log] + opts)
    recs = [json.loads(l[5:]) for l in open(log)]
    losses = [r["data"]["loss"] for r in recs if isinstance(r.get("data"), dict) and "loss" in r["data"]]
    aucs = [r["data"]["auc"] for r in recs if isinstance(r.get("data"), dict) and "auc" in r["data"]]
    assert len(losses) >= 2 and all(math.isfinite(x) and 0.3 < x < 3.0 for x in losses), (tag, losses)
    assert aucs and all(0.0 <= a <= 1.0 for a in aucs), (tag, aucs)
    assert "average_train_throughput" in recs[-1]["data"]
    return losses


def test_all_configs_on_the_default_spec(cuda, tmp_path):
    """test_all_configs.sh:17-27 on feature_specs/default.yaml: {optimized_mlp} x {cuda_graphs=True} x {cuda_dot, dot} x {amp}."""
    data = str(tmp_path / "data")
    synth_from_fixture("default.yaml", data, rows=BATCH * STEPS)
    first = {}
    for mlp in ("True", "False"):
        for dot in ("cuda_dot", "dot"):
            for amp in ("True", "False"):
                opts = ["--optimized_mlp=%s" % mlp, "--cuda_graphs=True", "--interaction_op=%s" % dot,
                        "--embedding_type=joint_sparse", "--amp=%s" % amp]
                first[(mlp, dot, amp)] = _train(tmp_path, data, "default_%s_%s_%s" % (mlp, dot, amp), opts)[0]
    # the same data and seed: every configuration starts from the same loss, to its arithmetic (--amp=True computes in fp16 under the
    # loss scaler, --amp=False in bf16 without one: there is no fp32 compute path here)
    ref = first[("True", "cuda_dot", "False")]
    print(first)
    assert all(abs(v - ref) <= 5e-3 * abs(ref) for v in first.values()), first


@pytest.mark.parametrize("name", [n for n in FSPECS if n != "default.yaml"])
def test_every_feature_spec_trains(cuda, tmp_path, name):
    """test_fspecs.sh (--embedding_type=joint_sparse --interaction_op=dot) and test_custom_dot.sh (--embedding_type=joint_sparse)
    over every other fixture: 10 / 20 numerical features, 10 / 30 tables, renamed features and label, other file paths, int8 ..
    int64 index storage, and the criteo_f15 cardinalities (32.7 M embedding rows)."""
    data = str(tmp_path / "data")
    spec, _, _, _ = synth_from_fixture(name, data, rows=BATCH * STEPS)
    _train(tmp_path, data, "fspecs", ["--embedding_type=joint_sparse", "--interaction_op=dot"])
    _train(tmp_path, data, "custom_dot", ["--embedding_type=joint_sparse", "--amp"])
