"""The engines' world_size > 1 code executed for real: two ranks run 3 steps of each trainer, compared with the
one-rank result.

With >= 2 GPUs: one GPU per rank, backend nccl (= RCCL over xGMI).  On a ONE-GPU box (this pool): both ranks share
cuda:0 and the collectives go through gloo staged via host memory (utils/comm.py) -- RCCL refuses two ranks on one
device; the engines' exchange code, bucket order, broadcasts, found-inf agreement and every HIP kernel are the
production path either way.
 * BERT: ranks hold different halves of a batch of 8 (dropout 0): mean of the rank losses == the 1-rank loss on the
   full batch at every step (MLM: 20 masked tokens per sequence on every rank, so the mean of means is the global mean);
   replicas are built from DIFFERENT seeds and must leave with identical weights (broadcast at construction).
 * RN50: the same batch on both ranks (BatchNorm is per rank, as in the reference): losses == the 1-rank run.
 * DLRM: tables split by get_device_mapping, all-to-all forward / backward, data-parallel top MLP: losses == one rank
   holding every table in the same device feature order.
 * WaveGlow (row f1): the ranks hold the two halves of a batch of 4 segments: mean of the rank losses == the 1-rank loss on
   the full batch; replicas built from different seeds leave with identical weights.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def _two_ranks(scenario, tmp_path):
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    out = str(tmp_path / ("%s.json" % scenario))
    port = 29600 + (os.getpid() + hash(scenario)) % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(HERE, "_multirank_worker.py"), scenario, backend, out]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "2-rank %s run failed (%s):\n%s\n%s" % (scenario, backend, r.stdout[-3000:], r.stderr[-6000:])
    return json.load(open(out)), backend


def _one_rank(scenario, cuda):
    import _multirank_worker as W
    return W.SCENARIOS[scenario](0, 1, cuda, 3)


@pytest.mark.parametrize("scenario,rtol", [("bert", 3e-4), ("bert_acc", 3e-4), ("rn50", 3e-4), ("rn50_abandon", 3e-4), ("dlrm", 3e-4),
                                           ("waveglow", 1e-3)])
def test_two_ranks_match_one_rank(cuda, tmp_path, scenario, rtol):
    two, backend = _two_ranks(scenario, tmp_path)
    one = _one_rank(scenario, cuda)
    print(scenario, backend, "2-rank", two[0]["losses"], "1-rank", one["losses"])
    np.testing.assert_allclose(two[0]["losses"], one["losses"], rtol=rtol)
    # replicas agree with each other exactly and with the single-rank weights closely
    assert two[0]["probe"] == two[1]["probe"], "data-parallel replicas diverged"
    ref = np.asarray(one["probe"])
    np.testing.assert_allclose(np.asarray(two[0]["probe"]), ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max())
    if scenario == "rn50_abandon":
        assert two[0]["raised"] and two[1]["raised"] and one["raised"]      # the injected failure did interrupt the backward pass
    if scenario in ("bert", "bert_acc", "rn50", "rn50_abandon", "waveglow"):
        assert two[0]["nbuckets"] > 1          # several gradient buckets were reduced during the backward pass


def test_row_sharded_dlrm_two_ranks_match_one_rank_table_wise(cuda, tmp_path):
    """BASELINE.json configs[3] "embedding tables row-sharded": two ranks with every table cut by row range (ids exchanged, then
    vectors: dlrm/row_sharded.py) against ONE rank holding every table whole on the default trainer -- per-step losses, the
    data-parallel weights, and the UPDATED embedding rows (each owned by exactly one of the two ranks)."""
    two, backend = _two_ranks("dlrm_row", tmp_path)
    one = _one_rank("dlrm_row", cuda)
    print("dlrm_row", backend, "2-rank", two[0]["losses"], "1-rank", one["losses"])
    np.testing.assert_allclose(two[0]["losses"], one["losses"], rtol=3e-4)
    assert two[0]["probe"] == two[1]["probe"], "data-parallel replicas diverged"
    ref = np.asarray(one["probe"])
    np.testing.assert_allclose(np.asarray(two[0]["probe"]), ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max())
    checked = 0
    for t, ref_rows in one["rows"].items():
        for i, want in enumerate(ref_rows):
            got = [r["rows"][str(t)][str(i)] for r in two if str(i) in r["rows"][str(t)]]
            assert len(got) == 1, "row %d of table %s owned by exactly one rank" % (i, t)
            np.testing.assert_allclose(got[0], want, rtol=2e-3, atol=2e-4)
            checked += 1
    assert checked >= 60


def test_rccl_branch_on_a_single_rank_communicator(cuda, tmp_path):
    """the `nccl` branches of utils/comm.py and the engines' bucket hooks on a real RCCL communicator (one rank:
    a one-GPU box cannot host two), see _multirank_worker.run_rccl_single_rank."""
    out = str(tmp_path / "rccl1.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29950 + os.getpid() % 40))
    r = subprocess.run([sys.executable, os.path.join(HERE, "_multirank_worker.py"), "rccl1", "nccl", out], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "single-rank RCCL run failed:\n%s\n%s" % (r.stdout[-3000:], r.stderr[-6000:])
    res = json.load(open(out))[0]
    print(res)
    assert res["identity"] and res["a2a"] and res["buckets_fp32"] and res["buckets_bf16wire"]
    # DLE_COMM=rccl: the same wrappers through the C-ABI binding of librccl.so (csrc/rccl_comm.hip, utils/rccl.py): ncclCommCount
    # reports the one rank that joined, every collective is the identity, the bucket hooks and the RN50 trainer's multi-rank path
    # reproduce the torch-ProcessGroup results
    assert res["direct_count"] == 1 and res["direct_ranks_seen"] == 1
    assert res["direct_identity"] and res["direct_a2a"] and res["direct_buckets"]
    np.testing.assert_allclose(res["direct_rn50"]["losses"], res["rn50"]["one"], rtol=1e-4)
    assert res["direct_rn50"]["nbuckets"] > 1
    for name in ("rn50", "bert"):
        np.testing.assert_allclose(res[name]["flag2"], res[name]["one"], rtol=1e-4)
        ref = np.asarray(res[name]["probe_one"])
        np.testing.assert_allclose(np.asarray(res[name]["probe_flag2"]), ref, rtol=1e-3, atol=1e-3 * np.abs(ref).max())
        assert res[name]["nbuckets"] > 1
