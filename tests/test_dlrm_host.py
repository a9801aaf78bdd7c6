"""Host-side DLRM logic of the product package (no GPU): placement vs the reference's outputs (bit exact),
exchange plan, LR schedule, and the multi-rank all-to-all layout over gloo (world_size 2)."""
import json
import os

import numpy as np
import pytest
import torch

from deeplearningexamples_amd.dlrm import placement as P
from deeplearningexamples_amd.dlrm.utils import LearningRateScheduler


def _load(golden_dir, name):
    return json.load(open(os.path.join(golden_dir, name)))


def test_product_placement_bit_exact(golden_dir):
    g = _load(golden_dir, "dlrm_placement.json")
    for case in g["device_mapping"]:
        assert P.get_device_mapping(case["sizes"], case["num_gpus"]) == case["result"], (case["name"], case["num_gpus"])
    for case in g["gpu_batch_sizes"]:
        try:
            got = list(P.get_gpu_batch_sizes(case["global_batch"], case["num_gpus"]))
        except RuntimeError:
            got = None
        assert got == case["result"], case
    for case in g["argsort"]:
        assert P.argsort(case["seq"]) == case["asc"]
        assert P.argsort(case["seq"], True) == case["desc"]


def test_exchange_plan_sizes():
    plan = P.ExchangePlan([8192] * 8, [1, 4, 4, 4, 4, 4, 4, 2], 128, rank=3)
    assert plan.global_batch == 65536 and plan.local_batch == 8192 and plan.n_total == 27 and plan.n_local == 4
    assert plan.fwd_send_splits == [8192 * 4 * 128] * 8                      # 8 MiB in fp16
    assert sum(plan.fwd_recv_splits) == 8192 * 27 * 128
    assert plan.recv_feature_base == [0, 1, 5, 9, 13, 17, 21, 25]
    assert [plan.source_of_feature(s) for s in (0, 1, 4, 5, 26)] == [0, 1, 1, 2, 7]


def test_lr_schedule_matches_reference_formula():
    # dlrm/scripts/utils.py:258-276 evaluated by hand
    s = LearningRateScheduler(warmup_steps=4, warmup_factor=0, decay_steps=4, decay_start_step=6, decay_power=2)
    got = [s.step() for _ in range(12)]
    exp = [0.25, 0.5, 0.75, 1.0, 1, 1, (3 / 4) ** 2, (2 / 4) ** 2, (1 / 4) ** 2, 0.0, 0, 0]
    np.testing.assert_allclose(got, exp)


def _a2a_worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        batch_sizes, vectors, d = [6, 10], [3, 2], 4
        plan = P.ExchangePlan(batch_sizes, vectors, d, rank)
        gb = sum(batch_sizes)
        # feature f of sample b carries the value 1000*f + b in every column
        feat0 = plan.recv_feature_base[rank]
        local = torch.empty(gb, vectors[rank], d)
        for j in range(vectors[rank]):
            local[:, j, :] = (1000 * (feat0 + j) + torch.arange(gb).float())[:, None]
        recv = torch.empty(sum(plan.fwd_recv_splits))
        dist.all_to_all_single(recv, local.reshape(-1), plan.fwd_recv_splits, plan.fwd_send_splits)
        x = torch.empty(plan.local_batch, plan.n_total, d)
        for s in range(world):
            blk = recv[plan.recv_block_start[s]:plan.recv_block_start[s] + plan.fwd_recv_splits[s]]
            x[:, plan.recv_feature_base[s]:plan.recv_feature_base[s] + vectors[s], :] = \
                blk.view(plan.local_batch, vectors[s], d)
        b0 = plan.batch_start[rank]
        exp = torch.empty_like(x)
        for f in range(plan.n_total):
            exp[:, f, :] = (1000 * f + b0 + torch.arange(plan.local_batch).float())[:, None]
        ok_fwd = torch.equal(x, exp)
        # reverse direction returns every rank's own slice of the gradient
        send = torch.empty(sum(plan.fwd_recv_splits))
        for s in range(world):
            blk = send[plan.recv_block_start[s]:plan.recv_block_start[s] + plan.fwd_recv_splits[s]]
            blk.view(plan.local_batch, vectors[s], d).copy_(x[:, plan.recv_feature_base[s]:plan.recv_feature_base[s] + vectors[s], :])
        back = torch.empty(gb, vectors[rank], d)
        dist.all_to_all_single(back.view(-1), send, plan.fwd_send_splits, plan.fwd_recv_splits)
        ret[rank] = bool(ok_fwd and torch.equal(back, local))
    finally:
        dist.destroy_process_group()


def test_all_to_all_layout_two_ranks_gloo():
    import torch.multiprocessing as mp
    port = 29500 + os.getpid() % 2000
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_a2a_worker, args=(2, port, ret), nprocs=2, join=True)
        assert ret[0] and ret[1]


def test_roc_auc_matches_sklearn_and_reference():
    """dlrm/utils.roc_auc_score vs sklearn (the definition the reference restates) and, when the tree is mounted, vs the
    reference's own function (dlrm/scripts/utils.py:289-320): ties, skewed classes, one-class input."""
    import numpy as np
    import torch
    from sklearn.metrics import roc_auc_score as sk_auc
    from deeplearningexamples_amd.dlrm.utils import roc_auc_score
    g = torch.Generator().manual_seed(0)
    cases = []
    y = (torch.rand(5000, generator=g) < 0.3).float()
    cases.append((y, torch.randn(5000, generator=g) + y))                                  # informative scores
    cases.append((y, torch.randint(0, 7, (5000,), generator=g).float()))                   # heavy ties
    cases.append(((torch.rand(4000, generator=g) < 0.01).float(), torch.rand(4000, generator=g)))
    for yt, ys in cases:
        assert abs(roc_auc_score(yt, ys) - sk_auc(yt.numpy(), ys.numpy())) < 1e-9
    assert np.isnan(roc_auc_score(torch.ones(10), torch.rand(10, generator=g)))
    from oracle import _ref_import as R
    if R.have_reference():
        import importlib.util, os
        spec = importlib.util.spec_from_file_location("ref_dlrm_utils_auc", os.path.join(
            R.REF, "PyTorch/Recommendation/DLRM/dlrm/scripts/utils.py"))
        src = open(spec.origin).read()
        ns = {"torch": torch}
        start = src.index("def roc_auc_score")
        exec(compile(src[start:], spec.origin, "exec"), ns)                                # the function alone (the module imports dllogger)
        for yt, ys in cases[:2]:
            assert abs(roc_auc_score(yt, ys) - ns["roc_auc_score"](yt.clone(), ys.clone())) < 1e-6


# ---- row-sharded placement (BASELINE.json configs[3] as worded; deeplearningexamples_amd/dlrm/row_sharded.py)
def test_row_shard_plan_partitions_every_table():
    sizes = [300, 50, 7, 2000, 11, 640, 1, 8]
    for world in (1, 2, 3, 8):
        plan = P.RowShardPlan(sizes, world)
        for t, n in enumerate(sizes):
            owned = []
            for r in range(world):
                lo, hi = plan.rows_of(r, t)
                assert hi - lo == plan.local_sizes[r][t]
                owned += list(range(lo, hi))
            assert owned == list(range(n)), "every row of every table is owned exactly once, in order"
        shard, off = plan.tensors("cpu")
        g = torch.Generator().manual_seed(world)
        ids = torch.stack([torch.randint(0, n, (64,), generator=g) for n in sizes], 1)
        owner, row = P.RowShardPlan.route(ids, shard, off)
        for b in range(ids.shape[0]):
            for t in range(len(sizes)):
                r = int(owner[b, t])
                lo, hi = plan.rows_of(r, t)
                assert lo <= int(ids[b, t]) < hi
                assert int(row[b, t]) == plan.local_offsets[r][t] + int(ids[b, t]) - lo
                assert 0 <= int(row[b, t]) < plan.local_rows[r]
        order, counts = P.RowShardPlan.bucket(owner.reshape(-1), world)
        flat = owner.reshape(-1)[order]
        assert torch.equal(flat, flat.sort().values) and counts.tolist() == [int((flat == r).sum()) for r in range(world)]
        for r in range(world):                                   # stable: lookups bound for one rank keep their order
            sel = order[flat == r]
            assert torch.equal(sel, sel.sort().values)


def _row_shard_worker(rank, world, port, ret):
    import torch.distributed as dist
    from deeplearningexamples_amd.dlrm import row_sharded as RS
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sizes, dim, batches = [37, 5, 1, 260, 11], 4, [6, 10]
        plan = P.RowShardPlan(sizes, world)
        g = torch.Generator().manual_seed(7)                                     # the same on every rank
        full = [torch.randn(n, dim, generator=g) for n in sizes]
        cat = torch.stack([torch.randint(0, n, (sum(batches),), generator=g) for n in sizes], 1)
        lo = sum(batches[:rank])
        mine = cat[lo:lo + batches[rank]].contiguous()
        local = torch.cat([full[t][slice(*plan.rows_of(rank, t))] for t in range(len(sizes))])
        shard, off = plan.tensors("cpu")
        back, (order, c_send, c_recv, recv_rows) = RS.lookup_exchange(mine, shard, off, world, None, lambda rows: local[rows], dim,
                                                                      torch.float32)
        got = torch.empty(mine.numel(), dim)
        got[order] = back                                                        # undo the send order
        exp = torch.stack([full[t][mine[b, t]] for b in range(mine.shape[0]) for t in range(len(sizes))])
        ok_fwd = torch.equal(got, exp)
        # backward: "gradient" of lookup (b, t) = its vector + 1; the owners accumulate what they receive on their joint rows
        g_send = (exp + 1.0)[order]
        g_recv = RS.grad_exchange(g_send, c_send, c_recv, world, None)
        acc = torch.zeros_like(local).index_add_(0, recv_rows, g_recv)
        # brute force over the GLOBAL batch: what this rank's rows must have received
        want = torch.zeros_like(local)
        for b in range(cat.shape[0]):
            for t in range(len(sizes)):
                i = int(cat[b, t])
                a, z = plan.rows_of(rank, t)
                if a <= i < z:
                    want[plan.local_offsets[rank][t] + i - a] += full[t][i] + 1.0
        ret[rank] = bool(ok_fwd and torch.allclose(acc, want, atol=1e-5) and sum(c_send) == mine.numel())
    finally:
        dist.destroy_process_group()


def test_row_sharded_exchange_two_ranks_gloo():
    """ids routed by row range, all-to-all of the ids, gather on the owner, all-to-all of the vectors back == a direct lookup in the
    full tables; the reverse exchange delivers every gradient row to the owner of its table row (world size 2, gloo)."""
    import torch.multiprocessing as mp
    port = 31500 + os.getpid() % 2000
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_row_shard_worker, args=(2, port, ret), nprocs=2, join=True)
        assert ret[0] and ret[1]


@pytest.mark.skipif(not os.path.isdir("/root/reference/PyTorch"), reason="reference tree not mounted")
def test_module_signatures_follow_the_reference_classes():
    """DistributedDlrm / DlrmBottom / DlrmTop (model/distributed.py:106-180, nn/parts.py:25-100): the reference's argument NAMES in
    the reference's ORDER (extra, defaulted arguments may follow), read out of its sources."""
    import ast
    import inspect
    from deeplearningexamples_amd.dlrm import model as M
    root = "/root/reference/PyTorch/Recommendation/DLRM/dlrm/"

    def ref_args(path, cls, fn):
        for n in ast.walk(ast.parse(open(root + path).read())):
            if isinstance(n, ast.ClassDef) and n.name == cls:
                for f in n.body:
                    if isinstance(f, ast.FunctionDef) and f.name == fn:
                        return [a.arg for a in f.args.args[1:]]
        raise AssertionError((path, cls, fn))

    for path, cls, fn in (("model/distributed.py", "DistributedDlrm", "__init__"), ("model/distributed.py", "DistributedDlrm", "forward"),
                          ("nn/parts.py", "DlrmBottom", "__init__"), ("nn/parts.py", "DlrmBottom", "forward"),
                          ("nn/parts.py", "DlrmTop", "__init__")):
        want = ref_args(path, cls, fn)
        got = list(inspect.signature(getattr(getattr(M, cls), fn)).parameters)[1:]
        assert got[:len(want)] == want or (fn == "forward" and cls == "DlrmBottom" and got[:2] == want[:2]), (cls, fn, want, got)
    assert hasattr(M.DistributedDlrm, "from_dict")
