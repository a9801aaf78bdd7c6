"""Data-parallel gradient buckets: cutting at parameter boundaries + mean all-reduce over gloo, world_size 2."""
import os

import torch

from deeplearningexamples_amd.utils.buckets import GradBuckets, cut_buckets


def test_cut_buckets_boundaries():
    named = [("a", 10), ("b", 3), ("c", 20), ("d", 1), ("e", 5)]
    b = cut_buckets(named, bucket_bytes=12 * 4)
    assert b == [(0, 13, "b"), (13, 33, "c"), (33, 39, "e")]
    assert cut_buckets(named, 10 ** 9) == [(0, 39, "e")]
    # RN50-sized: 25.56 M fp32 gradients in 25 MB buckets -> 4-5 buckets
    from oracle.resnet_oracle import param_shapes
    import numpy as np
    sizes = [(n, int(np.prod(s))) for n, s in param_shapes()][::-1]
    nb = len(cut_buckets(sizes, 25 << 20))
    assert 4 <= nb <= 5 and sum(n for _, n in sizes) == 25557032


def _worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        named = [("w3", 1000), ("b3", 10), ("w2", 3000), ("b2", 30), ("w1", 500)]
        total = sum(n for _, n in named)
        flat = torch.arange(total, dtype=torch.float32) * (rank + 1)
        gb = GradBuckets(flat, named, bucket_mb=0, group=None, comm_stream=None)       # every parameter its own bucket
        gb.buckets = cut_buckets(named, 4000 * 4)
        gb._by_last = {b[2]: i for i, b in enumerate(gb.buckets)}
        fired = [gb.grad_ready(n) for n, _ in named]
        gb.wait()
        exp = torch.arange(total, dtype=torch.float32) * (1 + 2) / 2
        ret[rank] = bool(torch.allclose(flat, exp)) and fired == [False, False, True, False, True]
    finally:
        dist.destroy_process_group()


def _worker_wire(rank, world, port, ret):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        named = [("w1", 500), ("b2", 30), ("w2", 3000), ("b3", 10), ("w3", 1000)]     # named_parameters() order
        total = sum(n for _, n in named)
        g = torch.Generator().manual_seed(5)
        base = torch.randn(total, generator=g)
        flat = base * (rank + 1) * 0.25
        # BERT layout: backward completes the buffer from its END; 16-bit wire format (fp16: gloo has no bf16 reduction)
        gb = GradBuckets(flat, named, bucket_mb=0, reverse=True, wire_dtype=torch.float16)
        gb.buckets = [(total - e, total - s, nm) for (s, e, nm) in cut_buckets(list(reversed(named)), 2000 * 4)]
        gb._by_last = {b[2]: i for i, b in enumerate(gb.buckets)}
        fired = [gb.grad_ready(n) for n, _ in reversed(named)]
        gb.wait()
        exact = base * (1 + 2) * 0.25 / 2                      # the fp32 mean
        # what the wire format computes: each rank's share rounded to fp16, summed in fp16
        exp = ((base * 0.25 / 2).half() + (base * 0.5 / 2).half()).float()
        ret[rank] = (bool(torch.equal(flat, exp)) and float((flat - exact).abs().max()) < 2e-3 and sum(fired) == len(gb.buckets)
                     and len(gb.buckets) >= 2)
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_16bit_wire_two_ranks_gloo():
    import torch.multiprocessing as mp
    port = 31600 + os.getpid() % 2000
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker_wire, args=(2, port, ret), nprocs=2, join=True)
        assert ret[0] and ret[1]


def test_bucketed_mean_allreduce_two_ranks_gloo():
    import torch.multiprocessing as mp
    port = 29600 + os.getpid() % 2000
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
        assert ret[0] and ret[1]
