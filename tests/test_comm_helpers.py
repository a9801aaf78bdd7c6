"""utils/comm.py over gloo, world_size 2, on CPU: replicas built from different seeds are equal after
broadcast_parameters_ (data-parallel replicas were never synchronised), mean / max all-reduce, the
split all-to-all of the DLRM exchange, reverse-order gradient buckets."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deeplearningexamples_amd.utils import comm
    from deeplearningexamples_amd.utils.buckets import GradBuckets
    torch.manual_seed(10 + rank)
    m = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.BatchNorm1d(7))
    m[1].num_batches_tracked.fill_(3 + rank)
    comm.broadcast_parameters_(list(m.parameters()) + list(m.buffers()), 0)
    flat = torch.cat([t.detach().reshape(-1).float() for t in list(m.parameters()) + list(m.buffers())])
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    ok_bcast = bool(torch.equal(both[0], both[1]))
    t = torch.full((4,), float(rank + 1))
    comm.allreduce_mean_(t)
    f = torch.tensor([float(rank)])
    comm.allreduce_max_(f)
    # all-to-all with uneven splits: rank r sends (r + 1) * (p + 1) elements to peer p
    send_splits = [(rank + 1) * (p + 1) for p in range(world)]
    recv_splits = [(s + 1) * (rank + 1) for s in range(world)]
    inp = torch.cat([torch.full((n,), 10.0 * rank + p) for p, n in enumerate(send_splits)])
    out = torch.empty(sum(recv_splits))
    comm.all_to_all_single(out, inp, recv_splits, send_splits)
    exp = torch.cat([torch.full((n,), 10.0 * s + rank) for s, n in enumerate(recv_splits)])
    # the host-staged form used for device tensors on a gloo group (two ranks on one GPU): uneven blocks, 16-bit payload
    out16 = comm._pairwise_exchange_host(torch.empty(sum(recv_splits), dtype=torch.bfloat16), inp.to(torch.bfloat16),
                                         recv_splits, send_splits)
    staged_ok = bool(torch.equal(out16, exp.to(torch.bfloat16)))
    # reverse buckets: parameters complete from the END of the flat buffer
    names = [("a", 300), ("b", 200), ("c", 500), ("d", 100)]
    g = torch.full((1100,), float(rank))
    b = GradBuckets(g, names, bucket_mb=0.001, reverse=True)
    fired = [n for n, _ in reversed(names) if b.grad_ready(n)]
    b.wait()
    q.put((rank, ok_bcast, t.tolist(), f.item(), bool(torch.equal(out, exp)) and staged_ok, fired, bool(torch.all(g == 0.5)),
           [(s, e) for s, e, _ in b.buckets]))
    dist.barrier()
    dist.destroy_process_group()


def test_comm_helpers_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + os.getpid() % 150
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    for rank, ok_bcast, mean, mx, a2a_ok, fired, reduced, buckets in res:
        assert ok_bcast and mean == [1.5] * 4 and mx == 1.0 and a2a_ok and reduced
        assert buckets[0][1] == 1100 and buckets[-1][0] == 0 and len(fired) == len(buckets) > 1
