"""Worker of tests/test_gpu_multirank.py (NOT a test module): one rank of a 2-process run of a trainer.

    python -m torch.distributed.run --nproc-per-node 2 ... tests/_multirank_worker.py <scenario> <backend> <out.json>

backend nccl: one GPU per rank over RCCL (needs >= 2 GPUs).  backend gloo: both ranks on cuda:0 -- RCCL refuses two
ranks on one device, so the collectives are staged through host memory (utils/comm.py); everything else (the
engines' multi-rank code, every HIP kernel) is the production path.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def run_bert(rank, world, dev, steps):
    from oracle import bert_oracle as BO
    from deeplearningexamples_amd.bert.model import BertForPreTraining
    from deeplearningexamples_amd.bert.engine import BertTrainer
    c = BO.BERT_STEP_CONFIG
    torch.manual_seed(100 + rank)                           # replicas are built DIFFERENTLY; the trainer must sync them
    model = BertForPreTraining(c["cfg"], device=dev)
    if rank == 0:
        model.load_state_dict({k: v.clone() for k, v in BO.seeded_state(c["cfg"], c["seed"]).items()}, strict=False)
    tr = BertTrainer(model, lr=c["lr"], warmup=c["warmup"], total_steps=c["total_steps"], compute_dtype=torch.bfloat16,
                     hidden_dropout=0.0, attention_dropout=0.0, world_size=world, rank=rank, bucket_mb=1)
    full = BO.seeded_batch(c["cfg"], c["seed"] + 1, 8)
    per = 8 // world
    mine = [t[rank * per:(rank + 1) * per].contiguous().to(dev) for t in full]
    losses = []
    for _ in range(steps):
        loss = tr.train_step(*mine)
        if world > 1:
            from deeplearningexamples_amd.utils import comm
            loss = comm.allreduce_mean_(loss.clone())
        losses.append(float(loss.item()))
    named = dict(model.named_parameters())
    probe = named["bert.encoder.layer.0.attention.self.query.weight"].detach().float().cpu().numpy()[:2].tolist()
    return {"losses": losses, "probe": probe, "nbuckets": len(tr.buckets.buckets) if tr.buckets else 0}


def run_bert_acc(rank, world, dev, steps):
    """Gradient accumulation (run_pretraining.py:679-681, --gradient_accumulation_steps): every rank runs 2 micro-batches per
    optimizer step, only the last one communicates (buckets fired during ITS backward), the accumulation count joins the loss
    scale.  world 1: the same four micro-batches on one rank (divisor 4) -> the mean over ranks of (gA + gB) / 2."""
    from oracle import bert_oracle as BO
    from deeplearningexamples_amd.bert.model import BertForPreTraining
    from deeplearningexamples_amd.bert.engine import BertTrainer
    c = BO.BERT_STEP_CONFIG
    torch.manual_seed(100 + rank)
    model = BertForPreTraining(c["cfg"], device=dev)
    if rank == 0:
        model.load_state_dict({k: v.clone() for k, v in BO.seeded_state(c["cfg"], c["seed"]).items()}, strict=False)
    tr = BertTrainer(model, lr=c["lr"], warmup=c["warmup"], total_steps=c["total_steps"], compute_dtype=torch.bfloat16,
                     hidden_dropout=0.0, attention_dropout=0.0, world_size=world, rank=rank, bucket_mb=1)
    full = BO.seeded_batch(c["cfg"], c["seed"] + 1, 8)
    quarters = [[t[q * 2:(q + 1) * 2].contiguous().to(dev) for t in full] for q in range(4)]
    mine = quarters if world == 1 else quarters[rank * 2:(rank + 1) * 2]
    tr.grad_divisor = len(mine)
    losses = []
    for _ in range(steps):
        tot = 0.0
        for i, mb in enumerate(mine):
            tr._reduce_now = i == len(mine) - 1
            loss, dlog, dnsp = tr.forward(*mb)
            tr.backward(dlog, dnsp, accumulate=i > 0)
            tot = tot + loss / len(mine)
        tr.optimizer_step()
        if world > 1:
            from deeplearningexamples_amd.utils import comm
            tot = comm.allreduce_mean_(tot.clone())
        losses.append(float(tot.item()))
    named = dict(model.named_parameters())
    probe = named["bert.encoder.layer.0.attention.self.query.weight"].detach().float().cpu().numpy()[:2].tolist()
    return {"losses": losses, "probe": probe, "nbuckets": len(tr.buckets.buckets) if tr.buckets else 0}


def run_rn50(rank, world, dev, steps):
    from oracle import resnet_oracle as RO
    from deeplearningexamples_amd.convnets.resnet import ResNet50
    from deeplearningexamples_amd.convnets.engine import ResNetTrainer
    c = RO.RN50_STEP_CONFIG
    torch.manual_seed(200 + rank)
    model = ResNet50(device=dev)
    if rank == 0:
        model.load_state_dict({k: v.clone() for k, v in RO.seeded_state(c["seed"]).items()}, strict=False)
    tr = ResNetTrainer(model, lr=c["lr"], compute_dtype=torch.bfloat16, static_loss_scale=128.0, world_size=world,
                       bucket_mb=4)
    # the SAME batch on every rank: the mean of identical gradients is the single-rank gradient, BatchNorm statistics
    # (per rank, as in the reference: no SyncBN) are identical too -> losses must equal the 1-rank run
    x, y = RO.seeded_batch(c["seed"] + 100, 8, c["size"])
    x, y = x.to(dev), y.to(dev)
    losses = [float(tr.train_step(x, y).item()) for _ in range(steps)]
    probe = model.fc.bias.detach().cpu().numpy()[:8].tolist()
    return {"losses": losses, "probe": probe, "nbuckets": len(tr.buckets.buckets) if tr.buckets else 0}


def run_rn50_abandon(rank, world, dev, steps):
    """A step that dies between forward and backward, and one that dies INSIDE its backward pass (after some units have set their
    cross-unit state -- a BatchNorm reduction taken by the neighbour's kernel, gradient buckets already launched), must not
    poison the steps that follow: same losses and weights as on one rank that went through the same sequence."""
    from oracle import resnet_oracle as RO
    from deeplearningexamples_amd import functional as F
    from deeplearningexamples_amd.convnets.resnet import ResNet50
    from deeplearningexamples_amd.convnets.engine import ResNetTrainer
    c = RO.RN50_STEP_CONFIG
    torch.manual_seed(300 + rank)
    model = ResNet50(device=dev)
    if rank == 0:
        model.load_state_dict({k: v.clone() for k, v in RO.seeded_state(c["seed"]).items()}, strict=False)
    tr = ResNetTrainer(model, lr=c["lr"], compute_dtype=torch.bfloat16, static_loss_scale=128.0, world_size=world, bucket_mb=4)
    x, y = RO.seeded_batch(c["seed"] + 100, 8, c["size"])
    x, y = x.to(dev), y.to(dev)
    losses = [float(tr.train_step(x, y).item())]
    tr.forward(x)                                            # (1) the exception arrives between forward and backward
    logits = tr.forward(x)                                   # (2) ... or inside the backward pass, half way down the network
    _, dlogits = F.softmax_xent(logits, y, smoothing=tr.smoothing, grad_scale=tr.scaler.scale if tr.scaler.enabled else None,
                                grad_dtype=tr.dtype)
    victim = tr.blocks[len(tr.blocks) // 2][0]
    orig = victim.backward

    def boom(*a, **k):
        raise RuntimeError("injected failure inside the backward pass")
    victim.backward = boom
    tr._reduce_now = True
    raised = False
    try:
        tr.backward(dlogits)
    except RuntimeError:
        raised = True
    victim.backward = orig
    torch.cuda.synchronize(dev)
    losses += [float(tr.train_step(x, y).item()) for _ in range(2)]
    probe = model.fc.bias.detach().cpu().numpy()[:8].tolist()
    return {"losses": losses, "probe": probe, "nbuckets": len(tr.buckets.buckets) if tr.buckets else 0, "raised": raised}


DLRM_MR = dict(num=13, sizes=[300, 50, 7, 2000, 11, 640], dim=128, bottom=[64, 128], top=[128, 64, 1], lr=0.5, batch=256,
               seed=5)


def dlrm_device_order(world):
    from deeplearningexamples_amd.dlrm import placement as P
    mapping = P.get_device_mapping(DLRM_MR["sizes"], world)
    return mapping, [t for bucket in mapping["embedding"] for t in bucket]


def run_dlrm(rank, world, dev, steps):
    """world 2: tables placed by get_device_mapping.  world 1: ONE rank holding the tables in the 2-rank DEVICE order
    (the interaction sees features in device order, dlrm/model/distributed.py:135-138), same weights, same batch."""
    from oracle import dlrm_step_oracle as SO
    from deeplearningexamples_amd.dlrm import placement as P
    from deeplearningexamples_amd.dlrm.model import DistributedDlrm
    from deeplearningexamples_amd.dlrm.engine import DlrmTrainer
    c = DLRM_MR
    mapping2, order = dlrm_device_order(2)
    sizes_dev = [c["sizes"][t] for t in order]
    state = SO.seeded_dlrm_state(sizes_dev, c["dim"], c["bottom"], c["top"], c["num"], c["seed"])
    num, cat, click = SO.seeded_dlrm_batch(sizes_dev, c["num"], c["batch"], c["seed"] + 1)
    off = np.concatenate([[0], np.cumsum(sizes_dev)])
    if world == 1:
        my = list(range(len(sizes_dev)))
        has_bottom, vectors, batches = True, None, [c["batch"]]
    else:
        # position of this rank's tables inside the device-ordered list
        start = sum(len(b) for b in mapping2["embedding"][:rank])
        my = list(range(start, start + len(mapping2["embedding"][rank])))
        has_bottom = rank == mapping2["bottom_mlp"]
        vectors = mapping2["vectors_per_gpu"]
        batches = P.get_gpu_batch_sizes(c["batch"], world)
    torch.manual_seed(300 + rank)
    model = DistributedDlrm(num_numerical_features=c["num"], categorical_feature_sizes=[sizes_dev[i] for i in my],
                            bottom_mlp_sizes=c["bottom"] if has_bottom else None, top_mlp_sizes=c["top"],
                            vectors_per_gpu=vectors, embedding_device_mapping=mapping2["embedding"] if world > 1 else None,
                            world_num_categorical_features=len(sizes_dev), embedding_dim=c["dim"], device=dev,
                            compute_dtype=torch.float16, world_size=world)
    with torch.no_grad():
        if has_bottom:
            for i, l in enumerate(model.bottom_model.mlp.linears):
                l.weight.copy_(state["bottom_mlp.%d.weight" % i]); l.bias.copy_(state["bottom_mlp.%d.bias" % i])
        if rank == 0:                                        # the data-parallel top MLP: only rank 0 gets the seeded weights
            for i, l in enumerate(model.top_model.mlp.linears):
                l.weight.copy_(state["top_mlp.%d.weight" % i]); l.bias.copy_(state["top_mlp.%d.bias" % i])
            model.top_model.out.weight.copy_(state["out.weight"]); model.top_model.out.bias.copy_(state["out.bias"])
        if my:
            rows = np.concatenate([np.arange(off[i], off[i + 1]) for i in my])
            model.bottom_model.embeddings.weight.copy_(state["embedding"][torch.from_numpy(rows)])
    model.refresh_working_copies()
    tr = DlrmTrainer(model, lr=c["lr"], batch_sizes_per_gpu=batches, vectors_per_gpu=vectors, rank=rank,
                     world_size=world, amp=True)
    numd = num.to(dev) if has_bottom else None
    catd = cat[:, my].contiguous().to(dev) if my else None
    clickd = click.to(dev)
    losses = []
    for _ in range(steps):
        loss = tr.train_step(numd, catd, clickd)
        if world > 1:
            from deeplearningexamples_amd.utils import comm
            loss = comm.allreduce_mean_(loss.clone())        # equal per-rank batch sizes -> mean of means
        losses.append(float(loss.item()))
    probe = model.top_model.out.weight.detach().float().cpu().numpy().reshape(-1)[:8].tolist()
    return {"losses": losses, "probe": probe}


def run_dlrm_row(rank, world, dev, steps):
    """world 2: every table ROW-SHARDED over the ranks (dlrm/row_sharded.py: ids routed by row range, all-to-all of ids, then of the
    vectors; data-parallel bottom + top MLP).  world 1: the default trainer holding every table whole.  Same weights, same batch."""
    from oracle import dlrm_step_oracle as SO
    from deeplearningexamples_amd.dlrm import placement as P
    from deeplearningexamples_amd.dlrm.model import DistributedDlrm
    from deeplearningexamples_amd.dlrm.engine import DlrmTrainer
    from deeplearningexamples_amd.dlrm import row_sharded as RS
    c = DLRM_MR
    sizes = c["sizes"]
    state = SO.seeded_dlrm_state(sizes, c["dim"], c["bottom"], c["top"], c["num"], c["seed"])
    num, cat, click = SO.seeded_dlrm_batch(sizes, c["num"], c["batch"], c["seed"] + 1)
    off = np.concatenate([[0], np.cumsum(sizes)])
    torch.manual_seed(300 + rank)
    if world == 1:
        model = DistributedDlrm(num_numerical_features=c["num"], categorical_feature_sizes=sizes, bottom_mlp_sizes=c["bottom"],
                                top_mlp_sizes=c["top"], embedding_dim=c["dim"], device=dev, compute_dtype=torch.float16)
    else:
        model, plan = RS.build_row_sharded_model(c["num"], sizes, c["bottom"], c["top"], rank, world, embedding_dim=c["dim"], device=dev,
                                                 compute_dtype=torch.float16)
    with torch.no_grad():
        if rank == 0:                                        # data-parallel MLPs: only rank 0 gets the seeded weights
            for i, l in enumerate(model.bottom_model.mlp.linears):
                l.weight.copy_(state["bottom_mlp.%d.weight" % i]); l.bias.copy_(state["bottom_mlp.%d.bias" % i])
            for i, l in enumerate(model.top_model.mlp.linears):
                l.weight.copy_(state["top_mlp.%d.weight" % i]); l.bias.copy_(state["top_mlp.%d.bias" % i])
            model.top_model.out.weight.copy_(state["out.weight"]); model.top_model.out.bias.copy_(state["out.bias"])
        full = [state["embedding"][int(off[t]):int(off[t + 1])] for t in range(len(sizes))]
        if world == 1:
            model.bottom_model.embeddings.weight.copy_(state["embedding"])
        else:
            RS.load_row_shards(model, plan, rank, full)
    model.refresh_working_copies()
    if world == 1:
        tr = DlrmTrainer(model, lr=c["lr"], batch_sizes_per_gpu=[c["batch"]], amp=True)
    else:
        tr = RS.RowShardedDlrmTrainer(model, plan, c["lr"], P.get_gpu_batch_sizes(c["batch"], world), rank=rank, world_size=world,
                                      amp=True)
    numd, catd, clickd = num.to(dev), cat.to(dev), click.to(dev)
    losses = []
    for _ in range(steps):
        loss = tr.train_step(numd, catd, clickd)
        if world > 1:
            from deeplearningexamples_amd.utils import comm
            loss = comm.allreduce_mean_(loss.clone())        # equal per-rank batch sizes -> mean of means
        losses.append(float(loss.item()))
    probe = model.top_model.out.weight.detach().float().cpu().numpy().reshape(-1)[:8].tolist()
    # the touched embedding rows after the steps, in GLOBAL row order (rank-local shards are re-assembled by the test)
    emb = model.bottom_model.embeddings.weight.detach().float().cpu()
    if world == 1:
        rows = {t: emb[int(off[t]):int(off[t]) + min(sizes[t], 16), :4].numpy().tolist() for t in range(len(sizes))}
    else:
        rows = {}
        for t in range(len(sizes)):
            lo, hi = plan.rows_of(rank, t)
            o = plan.local_offsets[rank][t]
            keep = [i for i in range(lo, hi) if i < min(sizes[t], 16)]
            rows[t] = {i: emb[o + i - lo, :4].numpy().tolist() for i in keep}
    return {"losses": losses, "probe": probe, "rows": rows}


def run_waveglow(rank, world, dev, steps):
    from oracle import waveglow_oracle as WO
    from deeplearningexamples_amd.waveglow.engine import WaveGlowTrainer
    from deeplearningexamples_amd.waveglow.model import WaveGlow
    c = WO.WAVEGLOW_CASE
    torch.manual_seed(300 + rank)                           # replicas are built DIFFERENTLY; the trainer must sync them
    model = WaveGlow(**c["cfg"], device=dev)
    if rank == 0:
        model.load_reference_state(WO.seeded_state(c["cfg"], c["seed"]))
    # lr: Adam moves every element by ~lr per step whatever the gradient size; small enough that an element whose 16-bit gradient
    # changes sign between the 1-rank and the 2-rank evaluation stays inside the probe tolerance of the test
    tr = WaveGlowTrainer(model, lr=2e-5, grad_clip_thresh=0.5, compute_dtype=torch.float16, init_loss_scale=1024.0,
                         world_size=world, bucket_mb=1)
    mel, audio = WO.seeded_inputs(dict(c, batch=4))
    per = 4 // world                                        # equal halves: the mean of the rank losses is the full-batch loss
    mine = [t[rank * per:(rank + 1) * per].contiguous().to(dev) for t in (mel, audio)]
    losses = []
    for _ in range(steps):
        loss = tr.train_step(*mine)
        if world > 1:
            from deeplearningexamples_amd.utils import comm
            loss = comm.allreduce_mean_(loss.clone())
        losses.append(float(loss.item()))
    probe = model.state_dict()["WN.1.in_layers.1.weight_v"].detach().float().cpu().numpy().reshape(-1)[:16].tolist()
    return {"losses": losses, "probe": probe, "nbuckets": len(tr.buckets.buckets) if tr.buckets else 0}


def run_rccl_single_rank(rank, world, dev, steps):
    """ONE rank, backend nccl: every `nccl` branch of utils/comm.py (AVG all-reduce, SUM / MAX, broadcast, all_to_all_single with
    split lists, `device_id` initialisation) and the engines' bucket hooks on the communication stream execute on a real RCCL
    communicator -- on a one-GPU box, where two ranks cannot share the device.  A one-rank collective is the identity, so a
    trainer built with world_size = 2 over this group (its multi-rank code path: parameter broadcast, buckets fired during
    backward on the side stream, bucket wait before the optimizer) must reproduce the world_size = 1 trainer (to the run-to-run noise of the step's few atomics)."""
    from deeplearningexamples_amd.utils import comm
    from deeplearningexamples_amd.utils.buckets import GradBuckets
    assert dist.get_backend() == "nccl" and world == 1
    out = {}
    g = torch.Generator().manual_seed(1)
    t = torch.randn(1 << 20, generator=g).to(dev)
    ref = t.clone()
    comm.allreduce_mean_(t); comm.allreduce_sum_(t); comm.allreduce_max_(t); comm.broadcast_(t, 0)
    out["identity"] = bool(torch.equal(t, ref))
    src = torch.randn(7 * 128, generator=g).to(dev).half()
    dst = torch.zeros_like(src)
    comm.all_to_all_single(dst, src, [7 * 128], [7 * 128])
    out["a2a"] = bool(torch.equal(dst, src))
    flat = torch.randn(3 << 20, generator=g).to(dev)
    keep = flat.clone()
    stream = torch.cuda.Stream()
    for wire in (None, torch.bfloat16):
        b = GradBuckets(flat, [("p%d" % i, 1 << 20) for i in range(3)], bucket_mb=2, comm_stream=stream, reverse=True, wire_dtype=wire)
        for i in (2, 1, 0):
            b.grad_ready("p%d" % i)
        b.wait()
        torch.cuda.synchronize()
        tol = 0.0 if wire is None else 4e-3
        out["buckets_%s" % ("fp32" if wire is None else "bf16wire")] = bool((flat - keep).abs().max() <= tol * keep.abs().max())
        flat.copy_(keep)
    # ---- the same wrappers over the C-ABI RCCL path (DLE_COMM=rccl: csrc/rccl_comm.hip through utils/rccl.py)
    os.environ["DLE_COMM"] = "rccl"
    try:
        from deeplearningexamples_amd.utils import rccl
        c = rccl.comm_for(None)
        out["direct_count"] = c.count()
        out["direct_ranks_seen"] = comm.ranks_seen(dev)
        t = ref.clone()
        comm.allreduce_mean_(t); comm.allreduce_sum_(t); comm.allreduce_max_(t); comm.broadcast_(t, 0)
        f = torch.full((1,), 3.0, device=dev); comm.allreduce_max_(f)
        i64 = torch.arange(5, device=dev); i64b = torch.zeros_like(i64); comm.all_to_all_single(i64b, i64, [5], [5])
        dst = torch.zeros_like(src)
        comm.all_to_all_single(dst, src, [7 * 128], [7 * 128])
        torch.cuda.synchronize()
        out["direct_identity"] = bool(torch.equal(t, ref) and float(f.item()) == 3.0 and torch.equal(i64b, i64))
        out["direct_a2a"] = bool(torch.equal(dst, src))
        b = GradBuckets(flat, [("p%d" % i, 1 << 20) for i in range(3)], bucket_mb=2, comm_stream=stream, reverse=True, wire_dtype=torch.bfloat16)
        for i in (2, 1, 0):
            b.grad_ready("p%d" % i)
        b.wait()
        torch.cuda.synchronize()
        out["direct_buckets"] = bool((flat - keep).abs().max() <= 4e-3 * keep.abs().max())
        flat.copy_(keep)
        d = run_rn50(0, 2, dev, steps)                       # the RN50 trainer's multi-rank path over the direct wrappers
        out["direct_rn50"] = {"losses": d["losses"], "nbuckets": d["nbuckets"]}
    finally:
        os.environ["DLE_COMM"] = "torch"
    # the trainers: world_size flag 2 over the one-rank RCCL group == world_size 1
    for name, fn in (("rn50", run_rn50), ("bert", run_bert)):
        a = fn(0, 1, dev, steps)
        b2 = fn(0, 2, dev, steps) if name == "rn50" else _bert_flag2(dev, steps)
        out[name] = {"one": a["losses"], "flag2": b2["losses"], "probe_one": a["probe"], "probe_flag2": b2["probe"],
                     "nbuckets": b2["nbuckets"]}
    return out


def _bert_flag2(dev, steps):
    """run_bert's one-rank batch (all 8 sequences) with the trainer in its multi-rank configuration."""
    from oracle import bert_oracle as BO
    from deeplearningexamples_amd.bert.model import BertForPreTraining
    from deeplearningexamples_amd.bert.engine import BertTrainer
    c = BO.BERT_STEP_CONFIG
    torch.manual_seed(100)
    model = BertForPreTraining(c["cfg"], device=dev)
    model.load_state_dict({k: v.clone() for k, v in BO.seeded_state(c["cfg"], c["seed"]).items()}, strict=False)
    tr = BertTrainer(model, lr=c["lr"], warmup=c["warmup"], total_steps=c["total_steps"], compute_dtype=torch.bfloat16,
                     hidden_dropout=0.0, attention_dropout=0.0, world_size=2, rank=0, bucket_mb=1)
    mine = [t.contiguous().to(dev) for t in BO.seeded_batch(c["cfg"], c["seed"] + 1, 8)]
    losses = [float(tr.train_step(*mine).item()) for _ in range(steps)]
    named = dict(model.named_parameters())
    probe = named["bert.encoder.layer.0.attention.self.query.weight"].detach().float().cpu().numpy()[:2].tolist()
    return {"losses": losses, "probe": probe, "nbuckets": len(tr.buckets.buckets)}


SCENARIOS = {"bert_acc": run_bert_acc, "bert": run_bert, "rn50": run_rn50, "rn50_abandon": run_rn50_abandon, "dlrm": run_dlrm, "dlrm_row": run_dlrm_row, "waveglow": run_waveglow, "rccl1": run_rccl_single_rank}


def main():
    scenario, backend, out = sys.argv[1:4]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    kw = {"device_id": dev} if backend == "nccl" else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    res = SCENARIOS[scenario](rank, world, dev, 3)
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        json.dump(gathered, open(out, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
