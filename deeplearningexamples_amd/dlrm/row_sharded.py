"""DLRM train step with ROW-SHARDED embedding tables (opt-in: `--embedding_sharding row`).

BASELINE.json configs[3] names "embedding tables row-sharded over 8 GPUs, RCCL all-to-all over xGMI"; the reference itself
places WHOLE tables on ranks (Recommendation/DLRM/dlrm/utils/distributed.py:102-176) and dlrm/engine.py follows it.  This is
the other placement SURVEY.md 8(e) describes -- "index bucketing by row range + a2a of indices and vectors" -- behind the same
trainer interface:

    every table is cut into `world` row ranges (placement.RowShardPlan); every rank is data parallel over its batch slice
    ids [B_r, T] --route--> (owner rank, joint row on the owner)            integer arithmetic, bit-exact (torch int64 ops)
                 --bucket--> stable sort by owner + per-owner counts        (counts cross once: host-sized exchange buffers)
    all-to-all #1: row ids to their owners (int64)            [dlrm/model/distributed.py:68 is the reference's exchange point]
    owner: gather of the requested rows (dle_emb_gather_fwd, the kernel of the table-wise path)
    all-to-all #2: the vectors back (16-bit) --rows_scatter--> the interaction input [B_r, 1 + T, D], slot 0 = bottom MLP
    backward: gradient rows gathered in the send order --all-to-all #3--> owners, duplicate-free sparse SGD on the owner

The bottom MLP is data parallel here (every rank runs it on its batch slice; its gradients are averaged with the top MLP's),
the embedding update uses lr / world like the reference's model-parallel parts (dlrm/scripts/main.py:444-452: the local loss is a
mean over B / world samples).  The per-owner counts are data dependent, so the step has one host read per call and is not HIP-graph
capturable -- the price of the placement; the table-wise path stays the default and the benchmarked one.
"""
from typing import Sequence

import torch

from .. import functional as F
from ..utils import comm
from ..utils.comm import allreduce_mean_
from .engine import DlrmTrainer
from .model import DistributedDlrm
from .placement import RowShardPlan


def build_row_sharded_model(num_numerical_features, table_sizes: Sequence[int], bottom_mlp_sizes, top_mlp_sizes, rank, world_size,
                            embedding_dim=128, device="cuda", compute_dtype=torch.float16):
    """The per-rank model of the row-sharded placement: this rank's row ranges of every table as ONE joint matrix (the update
    then runs its generic duplicate-folding path), bottom + top MLP on every rank.  -> (model, RowShardPlan)"""
    plan = RowShardPlan(table_sizes, world_size)
    model = DistributedDlrm(num_numerical_features=num_numerical_features, categorical_feature_sizes=[max(plan.local_rows[rank], 1)],
                            bottom_mlp_sizes=bottom_mlp_sizes, top_mlp_sizes=top_mlp_sizes,
                            world_num_categorical_features=len(table_sizes), embedding_dim=embedding_dim, device=device,
                            compute_dtype=compute_dtype, world_size=1)
    # (DlrmBottom initialises a table with uniform(+-sqrt(1 / rows)): redo it per logical table so that a row's distribution
    #  does not depend on the placement)
    with torch.no_grad():
        w = model.bottom_model.embeddings.weight
        for t, n in enumerate(table_sizes):
            lo, cnt = plan.local_offsets[rank][t], plan.local_sizes[rank][t]
            if cnt:
                torch.nn.init.uniform_(w[lo:lo + cnt], -(1.0 / n) ** 0.5, (1.0 / n) ** 0.5)
    return model, plan


def load_row_shards(model, plan: RowShardPlan, rank, full_tables):
    """Copy this rank's row ranges out of `full_tables` (one [N_t, D] tensor per table) into the joint matrix."""
    with torch.no_grad():
        w = model.bottom_model.embeddings.weight
        for t, full in enumerate(full_tables):
            lo, hi = plan.rows_of(rank, t)
            o = plan.local_offsets[rank][t]
            if hi > lo:
                w[o:o + hi - lo].copy_(full[lo:hi])


def _a2a(out, inp, out_splits, in_splits, world, group):
    if world > 1:
        comm.all_to_all_single(out, inp, out_splits, in_splits, group=group)
    else:
        out.copy_(inp)
    return out


def lookup_exchange(cat, shard_t, off_t, world, group, gather, dim, dtype):
    """The forward exchange of the row-sharded placement for this rank's ids cat [B, T]: route + bucket, all-to-all of the row ids,
    `gather(rows int64 [n]) -> [n, dim]` on the owner, all-to-all of the vectors back.
    -> (vectors [B T, dim] in SEND order, (order, send counts, receive counts, received rows)): lookup i of the send order is
    (sample, table) = divmod(order[i], T).  Host logic + collectives only (the CPU suite runs it on gloo with a torch gather)."""
    owner, row = RowShardPlan.route(cat, shard_t, off_t)
    order, counts = RowShardPlan.bucket(owner.reshape(-1), world)
    recv_counts = torch.empty_like(counts)
    _a2a(recv_counts, counts, [1] * world, [1] * world, world, group)
    c_send, c_recv = counts.tolist(), recv_counts.tolist()                  # (the step's one host read: buffer sizes)
    send_rows = row.reshape(-1)[order].contiguous()
    recv_rows = torch.empty(sum(c_recv), dtype=torch.int64, device=cat.device)
    _a2a(recv_rows, send_rows, c_recv, c_send, world, group)
    vec = gather(recv_rows) if recv_rows.numel() else torch.empty((0, dim), dtype=dtype, device=cat.device)
    back = torch.empty((cat.numel(), dim), dtype=dtype, device=cat.device)
    _a2a(back.view(-1), vec.reshape(-1), [c * dim for c in c_send], [c * dim for c in c_recv], world, group)
    return back, (order, c_send, c_recv, recv_rows)


def grad_exchange(g_send, c_send, c_recv, world, group):
    """The backward exchange: gradient rows in the send order of lookup_exchange -> the owners (in THEIR receive order, i.e.
    aligned with the `received rows` of the forward exchange)."""
    dim = g_send.shape[1]
    g_recv = torch.empty((sum(c_recv), dim), dtype=g_send.dtype, device=g_send.device)
    _a2a(g_recv.view(-1), g_send.reshape(-1), [c * dim for c in c_recv], [c * dim for c in c_send], world, group)
    return g_recv


class RowShardedDlrmTrainer(DlrmTrainer):
    def __init__(self, model: DistributedDlrm, plan: RowShardPlan, lr: float, batch_sizes_per_gpu: Sequence[int], rank=0,
                 world_size=1, **kw):
        t = len(plan.sizes)
        super().__init__(model, lr, batch_sizes_per_gpu, vectors_per_gpu=[t + 1] * world_size, rank=rank, world_size=world_size, **kw)
        if model._hash_indices:
            raise ValueError("row-sharded placement: --hash_indices is not supported (ids are routed by value)")
        self.rplan, self.tables = plan, t
        self.bottom_dp = True                          # _dense_step: the bottom MLP steps with the data-parallel learning rate
        self.shard_t, self.off_t = plan.tensors(self.device)
        if world_size > 1:
            comm.broadcast_parameters_(list(model.bottom_model.mlp.parameters()), 0, self.pg)
            model.refresh_working_copies()
        self._dst_cache = {}

    def _dst_rows(self, b):
        """Row of x.view(B (1 + T), D) that lookup (sample, table) lands in: slot 0 of every sample is the bottom MLP's."""
        if b not in self._dst_cache:
            t = self.tables
            self._dst_cache[b] = (torch.arange(b, device=self.device)[:, None] * (t + 1) + 1 +
                                  torch.arange(t, device=self.device)[None, :]).reshape(-1)
        return self._dst_cache[b]

    def lookup(self, cat):
        """ids [B, T] of this rank's samples -> (vectors [B T, D] in SEND order, state for the backward exchange)."""
        emb = self.model.bottom_model.embeddings
        cd = self.model.compute_dtype
        return lookup_exchange(cat, self.shard_t, self.off_t, self.world, self.pg,
                               lambda rows: F.emb_gather_fwd(emb.weight.data, rows.view(-1, 1), out_dtype=cd).view(-1, emb.embedding_dim),
                               emb.embedding_dim, cd)

    def train_step(self, numerical_features, categorical_features, click):
        """One optimisation step; every rank is handed the GLOBAL batch (numerical [B, F], categorical int64 [B, T], click [B])
        and works on its slice.  Returns this rank's device-resident fp32 loss [1] (mean over its slice)."""
        m, p, sc = self.model, self.plan, self.scaler
        lo, hi = p.batch_start[self.rank], p.batch_start[self.rank + 1]
        num, cat, labels = numerical_features[lo:hi], categorical_features[lo:hi].contiguous(), click[lo:hi]
        b, t, d = hi - lo, self.tables, m._embedding_dim
        bm = m.bottom_model
        x = torch.empty((b, t + 1, d), dtype=m.compute_dtype, device=self.device)
        back, (order, c_send, c_recv, recv_rows) = self.lookup(cat)
        dst = self._dst_rows(b)[order]
        F.rows_scatter_(x.view(-1, d), back, dst)
        bm.mlp(F.cast_rows(num, m.compute_dtype, cols_out=bm.mlp.k_padded(0)), out=x[:, 0, :])
        if self.fuse_head:
            loss = m.top_model.forward_loss_backward_head(x, labels, grad_scale=sc.scale if sc.enabled else None,
                                                          grads=self.top_grads.views[:-1], out_grads=self.top_grads.views[-1])
            dlogits = None
        else:
            logits = m.top_model(x)
            loss, dlogits = F.bce_with_logits(logits, labels, grad_scale=sc.scale if sc.enabled else None)
            dlogits = dlogits.view(-1, 1)
        grad_x = m.top_model.backward(dlogits, grads=self.top_grads.views[:-1], out_grads=self.top_grads.views[-1])
        if sc.enabled:
            F.check_nonfinite_(grad_x, sc.found_inf)
            if self.world > 1:
                comm.allreduce_max_(sc.found_inf, self.pg)
        # gradient rows in the send order of the forward exchange, back to the owners of the rows
        g_send = F.rows_gather(grad_x.view(-1, d), dst)
        g_recv = grad_exchange(g_send, c_send, c_recv, self.world, self.pg)
        if recv_rows.numel() and not self.freeze_embeddings:
            emb = bm.embeddings
            F.emb_sgd_dedup_(emb.weight.data, recv_rows.view(-1, 1), g_recv.view(-1, 1, d), emb.workspace(), self.lr_mp,
                             scale=sc.inv_scale if sc.enabled else None, skip_flag=sc.found_inf if sc.enabled else None)
        bm.mlp.backward(grad_x[:, 0, :], grads=self.bot_grads.views)
        if self.world > 1:
            allreduce_mean_(self.top_grads.flat, self.pg)
            allreduce_mean_(self.bot_grads.flat, self.pg)
        if sc.enabled:
            F.check_nonfinite_(self.top_grads.flat, sc.found_inf)
            F.check_nonfinite_(self.bot_grads.flat, sc.found_inf)
            if self.world > 1:
                comm.allreduce_max_(sc.found_inf, self.pg)
        if not self.freeze_mlps:
            self._dense_step()
        sc.update()
        return loss
