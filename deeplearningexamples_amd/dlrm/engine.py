"""DLRM train step on MI355X: forward, loss, backward, sparse + dense SGD, loss-scale update.

Mirrors the reference's step (Recommendation/DLRM/dlrm/scripts/main.py):
    forward_backward  :585-594   autocast forward, BCEWithLogits on this rank's label slice, scaled backward
    weight_update     :596-608   scaler.step(mlp_optimizer); scaler.unscale_(embedding_optimizer);
                                 embedding_optimizer.step(); scaler.update()
    LR compensation   :444-452   model-parallel parts (embeddings, bottom MLP) use lr / world_size
    CudaGraphWrapper  :194-274   whole-step capture  (here: HIP graph through torch.cuda.CUDAGraph)
and the hybrid-parallel exchange (dlrm/model/distributed.py:25-98): one all-to-all forward, one backward,
plus the data-parallel all-reduce (mean) of the top-MLP gradients, both through torch.distributed
(backend "nccl" == RCCL over xGMI) on side streams overlapped with the remaining backward work.

One deliberate difference: on a gradient overflow the reference still applies the embedding update
(embedding_optimizer.step() is called directly, main.py:606); here the sparse update is skipped together
with the dense one (found_inf gates both), the scale is backed off exactly as GradScaler does.
"""
from typing import Optional, Sequence

import os

import torch
import torch.distributed as dist

from .. import functional as F
from .. import multi_tensor as mt
from ..utils import comm
from ..utils.comm import allreduce_mean_
from .model import DistributedDlrm
from .placement import ExchangePlan


class GradScalerState:
    """Device-resident loss-scale state (torch.cuda.amp.GradScaler semantics, main.py:497)."""

    def __init__(self, device, enabled=True, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5,
                 growth_interval=int(1e9)):
        self.enabled = enabled
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        s = init_scale if enabled else 1.0
        self.scale = torch.full((1,), s, dtype=torch.float32, device=device)
        self.inv_scale = torch.full((1,), 1.0 / s, dtype=torch.float32, device=device)
        self.found_inf = torch.zeros(1, dtype=torch.float32, device=device)
        self.growth_tracker = torch.zeros(1, dtype=torch.int32, device=device)

    def update(self):
        if self.enabled:
            F.amp_update_scale_(self.scale, self.growth_tracker, self.found_inf, self.inv_scale,
                                self.growth_factor, self.backoff_factor, min(self.growth_interval, 2 ** 31 - 1))


class _FlatGrads:
    """fp32 gradients of a list of Linear layers as views of ONE flat buffer (a single all-reduce bucket)."""

    def __init__(self, linears, device, storage=None):
        n = sum(l.weight.numel() + l.bias.numel() for l in linears)
        self.flat = torch.zeros(n, dtype=torch.float32, device=device) if storage is None else storage
        assert self.flat.numel() == n
        self.views = []
        o = 0
        for l in linears:
            gw = self.flat[o:o + l.weight.numel()].view_as(l.weight)
            o += l.weight.numel()
            gb = self.flat[o:o + l.bias.numel()]
            o += l.bias.numel()
            self.views.append((gw, gb))

    def tensors(self):
        return [t for pair in self.views for t in pair]


class DlrmTrainer:
    def __init__(self, model: DistributedDlrm, lr: float, batch_sizes_per_gpu: Sequence[int],
                 vectors_per_gpu: Optional[Sequence[int]] = None, rank: int = 0, world_size: int = 1,
                 amp: bool = True, init_scale: float = 65536.0, freeze_mlps=False, freeze_embeddings=False,
                 process_group=None):
        self.model = model
        self.rank, self.world = rank, world_size
        self.device = model.top_model.out.weight.device
        self.pg = process_group
        self.freeze_mlps, self.freeze_embeddings = freeze_mlps, freeze_embeddings
        d = model._embedding_dim
        if vectors_per_gpu is None:
            vectors_per_gpu = [model.bottom_model.num_feature_vectors]
        self.plan = ExchangePlan(batch_sizes_per_gpu, vectors_per_gpu, d, rank)
        self.scaler = GradScalerState(self.device, enabled=amp and model.compute_dtype == torch.float16,
                                      init_scale=init_scale)
        self.base_lr = lr
        self.lr_factor = 1.0
        # device-resident learning rates (no host sync / graph friendly)
        self.lr_dp = torch.full((1,), lr, dtype=torch.float32, device=self.device)               # top model
        self.lr_mp = torch.full((1,), lr / world_size, dtype=torch.float32, device=self.device)  # bottom parts
        top = model.top_model
        self.top_linears = top.mlp.linears + [top.out]
        bm = model.bottom_model.mlp
        self.bot_linears = bm.linears if bm is not None else []
        # ONE buffer behind the dense gradients of both MLPs (top first: the data-parallel all-reduce bucket stays a contiguous
        # view): the GradScaler's inf / nan sweep over them is one launch
        count = lambda ls: sum(l.weight.numel() + l.bias.numel() for l in ls)
        n_top, n_bot = count(self.top_linears), count(self.bot_linears)
        pad = (n_top + 63) // 64 * 64                 # (the second view starts on a 256-byte boundary; the gap stays zero)
        self.dense_grads = torch.zeros(pad + n_bot, dtype=torch.float32, device=self.device)
        self.top_grads = _FlatGrads(self.top_linears, self.device, self.dense_grads[:n_top])
        self.bot_grads = _FlatGrads(self.bot_linears, self.device, self.dense_grads[pad:]) if bm is not None else None
        if world_size > 1:
            # the top MLP is the data-parallel part: every replica starts from rank 0's weights (the reference wraps it
            # in torch DDP, dlrm/scripts/main.py:463-466); embeddings / bottom MLP are model parallel and stay local
            comm.broadcast_parameters_(list(model.top_model.parameters()), 0, process_group)
        model.refresh_working_copies()
        self._tables = {}
        self._build_tables()
        self.noop = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.comm_stream = torch.cuda.Stream(device=self.device) if world_size > 1 else None
        self.moving_loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        import os
        # last layer + loss + their backward as one kernel (DLE_DLRM_FUSE_HEAD=0: the separate GEMM / loss launches)
        self.fuse_head = (os.environ.get("DLE_DLRM_FUSE_HEAD", "1") != "0" and self.device.type == "cuda"
                          and model.top_model.head_fusable())

    # ------------------------------------------------------------------ optimizer plumbing
    def _build_tables(self):
        top = self.model.top_model
        w16 = top.mlp.working_copies() + [top.out_working_copy()]
        # weights whose 16-bit copy has the same shape get the copy refreshed inside the SGD kernel
        def split(linears, grads, copies):
            same = [i for i, l in enumerate(linears) if copies[i].shape == l.weight.shape]
            g, p, c = [], [], []
            for i in same:
                g.append(grads.views[i][0]); p.append(linears[i].weight.data); c.append(copies[i])
            gb = [grads.views[i][1] for i in range(len(linears))]
            pb = [l.bias.data for l in linears]
            padded = [i for i in range(len(linears)) if i not in same]
            gp = [grads.views[i][0] for i in padded]
            pp = [linears[i].weight.data for i in padded]
            return (mt.TensorTable([g, p, c], mt.streaming_chunk([g])) if g else None,
                    mt.TensorTable([gb + gp, pb + pp], mt.streaming_chunk([gb + gp])), padded)
        self.t_top_w, self.t_top_b, self.top_padded = split(self.top_linears, self.top_grads, w16)
        if self.bot_grads is not None:
            bw16 = self.model.bottom_model.mlp.working_copies()
            self.t_bot_w, self.t_bot_b, self.bot_padded = split(self.bot_linears, self.bot_grads, bw16)
        self.t_top_all = mt.TensorTable([[self.top_grads.flat]])
        # One learning rate for both MLPs (one rank, or the row-sharded placement's data-parallel bottom MLP): ONE update launch
        # over every dense tensor -- weights with a same-shape 16-bit copy refresh it in the kernel, biases and the K-padded first
        # layers carry no copy (their padded copies follow by cast_rows).  Nine launches at the tail of the step become four.
        self.t_dense = None
        if self.bot_grads is not None and (self.world == 1 or getattr(self, "bottom_dp", False)) and \
                os.environ.get("DLE_DLRM_ONE_DENSE_STEP", "1") != "0":
            bw = self.model.bottom_model.mlp.working_copies()
            g, pp, c = [], [], []
            for lins, grads, copies in ((self.top_linears, self.top_grads, w16), (self.bot_linears, self.bot_grads, bw)):
                for i, l in enumerate(lins):
                    g += [grads.views[i][0], grads.views[i][1]]
                    pp += [l.weight.data, l.bias.data]
                    c += [copies[i] if copies[i].shape == l.weight.shape else None, None]
            self.t_dense = mt.TensorTable([g, pp, c], mt.streaming_chunk([g]))

    def set_lr_factor(self, factor: float):
        """LearningRateScheduler.step() (dlrm/scripts/utils.py:278-286): lr = base * factor per group."""
        if factor != self.lr_factor:
            self.lr_factor = factor
            self.lr_dp.fill_(self.base_lr * factor)
            self.lr_mp.fill_(self.base_lr * factor / self.world)

    def _dense_step(self):
        sc = self.scaler
        skip = sc.found_inf if sc.enabled else None
        inv = sc.inv_scale if sc.enabled else None
        if self.t_dense is not None:
            mt.sgd(self.t_dense, self.lr_dp, skip_flag=skip, inv_scale=inv, has_momentum=False, model_copy=True)
            top = self.model.top_model
            copies = top.mlp.working_copies() + [top.out_working_copy()]
            for i in self.top_padded:
                F.cast_rows(self.top_linears[i].weight.data, copies[i].dtype, cols_out=copies[i].shape[1], out=copies[i])
            bm = self.model.bottom_model.mlp
            for i in self.bot_padded:
                cp = bm.working_copies()[i]
                F.cast_rows(self.bot_linears[i].weight.data, cp.dtype, cols_out=cp.shape[1], out=cp)
            return
        if self.t_top_w is not None:
            mt.sgd(self.t_top_w, self.lr_dp, skip_flag=skip, inv_scale=inv, has_momentum=False, model_copy=True)
        mt.sgd(self.t_top_b, self.lr_dp, skip_flag=skip, inv_scale=inv, has_momentum=False)
        top = self.model.top_model
        for i in self.top_padded:
            lin = self.top_linears[i]
            copies = top.mlp.working_copies() + [top.out_working_copy()]
            F.cast_rows(lin.weight.data, copies[i].dtype, cols_out=copies[i].shape[1], out=copies[i])
        if self.bot_grads is not None:
            # (row-sharded placement, dlrm/row_sharded.py: the bottom MLP is data parallel there)
            lr_bot = self.lr_dp if getattr(self, "bottom_dp", False) else self.lr_mp
            if self.t_bot_w is not None:
                mt.sgd(self.t_bot_w, lr_bot, skip_flag=skip, inv_scale=inv, has_momentum=False, model_copy=True)
            mt.sgd(self.t_bot_b, lr_bot, skip_flag=skip, inv_scale=inv, has_momentum=False)
            bm = self.model.bottom_model.mlp
            for i in self.bot_padded:
                c = bm.working_copies()[i]
                F.cast_rows(self.bot_linears[i].weight.data, c.dtype, cols_out=c.shape[1], out=c)

    # ------------------------------------------------------------------ hybrid-parallel exchange
    def _exchange(self, out, inp, out_splits, in_splits):
        """The all-to-all on the COMMUNICATION stream, ordered after what the compute stream has enqueued; the caller waits
        (`_exchange_wait`) right before the first consumer, so independent compute launched in between overlaps the transfer."""
        cur = torch.cuda.current_stream()
        self.comm_stream.wait_stream(cur)
        with torch.cuda.stream(self.comm_stream):
            comm.all_to_all_single(out, inp, out_splits, in_splits, group=self.pg)
            self._a2a_event = torch.cuda.Event()
            self._a2a_event.record(self.comm_stream)
        for t in (out, inp):
            t.record_stream(self.comm_stream)

    def _exchange_wait(self):
        """The compute stream waits for the LAST all-to-all only (not for what was queued behind it on the communication stream)."""
        torch.cuda.current_stream().wait_event(self._a2a_event)

    def _bottom_to_top(self, local_out, plan=None):
        """[B_global, n_r, D] -> [B_r, n_total, D] (device feature order).  all_to_all_single over RCCL, then ONE unpack launch
        (the reference's torch.cat(dim=1) of the received blocks, dlrm/model/distributed.py:70-75)."""
        p = plan or self.plan
        recv = torch.empty(sum(p.fwd_recv_splits), dtype=local_out.dtype, device=local_out.device)
        self._exchange(recv, local_out.view(-1), p.fwd_recv_splits, p.fwd_send_splits)
        self._exchange_wait()
        x = torch.empty((p.local_batch, p.n_total, p.dim), dtype=local_out.dtype, device=local_out.device)
        F.a2a_blocks(recv, x, p.local_batch, [v * p.dim for v in p.vectors], pack=False)
        return x

    def _top_to_bottom_start(self, grad_x):
        """Reverse exchange of the gradient: [B_r, n_total, D] -> [B_global, n_r, D].  ONE pack launch, then the all-to-all on
        the communication stream; returns the receive buffer -- call _exchange_wait() before reading it."""
        p = self.plan
        send = torch.empty(sum(p.fwd_recv_splits), dtype=grad_x.dtype, device=grad_x.device)
        F.a2a_blocks(send, grad_x.contiguous(), p.local_batch, [v * p.dim for v in p.vectors], pack=True)
        out = torch.empty((p.global_batch, p.n_local, p.dim), dtype=grad_x.dtype, device=grad_x.device)
        self._exchange(out.view(-1), send, p.fwd_send_splits, p.fwd_recv_splits)
        return out

    # ------------------------------------------------------------------ validation (dlrm/scripts/main.py:733-835 dist_evaluate)
    @torch.no_grad()
    def evaluate(self, batches, batch_sizes_per_gpu=None):
        """Forward only over `batches` = iterable of (numerical or None, categorical or None, click[, valid rows]); with several
        ranks every rank's logits are gathered; AUC (utils.roc_auc_score) and the BCE loss over the whole set -> (auc, loss) on
        the main process, (None, None) elsewhere.  Batches may use a different (static) size than the train step."""
        from .utils import roc_auc_score
        m = self.model
        y_true, y_score = [], []
        plan = None
        for item in batches:
            num, cat, click = item[:3]
            valid = item[3] if len(item) > 3 else click.shape[0]
            x = m.bottom_model(num, cat)
            if self.world > 1:
                if plan is None or plan.global_batch != x.shape[0]:
                    sizes = list(batch_sizes_per_gpu) if batch_sizes_per_gpu and sum(batch_sizes_per_gpu) == x.shape[0] else \
                        [x.shape[0] // self.world] * self.world
                    plan = ExchangePlan(sizes, self.plan.vectors, self.plan.dim, self.rank)
                x = self._bottom_to_top(x, plan)
            out = m.top_model(x).reshape(-1).float()
            if self.world > 1:
                parts = [torch.empty(b, dtype=out.dtype, device=out.device) for b in plan.batch_sizes]
                if dist.get_backend(self.pg) == "nccl":
                    dist.all_gather(parts, out, group=self.pg)
                else:                                   # one-GPU test rig: staged through the host (utils/comm.py)
                    hp = [t.cpu() for t in parts]
                    dist.all_gather(hp, out.cpu(), group=self.pg)
                    parts = [t.to(out.device) for t in hp]
                out = torch.cat(parts)
            y_true.append(click.reshape(-1).float()[:valid])
            y_score.append(out[:valid])
        if self.rank != 0:
            if self.world > 1:
                dist.barrier(group=self.pg)
            return None, None
        y_true, y_score = torch.cat(y_true), torch.cat(y_score)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(y_score, y_true)
        auc = roc_auc_score(y_true, torch.sigmoid(y_score))
        if self.world > 1:
            dist.barrier(group=self.pg)
        return auc, float(loss.item())

    # ------------------------------------------------------------------ the step
    def _wgrad_stream(self):
        """Second stream for the top model's weight gradients (one rank; None with DLE_DLRM_TWO_STREAMS=0 or off the GPU)."""
        import os
        dev = self.scaler.scale.device
        if dev.type != "cuda" or os.environ.get("DLE_DLRM_TWO_STREAMS", "1") == "0":
            return None
        if getattr(self, "_wside", None) is None:
            self._wside = torch.cuda.Stream(device=dev)
        return self._wside

    def train_step(self, numerical_features, categorical_features, click):
        """One optimisation step on a (global) batch.  Returns the device-resident fp32 loss [1]."""
        m, p, sc = self.model, self.plan, self.scaler
        bottom_out = m.bottom_model(numerical_features, categorical_features)
        x = self._bottom_to_top(bottom_out) if self.world > 1 else bottom_out
        labels = click[p.batch_start[self.rank]:p.batch_start[self.rank + 1]] if self.world > 1 else click
        if self.fuse_head:
            # last layer + loss + their backward in one pass over the last hidden activation (csrc/dlrm_head.hip)
            loss = m.top_model.forward_loss_backward_head(x, labels, grad_scale=sc.scale if sc.enabled else None,
                                                          grads=self.top_grads.views[:-1], out_grads=self.top_grads.views[-1])
            dlogits = None
        else:
            logits = m.top_model(x)
            loss, dlogits = F.bce_with_logits(logits, labels, grad_scale=sc.scale if sc.enabled else None)
            dlogits = dlogits.view(-1, 1)
        # one rank: the interaction backward itself reports inf / nan in the gradient it writes (no sweep over 450 MB)
        fused_check = sc.enabled and self.world == 1
        if self.world > 1:
            # data-gradient chain of the top model first; the gradient all-to-all (xGMI) then runs on the communication stream
            # UNDER the top model's weight gradients, and the data-parallel mean of those follows it on the same stream
            grad_x, finish = m.top_model.backward(dlogits, grads=self.top_grads.views[:-1],
                                                  out_grads=self.top_grads.views[-1], defer_wgrad=True)
            grad_bottom = self._top_to_bottom_start(grad_x)
            finish()
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                allreduce_mean_(self.top_grads.flat, self.pg)
            # (stream order on comm_stream: all-to-all, then the all-reduce; the compute stream needs only the former here)
            self._exchange_wait()
        else:
            # one rank: the top model's weight / bias gradients are leaves -- they run on a second stream beside the bottom model's
            # backward (embedding update on its own stream there), joined before anything reads the flat gradient buffers
            wside = self._wgrad_stream()
            grad_x, finish = m.top_model.backward(dlogits, grads=self.top_grads.views[:-1],
                                                  out_grads=self.top_grads.views[-1],
                                                  found_inf=sc.found_inf if fused_check else None, defer_wgrad=True)
            if wside is not None:
                wside.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(wside):
                    finish()
            else:
                finish()
            grad_bottom = grad_x
        if sc.enabled:
            if not fused_check:
                F.check_nonfinite_(grad_bottom, sc.found_inf)
            if self.world > 1:
                # grad_bottom is this rank's slice of a model-parallel gradient: ranks must agree on skipping the step
                # (torch's GradScaler all-reduces found_inf across the process group in the same way)
                comm.allreduce_max_(sc.found_inf, self.pg)
        m.bottom_model.backward(grad_bottom, self.lr_mp, inv_scale=sc.inv_scale if sc.enabled else None,
                                skip_flag=sc.found_inf if sc.enabled else None,
                                mlp_grads=self.bot_grads.views if self.bot_grads is not None else None,
                                freeze_embeddings=self.freeze_embeddings)
        if self.world > 1:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        elif self._wgrad_stream() is not None:
            torch.cuda.current_stream().wait_stream(self._wgrad_stream())
        if sc.enabled:
            F.check_nonfinite_(self.dense_grads, sc.found_inf)      # the dense gradients of both MLPs: one buffer, one sweep
            if self.world > 1:
                comm.allreduce_max_(sc.found_inf, self.pg)
        if not self.freeze_mlps:
            self._dense_step()
        sc.update()
        return loss
