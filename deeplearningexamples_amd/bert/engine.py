"""BERT pre-training step on MI355X: BertForPreTraining forward, MLM + NSP loss, backward, LAMB.

Mirrors the reference's step (LanguageModeling/BERT/):
    run_pretraining.py:518-524   take_training_step: autocast forward, criterion, scaled backward
    run_pretraining.py:527-536   take_optimizer_step: lr_scheduler.step(); grad_scaler.step(LAMB); update; zero_grad
    run_pretraining.py:75-95     BertPretrainingCriterion (dense MLM rows, CE ignore_index -1, + NSP CE)
    modeling.py:263-595,788-958  the model (module tree / parameter names in model.py)
    lamb_amp_opt/fused_lamb/fused_lamb.py:131-258  FusedLAMBAMP.step: global grad norm (of the SCALED grads) vs
                                 max_grad_norm * scale, step += (found_inf == 0), two param groups
    schedulers.py:109-136        PolyWarmUpScheduler (degree 0.5)
    run_pretraining.py:679-681   gradient accumulation with no_sync (here: `accumulate` micro-steps add into the flat
                                 fp32 gradient; the data-parallel all-reduce runs once per optimizer step)
Token-major layout [B*S, H]; the per-(sequence, head) attention contractions are batched GEMMs over strided
slices of the fused QKV activation, scores/probabilities are kept for the backward pass.
"""
import math
import os

import torch
import torch.distributed as dist

from .. import _cabi as C
from .. import functional as F
from .. import multi_tensor as mt
from ..dlrm.engine import GradScalerState
from ..utils.buckets import GradBuckets
from ..utils import comm
from .model import BertForPreTraining

NO_DECAY = ("bias", "gamma", "beta", "LayerNorm")      # run_pretraining.py:423


def poly_warmup_lr(step_after, base_lr, warmup, total_steps, degree=0.5):
    """PolyWarmUpScheduler.get_lr (LanguageModeling/BERT/schedulers.py:123-136) with last_epoch = step_after:
    linear warm-up over the first `warmup` fraction of `total_steps`, then base_lr * (1 - progress) ** degree
    (clamped at 0 past the end of the schedule instead of going complex)."""
    progress = step_after / total_steps
    if progress < warmup:
        return base_lr * progress / warmup
    return base_lr * (max(1.0 - progress, 0.0) ** degree)


class BertTrainer:
    def __init__(self, model: BertForPreTraining, lr=6e-3, warmup=0.2843, total_steps=7038, weight_decay=0.01,
                 max_grad_norm=1.0, compute_dtype=torch.bfloat16, init_loss_scale=2.0 ** 20, world_size=1,
                 process_group=None, hidden_dropout=None, attention_dropout=None, seed=42, rank=0, static_batch=False,
                 max_predictions_per_seq=None,
                 bucket_mb=64, allreduce_dtype=None):
        self.model, self.cfg = model, model.config
        self.dev = model.bert.embeddings.word_embeddings.weight.device
        self.dtype = compute_dtype
        self.base_lr, self.warmup, self.total = lr, warmup, total_steps
        self.wd, self.max_norm = weight_decay, max_grad_norm
        self.world, self.pg = world_size, process_group
        # nn.Dropout sites of the model (modeling.py:276,320,392,428; bert_config.json: 0.1 / 0.1).  Every call
        # draws a fresh 64-bit offset of the counter-based RNG; ranks use different seeds (different data, same as
        # torch.manual_seed(seed + rank) in run_pretraining.py:342).
        self.p_hidden = model.config.get("hidden_dropout", 0.1) if hidden_dropout is None else hidden_dropout
        self.p_attn = model.config.get("attention_dropout", 0.1) if attention_dropout is None else attention_dropout
        self.rng_seed, self._rng_offset = int(seed) + int(rank), 0
        self._rng_base = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.scaler = GradScalerState(self.dev, enabled=compute_dtype == torch.float16, init_scale=init_loss_scale,
                                      growth_interval=2000)
        dev = self.dev
        # static_batch: the caller promises that the SAME device tensors hold the same batch at every step (synthetic
        # benchmark / graph-captured input buffers, run_pretraining.py:602-640) -- only then are the masked-row indices
        # of a batch reused; by default they are rebuilt at every step (a loader may refill the same addresses)
        self.static_batch = static_batch
        # max_predictions_per_seq (run_pretraining.py --max_predictions_per_seq): the dense MLM head then works on
        # EXACTLY that many rows per sequence, chosen on the device (masked positions first, unmasked ones -- label -1,
        # ignored by the criterion, zero gradient -- as padding): no host synchronisation, a static shape, HIP-graph
        # capturable.  None: the masked rows are counted on the host like the reference's index_select (modeling.py:590).
        self.max_pred = max_predictions_per_seq
        self.fused_attention = True     # False: batched GEMMs + softmax kernels (also taken for shapes outside the fused envelope)
        # the S = 128 attention backward READING the forward pass's keep bits instead of re-drawing them (dle_attention_bwd_keep):
        # measured 68.10 / 67.99 ms per step against 68.12 / 68.19 re-drawn (batch 256, same box, alternating runs) -- the kernel is
        # not bound by its Philox calls after all; opt-in (DLE_BERT_ATTN_KEEP=1), the default keeps no mask
        self.attn_keep_mask = os.environ.get("DLE_BERT_ATTN_KEEP", "0") == "1"
        self.keep_activations = False   # tests: keep the dropout keep masks of the last step
        model.fuse_qkv_storage()
        if world_size > 1:
            # replicas start from rank 0's weights (torch DDP's constructor, run_pretraining.py:455-460)
            comm.broadcast_parameters_(list(model.parameters()), 0, process_group)
        named = self._ordered_named_parameters(model)
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        total = sum(p.numel() for p in self.params)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.gview, o = {}, 0
        for n, p in named:
            self.gview[n] = self.flat_grad[o:o + p.numel()].view(p.shape)
            o += p.numel()
        self._qkv_grad_contiguity_check()
        self.exp_avg = {n: torch.zeros_like(p.data) for n, p in named}
        self.exp_avg_sq = {n: torch.zeros_like(p.data) for n, p in named}
        # 16-bit working copies of every matrix that feeds a GEMM
        self.w16 = {}
        h = self.cfg["hidden"]
        for l, layer in enumerate(model.bert.encoder.layer):
            pre = "bert.encoder.layer.%d." % l
            qkv16 = torch.empty((3 * h, h), dtype=compute_dtype, device=dev)
            for i, nm in enumerate(("query", "key", "value")):
                self.w16[pre + "attention.self.%s.weight" % nm] = qkv16[i * h:(i + 1) * h]
            layer.qkv16 = qkv16
            for nm in ("attention.output.dense", "intermediate.dense_act", "output.dense"):
                self.w16[pre + nm + ".weight"] = torch.empty_like(dict(named)[pre + nm + ".weight"].data, dtype=compute_dtype)
        for nm in ("bert.pooler.dense_act.weight", "cls.predictions.transform.dense_act.weight",
                   "bert.embeddings.word_embeddings.weight"):
            self.w16[nm] = torch.empty_like(dict(named)[nm].data, dtype=compute_dtype)
        self.names = [n for n, _ in named]
        self.nsp16 = torch.zeros((8, h), dtype=compute_dtype, device=dev)          # 2 -> 8 rows (16-byte rows of dlogits)
        self.w16["cls.seq_relationship.weight"] = self.nsp16[:2]
        self.nsp_bias8 = torch.zeros(8, dtype=torch.float32, device=dev)
        self.refresh_working_copies()
        self._build_tables()
        self.step_t = torch.zeros(1, dtype=torch.int32, device=dev)
        self.lr_t = torch.zeros((), dtype=torch.float32, device=dev)
        self.max_norm_t = torch.full((1,), max_grad_norm, dtype=torch.float32, device=dev)
        self.one = torch.ones(1, dtype=torch.float32, device=dev)
        self.noop = torch.zeros(1, dtype=torch.int32, device=dev)
        self._grad_divisor = 1         # gradient-accumulation micro-steps summed into the flat gradient (property below)
        self._batch_key, self._sel, self._idx0, self._mask_add, self._dense_labels = None, None, None, None, None
        self.comm_stream = torch.cuda.Stream(device=dev) if world_size > 1 else None
        # gradient buckets over the flat buffer, all-reduced (mean) on the side stream while the backward pass is still
        # running; the buffer follows named_parameters(), backward completes it from the end -> reverse buckets
        self.buckets = GradBuckets(self.flat_grad, [(n, p.numel()) for n, p in zip(self.names, self.params)], bucket_mb,
                                   process_group, self.comm_stream, reverse=True, wire_dtype=allreduce_dtype) if world_size > 1 else None
        self._reduce_now = False       # set for the micro-step whose gradients are final (last accumulation step)
        base_t = torch.tensor(float(lr), dtype=torch.float32, device=dev)
        self._lr_consts = (base_t, torch.tensor(float(warmup), device=dev), torch.tensor(float(total_steps), device=dev))

    # ------------------------------------------------------------------ plumbing
    @staticmethod
    def _ordered_named_parameters(model):
        """named_parameters() with each layer's q/k/v weights (then biases) adjacent, so their gradients form one
        [3H, H] / [3H] block of the flat gradient buffer (the fused QKV wgrad writes it in one GEMM)."""
        named = list(model.named_parameters())
        out, seen = [], set()
        d = dict(named)
        for n, p in named:
            if n in seen:
                continue
            if ".attention.self." in n:
                pre = n[:n.index(".attention.self.") + len(".attention.self.")]
                for suffix in ("weight", "bias"):
                    for nm in ("query", "key", "value"):
                        k = pre + nm + "." + suffix
                        out.append((k, d[k])); seen.add(k)
            else:
                out.append((n, p)); seen.add(n)
        return out

    def _qkv_grad_contiguity_check(self):
        for l in range(self.cfg["layers"]):
            pre = "bert.encoder.layer.%d.attention.self." % l
            gq, gk, gv = (self.gview[pre + n + ".weight"] for n in ("query", "key", "value"))
            assert gk.data_ptr() == gq.data_ptr() + gq.numel() * 4 and gv.data_ptr() == gk.data_ptr() + gk.numel() * 4
            bq, bk, bv = (self.gview[pre + n + ".bias"] for n in ("query", "key", "value"))
            assert bk.data_ptr() == bq.data_ptr() + bq.numel() * 4 and bv.data_ptr() == bk.data_ptr() + bk.numel() * 4

    def refresh_working_copies(self):
        named = dict(self.model.named_parameters())
        for n, c in self.w16.items():
            F.cast(named[n].data, self.dtype, out=c)
        self.nsp_bias8[:2].copy_(named["cls.seq_relationship.bias"].data)

    def _build_tables(self):
        named = dict(zip(self.names, self.params))
        groups = {"decay_copy": ([], [], [], [], []), "decay": ([], [], [], []), "nodecay": ([], [], [], [])}
        for n, p in named.items():
            key = "nodecay" if any(nd in n for nd in NO_DECAY) else ("decay_copy" if n in self.w16 else "decay")
            lists = groups[key]
            for lst, t in zip(lists, (self.gview[n], p.data, self.exp_avg[n], self.exp_avg_sq[n])):
                lst.append(t)
            if key == "decay_copy":
                lists[4].append(self.w16[n])
        # LAMB's per-tensor norms taken inside stage 1 (csrc/multi_tensor.hip mt_lamb_stage1<NORMS>); DLE_BERT_LAMB_NORMS=0 keeps the
        # two l2norm sweeps of the reference's call sequence
        self.fuse_lamb_norms = os.environ.get("DLE_BERT_LAMB_NORMS", "1") != "0"
        self.tables = {}
        for key, lists in groups.items():
            if not lists[0]:
                continue
            g, p, m, v = lists[:4]
            self.tables[key] = dict(
                wd=0.0 if key == "nodecay" else self.wd,
                t_p=mt.TensorTable([p]), t_g=mt.TensorTable([g]), t_s1=mt.TensorTable([g, p, m, v]),
                t_s2=mt.TensorTable([g, p, lists[4]] if key == "decay_copy" else [g, p]))
        self.t_all_grads = mt.TensorTable([[self.flat_grad]])

    def _prepare_batch(self, input_ids, attention_mask, labels):
        key = (input_ids.data_ptr(), labels.data_ptr(), attention_mask.data_ptr(), tuple(input_ids.shape))
        if self.static_batch and key == self._batch_key:
            return
        b, s = input_ids.shape
        flat = labels.reshape(-1)
        if self.max_pred is not None:
            # stable sort of (label == -1): masked positions first, in sequence order; the first max_pred columns
            order = torch.sort((labels == -1).to(torch.int8), dim=1, stable=True).indices[:, :self.max_pred]
            self._sel = (order + torch.arange(b, device=labels.device, dtype=torch.int64)[:, None] * s).reshape(-1)
        else:
            self._sel = torch.nonzero(flat != -1).squeeze(1)        # data-dependent size: one host sync per new batch,
        self._dense_labels = flat[self._sel].contiguous()          # like the reference's index_select (modeling.py:590)
        self._idx0 = torch.arange(b, device=self.dev, dtype=torch.int64) * s
        self._mask_add = ((1.0 - attention_mask.to(torch.float32)) * -10000.0).contiguous()
        self._batch_key = key

    def _next_offset(self):
        """Index of the next dropout call INSIDE the current step.  The per-step advance lives in the device word
        self._rng_base, which the kernels add to this index (and which backward() bumps on the device): a HIP-graph
        captured step therefore draws fresh masks at every replay, like torch's graph-safe Philox state."""
        self._rng_offset += 1
        return self._rng_offset

    # ------------------------------------------------------------------ forward
    def forward(self, input_ids, token_type_ids, attention_mask, labels, next_sentence_labels):
        cfg, m, dt = self.cfg, self.model, self.dtype
        h, nh, inter, v = cfg["hidden"], cfg["heads"], cfg["intermediate"], cfg["vocab"]
        d = h // nh
        b, s = input_ids.shape
        t = b * s
        self._prepare_batch(input_ids, attention_mask, labels)
        self._rng_offset = 0
        emb = m.bert.embeddings
        ids, tts = input_ids.reshape(-1).contiguous(), token_type_ids.reshape(-1).contiguous()
        z0 = F.embed_sum(emb.word_embeddings.weight.data, emb.position_embeddings.weight.data,
                         emb.token_type_embeddings.weight.data, ids, tts, s, dt)
        x, _, mean0, rstd0 = F.layernorm_fwd(z0, emb.LayerNorm.weight.data, emb.LayerNorm.bias.data)
        ph, pa, seed = self.p_hidden, self.p_attn, self.rng_seed
        mask0 = None
        if ph > 0:
            x, mask0 = F.dropout_fwd(x, ph, seed, self._next_offset(), offset_base=self._rng_base)
        sv = {"ids": ids, "tt": tts, "z0": z0, "ln0": (mean0, rstd0), "mask0": mask0, "layers": [], "b": b, "s": s}
        scale = 1.0 / math.sqrt(d)
        fused_attn = self.fused_attention and F.attention_supported(s, d)
        sv["fused_attn"] = fused_attn
        for l, layer in enumerate(m.bert.encoder.layer):
            pre = "bert.encoder.layer.%d." % l
            att = layer.attention
            qkv = F.gemm(x, layer.qkv16, t, 3 * h, h, True, True, bias=layer.qkv_bias)
            probs = pdrop = stats = None
            mask_a, mask_1, mask_2 = None, None, None
            off_a = self._next_offset() if pa > 0 else 0
            if fused_attn:
                # QK^T, scale + mask, softmax, dropout and P V in one kernel: no [B, heads, S, S] tensor in HBM
                # (DLE_BERT_ATTN_KEEP=1, S = 128: the keep bits are kept -- 1 bit per probability, 8 MB per layer at batch 256 -- and
                #  READ by the backward kernel instead of re-drawn)
                keep_bits = self.keep_activations or (self.attn_keep_mask and s == 128 and pa > 0)
                ctx, stats, mask_a = F.attention_fwd(qkv, self._mask_add, b, s, nh, scale, pa, seed, off_a,
                                                     want_mask=keep_bits, offset_base=self._rng_base)
            else:
                probs = torch.empty((b * nh, s, s), dtype=dt, device=self.dev)
                F.gemm_batched(qkv, qkv[:, h:], probs, s, s, d, 3 * h, 3 * h, s, True, True, b * nh, nh,
                               (s * 3 * h, d), (s * 3 * h, d), (nh * s * s, s * s))
                if pa > 0:
                    pdrop, mask_a = F.softmax_dropout_fwd_(probs, self._mask_add, nh * s, scale, pa, seed, off_a,
                                                           offset_base=self._rng_base)
                else:
                    F.softmax_fwd_(probs, self._mask_add, nh * s, scale)
                    pdrop = probs
                ctx = torch.empty((t, h), dtype=dt, device=self.dev)
                F.gemm_batched(pdrop, qkv[:, 2 * h:], ctx, s, d, s, s, 3 * h, h, True, False, b * nh, nh,
                               (nh * s * s, s * s), (s * 3 * h, d), (s * h, d))
            ao = F.gemm(ctx, self.w16[pre + "attention.output.dense.weight"], t, h, h, True, True,
                        bias=att.output.dense.bias.data)
            if ph > 0:
                x1, z1, m1, r1, mask_1 = F.dropout_add_layernorm_fwd(ao, att.output.LayerNorm.weight.data,
                                                                      att.output.LayerNorm.bias.data, x, ph, seed,
                                                                      self._next_offset(), offset_base=self._rng_base)
            else:
                x1, z1, m1, r1 = F.layernorm_fwd(ao, att.output.LayerNorm.weight.data, att.output.LayerNorm.bias.data, residual=x)
            # the epilogue leaves gelu'(pre-activation) behind (not the pre-activation): the backward GEMM multiplies
            pre_act = torch.empty((t, inter), dtype=dt, device=self.dev)
            it = F.gemm(x1, self.w16[pre + "intermediate.dense_act.weight"], t, inter, h, True, True,
                        bias=layer.intermediate.dense_act.bias.data, act=C.ACT_GELU_DAUX, aux=pre_act)
            o2 = F.gemm(it, self.w16[pre + "output.dense.weight"], t, h, inter, True, True, bias=layer.output.dense.bias.data)
            if ph > 0:
                x2, z2, m2, r2, mask_2 = F.dropout_add_layernorm_fwd(o2, layer.output.LayerNorm.weight.data,
                                                                      layer.output.LayerNorm.bias.data, x1, ph, seed,
                                                                      self._next_offset(), offset_base=self._rng_base)
            else:
                x2, z2, m2, r2 = F.layernorm_fwd(o2, layer.output.LayerNorm.weight.data, layer.output.LayerNorm.bias.data, residual=x1)
            sv["layers"].append(dict(x=x, qkv=qkv, probs=probs, pdrop=pdrop, stats=stats, off_a=off_a, ctx=ctx, z1=z1, ln1=(m1, r1), x1=x1,
                                     pre=pre_act, it=it, z2=z2, ln2=(m2, r2), mask_a=mask_a, mask_1=mask_1, mask_2=mask_2))
            x = x2
        sv["seq"] = x
        # pooler + NSP head
        first = F.rows_gather(x, self._idx0)
        pooled = F.gemm(first, self.w16["bert.pooler.dense_act.weight"], b, h, h, True, True,
                        bias=m.bert.pooler.dense_act.bias.data, act=C.ACT_TANH)
        nsp_logits = F.gemm(pooled, self.nsp16, b, 8, h, True, True, out_dtype=torch.float32, bias=self.nsp_bias8)
        # dense MLM head on the masked rows
        n = self._sel.numel()
        hm = F.rows_gather(x, self._sel)
        tr = m.cls.predictions.transform
        tpre = torch.empty((n, h), dtype=dt, device=self.dev)
        tg = F.gemm(hm, self.w16["cls.predictions.transform.dense_act.weight"], n, h, h, True, True,
                    bias=tr.dense_act.bias.data, act=C.ACT_GELU, aux=tpre)
        tl, _, mt_, rt = F.layernorm_fwd(tg, tr.LayerNorm.weight.data, tr.LayerNorm.bias.data)
        logits = F.gemm(tl, self.w16["bert.embeddings.word_embeddings.weight"], n, v, h, True, True,
                        out_dtype=torch.float32, bias=m.cls.predictions.bias.data)
        sv.update(first=first, pooled=pooled, hm=hm, tpre=tpre, tg=tg, lnt=(mt_, rt), tl=tl)
        self._sv = sv
        gs = self.scaler.scale if self.scaler.enabled else None
        loss_mlm, dlogits = F.softmax_xent(logits, self._dense_labels, ignore_index=-1, grad_scale=gs, grad_dtype=dt)
        loss_nsp, dnsp = F.softmax_xent(nsp_logits[:, :2], next_sentence_labels, ignore_index=-1, grad_scale=gs,
                                        grad_dtype=dt, ld_out=8)
        return loss_mlm + loss_nsp, dlogits, dnsp

    @property
    def grad_divisor(self):
        return self._grad_divisor

    @grad_divisor.setter
    def grad_divisor(self, n):
        """Accumulation count: joins the loss scale in optimizer_step; with a 16-bit wire format the buckets also pre-divide by it
        before rounding (the reference's wire copy carries loss / count gradients: same fp16 headroom, same overflow pattern)."""
        self._grad_divisor = int(n)
        if getattr(self, "buckets", None) is not None and self.buckets.wire_dtype is not None:
            self.buckets.wire_divisor = float(n)

    # ------------------------------------------------------------------ backward
    def _leaf_stream(self):
        """Second stream for the leaves of the backward graph (weight / bias gradients): beside the data-gradient chain they fill
        the matrix pipes while the chain runs its HBM-bound LayerNorm / attention / dropout kernels.  None: DLE_BERT_WGRAD_STREAM=0
        or off the GPU."""
        if self.dev.type != "cuda" or os.environ.get("DLE_BERT_WGRAD_STREAM", "1") == "0":
            return None
        if getattr(self, "_wstream", None) is None:
            self._wstream = torch.cuda.Stream(device=self.dev)
            self._leaf_keep = []
            if self.buckets is not None:
                self.buckets.extra_streams.append(self._wstream)
        return self._wstream

    def _on_leaf_stream(self, fn, *operands):
        """Run fn() on the leaf stream after everything enqueued so far; its operands stay alive until the streams join."""
        ws = self._leaf_stream()
        if ws is None:
            fn()
            return
        ws.wait_stream(torch.cuda.current_stream())
        self._leaf_keep.append(operands)
        with torch.cuda.stream(ws):
            fn()

    def _join_leaf_stream(self):
        ws = self._leaf_stream()
        if ws is not None:
            torch.cuda.current_stream().wait_stream(ws)
            self._leaf_keep.clear()

    def _wgrad(self, name, g, x, accumulate):
        gw = self.gview[name]
        nout, kin = gw.shape
        self._on_leaf_stream(lambda: F.gemm(g, x, nout, kin, g.shape[0], False, False, out=gw,
                                            splitk=F.pick_splitk(nout, kin, g.shape[0], 1024), accumulate=accumulate), g, x)

    def _bgrad(self, name, g, accumulate):
        self._on_leaf_stream(lambda: F.colsum(g, out=self.gview[name], accumulate=accumulate), g)

    def backward(self, dlogits, dnsp, accumulate=False):
        cfg, m, sv, dt = self.cfg, self.model, self._sv, self.dtype
        h, nh, inter, v = cfg["hidden"], cfg["heads"], cfg["intermediate"], cfg["vocab"]
        d = h // nh
        b, s = sv["b"], sv["s"]
        t = b * s
        acc = accumulate
        n = dlogits.shape[0]
        # ---- MLM head
        self._wgrad("bert.embeddings.word_embeddings.weight", dlogits, sv["tl"], acc)          # tied decoder
        self._bgrad("cls.predictions.bias", dlogits, acc)
        dtl = F.gemm(dlogits, self.w16["bert.embeddings.word_embeddings.weight"], n, h, v, True, False)
        tr = m.cls.predictions.transform
        dtg = F.layernorm_bwd(dtl, sv["tg"], sv["lnt"][0], sv["lnt"][1], tr.LayerNorm.weight.data,
                              self.gview["cls.predictions.transform.LayerNorm.weight"],
                              self.gview["cls.predictions.transform.LayerNorm.bias"], acc)
        dtpre = F.act_bwd(dtg, sv["tpre"], C.ACT_GELU_BWD)
        self._wgrad("cls.predictions.transform.dense_act.weight", dtpre, sv["hm"], acc)
        self._bgrad("cls.predictions.transform.dense_act.bias", dtpre, acc)
        dhm = F.gemm(dtpre, self.w16["cls.predictions.transform.dense_act.weight"], n, h, h, True, False)
        dseq = torch.zeros((t, h), dtype=dt, device=self.dev)
        F.rows_scatter_(dseq, dhm, self._sel)
        # ---- NSP head + pooler
        gnsp8 = torch.empty((8, h), dtype=torch.float32, device=self.dev)
        F.gemm(dnsp, sv["pooled"], 8, h, b, False, False, out=gnsp8)
        gw = self.gview["cls.seq_relationship.weight"]
        if acc:
            gw.add_(gnsp8[:2])
        else:
            gw.copy_(gnsp8[:2])
        gb8 = F.colsum(dnsp)
        gb = self.gview["cls.seq_relationship.bias"]
        if acc:
            gb.add_(gb8[:2])
        else:
            gb.copy_(gb8[:2])
        dpool_pre = F.gemm(dnsp, self.nsp16, b, h, 8, True, False, act=C.ACT_TANH_BWD, mask_src=sv["pooled"])
        self._wgrad("bert.pooler.dense_act.weight", dpool_pre, sv["first"], acc)
        self._bgrad("bert.pooler.dense_act.bias", dpool_pre, acc)
        dfirst = F.gemm(dpool_pre, self.w16["bert.pooler.dense_act.weight"], b, h, h, True, False)
        F.rows_scatter_(dseq, dfirst, self._idx0, accumulate=True)
        self._grads_final(("cls.", "bert.pooler."))          # both heads are done (they sit at the END of the flat buffer)
        # ---- encoder
        dx = dseq
        scale = 1.0 / math.sqrt(d)
        for l in range(cfg["layers"] - 1, -1, -1):
            layer, a = m.bert.encoder.layer[l], sv["layers"][l]
            pre = "bert.encoder.layer.%d." % l
            # dz2 flows unchanged into the residual branch; the dense branch sees it through the dropout mask.  With
            # dropout the LayerNorm backward writes both and the dense bias gradient (no dropout / column-sum passes)
            if a["mask_2"] is not None:
                dz2, do2 = F.dropout_add_layernorm_bwd(dx, a["z2"], a["ln2"][0], a["ln2"][1], layer.output.LayerNorm.weight.data,
                                                       a["mask_2"], self.p_hidden, self.gview[pre + "output.LayerNorm.weight"],
                                                       self.gview[pre + "output.LayerNorm.bias"],
                                                       dbias=self.gview[pre + "output.dense.bias"], accumulate=acc)
            else:
                dz2 = F.layernorm_bwd(dx, a["z2"], a["ln2"][0], a["ln2"][1], layer.output.LayerNorm.weight.data,
                                      self.gview[pre + "output.LayerNorm.weight"], self.gview[pre + "output.LayerNorm.bias"], acc)
                do2 = dz2
                self._bgrad(pre + "output.dense.bias", do2, acc)
            self._wgrad(pre + "output.dense.weight", do2, a["it"], acc)
            # ... times the stored GELU derivative; the epilogue also leaves the column sums = the bias gradient of dense_act
            dpre = F.gemm_colsum(do2, self.w16[pre + "output.dense.weight"], t, inter, h, a["pre"],
                                 self.gview[pre + "intermediate.dense_act.bias"], act=C.ACT_MUL, accumulate=acc) \
                if os.environ.get("DLE_BERT_FUSE_BIAS_GRAD", "1") != "0" else None
            fused_b = dpre is not None
            if not fused_b:
                dpre = F.gemm(do2, self.w16[pre + "output.dense.weight"], t, inter, h, True, False, act=C.ACT_MUL,
                              mask_src=a["pre"])
            self._wgrad(pre + "intermediate.dense_act.weight", dpre, a["x1"], acc)
            if not fused_b:
                self._bgrad(pre + "intermediate.dense_act.bias", dpre, acc)
            dx1 = F.gemm(dpre, self.w16[pre + "intermediate.dense_act.weight"], t, h, inter, True, False, act=C.ACT_ADD,
                         mask_src=dz2)
            if a["mask_1"] is not None:
                dz1, dao = F.dropout_add_layernorm_bwd(dx1, a["z1"], a["ln1"][0], a["ln1"][1],
                                                       layer.attention.output.LayerNorm.weight.data, a["mask_1"], self.p_hidden,
                                                       self.gview[pre + "attention.output.LayerNorm.weight"],
                                                       self.gview[pre + "attention.output.LayerNorm.bias"],
                                                       dbias=self.gview[pre + "attention.output.dense.bias"], accumulate=acc)
            else:
                dz1 = F.layernorm_bwd(dx1, a["z1"], a["ln1"][0], a["ln1"][1], layer.attention.output.LayerNorm.weight.data,
                                      self.gview[pre + "attention.output.LayerNorm.weight"],
                                      self.gview[pre + "attention.output.LayerNorm.bias"], acc)
                dao = dz1
                self._bgrad(pre + "attention.output.dense.bias", dao, acc)
            self._wgrad(pre + "attention.output.dense.weight", dao, a["ctx"], acc)
            dctx = F.gemm(dao, self.w16[pre + "attention.output.dense.weight"], t, h, h, True, False)
            qkv, probs = a["qkv"], a["probs"]
            if sv["fused_attn"]:
                cs = torch.empty((b * (s // 128), 3 * h), dtype=torch.float32, device=self.dev)     # per (sequence, 128-row block)
                dqkv = F.attention_bwd(qkv, dctx, self._mask_add, a["stats"], b, s, nh, scale, self.p_attn,
                                       self.rng_seed, a["off_a"], offset_base=self._rng_base, colsum_partial=cs,
                                       keep_mask=a["mask_a"] if (self.attn_keep_mask and s == 128) else None)
            else:
                dprobs = torch.empty_like(probs)
                F.gemm_batched(dctx, qkv[:, 2 * h:], dprobs, s, s, d, h, 3 * h, s, True, True, b * nh, nh,
                               (s * h, d), (s * 3 * h, d), (nh * s * s, s * s))
                if a["mask_a"] is not None:
                    F.softmax_dropout_bwd_(probs, dprobs, a["mask_a"], scale, self.p_attn)
                else:
                    F.softmax_bwd_(probs, dprobs, scale)
                dqkv = torch.empty((t, 3 * h), dtype=dt, device=self.dev)
                F.gemm_batched(dprobs, qkv[:, h:], dqkv, s, d, s, s, 3 * h, 3 * h, True, False, b * nh, nh,
                               (nh * s * s, s * s), (s * 3 * h, d), (s * 3 * h, d))                         # dQ = dS K
                F.gemm_batched(dprobs, qkv, dqkv[:, h:], s, d, s, s, 3 * h, 3 * h, False, False, b * nh, nh,
                               (nh * s * s, s * s), (s * 3 * h, d), (s * 3 * h, d))                         # dK = dS^T Q
                F.gemm_batched(a["pdrop"], dctx, dqkv[:, 2 * h:], s, d, s, s, h, 3 * h, False, False, b * nh, nh,
                               (nh * s * s, s * s), (s * h, d), (s * 3 * h, d))                             # dV = dropout(P)^T dO
            gq = self.gview[pre + "attention.self.query.weight"]
            gqkv = torch.as_strided(gq, (3 * h, h), (h, 1))
            gbq = self.gview[pre + "attention.self.query.bias"]
            csrc = cs if sv["fused_attn"] else dqkv
            # (fused path: the attention backward left per-sequence column sums [B, 3H]; fold those B rows instead of T)
            self._on_leaf_stream(lambda dqkv=dqkv, ax=a["x"], gqkv=gqkv, csrc=csrc, gbq=gbq: (
                F.gemm(dqkv, ax, 3 * h, h, t, False, False, out=gqkv, splitk=F.pick_splitk(3 * h, h, t, 1024), accumulate=acc),
                F.colsum(csrc, out=torch.as_strided(gbq, (3 * h,), (1,)), accumulate=acc)), dqkv, a["x"], csrc)
            dx = F.gemm(dqkv, layer.qkv16, t, h, 3 * h, True, False, act=C.ACT_ADD, mask_src=dz1)
            self._grads_final((pre,))
        # ---- embeddings
        # The tied decoder's weight gradient (first launch on the leaf stream, accumulate=False: it OVERWRITES the word-embedding
        # gradient) must have landed before the scatter-add below adds the lookup gradient into the same buffer: join the leaf
        # stream here, not after the embedding kernels (nothing ordered the two writers before; under graph capture there was no
        # edge between the two nodes).
        self._join_leaf_stream()                   # every weight / bias gradient has landed
        emb = m.bert.embeddings
        if sv["mask0"] is not None:
            dx = F.dropout_bwd(dx, sv["mask0"], self.p_hidden)
        dz0 = F.layernorm_bwd(dx, sv["z0"], sv["ln0"][0], sv["ln0"][1], emb.LayerNorm.weight.data,
                              self.gview["bert.embeddings.LayerNorm.weight"], self.gview["bert.embeddings.LayerNorm.bias"], acc)
        # lookup gradient of the word embeddings: rows of dz0 added into the (tied) embedding gradient.  Duplicate-free row update
        # (the DLRM sparse-update kernels with lr = -1: one 4-byte exchange per token threads the tokens of a vocabulary row into a
        # list, its head sums them and does ONE plain read-modify-write of the row) instead of 33 M fp32 atomics (0.88 ms / step)
        gw_word = self.gview["bert.embeddings.word_embeddings.weight"]
        if getattr(self, "_emb_ws", None) is None:
            self._emb_ws = F.EmbUpdateWorkspace([0, gw_word.shape[0]], h, self.dev)
        if os.environ.get("DLE_BERT_EMB_DEDUP", "1") != "0":
            F.emb_sgd_dedup_(gw_word, sv["ids"].view(-1, 1), dz0, self._emb_ws, lr=-1.0)
        else:
            F.embed_scatter_add_(gw_word, dz0, sv["ids"])
        gpos = self.gview["bert.embeddings.position_embeddings.weight"]
        if not acc:
            gpos[s:].zero_()
        F.colsum(dz0.view(b, s * h), out=gpos[:s].view(-1), accumulate=acc)
        F.rows_select_sum(dz0, sv["tt"], cfg["type_vocab"], self.gview["bert.embeddings.token_type_embeddings.weight"], acc)
        self._grads_final(("bert.embeddings.",))
        self._last_sv = sv if self.keep_activations else None     # tests read the dropout masks
        self._sv = None
        if self._rng_offset:
            self._rng_base += self._rng_offset     # on the device: the next step (or graph replay) draws new masks

    def _grads_final(self, prefixes):
        """The gradients of every parameter whose name starts with one of `prefixes` are complete: launch the
        all-reduce of the buckets they close (side stream, overlapped with the rest of the backward pass)."""
        if self.buckets is None or not self._reduce_now:
            return
        for n in self.names:
            if n.startswith(prefixes):
                self.buckets.grad_ready(n)

    # ------------------------------------------------------------------ optimizer
    @property
    def opt_steps(self):
        """Optimizer steps APPLIED so far = the LAMB step word on the device (one .item(): call it at log / checkpoint time
        only).  It does not advance on overflow-skipped steps and keeps advancing under HIP-graph replay, where host-side
        counters inside the captured step would freeze."""
        return int(self.step_t.item())

    def current_lr(self):
        """The rate the NEXT optimizer step will apply (logging): PolyWarmUpScheduler.step reads param_group['step'] + 1
        (schedulers.py:123-131), evaluated from the device step word, so the logged value follows skipped steps and graph
        replays exactly like the rate the kernels use (_device_lr)."""
        return poly_warmup_lr(self.opt_steps + 1, self.base_lr, self.warmup, self.total)

    def _device_lr(self):
        base, warm, total = self._lr_consts
        progress = (self.step_t.to(torch.float32) + 1.0) / total
        lr = torch.where(progress < warm, base * progress / warm, base * torch.clamp(1.0 - progress, min=0.0) ** 0.5)
        self.lr_t.copy_(lr.reshape(()))

    def _unused_norms(self, n):
        buf = getattr(self, "_norm_pad", None)
        if buf is None or buf.numel() < n:
            buf = self._norm_pad = torch.ones(n, dtype=torch.float32, device=self.dev)
        return buf

    def optimizer_step(self):
        sc = self.scaler
        if self.buckets is not None:
            # every bucket was launched during the backward pass of the last micro-step, on the communication stream, and
            # reduces flat_grad IN PLACE: nothing on the compute stream may touch the buffer before this wait
            self.buckets.wait()
        self.noop.zero_()
        if sc.enabled:
            F.check_nonfinite_(self.flat_grad, sc.found_inf)
            self.noop.copy_(sc.found_inf.to(torch.int32))
        self._device_lr()                # from step_t BEFORE it advances (lr_scheduler.step() precedes optimizer.step())
        self.step_t += (1 - self.noop)
        gnorm, _ = mt.l2norm(self.t_all_grads, self.noop)
        scale = sc.scale if sc.enabled else self.one
        inv = sc.inv_scale if sc.enabled else self.one
        if self.grad_divisor != 1:
            # each micro-step's loss is divided by the accumulation count in the reference (run_pretraining.py:521); here the
            # micro-steps are summed undivided and the count joins the loss scale (no pass over the 1.3 GB buffer): the
            # gradients LAMB sees are flat_grad * inv_scale / count, the norm test compares against max_norm * scale * count
            scale = scale * float(self.grad_divisor)
            inv = inv * (1.0 / self.grad_divisor)
        max_norm = self.max_norm_t * scale
        for key, tb in self.tables.items():
            if self.fuse_lamb_norms:
                # the per-tensor norms of p and of the update leave with stage 1 (two sweeps over 1.34 GB each less); a group
                # without weight decay steps with ratio = lr (multi_tensor_lamb.cu:277-282): its norms are never read
                if tb["wd"] != 0.0:
                    pn, un = mt.lamb_stage1_norms(tb["t_s1"], self.noop, 0.9, 0.999, 1.0 - 0.9, self.step_t, True, 1e-6, 1,
                                                  tb["wd"], gnorm, max_norm, inv)
                else:
                    mt.lamb_stage1(tb["t_s1"], self.noop, 0.9, 0.999, 1.0 - 0.9, self.step_t, True, 1e-6, 1, tb["wd"], gnorm,
                                   max_norm, inv)
                    pn = un = self._unused_norms(tb["t_s1"].n)
                mt.lamb_stage2(tb["t_s2"], self.noop, pn, un, self.lr_t, tb["wd"], False)
                continue
            _, pn = mt.l2norm(tb["t_p"], self.noop, per_tensor=True)
            mt.lamb_stage1(tb["t_s1"], self.noop, 0.9, 0.999, 1.0 - 0.9, self.step_t, True, 1e-6, 1, tb["wd"], gnorm,
                           max_norm, inv)
            _, un = mt.l2norm(tb["t_g"], self.noop, per_tensor=True)
            mt.lamb_stage2(tb["t_s2"], self.noop, pn, un, self.lr_t, tb["wd"], False)
        self.nsp_bias8[:2].copy_(self.model.cls.seq_relationship.bias.data)
        sc.update()

    def train_step(self, input_ids, token_type_ids, attention_mask, labels, next_sentence_labels):
        """One optimizer step on one micro-batch.  Returns the device-resident fp32 loss [1]."""
        loss, dlogits, dnsp = self.forward(input_ids, token_type_ids, attention_mask, labels, next_sentence_labels)
        self._reduce_now = True
        self.backward(dlogits, dnsp)
        self.optimizer_step()
        return loss
