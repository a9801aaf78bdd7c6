"""ResNet-50 train step on MI355X: forward, label-smoothing loss, backward, SGD with momentum.

Mirrors the reference's step (Classification/ConvNets/image_classification/):
    training.py:86-96     Executor._fwd_bwd_fn: autocast forward, loss / divide_loss, scaler.scale(loss).backward()
    training.py:167-186   Trainer.train_step: scaler.step(optimizer); scaler.update(); zero_grad; synchronize
    optimizers.py:34-56   SGD(momentum, nesterov, weight_decay), no weight decay for names containing "bn"
    optimizers.py:82-130  step / linear / cosine LR policies with linear warm-up (per epoch)
    utils.py:117-123      loss all-reduce across ranks (here: only when asked, never inside the step)
Data parallelism: one process per GPU, the fp32 gradient bucket(s) are all-reduced (mean) with
torch.distributed (backend nccl = RCCL over xGMI) on a side stream, bucket by bucket while the backward of the
earlier layers is still running (the reference uses torch DDP's reducer, training.py:78-84).
The step is a fixed kernel sequence (HIP-graph capturable): activations NHWC 16-bit, statistics and master
weights fp32, gradients fp32 in ONE flat buffer whose element order matches the parameters' memory order.
"""
import math
from typing import Optional

import numpy as np
import os

import torch
import torch.distributed as dist

from .. import functional as F
from .. import multi_tensor as mt
from ..dlrm.engine import GradScalerState
from ..utils.buckets import GradBuckets
from ..utils import comm
from .resnet import Deferred, ResNet50


def lr_cosine_policy(base_lr, warmup_length, epochs, end_lr=0.0):
    def fn(iteration, epoch):
        if epoch < warmup_length:
            return base_lr * (epoch + 1) / warmup_length
        e, es = epoch - warmup_length, epochs - warmup_length
        return end_lr + 0.5 * (1 + np.cos(np.pi * e / es)) * (base_lr - end_lr)
    return fn


def lr_step_policy(base_lr, steps, decay_factor, warmup_length):
    def fn(iteration, epoch):
        if epoch < warmup_length:
            return base_lr * (epoch + 1) / warmup_length
        lr = base_lr
        for s in steps:
            if epoch >= s:
                lr *= decay_factor
        return lr
    return fn


def lr_linear_policy(base_lr, warmup_length, epochs):
    def fn(iteration, epoch):
        if epoch < warmup_length:
            return base_lr * (epoch + 1) / warmup_length
        return base_lr * (1 - (epoch - warmup_length) / (epochs - warmup_length))
    return fn


class ResNetTrainer:
    def __init__(self, model: ResNet50, lr: float, momentum=0.875, weight_decay=3.0517578125e-05, nesterov=False,
                 label_smoothing=0.1, compute_dtype=torch.bfloat16, bn_weight_decay=False, static_loss_scale=1.0,
                 world_size=1, process_group=None, bucket_mb=25, grad_acc_steps=1):
        self.model = model
        # gradient accumulation (main.py:405-416 batch_size_multiplier -> training.py divide_loss / grad_acc_steps): the optimizer
        # steps every grad_acc_steps calls of train_step on the sum of the micro-batch gradients, each divided by grad_acc_steps
        self.grad_acc_steps = max(1, int(grad_acc_steps))
        self.steps_since_update = 0
        self.flat_acc = None
        self.dev = model.fc.weight.device
        self.dtype = compute_dtype
        self.momentum, self.wd, self.nesterov, self.smoothing = momentum, weight_decay, nesterov, label_smoothing
        self.world, self.pg = world_size, process_group
        self.bn_weight_decay = bn_weight_decay
        if world_size > 1:
            # data-parallel replicas start from rank 0's weights and BatchNorm buffers (torch DDP's constructor does
            # this in the reference, training.py:78-84) -- whatever seed each rank built its model with
            comm.broadcast_parameters_(list(model.parameters()) + list(model.buffers()), 0, process_group)
        self.stem, self.blocks = model.units()
        self.scaler = GradScalerState(self.dev, enabled=compute_dtype == torch.float16, init_scale=static_loss_scale,
                                      growth_interval=int(1e9))   # static scale, as configs.yml AMP (128)
        self.lr = torch.full((1,), lr, dtype=torch.float32, device=self.dev)
        self.first_step = True
        self.steps_done = 0
        # ---- parameters in backward-completion order (fc first ... stem last) for bucketed all-reduce
        named = dict(model.named_parameters())
        order = ["fc.weight", "fc.bias"]
        for (u1, u2, u3, ud) in reversed(self.blocks):
            for u in ([u3, ud] if ud is not None else [u3]) + [u2, u1]:
                order += [u.name_conv + ".weight", u.name_bn + ".weight", u.name_bn + ".bias"]
        order += ["conv1.weight", "bn1.weight", "bn1.bias"]
        assert sorted(order) == sorted(named), "parameter bookkeeping out of sync with the module tree"
        self.names = order
        self.params = [named[n] for n in order]
        total = sum(p.numel() for p in self.params)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=self.dev)
        self.flat_mom = torch.zeros(total, dtype=torch.float32, device=self.dev)
        self.gview, self.mview, self.offset = {}, {}, {}
        o = 0
        for n, p in zip(order, self.params):
            self.offset[n] = o
            self.gview[n] = self.flat_grad[o:o + p.numel()]
            self.mview[n] = self.flat_mom[o:o + p.numel()]
            o += p.numel()
        self.comm_stream = torch.cuda.Stream(device=self.dev) if world_size > 1 else None
        self.buckets = GradBuckets(self.flat_grad, [(n, p.numel()) for n, p in zip(order, self.params)], bucket_mb,
                                   process_group, self.comm_stream) if world_size > 1 else None
        # ---- 16-bit working copies (KRSC order == the channels_last master's memory order)
        self.w16 = {}
        units = [self.stem] + [u for blk in self.blocks for u in blk if u is not None]
        self.units = units
        for u in units:
            w = named[u.name_conv + ".weight"]
            ko, ci, r, s = w.shape
            cp = (ci + 7) // 8 * 8
            u.w16 = torch.zeros((ko, r, s, cp), dtype=compute_dtype, device=self.dev)
            self.w16[u.name_conv + ".weight"] = u.w16
            u.ggamma = self.gview[u.name_bn + ".weight"]
            u.gbeta = self.gview[u.name_bn + ".bias"]
            if cp == ci:
                u.gw = self.gview[u.name_conv + ".weight"].view(ko, r, s, ci)
            else:                                   # stem: gradient of the channel-padded weight, cropped later
                u.gw = torch.zeros((ko, r, s, cp), dtype=torch.float32, device=self.dev)
        # weight gradients on a second stream (resnet.ConvBN.backward); DLE_RN50_WGRAD_STREAM=0 keeps everything on one stream
        self.wgrad_stream = torch.cuda.Stream(device=self.dev) if (self.dev.type == "cuda" and
                                                                   os.environ.get("DLE_RN50_WGRAD_STREAM", "1") != "0") else None
        self._wgrad_keepalive = []
        for u in units:
            u.wgrad_stream, u.keepalive = self.wgrad_stream, self._wgrad_keepalive
        # the downsample branch of a layer's first block (1x1 convolution + BatchNorm beside conv1 -> conv2 -> conv3) on a third
        # stream, forward and backward; DLE_RN50_BRANCH_STREAM=0 keeps it in line
        self.branch_stream = torch.cuda.Stream(device=self.dev) if (self.dev.type == "cuda" and
                                                                    os.environ.get("DLE_RN50_BRANCH_STREAM", "1") != "0") else None
        self._branch_keep = []
        if self.buckets is not None:
            self.buckets.extra_streams += [st for st in (self.wgrad_stream, self.branch_stream) if st is not None]
        # the stem's own kernels (csrc/stem.hip) on a 4-channel image, for inputs up to 224 pixels wide; DLE_RN50_STEM4=0 keeps
        # the generic implicit-GEMM path (8-channel image) -- which wider inputs take in any case
        self.stem4 = os.environ.get("DLE_RN50_STEM4", "1") != "0"
        # the stride-2 downsample branch's data gradient stays on its own grid and is added at the even pixels by conv1's
        # data-gradient kernel (no zero-stuffed tensor); DLE_RN50_FUSE_UP2=0 materialises it as before
        self.fuse_up2 = os.environ.get("DLE_RN50_FUSE_UP2", "1") != "0"
        # BatchNorm-apply of bn2 / bn3 on the operand load of the consuming 1x1 convolution (conv + BN + ReLU as one unit,
        # csrc/conv_bnload.hip); DLE_RN50_FUSE_BN=0 keeps the stand-alone apply passes
        self.fuse_bn = os.environ.get("DLE_RN50_FUSE_BN", "1") != "0"
        # the backward reduction of a block's bn3 taken in the epilogue of the NEXT block's conv1 data gradient (gemm_expand BRED)
        self.fuse_bnred = os.environ.get("DLE_RN50_FUSE_BNRED", "1") != "0"
        # the downsample branch's BatchNorm (no ReLU) applied where bn3's apply LOADS the residual (bn_apply2_pf_kernel, conv_bnload
        # RES = 2): the branch's 16-bit output is never written; DLE_RN50_FUSE_DSBN=0 keeps its stand-alone apply pass
        self.fuse_dsbn = os.environ.get("DLE_RN50_FUSE_DSBN", "1") != "0"
        # ... and in the backward direction: bn3 and the downsample BatchNorm of a layer's first block receive the SAME gradient
        # (g under the block's output keep bits); their two reductions read g and the mask once (bn_reduce2_kernel);
        # DLE_RN50_FUSE_DSRED=0 keeps the two launches
        self.fuse_dsred = os.environ.get("DLE_RN50_FUSE_DSRED", "1") != "0"
        # the stem's backward chain max pooling -> ReLU -> BatchNorm without the pooling gradient's full-resolution tensor;
        # DLE_RN50_FUSE_POOLBWD=0 keeps dle_maxpool_bwd + the two BatchNorm passes
        self.fuse_pool_bwd = os.environ.get("DLE_RN50_FUSE_POOLBWD", "1") != "0"
        self.stem.w2 = torch.zeros((64, 7, 8, 4), dtype=compute_dtype, device=self.dev)
        self.stem.gw_flat = self.gview["conv1.weight"]
        self.fc_w16 = torch.empty(model.fc.weight.shape, dtype=compute_dtype, device=self.dev)
        self.w16["fc.weight"] = self.fc_w16
        self.refresh_working_copies()
        self._build_tables()

    # ------------------------------------------------------------------ parameter plumbing
    @staticmethod
    def _phys(p):
        """Flat view of a parameter in MEMORY order (channels_last conv weights -> KRSC)."""
        if p.dim() == 4:
            return p.data.permute(0, 2, 3, 1).reshape(-1)
        return p.data.reshape(-1)

    def refresh_working_copies(self):
        for u in self.units:
            w = dict(self.model.named_parameters())[u.name_conv + ".weight"]
            ko, ci, r, s = w.shape
            F.cast_rows(self._phys(w).view(ko * r * s, ci), self.dtype, cols_out=u.w16.shape[-1],
                        out=u.w16.view(ko * r * s, -1))
        F.cast(self.model.fc.weight.data, self.dtype, out=self.fc_w16)
        F.stem_pack_weight(self.model.conv1.weight.data, self.dtype, out=self.stem.w2)

    def _stem_image(self, images):
        """Input batch -> the 16-bit NHWC image the stem reads: 4 channels (8 bytes per pixel) for its own kernels, 8 for the
        generic convolution path (DLE_RN50_STEM4=0, or images wider than 224 pixels)."""
        cp = 4 if (self.stem4 and images.shape[-1] <= 224) else 8
        if images.dtype == torch.uint8:
            # decoded images straight from the loader: normalisation fused with the layout change (dataloaders.py:354-384)
            if getattr(self, "_mean_std", None) is None:
                from .dataloaders import IMAGENET_MEAN, IMAGENET_STD
                self._mean_std = (torch.tensor(IMAGENET_MEAN, device=self.dev) * 255.0, torch.tensor(IMAGENET_STD, device=self.dev) * 255.0)
            return F.u8_nchw_normalize_nhwc(images, self._mean_std[0], self._mean_std[1], self.dtype, cp)
        return F.nchw_to_nhwc(images, self.dtype, cp)

    def _build_tables(self):
        named = dict(zip(self.names, self.params))
        decay_copy, decay_plain, nodecay = ([], [], [], []), ([], [], []), ([], [], [])
        for n, p in named.items():
            g, m, ph = self.gview[n], self.mview[n], self._phys(p)
            assert ph.data_ptr() == p.data_ptr(), "parameter %s is not dense in memory order" % n
            if "bn" in n and not self.bn_weight_decay:                 # optimizers.py:42-43 (name based; --bn-weight-decay)
                for lst, t in zip(nodecay, (g, ph, m)):
                    lst.append(t)
            elif n in self.w16 and self.w16[n].numel() == p.numel():
                for lst, t in zip(decay_copy, (g, ph, m, self.w16[n].view(-1))):
                    lst.append(t)
            else:
                for lst, t in zip(decay_plain, (g, ph, m)):
                    lst.append(t)
        self.t_decay_copy = mt.TensorTable(list(decay_copy))
        self.t_decay_plain = mt.TensorTable(list(decay_plain))
        self.t_nodecay = mt.TensorTable(list(nodecay))

    def set_lr(self, lr: float):
        self.lr.fill_(lr)

    # ------------------------------------------------------------------ communication
    def set_side_streams(self, enabled):
        """Second / third stream on or off (off: every kernel in line on the current stream -- bench.py's per-kernel timing pass)."""
        if not hasattr(self, "_streams_saved"):
            self._streams_saved = (self.wgrad_stream, self.branch_stream)
        torch.cuda.synchronize(self.dev)
        self.wgrad_stream, self.branch_stream = self._streams_saved if enabled else (None, None)
        for u in self.units:
            u.wgrad_stream = self.wgrad_stream

    def _maybe_reduce(self, finished_param_name):
        """Launch the all-reduce of a gradient bucket once its last gradient has been produced."""
        if self.buckets is not None and getattr(self, "_reduce_now", True):
            self.buckets.grad_ready(finished_param_name)

    # ------------------------------------------------------------------ the step
    def _input(self, images):
        if images.dim() == 4 and not images.is_contiguous():
            images = images.contiguous()         # --memory-format nhwc loaders hand over channels_last tensors
        return images

    def infer(self, images):
        """Evaluation-mode forward (Executor.forward under model.eval(), training.py:98-105): running BatchNorm statistics,
        no state kept.  -> fp32 logits [N, classes]."""
        x = self._stem_image(self._input(images))
        h, _ = F.maxpool_fwd(self.stem.forward_eval(x))
        for (u1, u2, u3, ud) in self.blocks:
            res = ud.forward_eval(h) if ud is not None else h
            h = u3.forward_eval(u2.forward_eval(u1.forward_eval(h)), residual=res)
        pooled = F.avgpool_fwd(h)
        return F.gemm(pooled, self.fc_w16, pooled.shape[0], self.fc_w16.shape[0], self.fc_w16.shape[1], True, True,
                      out_dtype=torch.float32, bias=self.model.fc.bias.data)

    def eval_step(self, images, target):
        """-> (loss [1] (plain cross entropy, as NLLMultiLabelSmooth / LabelSmoothing / CrossEntropyLoss evaluate), logits)."""
        logits = self.infer(images)
        loss, _ = F.softmax_xent(logits, target, smoothing=0.0)
        return loss, logits

    def forward(self, images):
        """images fp32 NCHW (as produced by the reference's loaders) -> fp32 logits [N, classes]."""
        x = self._stem_image(self._input(images))
        d0 = self.stem.forward(x, defer=self.fuse_bn)
        if isinstance(d0, Deferred) and d0.t.shape[1] % 2 == 0 and d0.t.shape[2] % 2 == 0:
            # bn1 + ReLU + max pooling in one pass: the stem's 16-bit activation is never written (csrc/convnet.hip)
            m0, self._amax, mask0 = F.bn_relu_maxpool_fwd(d0.t, d0.mean, d0.rstd, self.stem.bn.weight.data, self.stem.bn.bias.data)
            d0._done(None, mask0)
            self._pool_in_hw = d0.t.shape[1:3]
        else:
            a0 = d0.materialize() if isinstance(d0, Deferred) else d0
            m0, self._amax = F.maxpool_fwd(a0)
            self._pool_in_hw = a0.shape[1:3]
        h = m0
        bs = self.branch_stream
        fuse = self.fuse_bn
        for (u1, u2, u3, ud) in self.blocks:
            # conv + BN + ReLU as one unit where the consumer is a 1x1 convolution: bn2's apply runs inside conv3's kernel, bn3's
            # (+ residual) inside the NEXT block's conv1 (csrc/conv_bnload.hip); `h` is then a Deferred whose applied form `h.y`
            # exists after u1 has consumed it (side output of the fused kernel, or the stand-alone pass)
            o1 = u1.forward(h)
            hy = h.y if isinstance(h, Deferred) else h
            if ud is not None and bs is not None:
                cur = torch.cuda.current_stream()
                bs.wait_stream(cur)
                self._branch_keep.clear()        # (what the branch stream produced earlier has been consumed before this point)
                with torch.cuda.stream(bs):
                    res = ud.forward(hy, defer=self.fuse_dsbn)
                o2 = u2.forward(o1, defer=fuse)
                cur.wait_stream(bs)
                self._branch_keep.append(res)    # allocated on the branch stream, read on this one: alive until the next fork
            else:
                res = ud.forward(hy, defer=self.fuse_dsbn) if ud is not None else hy
                o2 = u2.forward(o1, defer=fuse)
            h = u3.forward(o2, residual=res, defer=fuse)
        if isinstance(h, Deferred):
            h = h.materialize()
        self._feat_hw = h.shape[1:3]
        self._pooled = F.avgpool_fwd(h)
        n = self._pooled.shape[0]
        logits = F.gemm(self._pooled, self.fc_w16, n, self.fc_w16.shape[0], self.fc_w16.shape[1], True, True,
                        out_dtype=torch.float32, bias=self.model.fc.bias.data)
        return logits

    def backward(self, dlogits):
        n = dlogits.shape[0]
        fcw = self.fc_w16
        if self.buckets is not None:
            self.buckets.fired = 0               # (a backward pass abandoned half way had launched some of its buckets)
        F.gemm(dlogits, self._pooled, fcw.shape[0], fcw.shape[1], n, False, False,
               out=self.gview["fc.weight"].view(fcw.shape), splitk=F.pick_splitk(fcw.shape[0], fcw.shape[1], n))
        F.colsum(dlogits, out=self.gview["fc.bias"])
        self._maybe_reduce("fc.weight"); self._maybe_reduce("fc.bias")
        gp = F.gemm(dlogits, fcw, n, fcw.shape[1], fcw.shape[0], True, False)
        g = F.avgpool_bwd(gp, self._feat_hw)
        for blk in self.blocks:                  # (a backward pass that was abandoned half way must not leave a unit believing its
            for u in blk:                        #  BatchNorm reduction has been taken by its neighbour)
                if u is not None:
                    u.reduce_done = False
        for bi in range(len(self.blocks) - 1, -1, -1):
            u1, u2, u3, ud = self.blocks[bi]
            # the block ends in relu(bn3(conv3) + shortcut): g * (out > 0) flows into BOTH branches.  It is never written:
            # bn3's backward applies the mask on load, the shortcut side gets (g, mask) and applies it where it is consumed
            mask3 = u3.relu_mask()
            if ud is not None and self.fuse_dsred and u3.saved is not None and ud.saved is not None and mask3 is not None \
                    and not u3.reduce_done and u3.saved[1].shape == ud.saved[1].shape:
                F.bn_bwd_reduce2(g, mask3, u3.saved[1], u3.saved[3], u3.saved[4], u3.ggamma, u3.gbeta,
                                 ud.saved[1], ud.saved[3], ud.saved[4], ud.ggamma, ud.gbeta)
                u3.reduce_done = ud.reduce_done = True
            bs = self.branch_stream if ud is not None else None
            if bs is not None:                   # the downsample branch's backward beside bn3 / conv3 / conv2's
                cur = torch.cuda.current_stream()
                bs.wait_stream(cur)
                self._branch_keep.clear()
                with torch.cuda.stream(bs):
                    gskip = ud.backward(g, dy_mask=mask3, compact_dx=self.fuse_up2)
                self._branch_keep.append((gskip, g))
            g3 = u3.backward(g, bnred=u2 if self.fuse_bnred else None)     # (bn2's reduction rides on conv3's fused data gradient)
            self._done(u3)
            if ud is not None and bs is None:
                gskip = ud.backward(g, dy_mask=mask3, compact_dx=self.fuse_up2)
            elif ud is None:
                gskip = (g, mask3)
            g2 = u2.backward(g3)
            self._done(u2)
            if bs is not None:
                cur.wait_stream(bs)
            if ud is not None:
                self._done(ud)
            # (no downsample branch: g is the gradient of the previous block's output -- its bn3 reduction rides on this GEMM)
            prev3 = self.blocks[bi - 1][2] if (self.fuse_bnred and ud is None and bi > 0) else None
            g = u1.backward(g2, dx_addend=gskip, bnred=prev3)
            self._done(u1)
        if self.fuse_pool_bwd:
            # the stem's BatchNorm backward gathers the pooling gradient from (g, argmax) itself: its 411 MB full-resolution form
            # (batch 256) is never written or re-read (csrc/convnet.hip pool_bn_bwd_kernel)
            self.stem.backward(None, need_dx=False, pooled=(g, self._amax))
        else:
            g = F.maxpool_bwd(g, self._amax, self._pool_in_hw)
            self.stem.backward(g, need_dx=False)
        gw = self.stem.gw
        ko, r, s, cp = gw.shape
        stem_generic = self.stem.saved_c != 4        # (set by the stem's backward: which image layout this step ran on)
        if self.wgrad_stream is not None:            # every weight gradient has landed before anything reads the flat buffer
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)
            self._wgrad_keepalive.clear()
        self._branch_keep.clear()
        if stem_generic:                             # gradient of the channel-padded weight, cropped into the flat buffer
            F.copy_rows(gw.view(ko * r * s, cp)[:, :3], self.gview["conv1.weight"].view(ko * r * s, 3))
        self._done(self.stem)

    def _done(self, u):
        for suffix, nm in ((".weight", u.name_conv), (".weight", u.name_bn), (".bias", u.name_bn)):
            self._maybe_reduce(nm + suffix)

    def optimizer_step(self):
        sc = self.scaler
        skip = sc.found_inf if sc.enabled else None
        inv = sc.inv_scale if sc.enabled else None
        kw = dict(momentum=self.momentum, nesterov=self.nesterov, first_step=self.first_step, skip_flag=skip,
                  inv_scale=inv, has_momentum=self.momentum != 0)
        mt.sgd(self.t_decay_copy, self.lr, weight_decay=self.wd, model_copy=True, **kw)
        mt.sgd(self.t_decay_plain, self.lr, weight_decay=self.wd, **kw)
        mt.sgd(self.t_nodecay, self.lr, weight_decay=0.0, **kw)
        # stem weight: refresh its working copies from the master (packed 4-channel form for the stem kernels; the channel-padded
        # 8-channel form only when the generic path is in use)
        u = self.stem
        w = self.model.conv1.weight
        F.stem_pack_weight(w.data, self.dtype, out=u.w2)
        ko, ci, r, s = w.shape                       # (the 8-channel copy serves inputs wider than 224 pixels / DLE_RN50_STEM4=0)
        F.cast_rows(self._phys(w).view(ko * r * s, ci), self.dtype, cols_out=u.w16.shape[-1], out=u.w16.view(ko * r * s, -1))
        self.first_step = False

    def train_step(self, images, target):
        """One call of the reference's Trainer.train_step (training.py:167-186): forward + backward of one (micro-)batch and,
        every grad_acc_steps calls, the optimizer step.  Returns the device-resident fp32 loss [1] (no host sync), divided by
        grad_acc_steps like the reference's `loss /= divide_loss`."""
        sc = self.scaler
        acc = self.grad_acc_steps
        self.steps_since_update += 1
        last = self.steps_since_update == acc
        logits = self.forward(images)
        gs = sc.scale if sc.enabled else None
        if isinstance(target, tuple):
            # --mixup (mixup.py:19-69): targets c * onehot(y) + (1 - c) * onehot(y[perm]); NLLMultiLabelSmooth is linear in the
            # target, so loss = c * L(y) + (1 - c) * L(y[perm]) with the label-smoothing loss L, and so is its gradient
            ya, yb, lam = target
            la, da = F.softmax_xent(logits, ya, smoothing=self.smoothing, grad_scale=gs, grad_dtype=torch.float32)
            lb, db = F.softmax_xent(logits, yb, smoothing=self.smoothing, grad_scale=gs, grad_dtype=torch.float32)
            loss = lam * la + (1.0 - lam) * lb
            F.axpby_(da.view(-1), db.view(-1), da.view(-1), lam, 1.0 - lam)
            dlogits = F.cast(da, self.dtype)
        else:
            loss, dlogits = F.softmax_xent(logits, target, smoothing=self.smoothing, grad_scale=gs, grad_dtype=self.dtype)
        self._reduce_now = acc == 1                 # buckets fire during backward only without accumulation
        self.backward(dlogits)
        if acc > 1:
            # micro-batch gradients meet in a second flat buffer: flat_acc = sum_k flat_grad_k / acc (the backward kernels
            # overwrite flat_grad, and BatchNorm's backward needs the per-micro-batch dgamma / dbeta it has just written)
            if self.flat_acc is None:
                self.flat_acc = torch.empty_like(self.flat_grad)
            first = self.steps_since_update == 1
            dst = self.flat_grad if last else self.flat_acc
            F.axpby_(self.flat_grad, None if first else self.flat_acc, dst, 1.0 / acc, 0.0 if first else 1.0)
            loss = loss / acc
            if not last:
                self.steps_done += 1
                return loss
            if self.world > 1:
                comm.allreduce_mean_(self.flat_grad, self.pg)          # one reduction per optimizer step
        elif self.buckets is not None:
            self.buckets.wait()
        if sc.enabled:
            F.check_nonfinite_(self.flat_grad, sc.found_inf)
        self.optimizer_step()
        sc.update()
        self.steps_since_update = 0
        self.steps_done += 1          # BatchNorm.num_batches_tracked is materialised by sync_counters()
        return loss

    def reduced_loss(self, loss):
        """Mean of the per-rank losses for logging (utils.py:117-123 reduce_tensor); not part of the step."""
        if self.world > 1:
            loss = comm.allreduce_mean_(loss.clone(), self.pg)
        return loss

    def sync_counters(self):
        """Write the step count into every BatchNorm's num_batches_tracked (before saving a checkpoint)."""
        for u in self.units:
            u.bn.num_batches_tracked.fill_(self.steps_done)
