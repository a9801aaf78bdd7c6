"""ResNet-50 v1.5 on the MI355X kernels: parameter container with the reference's module tree / state_dict
names + explicit forward/backward over NHWC 16-bit activations.

Mirrors (Classification/ConvNets/image_classification/):
    models/resnet.py:107-175   Bottleneck (1x1 -> 3x3(stride) -> 1x1, BN after each, ReLU, `out += residual`)
    models/resnet.py:211-322   ResNet: stem 7x7/2 + BN + ReLU + MaxPool(3,2,1), layers [3,4,6,3], avgpool, fc
    models/common.py:31-128    LayerBuilder: Conv2d(bias=False, padding=k//2, kaiming_normal fan_in), BN gamma 1 / 0
The nn.Conv2d / nn.BatchNorm2d / nn.Linear submodules exist only to own the parameters and buffers under the
reference's names (checkpoint compatible); conv weights are kept in channels_last memory so that the fp32
master, its gradient and the 16-bit working copy all share the KRSC element order the kernels read.
"""
from contextlib import nullcontext as _nullcontext
from typing import List

import os

import torch
from torch import nn

from .. import _cabi as C
from .. import functional as F

LAYERS, WIDTHS, EXPANSION = [3, 4, 6, 3], [64, 128, 256, 512], 4


def _conv(cin, cout, k, stride, device):
    m = nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=(k - 1) // 2, bias=False, device=device)
    nn.init.kaiming_normal_(m.weight, mode="fan_in", nonlinearity="relu")
    m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    return m


def _bn(c, device, zero_init=False):
    m = nn.BatchNorm2d(c, device=device)
    nn.init.constant_(m.weight, 0 if zero_init else 1)
    nn.init.constant_(m.bias, 0)
    return m


class Deferred:
    """A unit's output BEFORE its BatchNorm-apply: the convolution result t, the batch statistics, and the residual the apply will
    add.  The consumer either runs the apply inside its own convolution kernel (1x1 consumers: csrc/conv_bnload.hip -- conv + BN +
    ReLU as one unit, models/resnet.py:148-175) or materialises it with the stand-alone pass.  Either way `y` (the applied 16-bit
    output) and the unit's ReLU keep bits exist afterwards: the residual / downsample branches and the backward pass read them."""
    __slots__ = ("unit", "t", "mean", "rstd", "residual", "y")

    def __init__(self, unit, t, mean, rstd, residual):
        self.unit, self.t, self.mean, self.rstd, self.residual, self.y = unit, t, mean, rstd, residual, None

    def residual_args(self):
        """-> (residual tensor or None, its BatchNorm (mean, rstd, gamma, beta) or None).  A residual that is itself a Deferred is
        the downsample branch (conv -> BatchNorm, no ReLU, models/resnet.py:166-173): its BatchNorm is applied where the residual
        is LOADED (bn_apply2_pf_kernel / conv_bnload_kernel<RES = 2>) and its 16-bit output never exists."""
        r = self.residual
        if isinstance(r, Deferred):
            return r.t, (r.mean, r.rstd, r.unit.bn.weight.data, r.unit.bn.bias.data)
        return r, None

    def _done(self, y, mask):
        u = self.unit
        u.saved = (u.saved[0], self.t, mask, self.mean, self.rstd)
        self.y = y

    def materialize(self):
        if self.y is None:
            u = self.unit
            res, res_bn = self.residual_args()
            y, mask = F.bn_fwd_apply(self.t, self.mean, self.rstd, u.bn.weight.data, u.bn.bias.data, residual=res,
                                     relu=u.relu, want_mask=True, residual_bn=res_bn)
            self._done(y, mask)
        return self.y


class ConvBN:
    """One conv + BN (+ ReLU) (+ residual) unit of the step: forward keeps what backward needs."""

    def __init__(self, conv: nn.Conv2d, bn: nn.BatchNorm2d, name_conv: str, name_bn: str, relu: bool):
        self.conv, self.bn, self.name_conv, self.name_bn, self.relu = conv, bn, name_conv, name_bn, relu
        self.k = conv.kernel_size[0]
        self.stride, self.pad = conv.stride[0], conv.padding[0]
        self.cin, self.cout = conv.in_channels, conv.out_channels
        self.w16 = None                 # [Ko, R, S, Cp] 16-bit working copy (Cp = cin padded to 8)
        self.gw = self.ggamma = self.gbeta = None   # fp32 gradient views (set by the trainer)
        self.saved = None
        # the stem (7x7 / 2 on a 3-channel image) has its own kernels on a 4-channel image (csrc/stem.hip); the trainer sets
        # w2 (packed [64, 7, 8, 4] weights) and gw_flat (the flat fp32 gradient view in the master's KRSC order)
        self.w2 = self.gw_flat = None
        self.wgrad_stream = None                    # set by the trainer: weight gradients run beside the data-gradient chain
        self.fuse_bnbwd = os.environ.get("DLE_RN50_FUSE_BNBWD", "1") != "0"      # BatchNorm backward on the data gradient's operand load
        self.reduce_done = False                    # set by the unit that PRODUCED this unit's output gradient (backward(bnred=))
        self.keepalive = None                       # ... with the list that keeps their operands alive until the streams join

    def forward(self, x, residual=None, defer=False):
        """x: the input tensor, or the producer's Deferred output -- a 1x1 stride-1 unit then applies the producer's BatchNorm
        (+ residual) + ReLU on its own operand load (one kernel: csrc/conv_bnload.hip) where the shape allows, and materialises
        it otherwise.  defer: return this unit's output as a Deferred (conv + statistics done, apply left to the consumer)."""
        # conv + batch statistics in one pass over the activation (the statistics come out of the convolution
        # epilogue), then normalise + residual + ReLU; the backward pass rebuilds the ReLU mask from 1 bit per element
        t = None
        if isinstance(x, Deferred):
            d, r = x, None
            if self.k == 1 and self.stride == 1 and d.unit.relu and d.y is None:
                res, res_bn = d.residual_args()
                r = F.conv1x1_bnload_fwd(d.t, res, self.w16, d.mean, d.rstd, d.unit.bn.weight.data, d.unit.bn.bias.data,
                                         self.bn.running_mean, self.bn.running_var, eps=self.bn.eps, momentum=self.bn.momentum,
                                         res_bn=res_bn)
            if r is not None:
                t, x, bits, mean, rstd = r
                d._done(x, bits)
            else:
                x = d.materialize()
        if t is None:
            if x.shape[-1] == 4:                    # stem on its 4-channel image
                t, mean, rstd = F.stem_conv_fwd_bnstats(x, self.w2, self.bn.running_mean, self.bn.running_var, eps=self.bn.eps,
                                                        momentum=self.bn.momentum)
            else:
                t, mean, rstd = F.conv2d_fwd_bnstats(x, self.w16, self.stride, self.pad, self.bn.running_mean,
                                                     self.bn.running_var, eps=self.bn.eps, momentum=self.bn.momentum)
        if defer:
            self.saved = (x, t, None, mean, rstd)
            return Deferred(self, t, mean, rstd, residual)
        if isinstance(residual, Deferred):          # (the downsample branch left deferred, this unit not: apply both here)
            d = Deferred(self, t, mean, rstd, residual)
            self.saved = (x, t, None, mean, rstd)
            return d.materialize()
        y, mask = F.bn_fwd_apply(t, mean, rstd, self.bn.weight.data, self.bn.bias.data, residual=residual,
                                 relu=self.relu, want_mask=True)
        self.saved = (x, t, mask, mean, rstd)
        return y

    def forward_eval(self, x, residual=None):
        """Inference-mode unit (model.eval(): BatchNorm normalises with its running statistics, nothing is saved) -- the
        validation pass of training.py:257-311."""
        t = F.stem_conv_fwd(x, self.w2, want_stats=False)[0] if x.shape[-1] == 4 else F.conv2d_fwd(x, self.w16, self.stride, self.pad)
        rstd = torch.rsqrt(self.bn.running_var + self.bn.eps)
        y, _ = F.bn_fwd_apply(t, self.bn.running_mean, rstd, self.bn.weight.data, self.bn.bias.data, residual=residual,
                              relu=self.relu, want_mask=False)
        return y

    def backward(self, dy, need_dx=True, dx_addend=None, dy_mask=None, compact_dx=False, bnred=None, pooled=None):
        """dy: gradient w.r.t. the unit's output; dy_mask (optional): bit-packed keep bits to apply to dy first (the
        ReLU that follows the residual add sits on the OTHER branch's unit: its mask gates this branch's gradient too).
        dx_addend: a tensor, or (tensor, keep bits) = the residual-branch gradient dy * (y > 0) that is never
        materialised: the data-gradient GEMM adds it under the mask in its epilogue -- or ("up2", compact, (H, W)) = the gradient
        of a stride-2 1x1 branch on its own P x Q grid, added at the even pixels without its zero-stuffed form being written.
        compact_dx (1x1 stride-2 units): return that triple instead of the full-resolution dx.
        bnred (with a masked dx_addend): the conv + BN unit whose output gradient this dx IS (the previous block's conv3 / bn3) --
        its BatchNorm's backward reduction is taken in the epilogue that produces dx, and that unit's backward() skips it.
        pooled (the stem; dy is then None): (gradient of the max pooling's OUTPUT, argmax) -- the BatchNorm backward gathers the
        pooling gradient itself (F.pool_bn_bwd); outside that kernel's envelope the pooling backward runs here.
        Returns dx or None."""
        x, t, mask, mean, rstd = self.saved
        self.saved = None
        self.saved_c = x.shape[-1]
        rmask = mask if self.relu else dy_mask
        n, h, w, c = x.shape
        # conv3 / bn3 of the 56 x 56 stage: the BatchNorm backward runs on the operand load of the unit's own data gradient
        # (csrc/conv_bnbwd.hip) -- dt is written once for the weight gradient and never read back by the data gradient
        reduce_done, self.reduce_done = self.reduce_done, False
        fused = None
        if self.fuse_bnbwd and self.k == 1 and self.stride == 1 and need_dx and dx_addend is None:   # (compact_dx: stride 2 only)
            b2 = None
            if bnred is not None and bnred.saved is not None and bnred.relu:
                # dx of this unit is the gradient entering `bnred`'s BatchNorm (bn2 of the bottleneck): its reduction rides along
                b2 = (bnred.saved[1], bnred.saved[2], bnred.saved[3], bnred.saved[4], bnred.ggamma, bnred.gbeta)
            fused = F.bn_bwd_conv1x1_dgrad(dy, t, mean, rstd, self.bn.weight.data, self.ggamma, self.gbeta,
                                           self.w16.view(self.cout, c), relu_mask=rmask, reduce_done=reduce_done, bnred=b2)
        gt = None
        if pooled is not None:
            if self.relu and not reduce_done:
                gt = F.pool_bn_bwd(pooled[0], pooled[1], t, mean, rstd, self.bn.weight.data, self.ggamma, self.gbeta, rmask)
            if gt is None:
                dy = F.maxpool_bwd(pooled[0], pooled[1], t.shape[1:3])
        if gt is not None:
            pass
        elif fused is not None:
            gt = fused[0]
            if fused[2]:
                bnred.reduce_done = True
        else:
            gt, _ = F.bn_bwd(dy, None, t, mean, rstd, self.bn.weight.data, self.ggamma, self.gbeta, relu_mask=rmask,
                             reduce_done=reduce_done)
        up2 = isinstance(dx_addend, tuple) and dx_addend[0] == "up2"
        masked = isinstance(dx_addend, tuple) and not up2
        # The weight gradient is a leaf of the backward graph (nothing downstream reads it before the optimizer) while the data
        # gradient is on the critical chain: it goes to a second stream, where its split-K slices (one workgroup per CU, bound by
        # HBM latency rather than bandwidth) share the chip with the next unit's BatchNorm / data-gradient kernels.
        ws = self.wgrad_stream
        if ws is not None:
            ws.wait_stream(torch.cuda.current_stream())
            self.keepalive.append((gt, x))          # freed only after the trainer has joined the streams (no record_stream: that
                                                    # call is not capturable, and the step may be recorded into a HIP graph)
        with (torch.cuda.stream(ws) if ws is not None else _nullcontext()):
            if c == 4:                              # stem: straight into the flat gradient of the channels_last master
                F.stem_conv_wgrad(gt, x, self.gw_flat)
            elif self.k == 1 and self.stride == 1:
                m = n * h * w
                # long contraction, small output: the streaming kernel that holds the whole Ko x C block in one workgroup
                # (csrc/wgrad1x1.hip); every other shape: split-K tile GEMM
                if not F.wgrad1x1(gt.view(m, self.cout), x.view(m, c), self.gw.view(self.cout, c)):
                    F.gemm(gt.view(m, self.cout), x.view(m, c), self.cout, c, m, False, False, out=self.gw.view(self.cout, c),
                           splitk=F.pick_splitk(self.cout, c, m, target_blocks=1024))
            else:
                F.conv2d_wgrad(gt, x, (self.k, self.k), self.stride, self.pad, out=self.gw)
        dx = None
        if fused is not None:
            dx = fused[1].view(n, h, w, c)
        elif self.k == 1 and self.stride == 1:
            m = n * h * w
            g2 = gt.view(m, self.cout)
            if need_dx:
                if masked:
                    dxf = None
                    if bnred is not None and bnred.saved is not None and bnred.relu:
                        # dx is the gradient of the previous block's output: take its bn3's backward reduction here
                        _, pt, pmask, pmean, prstd = bnred.saved
                        dxf = F.gemm_masked_add_bnred(g2, self.w16.view(self.cout, c), m, c, self.cout, dx_addend[0].view(m, c),
                                                      dx_addend[1], pt.view(m, c), pmask, pmean, prstd, bnred.ggamma,
                                                      bnred.gbeta) if pt.numel() == m * c else None
                        if dxf is not None:
                            bnred.reduce_done = True
                    dx = (dxf if dxf is not None else
                          F.gemm(g2, self.w16.view(self.cout, c), m, c, self.cout, True, False, act=C.ACT_ADD_MASKED,
                                 mask_src=dx_addend[0].view(m, c), aux=dx_addend[1])).view(n, h, w, c)
                else:
                    if up2:
                        dx = F.gemm_add_upsampled2(g2, self.w16.view(self.cout, c), dx_addend[1], dx_addend[2])
                        if dx is not None:
                            return dx.view(n, h, w, c)
                        dx_addend = F.upsample_zero(dx_addend[1], dx_addend[2], 2)       # outside the fused kernel's envelope
                    dx = F.gemm(g2, self.w16.view(self.cout, c), m, c, self.cout, True, False,
                                act=C.ACT_ADD if dx_addend is not None else C.ACT_NONE,
                                mask_src=dx_addend.view(m, c) if dx_addend is not None else None).view(n, h, w, c)
        else:
            assert not masked and not up2, "masked / compact residual gradients enter through the 1x1 convolution of a bottleneck"
            if need_dx and compact_dx and self.k == 1 and self.stride == 2 and dx_addend is None and h % 2 == 0 and w % 2 == 0:
                dx = ("up2", F.conv1x1_s2_dgrad_compact(gt, self.w16), (h, w))
            else:
                dx = F.conv2d_dgrad(gt, self.w16, (h, w), self.stride, self.pad, addend=dx_addend) if need_dx else None
        return dx

    def relu_mask(self):
        """Keep bits of this unit's output ReLU (valid between forward and backward)."""
        return self.saved[2]


class Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, stride, downsample, device, last_bn_0_init=False):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 1, 1, device)
        self.bn1 = _bn(planes, device)
        self.conv2 = _conv(planes, planes, 3, stride, device)
        self.bn2 = _bn(planes, device)
        self.conv3 = _conv(planes, planes * EXPANSION, 1, 1, device)
        self.bn3 = _bn(planes * EXPANSION, device, zero_init=last_bn_0_init)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride


class ResNet50(nn.Module):
    def __init__(self, num_classes=1000, last_bn_0_init=False, device="cuda"):
        super().__init__()
        self.conv1 = _conv(3, 64, 7, 2, device)
        self.bn1 = _bn(64, device)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        inplanes = 64
        layers = []
        for i, (w, n) in enumerate(zip(WIDTHS, LAYERS)):
            blocks = []
            for b in range(n):
                stride = (1 if i == 0 else 2) if b == 0 else 1
                down = None
                if b == 0:
                    down = nn.Sequential(_conv(inplanes, w * EXPANSION, 1, stride, device), _bn(w * EXPANSION, device))
                blocks.append(Bottleneck(inplanes, w, stride, down, device, last_bn_0_init))
                inplanes = w * EXPANSION
            layers.append(nn.Sequential(*blocks))
        self.layers = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512 * EXPANSION, num_classes, device=device)

    def units(self) -> List:
        """(stem, [(conv1, conv2, conv3, downsample or None) per block]) as ConvBN units with parameter names."""
        stem = ConvBN(self.conv1, self.bn1, "conv1", "bn1", relu=True)
        blocks = []
        for li, layer in enumerate(self.layers):
            for bi, blk in enumerate(layer):
                pre = "layers.%d.%d." % (li, bi)
                u1 = ConvBN(blk.conv1, blk.bn1, pre + "conv1", pre + "bn1", True)
                u2 = ConvBN(blk.conv2, blk.bn2, pre + "conv2", pre + "bn2", True)
                u3 = ConvBN(blk.conv3, blk.bn3, pre + "conv3", pre + "bn3", True)     # ReLU after the residual add
                ud = None
                if blk.downsample is not None:
                    ud = ConvBN(blk.downsample[0], blk.downsample[1], pre + "downsample.0", pre + "downsample.1", False)
                blocks.append((u1, u2, u3, ud))
        return stem, blocks
