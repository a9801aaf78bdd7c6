"""WaveGlow train step on the gfx950 library: the AMP + data-parallel iteration of
SpeechSynthesis/Tacotron2/train.py:474-500 for `-m WaveGlow` -- model forward (waveglow/model.py:188-231), WaveGlowLoss
(waveglow/loss_function.py:30-48), scaled backward, GradScaler.unscale_ + clip_grad_norm_, torch.optim.Adam, GradScaler.update
-- as a fixed sequence of C-ABI launches on one HIP stream with an explicit backward (no autograd tape).  No CPU path.

Layout (see csrc/waveglow.hip): a series [B, C, T] of the reference is the matrix [B*T, C]; M = B * T / 8 rows of grouped
audio.  Every Conv1d is a dle_gemm over rows:
  * ConvTranspose1d(80, 80, 1024, stride 256): ONE GEMM [B*Fq, 4*80] x [256*80, 4*80]^T whose output, time-major [B, T, 80],
    read as [M, 640] IS the grouped spectrogram with its channels in (g, mel) order -- the cond weights are laid out to match;
  * the 8 cond_layers of ALL 12 flows: one GEMM spect [M, 640] -> [M, 12*8*2nc]; its backward is one data-gradient GEMM with
    K = 12*8*2nc and one weight-gradient GEMM, so the spectrogram gradient is accumulated in fp32 inside the contraction;
  * dilated in_layers: row gather (dle_wg_taps) + GEMM with K = 3 nc, the cond slice added in the epilogue (DLE_ACT_ADD); the
    gathered rows of every (flow, layer) are kept, and ALL in_layer weight gradients are one batched GEMM after the backward
    sweep (96 slices of 1024 x 1536 x M fill the chip; one by one they needed split-K slabs + a reduction each);
  * res_skip_layers: one GEMM per layer over a two-halves buffer [audio | skip sum], the previous halves added in the epilogue;
    their weight gradients: two batched GEMMs after the sweep over the kept per-layer output gradients;
  * start / end (n_half <= 4 channels): GEMMs with the narrow side zero-padded to 8.
Weight normalisation (fp32 masters -> 16-bit operands, GEMM-layout gradients -> dv, dg) and the 12 log-determinants run as ONE
table-driven launch each.  The flow state stays fp32 ([M, 8]); 16-bit tensors are the GEMM operands and WN activations, as
under autocast.
"""
import torch

from .. import _cabi as C
from .. import functional as F
from .. import multi_tensor as mt
from ..dlrm.engine import GradScalerState
from ..utils.buckets import GradBuckets
from . import ops
from .model import UPSAMPLE_KERNEL, UPSAMPLE_STRIDE, FlatViews, WaveGlow, flow_channels


class _Flow:
    """Per-flow 16-bit weight operands (rebuilt from the fp32 masters every step) and saved forward tensors."""
    __slots__ = ("c", "nh", "w_start", "w_in", "w_rs", "w_end", "dw_start", "dw_rs", "winv_t", "state", "y", "a0", "out", "o")


class WaveGlowTrainer:
    def __init__(self, model: WaveGlow, lr=1e-4, weight_decay=0.0, grad_clip_thresh=65504.0, sigma=1.0,
                 compute_dtype=torch.float16, amp=True, init_loss_scale=65536.0, growth_interval=2000, world_size=1,
                 process_group=None, bucket_mb=25):
        self.model, self.cfg = model, model.cfg
        self.dev = dev = model.store.flat.device
        self.dtype = dt = compute_dtype
        self.lr, self.wd, self.clip, self.sigma = float(lr), float(weight_decay), float(grad_clip_thresh), float(sigma)
        self.world, self.pg = world_size, process_group
        wn = self.cfg["WN_config"]
        self.nc, self.nl, self.ks = nc, nl, ks = wn["n_channels"], wn["n_layers"], wn["kernel_size"]
        self.mel, self.ng, self.nf = self.cfg["n_mel_channels"], self.cfg["n_group"], self.cfg["n_flows"]
        if nc % 8 or self.mel % 8 or ks % 2 == 0:
            raise ValueError("n_channels and n_mel_channels must be multiples of 8, kernel_size odd")
        self.chans = flow_channels(self.cfg)
        self.p = p = model.store                                         # fp32 masters (views of one flat buffer)
        self.g = g = FlatViews(model.layout, dev)                        # fp32 gradients, same offsets
        self.m = FlatViews(model.layout, dev)                            # Adam exp_avg
        self.v = FlatViews(model.layout, dev)                            # Adam exp_avg_sq
        self.scaler = GradScalerState(dev, enabled=amp, init_scale=init_loss_scale, growth_interval=growth_interval)
        self.step_t = torch.zeros(1, dtype=torch.int32, device=dev)
        self.noop = torch.zeros(1, dtype=torch.int32, device=dev)
        self.lr_t = torch.full((1,), self.lr, dtype=torch.float32, device=dev)
        self._tables = mt.TableCache()
        self.cond_cols = self.nf * nl * 2 * nc                           # columns of the all-flows cond / pre-activation matrices
        kc = self.mel * self.ng
        # 16-bit GEMM operands and fp32 GEMM-layout weight gradients: persistent, so that ONE table drives each direction
        self.w_cond = torch.zeros((self.cond_cols, kc), dtype=dt, device=dev)
        self.dw_cond = torch.zeros((self.cond_cols, kc), dtype=torch.float32, device=dev)
        self.w_in = torch.zeros((self.nf * nl, 2 * nc, ks * nc), dtype=dt, device=dev)
        self.dw_in = torch.zeros((self.nf * nl, 2 * nc, ks * nc), dtype=torch.float32, device=dev)
        # res_skip operands of every (flow, layer): rows [0, nc) = residual half, [nc, 2nc) = skip half; the last layer of a flow has
        # the skip half only and lives in the SECOND half of its slot, so that the skip rows of all layers sit at one stride
        self.w_rs_all = torch.zeros((self.nf, nl, 2 * nc, nc), dtype=dt, device=dev)
        self.dw_rs = torch.zeros((self.nf, nl, 2 * nc, nc), dtype=torch.float32, device=dev)
        self._bwd_shape = None                                           # persistent backward buffers (allocated per batch shape)
        self.flows, fwd, bwd, ld = [], [], [], []
        for k, (c, nh) in enumerate(self.chans):
            f = _Flow()
            f.c, f.nh = c, nh
            pre = "WN.%d." % k
            f.w_start = torch.zeros((nc, 8), dtype=dt, device=dev)
            f.dw_start = torch.zeros((nc, 8), dtype=torch.float32, device=dev)
            f.w_end = torch.zeros((8, nc), dtype=dt, device=dev)         # rows >= 2 nh stay zero
            f.w_in = [self.w_in[k * nl + i] for i in range(nl)]
            f.w_rs = [self.w_rs_all[k, i] if i < nl - 1 else self.w_rs_all[k, i, nc:] for i in range(nl)]
            f.dw_rs = [self.dw_rs[k, i] if i < nl - 1 else self.dw_rs[k, i, nc:] for i in range(nl)]

            def normed(name, w16, dw, cip=None, as_shape=None):
                v, dv = p[name + ".weight_v"], g[name + ".weight_v"]
                if as_shape is not None:
                    v, dv = v.view(as_shape), dv.view(as_shape)
                e = dict(v=v, g=p[name + ".weight_g"], w16=w16, dw=dw, dv=dv, dg=g[name + ".weight_g"], cip=cip)
                fwd.append(e)
                bwd.append(e)
            normed(pre + "start", f.w_start, f.dw_start, cip=8)
            fwd.append(dict(v=p[pre + "end.weight"], g=None, w16=f.w_end))       # plain weight; its gradient goes to its slot directly
            for i in range(nl):
                z = k * nl + i
                normed(pre + "in_layers.%d" % i, self.w_in[z], self.dw_in[z])
                # the reference's grouped spectrogram has channel = mel * 8 + g, the time-major rows here g * 80 + mel: the
                # [2nc, 640, 1] weight read as [2nc, 80 "channels", 8 "taps"] lands in exactly that order (tap-major operand)
                normed(pre + "cond_layers.%d" % i, self.w_cond[z * 2 * nc:(z + 1) * 2 * nc], self.dw_cond[z * 2 * nc:(z + 1) * 2 * nc],
                       as_shape=(2 * nc, self.mel, self.ng))
                normed(pre + "res_skip_layers.%d" % i, f.w_rs[i], f.dw_rs[i])
            ld.append((p.offsets["convinv.%d.conv.weight" % k][0], c))
            self.flows.append(f)
        self.wn_fwd = ops.WeightNormTable(fwd, dev)
        self.wn_bwd = ops.WeightNormTable(bwd, dev)
        self.ld_table = ops.LogdetTable(ld, dev)
        self.logdets = torch.zeros(self.nf, dtype=torch.float32, device=dev)
        self.signs = torch.ones(self.nf, dtype=torch.float32, device=dev)
        self.winv_t = torch.zeros((self.nf, 64), dtype=torch.float32, device=dev)
        for k, f in enumerate(self.flows):
            f.winv_t = self.winv_t[k, :f.c * f.c]
        self.buckets = None
        if world_size > 1:
            named = [(n, s) for n, _, s in model.layout]
            self.comm_stream = torch.cuda.Stream() if dev.type == "cuda" else None
            self.buckets = GradBuckets(g.flat, named, bucket_mb=bucket_mb, group=process_group, comm_stream=self.comm_stream,
                                       reverse=True)
            from ..utils.comm import broadcast_
            broadcast_(p.flat, 0, process_group)                         # DDP broadcasts rank 0's weights at wrap time

    # ------------------------------------------------------------------ weights: fp32 masters -> 16-bit GEMM operands
    def _prepare_weights(self):
        p = self.p
        self.w_up, self.b_up = ops.upsample_weight(p["upsample.weight"], p["upsample.bias"], self.dtype, UPSAMPLE_STRIDE)
        ops.weight_norm_fwd_batched(self.wn_fwd, self.dtype)
        ops.logdet_inv_batched(p.flat, self.ld_table, self.logdets, self.winv_t, self.signs)

    def _bias_block(self, store, first):
        """A (flow, layer)-major block of 2nc-wide biases that is contiguous in the flat buffers (model.param_layout)."""
        off, _ = store.offsets[first]
        return store.flat[off:off + self.cond_cols]

    # ------------------------------------------------------------------ forward
    def forward(self, mel, audio):
        """mel fp32 [B, 80, frames], audio fp32 [B, T] -> loss fp32 [1]; keeps what backward needs."""
        C.require_cuda(mel, audio)
        nc, nl, ng, ks = self.nc, self.nl, self.ng, self.ks
        b, t = audio.shape
        if t % ng or mel.dtype != torch.float32 or audio.dtype != torch.float32:
            raise ValueError("audio length must be a multiple of n_group; inputs are fp32")
        fq = (t + UPSAMPLE_STRIDE - 1) // UPSAMPLE_STRIDE
        if mel.shape[0] != b or mel.shape[1] != self.mel or mel.shape[2] < fq:
            # waveglow/model.py:198 asserts spect.size(2) >= audio.size(1); frames past the segment never reach it
            raise ValueError("mel must be [B, %d, >= %d frames]" % (self.mel, fq))
        self._prepare_weights()
        self.b, self.t, self.fq, self.tg = b, t, fq, t // ng
        m = b * self.tg
        self.M = m
        ntap = UPSAMPLE_KERNEL // UPSAMPLE_STRIDE
        mel_cl = F.nchw_to_nhwc(mel[:, :, :fq].contiguous().view(b, self.mel, fq, 1), self.dtype, c_padded=self.mel)
        self.col_mel = ops.taps(mel_cl.view(b * fq, self.mel), b, fq, ntap, -1, 0)
        up = F.gemm(self.col_mel, self.w_up, b * fq, UPSAMPLE_STRIDE * self.mel, ntap * self.mel, True, True, bias=self.b_up)
        if fq * UPSAMPLE_STRIDE == t:
            spect = up.view(m, self.mel * ng)
        else:                                                            # drop the tail of the last frame block (model.py:199-200)
            spect = torch.empty((b, t * self.mel), dtype=self.dtype, device=self.dev)
            F.copy_rows(up.view(b, fq * UPSAMPLE_STRIDE * self.mel)[:, :t * self.mel], spect)
            spect = spect.view(m, self.mel * ng)
        self.spect = spect
        # every cond_layer of every flow in one contraction
        self.cond = F.gemm(spect, self.w_cond, m, self.cond_cols, self.mel * ng, True, True,
                           bias=self._bias_block(self.p, "WN.0.cond_layers.0.bias"))
        self.s_all = torch.empty((m, self.cond_cols), dtype=self.dtype, device=self.dev)
        self.col_all = torch.empty((self.nf * nl, m, ks * nc), dtype=self.dtype, device=self.dev)     # gathered in_layer inputs
        self.acts_all = torch.empty((self.nf * nl, m, nc), dtype=self.dtype, device=self.dev)
        state = audio.contiguous().view(m, ng)
        parts = ops.coupling_partials(m)
        self.logs_partial = torch.zeros((self.nf, parts), dtype=torch.float32, device=self.dev)
        for k, f in enumerate(self.flows):
            pre = "WN.%d." % k
            f.state = state
            f.y, f.a0 = ops.invconv_fwd(state, self.p["convinv.%d.conv.weight" % k], f.c, self.dtype)
            # xo = [audio | running skip sum], two [M, 2nc] buffers in turn: ONE res_skip GEMM per layer writes both halves and
            # adds the previous layer's halves in its epilogue (model.py:150-156: audio = res + audio, output = output + skip)
            xo = [torch.empty((m, 2 * nc), dtype=self.dtype, device=self.dev) for _ in range(2)]
            F.gemm(f.a0, f.w_start, m, nc, 8, True, True, bias=self.p[pre + "start.bias"], out=xo[0][:, :nc])
            xo[0][:, nc:].zero_()
            cur = 0
            for i in range(nl):
                z = k * nl + i
                c0 = z * 2 * nc
                col = ops.taps(xo[cur][:, :nc], b, self.tg, ks, 2 ** i, ks // 2, out=self.col_all[z])
                s_i = self.s_all[:, c0:c0 + 2 * nc]
                F.gemm(col, f.w_in[i], m, 2 * nc, ks * nc, True, True, out=s_i, bias=self.p[pre + "in_layers.%d.bias" % i],
                       act=C.ACT_ADD, mask_src=self.cond[:, c0:c0 + 2 * nc])
                acts = ops.gate_fwd(s_i, nc, out=self.acts_all[z])
                b_rs = self.p[pre + "res_skip_layers.%d.bias" % i]
                if i < nl - 1:
                    F.gemm(acts, f.w_rs[i], m, 2 * nc, nc, True, True, bias=b_rs, act=C.ACT_ADD, mask_src=xo[cur], out=xo[1 - cur])
                else:                                                    # the last layer has the skip half only
                    F.gemm(acts, f.w_rs[i], m, nc, nc, True, True, bias=b_rs, act=C.ACT_ADD, mask_src=xo[cur][:, nc:],
                           out=xo[1 - cur][:, nc:])
                cur = 1 - cur
            f.out = out = xo[cur][:, nc:]
            f.o = F.gemm(out, f.w_end, m, 8, nc, True, True, bias=self.p.slot(pre + "end.bias"), out_dtype=torch.float32)
            state = ops.coupling_fwd(f.y, f.o, f.c, self.logs_partial[k])
        self.z = state
        self.loss = ops.loss(state, self.logs_partial.view(-1), self.logdets, self.sigma)
        return self.loss

    def _backward_buffers(self, m):
        """Persistent gradient buffers of the res_skip outputs for M = m rows + the table of their bias-gradient column sums."""
        if self._bwd_shape != m:
            nc, nl, dev, dt = self.nc, self.nl, self.dev, self.dtype
            self._d_res_all = torch.empty((self.nf, max(nl - 1, 1), m, nc), dtype=dt, device=dev)     # layers 0 .. nl-2
            self._d_skip_all = torch.empty((self.nf, m, nc), dtype=dt, device=dev)
            self._dskip_w = torch.empty((nl, m, nc), dtype=dt, device=dev)
            entries = []
            for k in range(self.nf):
                for i in range(nl):
                    gb = self.g["WN.%d.res_skip_layers.%d.bias" % (k, i)]
                    if i < nl - 1:
                        entries.append((self._d_res_all[k, i], gb[:nc]))
                        entries.append((self._d_skip_all[k], gb[nc:]))
                    else:
                        entries.append((self._d_skip_all[k], gb))
            self._rs_bias_table = F.ColsumTable(entries)
            self._bwd_shape = m
        return self._d_res_all, self._d_skip_all, self._dskip_w

    # ------------------------------------------------------------------ backward (gradients scaled by the loss scale)
    def backward(self):
        nc, nl, m, b, ks = self.nc, self.nl, self.M, self.b, self.ks
        p, g, scale = self.p, self.g, self.scaler.scale
        count = float(m * self.ng)
        dz = ops.dz_init(self.z, scale, 1.0 / (self.sigma * self.sigma * count))
        ds_all = torch.empty((m, self.cond_cols), dtype=self.dtype, device=self.dev)
        # Gradients of the res_skip outputs.  `output` is the plain sum of the skip halves, so ONE tensor per flow (d_skip) is the
        # gradient of every layer's skip half; the residual halves differ per layer.  Both are kept (persistent buffers: the
        # bias-gradient table holds their addresses) so that all res_skip weight gradients are two batched GEMMs and all bias
        # gradients one batched column sum after the sweep.
        d_res_all, d_skip_all, dskip_w = self._backward_buffers(m)
        batched = m % 8 == 0                 # the batched weight-gradient kernel wants the contraction (M rows) in 16-byte steps
        for k in range(self.nf - 1, -1, -1):
            f = self.flows[k]
            pre = "WN.%d." % k
            dy, d_o = ops.coupling_bwd(dz, f.y, f.o, scale, 1.0 / count, f.c, self.dtype)
            d_skip = d_skip_all[k]
            F.gemm(d_o, f.w_end, m, nc, 8, True, False, out=d_skip)
            F.gemm(d_o, f.out, 8, nc, m, False, False, out=g.slot(pre + "end.weight").view(8, nc), splitk=F.pick_splitk(8, nc, m))
            F.colsum(d_o, out=g.slot(pre + "end.bias"))
            # the skip half's share of EVERY layer's data gradient in one batched product (N = nl x nc columns of work instead of nl
            # half-empty launches): dskip_w[i] = d_skip x W_rs[i][skip rows]
            F.gemm_batched(d_skip, self.w_rs_all[k, 0, nc:], dskip_w, m, nc, nc, nc, nc, nc, True, False, nl, nl,
                           (0, 0), (0, 2 * nc * nc), (0, m * nc))
            d_x0 = None
            for i in range(nl - 1, -1, -1):
                last = i == nl - 1
                z = k * nl + i
                c0 = z * 2 * nc
                if last:                                                 # the last layer's res_skip has the skip half only
                    d_acts = dskip_w[i]
                else:
                    d_acts = F.gemm(d_res_all[k, i], f.w_rs[i][:nc], m, nc, nc, True, False, act=C.ACT_ADD, mask_src=dskip_w[i])
                if not batched:
                    F.gemm(d_skip, self.acts_all[z], nc, nc, m, False, False, out=f.dw_rs[i][-nc:], splitk=F.pick_splitk(nc, nc, m))
                    if not last:
                        F.gemm(d_res_all[k, i], self.acts_all[z], nc, nc, m, False, False, out=f.dw_rs[i][:nc],
                               splitk=F.pick_splitk(nc, nc, m))
                ds_i = ops.gate_bwd(d_acts, self.s_all[:, c0:c0 + 2 * nc], ds_all[:, c0:c0 + 2 * nc])
                dcol = F.gemm(ds_i, f.w_in[i], m, ks * nc, 2 * nc, True, False)
                add = None if last else d_res_all[k, i]                  # + the residual path's gradient (audio = res + audio)
                if i > 0:
                    ops.taps_bwd(dcol, b, self.tg, nc, ks, 2 ** i, ks // 2, out=d_res_all[k, i - 1], addend=add)
                else:
                    d_x0 = torch.empty((m, nc), dtype=self.dtype, device=self.dev)
                    ops.taps_bwd(dcol, b, self.tg, nc, ks, 1, ks // 2, out=d_x0, addend=add)
            # start
            da0 = F.gemm(d_x0, f.w_start, m, 8, nc, True, False, out_dtype=torch.float32)
            F.gemm(d_x0, f.a0, nc, 8, m, False, False, out=f.dw_start, splitk=F.pick_splitk(nc, 8, m))
            F.colsum(d_x0, out=g[pre + "start.bias"])
            dz = ops.invconv_bwd(dy, da0, f.state, p["convinv.%d.conv.weight" % k], f.winv_t,
                                 g["convinv.%d.conv.weight" % k], scale, 1.0 / self.ng, f.c)
        # ---- everything that spans the flows, once
        kc = self.mel * self.ng
        nz = self.nf * nl
        # res_skip bias gradients of every (flow, layer): [column sums of the residual half | column sums of d_skip]
        F.colsum_batched(self._rs_bias_table, m, nc, nc, self.dtype)
        # in_layer weight gradients of every (flow, layer): slice z = ds_all[:, z*2nc:(z+1)*2nc]^T x col_all[z]
        if batched:
            F.gemm_batched(ds_all, self.col_all, self.dw_in, 2 * nc, ks * nc, m, self.cond_cols, ks * nc, ks * nc, False, False,
                           nz, 1, (2 * nc, 0), (m * ks * nc, 0), (2 * nc * ks * nc, 0))
            # res_skip weight gradients: skip rows of slice (k, i) = d_skip[k]^T x acts_all[k, i] (all layers), residual rows =
            # d_res_all[k, i]^T x acts_all[k, i] (layers 0 .. nl-2)
            F.gemm_batched(d_skip_all, self.acts_all, self.dw_rs[0, 0, nc:], nc, nc, m, nc, nc, nc, False, False, nz, nl,
                           (m * nc, 0), (nl * m * nc, m * nc), (nl * 2 * nc * nc, 2 * nc * nc))
            if nl > 1:
                F.gemm_batched(d_res_all, self.acts_all, self.dw_rs, nc, nc, m, nc, nc, nc, False, False, self.nf * (nl - 1),
                               nl - 1, ((nl - 1) * m * nc, m * nc), (nl * m * nc, m * nc), (nl * 2 * nc * nc, 2 * nc * nc))
        else:                          # any other batch x segment goes slice by slice through dle_gemm (which has an unaligned path)
            for z in range(nz):
                F.gemm(ds_all[:, z * 2 * nc:(z + 1) * 2 * nc], self.col_all[z], 2 * nc, ks * nc, m, False, False, out=self.dw_in[z],
                       splitk=F.pick_splitk(2 * nc, ks * nc, m))
        # cond layers: one weight-gradient GEMM, one data-gradient GEMM (K = all cond columns)
        F.gemm(ds_all, self.spect, self.cond_cols, kc, m, False, False, out=self.dw_cond, splitk=F.pick_splitk(self.cond_cols, kc, m))
        # s = in_layer + cond_layer: both bias gradients are the column sums of ds, contiguous blocks in (flow, layer) order
        gb = self._bias_block(g, "WN.0.cond_layers.0.bias")
        F.colsum(ds_all, out=gb)
        self._bias_block(g, "WN.0.in_layers.0.bias").copy_(gb)
        ops.weight_norm_bwd_batched(self.wn_bwd)                         # every dv, dg from the GEMM-layout gradients
        d_spect = F.gemm(ds_all, self.w_cond, m, kc, self.cond_cols, True, False)
        # upsampling
        fq, t = self.fq, self.t
        cols = UPSAMPLE_STRIDE * self.mel
        if fq * UPSAMPLE_STRIDE == t:
            d_up = d_spect.view(b * fq, cols)
        else:
            d_up = torch.zeros((b, fq * cols), dtype=self.dtype, device=self.dev)
            F.copy_rows(d_spect.view(b, t * self.mel), d_up[:, :t * self.mel])
            d_up = d_up.view(b * fq, cols)
        kk = self.col_mel.shape[1]
        db = torch.empty((cols, kk), dtype=torch.float32, device=self.dev)
        F.gemm(d_up, self.col_mel, cols, kk, b * fq, False, False, out=db, splitk=F.pick_splitk(cols, kk, b * fq))
        ops.upsample_weight_bwd(db, g["upsample.weight"], UPSAMPLE_STRIDE)
        F.colsum(d_up.view(b * fq * UPSAMPLE_STRIDE, self.mel), out=g["upsample.bias"])
        if self.buckets is not None:
            # The cond / in_layer gradients of every flow exist only now (one GEMM for all flows), so the buckets go out after the
            # backward pass: ~1 GB of fp32 gradients at the reference's size = 2 x 7/8 x 1 GB over 7 x ~153 GB/s ~ 1.7 ms at N = 8.
            for _, _, name in self.buckets.buckets:
                self.buckets.grad_ready(name)

    # ------------------------------------------------------------------ optimizer (train.py:487-497)
    def optimizer_step(self):
        sc = self.scaler
        if self.buckets is not None:
            self.buckets.wait()
        t_g = self._tables.get("g", [[self.g.flat]])
        if sc.enabled:
            F.check_nonfinite_(self.g.flat, sc.found_inf)
        self.noop.copy_(sc.found_inf.to(torch.int32))
        self.step_t += (1 - self.noop)                                   # Adam's state step advances only when the step is taken
        gnorm, _ = mt.l2norm(t_g)
        self.grad_norm = gnorm
        t_adam = self._tables.get("adam", [[self.g.flat], [self.p.flat], [self.m.flat], [self.v.flat]],
                                  chunk=mt.streaming_chunk([[self.g.flat]]))
        mt.adam(t_adam, self.lr_t, 0.9, 0.999, 1e-8, self.wd, self.step_t, skip_flag=sc.found_inf if sc.enabled else None,
                inv_scale=sc.inv_scale if sc.enabled else None, grad_norm=gnorm, max_grad_norm=self.clip)
        sc.update()

    def set_lr(self, lr):
        """adjust_learning_rate (train.py:324-342) computes the value on the host once per iteration."""
        if lr != self.lr:
            self.lr = float(lr)
            self.lr_t.fill_(self.lr)

    def eval_loss(self, mel, audio):
        """The validation pass of train.py:273-318: forward + criterion, nothing updated.  WaveGlow has neither dropout nor
        BatchNorm, so eval mode runs the training forward; what it keeps for a backward pass is overwritten by the next forward."""
        return self.forward(mel, audio)

    def train_step(self, mel, audio):
        loss = self.forward(mel, audio)
        self.backward()
        self.optimizer_step()
        return loss
