"""Tacotron2 train step on the gfx950 library (second half): the AMP iteration of
SpeechSynthesis/Tacotron2/train.py:474-500 for `-m Tacotron2` -- Tacotron2.forward (tacotron2/model.py:667-681: embedding,
encoder, teacher-forced decoder, postnet), Tacotron2Loss (tacotron2/loss_function.py:31-46), scaled backward (BPTT through the
decoder and the encoder LSTMs), GradScaler.unscale_ + clip_grad_norm_, torch.optim.Adam -- as a fixed sequence of C-ABI launches
with an explicit backward.  No CPU path.

Layout: channels-last matrices.  Text side rows are (b, t_in) batch-major ([B*Ti, C]: convolutions = dle_wg_taps + dle_gemm, BatchNorm
over rows), decoder state is one row per sample.  What is NOT recurrent under teacher forcing runs once for all steps: the
prenet, the prenet's share of the attention-LSTM gates, the mel / gate projection, and every weight gradient (per-step gate
gradients and inputs are kept, so each weight gradient is ONE GEMM over To*B rows after the sweep).  Per decoder step, forward:
gates GEMM (+ the precomputed prenet share in the epilogue) -> dle_t2_lstm_fwd (dropout inside) -> query GEMM -> location taps +
GEMM (location conv and dense pre-multiplied into one [A, 31*2] matrix, the processed memory added in the epilogue) ->
dle_t2_attention_fwd (energies, masked softmax, context, cumulative weights) -> gates GEMM -> dle_t2_lstm_fwd.
Vectors that feed several consumers are written by the producing kernel straight into the consumers' operand buffers
(X_a[t] = [context | attention_hidden], X_d[t] = [attention_hidden | context | decoder_hidden], HC[:, t] = [decoder_hidden | context]).
--mask-padding (model.py:648-655, off by default): the frames past each sample's output length are overwritten before the loss
(mel outputs 0, gate energies 1e3) by dle_t2_mask_rows, and the gradients flowing back into them are zeroed by the same kernel --
what the reference's masked_fill_ means for autograd.  (The reference's own training run with the flag raises in backward: the
in-place fill touches a tensor the postnet's first convolution saved; oracle/make_golden.py asserts that and pins the gradients
with its parse_output applied to clones.)
Restriction: n_frames_per_step = 1 (the reference's default).
"""
import torch

from .. import _cabi as C
from .. import functional as F
from .. import multi_tensor as mt
from ..dlrm.engine import GradScalerState
from ..utils.buckets import GradBuckets
from ..waveglow import ops as wops
from ..waveglow.model import FlatViews
from . import ops
from .model import Tacotron2

NPAD = 8          # the mel + gate projection runs (n_mel + 1) outputs wide, padded to a multiple of 8


class _null_ctx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class Tacotron2Trainer:
    def __init__(self, model: Tacotron2, lr=1e-3, weight_decay=1e-6, grad_clip_thresh=1.0, compute_dtype=torch.float16, amp=True,
                 init_loss_scale=65536.0, growth_interval=2000, world_size=1, process_group=None, bucket_mb=25, seed=1234, rank=0,
                 mask_padding=False):
        self.model, self.cfg = model, model.cfg
        self.mask_padding = bool(mask_padding)
        self.training = True
        self.dev = dev = model.store.flat.device
        self.dtype = compute_dtype
        self.lr, self.wd, self.clip = float(lr), float(weight_decay), float(grad_clip_thresh)
        self.world, self.pg = world_size, process_group
        c = self.cfg
        self.E, self.A, self.Ha, self.Hd, self.P = (c["encoder_embedding_dim"], c["attention_dim"], c["attention_rnn_dim"],
                                                    c["decoder_rnn_dim"], c["prenet_dim"])
        self.NM, self.NF, self.KL = c["n_mel_channels"], c["attention_location_n_filters"], c["attention_location_kernel_size"]
        self.h = self.E // 2
        self.NO = (self.NM + 1 + NPAD - 1) // NPAD * NPAD
        for v in (self.E, self.A, self.Ha, self.Hd, self.P, self.NM, self.NF, self.h, c["postnet_embedding_dim"]):
            if v % 8:
                raise ValueError("every layer width must be a multiple of 8")
        self.p = model.store
        self.g = FlatViews(model.layout, dev)
        self.m = FlatViews(model.layout, dev)
        self.v = FlatViews(model.layout, dev)
        self.scaler = GradScalerState(dev, enabled=amp, init_scale=init_loss_scale, growth_interval=growth_interval)
        self.step_t = torch.zeros(1, dtype=torch.int32, device=dev)
        self.noop = torch.zeros(1, dtype=torch.int32, device=dev)
        self.lr_t = torch.full((1,), self.lr, dtype=torch.float32, device=dev)
        self._tables = mt.TableCache()
        # dropout: counter-based masks; the call index INSIDE a step is host state, the per-step advance a device word the kernels add
        # (a HIP-graph replay of the step then draws fresh masks, like torch's graph-safe Philox offsets)
        self.rng_seed, self._rng_calls = int(seed) + int(rank), 0
        self._rng_base = torch.zeros(1, dtype=torch.int64, device=dev)
        self._buf = dict(model.named_buffers())
        self.buckets = None
        if world_size > 1:
            self.comm_stream = torch.cuda.Stream() if dev.type == "cuda" else None
            self.buckets = GradBuckets(self.g.flat, [(n, s) for n, _, s in model.layout], bucket_mb=bucket_mb, group=process_group,
                                       comm_stream=self.comm_stream, reverse=True)
            from ..utils.comm import broadcast_
            broadcast_(self.p.flat, 0, process_group)
            self._rev_names = [n for n, _, _ in reversed(list(model.layout))]

    # ------------------------------------------------------------------ helpers
    def _z(self, *shape, dtype=None):
        return torch.zeros(shape, dtype=dtype or self.dtype, device=self.dev)

    def _e(self, *shape, dtype=None):
        return torch.empty(shape, dtype=dtype or self.dtype, device=self.dev)

    def _drop(self, x, p):
        self._rng_calls += 1
        return F.dropout_fwd(x, p, self.rng_seed, self._rng_calls, offset_base=self._rng_base)

    def _conv_w(self, name):
        """Conv1d weight [Co, Ci, K] -> GEMM operand [Co, K * Ci] (tap-major), 16-bit."""
        w = self.p[name]
        w16 = self._e(w.shape[0], w.shape[2] * w.shape[1])
        wops.weight_norm_fwd(w, None, w16)
        return w16

    def _cast(self, t):
        return F.cast(t, self.dtype)

    def _sum2(self, a, b):
        out = torch.empty_like(a)
        F.axpby_(a, b, out, 1.0, 1.0)
        return out

    def _prepare_weights(self):
        p, dt = self.p, self.dtype
        E, A, Ha, Hd, P, NM = self.E, self.A, self.Ha, self.Hd, self.P, self.NM
        w = {}
        w["emb"] = self._cast(p["embedding.weight"])
        for i in range(self.cfg["encoder_n_convolutions"]):
            w["enc%d" % i] = self._conv_w("encoder.convolutions.%d.0.conv.weight" % i)
        for sfx in ("", "_reverse"):
            w["eih" + sfx] = self._cast(p["encoder.lstm.weight_ih_l0" + sfx])
            w["ehh" + sfx] = self._cast(p["encoder.lstm.weight_hh_l0" + sfx])
            w["eb" + sfx] = self._sum2(p["encoder.lstm.bias_ih_l0" + sfx], p["encoder.lstm.bias_hh_l0" + sfx])
        w["pre0"] = self._cast(p["decoder.prenet.layers.0.linear_layer.weight"])
        w["pre1"] = self._cast(p["decoder.prenet.layers.1.linear_layer.weight"])
        wih, whh = p["decoder.attention_rnn.weight_ih"], p["decoder.attention_rnn.weight_hh"]
        w["a_pre"] = F.cast_rows(wih[:, :P], dt)                                      # the prenet's share of the gates
        w["a_cat"] = self._e(4 * Ha, E + Ha)                                           # [context | attention_hidden] share
        F.cast_rows(wih[:, P:], dt, out=w["a_cat"][:, :E])
        F.cast_rows(whh, dt, out=w["a_cat"][:, E:])
        w["a_b"] = self._sum2(p["decoder.attention_rnn.bias_ih"], p["decoder.attention_rnn.bias_hh"])
        w["d_cat"] = self._e(4 * Hd, Ha + E + Hd)
        F.cast_rows(p["decoder.decoder_rnn.weight_ih"], dt, out=w["d_cat"][:, :Ha + E])
        F.cast_rows(p["decoder.decoder_rnn.weight_hh"], dt, out=w["d_cat"][:, Ha + E:])
        w["d_b"] = self._sum2(p["decoder.decoder_rnn.bias_ih"], p["decoder.decoder_rnn.bias_hh"])
        att = "decoder.attention_layer."
        w["q"] = self._cast(p[att + "query_layer.linear_layer.weight"])
        w["mem"] = self._cast(p[att + "memory_layer.linear_layer.weight"])
        w["v"] = p[att + "v.linear_layer.weight"].view(-1)
        # location conv [F, 2, KL] (tap-major, channels padded to 8) and dense [A, F] pre-multiplied: one [A, KL*8] operand
        wc = p[att + "location_layer.location_conv.conv.weight"]
        w["loc_c"] = self._z(self.NF, self.KL * 8)
        wops.weight_norm_fwd(wc, None, w["loc_c"], cip=8)
        w["loc_d"] = self._cast(p[att + "location_layer.location_dense.linear_layer.weight"])
        w["loc"] = F.gemm(w["loc_d"], w["loc_c"], A, self.KL * 8, self.NF, True, False)
        w["proj"] = self._z(self.NO, Hd + E)
        F.cast_rows(p["decoder.linear_projection.linear_layer.weight"], dt, out=w["proj"][:NM])
        F.cast_rows(p["decoder.gate_layer.linear_layer.weight"], dt, out=w["proj"][NM:NM + 1])
        w["proj_b"] = self._z(self.NO, dtype=torch.float32)
        w["proj_b"][:NM].copy_(p["decoder.linear_projection.linear_layer.bias"])
        w["proj_b"][NM:NM + 1].copy_(p["decoder.gate_layer.linear_layer.bias"])
        for i in range(self.cfg["postnet_n_convolutions"]):
            w["post%d" % i] = self._conv_w("postnet.convolutions.%d.0.conv.weight" % i)
        # the recurrent cells' data gradients contract over the GATE dimension: transposed copies keep both operands of those
        # few-row products k-contiguous (csrc/gemm_smallm.hip streams [N, K] weight rows)
        w["a_catT"] = F.transpose_cast(w["a_cat"], dt)                                # [E + Ha, 4 Ha]
        w["d_catT"] = F.transpose_cast(w["d_cat"], dt)                                # [Ha + E + Hd, 4 Hd]
        w["qT"] = F.transpose_cast(w["q"], dt)                                        # [Ha, A]
        w["locT"] = F.transpose_cast(w["loc"], dt)                                    # [KL * 8, A]
        # compact form for the fused location term of the attention kernels: k = tap * 2 + channel (the 6 padding channels of
        # every tap dropped), zero padded to a multiple of 32
        self.fuse_loc = A % 32 == 0 and self.KL % 2 == 1
        if self.fuse_loc:
            kk = (2 * self.KL + 31) // 32 * 32
            w["loc2"] = self._z(A, kk)
            w["loc2"][:, :2 * self.KL].view(A, self.KL, 2).copy_(w["loc"].view(A, self.KL, 8)[:, :, :2])
            w["loc2T"] = F.transpose_cast(w["loc2"], dt)                              # [KK, A]
        for sfx in ("", "_reverse"):
            w["ehhT" + sfx] = F.transpose_cast(w["ehh" + sfx], dt)                    # [h, 4 h]
        self.w = w

    def _conv_bn(self, x, b, t, name, w16, act, p_drop):
        """Conv1d(k, pad (k-1)/2) + BatchNorm1d(train) + act + dropout on rows (b, t).  -> (y, saved)."""
        k = self.p[name + ".0.conv.weight"].shape[2]
        cout = self.p[name + ".0.conv.weight"].shape[0]
        col = wops.taps(x, b, t, k, 1, k // 2)
        pre = F.gemm(col, w16, b * t, cout, col.shape[1], True, True, bias=self.p[name + ".0.conv.bias"])
        bn = name + ".1"
        if not self.training:                      # model.eval(): running statistics, no dropout (the validation pass)
            rstd = torch.rsqrt(self._buf[bn + ".running_var"] + 1e-5)
            y, _ = F.bn_fwd_apply(pre, self._buf[bn + ".running_mean"], rstd, self.p[bn + ".weight"], self.p[bn + ".bias"],
                                  relu=(act == "relu"))
            return (ops.tanh_fwd(y) if act == "tanh" else y), {}
        y, mean, rstd = F.bn_fwd(pre, self.p[bn + ".weight"], self.p[bn + ".bias"], self._buf[bn + ".running_mean"],
                                 self._buf[bn + ".running_var"], eps=1e-5, momentum=0.1, relu=(act == "relu"))
        self._buf[bn + ".num_batches_tracked"] += 1
        if act == "tanh":
            y = ops.tanh_fwd(y)
        yd, mask = self._drop(y, p_drop)
        return yd, dict(col=col, pre=pre, mean=mean, rstd=rstd, y=y, mask=mask, act=act, k=k, cout=cout, name=name)

    # ------------------------------------------------------------------ forward
    def forward(self, text, text_lengths, mel, gate_target, output_lengths=None):
        """text int64 [B, Ti] (sorted by length, descending, 0-padded), text_lengths int64 [B], mel fp32 [B, n_mel, To] zero padded,
        gate_target fp32 [B, To], output_lengths int64 [B] (read under mask_padding only) -> loss fp32 [1]."""
        C.require_cuda(text, text_lengths, mel, gate_target, output_lengths)
        if self.mask_padding and output_lengths is None:
            raise ValueError("mask_padding needs the output lengths of the batch")
        E, A, Ha, Hd, P, NM, NO, h = self.E, self.A, self.Ha, self.Hd, self.P, self.NM, self.NO, self.h
        dt, cfg = self.dtype, self.cfg
        b, ti = text.shape
        to = mel.shape[2]
        self._prepare_weights()
        self._rng_calls = 0
        w = self.w
        sv = self.sv = dict(b=b, ti=ti, to=to, text=text.reshape(-1).contiguous(), lengths=text_lengths)
        # ---- encoder: embedding, 3 x (conv + BN + ReLU + dropout), bi-LSTM over the packed batch
        x = F.rows_gather(w["emb"], sv["text"])
        sv["enc"] = []
        for i in range(cfg["encoder_n_convolutions"]):
            x_in = x
            x, s = self._conv_bn(x, b, ti, "encoder.convolutions.%d" % i, w["enc%d" % i], "relu", 0.5)
            s["x_in"] = x_in
            sv["enc"].append(s)
        sv["enc_out"] = x
        memory = self._z(b * ti, E)                                      # [fwd | reverse] halves, zero at padded positions
        mem3 = memory.view(b, ti, E)
        steps = torch.arange(ti, device=self.dev)
        live_all = (steps[:, None] < text_lengths[None, :]).to(torch.float32).contiguous()          # [Ti, B]
        sv["live"] = live_all
        sv["lstm"] = {}
        for d, sfx in enumerate(("", "_reverse")):
            gx = F.gemm(x, w["eih" + sfx], b * ti, 4 * h, E, True, True, bias=w["eb" + sfx]).view(b, ti, 4 * h)
            gates = self._e(b, ti, 4 * h)                                # replaced by the gate activations step by step
            hprev = self._z(b, ti, h)                                    # the state each step started from (rows (b, t))
            c_all = self._z(ti + 1, b, h, dtype=torch.float32)          # cell state BEFORE the step processed k-th
            order = list(range(ti - 1, -1, -1) if d else range(ti))
            # the state a step starts from IS the row block hprev[:, t] (kept for the weight gradient): the cell writes the next
            # step's block directly -- no copy, no per-step allocation
            hlast = self._z(b, h)
            for k, t in enumerate(order):
                hstate = hprev[:, t]
                hnext = hprev[:, order[k + 1]] if k + 1 < ti else hlast
                F.gemm(hstate, w["ehh" + sfx], b, 4 * h, h, True, True, out=gates[:, t], act=C.ACT_ADD, mask_src=gx[:, t])
                ops.lstm_fwd(gates[:, t], c_all[k], c_all[k + 1], [hnext], live=live_all[t], h_prev=hstate,
                             out_dst=mem3[:, t, d * h:(d + 1) * h])
            sv["lstm"][sfx] = dict(gates=gates, hprev=hprev, c_all=c_all, order=order)
        sv["memory"] = memory
        pm = F.gemm(memory, w["mem"], b * ti, A, E, True, True)          # processed memory
        sv["pm"] = pm
        # ---- decoder, teacher forcing.  Prenet over the go frame + all target frames at once (model.py:473-476)
        dec_in = self._z(to + 1, b, NM)
        dec_in[1:].copy_(mel.permute(2, 0, 1))                           # input pipeline: time-major 16-bit copy of the targets
        r_all = (to + 1) * b
        l1 = F.gemm(dec_in.view(r_all, NM), w["pre0"], r_all, P, NM, True, True, act=C.ACT_RELU)
        l1d, m1 = self._drop(l1, 0.5)
        l2 = F.gemm(l1d, w["pre1"], r_all, P, P, True, True, act=C.ACT_RELU)
        l2d, m2 = self._drop(l2, 0.5)
        sv.update(dec_in=dec_in, l1=l1, l1d=l1d, m1=m1, l2=l2, l2d=l2d, m2=m2)
        g_pre = F.gemm(l2d[:to * b], w["a_pre"], to * b, 4 * Ha, P, True, True, bias=w["a_b"]).view(to, b, 4 * Ha)
        pa, pd = cfg["p_attention_dropout"], cfg["p_decoder_dropout"]
        keep_a = keep_d = None                                           # eval: only the prenet's dropout stays on (model.py:133)
        if self.training:
            _, keep_a = self._drop(torch.ones(to * b * Ha, dtype=dt, device=self.dev), pa)
            _, keep_d = self._drop(torch.ones(to * b * Hd, dtype=dt, device=self.dev), pd)
        x_a = self._z(to + 1, b, E + Ha)                                 # [context_{t-1} | attention_hidden_{t-1}]
        x_d = self._z(to + 1, b, Ha + E + Hd)                            # [attention_hidden_t | context_t | decoder_hidden_{t-1}]
        hc = self._z(b, to, Hd + E)                                      # [decoder_hidden_t | context_t], rows (b, t)
        ga = self._e(to, b, 4 * Ha)
        gd = self._e(to, b, 4 * Hd)
        ac = self._z(to + 1, b, Ha, dtype=torch.float32)
        dc = self._z(to + 1, b, Hd, dtype=torch.float32)
        awc = self._z(to + 1, b * ti, 8)                                 # (previous weights, cumulative weights, 0 x 6) per step
        aw = self._e(to, b, ti, dtype=torch.float32)
        tanh_all = self._e(to, b * ti, A)
        q_all = self._e(to, b, A, dtype=torch.float32)
        col = self._e(b * ti, self.KL * 8)
        pl = self._e(b * ti, A)
        for t in range(to):
            # gates product + cell + dropout in one launch (the cell is the epilogue of the few-row GEMM)
            ops.lstm_gemm_fwd(x_a[t], w["a_cat"], None, g_pre[t], ac[t], ac[t + 1], ga[t], [x_d[t][:, :Ha], x_a[t + 1][:, E:]],
                              keep=keep_a, keep_index=t * b * Ha, p=pa)
            F.gemm(x_d[t][:, :Ha], w["q"], b, A, Ha, True, True, out=q_all[t])
            if self.fuse_loc:
                # the location term (2-channel k = 31 convolution + dense, one [A, 64] operand) is formed inside the kernel
                ops.attention_fwd(q_all[t], pm, w["v"], memory, text_lengths, awc[t], tanh_all[t], aw[t], awc[t + 1],
                                  [x_d[t][:, Ha:Ha + E], x_a[t + 1][:, :E], hc[:, t, Hd:]], wloc=w["loc2"], kl=self.KL)
            else:
                wops.taps(awc[t], b, ti, self.KL, 1, self.KL // 2, out=col)
                F.gemm(col, w["loc"], b * ti, A, self.KL * 8, True, True, act=C.ACT_ADD, mask_src=pm, out=pl)
                ops.attention_fwd(q_all[t], pl, w["v"], memory, text_lengths, awc[t], tanh_all[t], aw[t], awc[t + 1],
                                  [x_d[t][:, Ha:Ha + E], x_a[t + 1][:, :E], hc[:, t, Hd:]])
            ops.lstm_gemm_fwd(x_d[t], w["d_cat"], w["d_b"], None, dc[t], dc[t + 1], gd[t], [x_d[t + 1][:, Ha + E:], hc[:, t, :Hd]],
                              keep=keep_d, keep_index=t * b * Hd, p=pd)
        sv.update(g_pre=g_pre, keep_a=keep_a, keep_d=keep_d, x_a=x_a, x_d=x_d, hc=hc, ga=ga, gd=gd, ac=ac, dc=dc, awc=awc, aw=aw,
                  tanh_all=tanh_all, q_all=q_all)
        # ---- mel + gate projection of every step at once, postnet, loss
        r = b * to
        out_all = F.gemm(hc.view(r, Hd + E), w["proj"], r, NO, Hd + E, True, True, bias=w["proj_b"], out_dtype=torch.float32)
        y = F.cast_rows(out_all[:, :NM], dt)
        sv["post"] = []
        npc = cfg["postnet_n_convolutions"]
        for i in range(npc):
            y_in = y
            y, s = self._conv_bn(y, b, to, "postnet.convolutions.%d" % i, w["post%d" % i], "tanh" if i < npc - 1 else "none", 0.5)
            s["x_in"] = y_in
            sv["post"].append(s)
        target = mel.permute(0, 2, 1).contiguous().view(r, NM)           # input pipeline: rows (b, t)
        scale = self.scaler.scale
        d_out = self._z(r, NO)
        d_post = self._e(r, NM)
        if self.mask_padding:
            ops.mask_rows(out_all, NM, output_lengths, b, to, 0.0)
            ops.mask_rows(out_all[:, NM:], 1, output_lengths, b, to, 1e3)
            ops.mask_rows(y, NM, output_lengths, b, to, 0.0)
        mel_l = ops.mel_loss(out_all, y, target, NM, scale, d_out, d_post)
        gate_l, dgate = F.bce_with_logits(out_all[:, NM:], gate_target.reshape(-1).contiguous(), grad_scale=scale, ld_logits=NO)
        d_out[:, NM].copy_(dgate)
        if self.mask_padding:
            ops.mask_rows(d_out, NM + 1, output_lengths, b, to, 0.0)
            ops.mask_rows(d_post, NM, output_lengths, b, to, 0.0)
        self.loss = mel_l + gate_l
        sv.update(out_all=out_all, d_out=d_out, d_post=d_post)
        return self.loss

    # ------------------------------------------------------------------ backward
    def _conv_bn_bwd(self, dy, s, b, t, w16, want_dx=True):
        """Backward of _conv_bn: dropout -> act -> BatchNorm -> conv.  Fills the parameter gradients, returns dx (16-bit)."""
        name, k, cout, g = s["name"], s["k"], s["cout"], self.g
        d = F.dropout_bwd(dy, s["mask"], 0.5)
        if s["act"] == "tanh":
            d = F.act_bwd(d, s["y"], C.ACT_TANH_BWD)
        bn = name + ".1"
        dpre, _ = F.bn_bwd(d, s["y"] if s["act"] == "relu" else None, s["pre"], s["mean"], s["rstd"], self.p[bn + ".weight"],
                           g[bn + ".weight"], g[bn + ".bias"])
        rows, kc = s["col"].shape
        dw = torch.empty((cout, kc), dtype=torch.float32, device=self.dev)
        F.gemm(dpre, s["col"], cout, kc, rows, False, False, out=dw, splitk=F.pick_splitk(cout, kc, rows))
        wops.weight_norm_bwd(dw, self.p[name + ".0.conv.weight"], None, g[name + ".0.conv.weight"], None)
        F.colsum(dpre, out=g[name + ".0.conv.bias"])
        if not want_dx:
            return None
        cin = kc // k
        dcol = F.gemm(dpre, w16, rows, kc, cout, True, False)
        dx = self._e(rows, cin)
        wops.taps_bwd(dcol, b, t, cin, k, 1, k // 2, out=dx)
        return dx

    def _wgrad(self, dy, x, out, rows):
        """out (fp32 [N, K]) = dy[rows, N]^T x[rows, K]."""
        n, k = out.shape
        F.gemm(dy, x, n, k, rows, False, False, out=out, splitk=F.pick_splitk(n, k, rows))

    def backward(self):
        sv, w, g, p = self.sv, self.w, self.g, self.p
        E, A, Ha, Hd, P, NM, NO, h = self.E, self.A, self.Ha, self.Hd, self.P, self.NM, self.NO, self.h
        b, ti, to, dt, cfg = sv["b"], sv["ti"], sv["to"], self.dtype, self.cfg
        f32 = torch.float32
        r = b * to
        self._rev_pos = 0
        # ---- postnet
        npc = cfg["postnet_n_convolutions"]
        dy = sv["d_post"]
        for i in range(npc - 1, -1, -1):
            dy = self._conv_bn_bwd(dy, sv["post"][i], b, to, w["post%d" % i])
        d_out = sv["d_out"]
        d_out[:, :NM].add_(dy)                                          # + the postnet branch (mel_post = mel_out + postnet(mel_out))
        # ---- projection (all steps)
        hc2 = sv["hc"].view(r, Hd + E)
        dwp = torch.empty((NO, Hd + E), dtype=f32, device=self.dev)
        self._wgrad(d_out, hc2, dwp, r)
        g["decoder.linear_projection.linear_layer.weight"].copy_(dwp[:NM])
        g["decoder.gate_layer.linear_layer.weight"].copy_(dwp[NM:NM + 1])
        dbp = torch.empty(NO, dtype=f32, device=self.dev)
        F.colsum(d_out, out=dbp)
        g["decoder.linear_projection.linear_layer.bias"].copy_(dbp[:NM])
        g["decoder.gate_layer.linear_layer.bias"].copy_(dbp[NM:NM + 1])
        dhc = F.gemm(d_out, w["proj"], r, Hd + E, NO, True, False, out_dtype=f32).view(b, to, Hd + E)
        self._grads_final(("postnet.", "decoder.gate_layer.", "decoder.linear_projection."))    # reduced under the whole sweep
        # ---- decoder BPTT
        memory, pm = sv["memory"], sv["pm"]
        x_a, x_d, ga, gd, ac, dc, aw, awc = sv["x_a"], sv["x_d"], sv["ga"], sv["gd"], sv["ac"], sv["dc"], sv["aw"], sv["awc"]
        d_memory = self._z(b * ti, E, dtype=f32)
        # context_t = weights_t x memory: d memory = sum_t weights_t (x) d_context_t.  When the shapes fit the batched kernel (Ti, E
        # multiples of 8) the per-step gradients are kept (16-bit, steps padded to a multiple of 8) and the sum is ONE batched GEMM
        # per sample after the sweep; otherwise the attention kernel accumulates it step by step.
        hoist = ti % 8 == 0 and E % 8 == 0
        to8 = (to + 7) // 8 * 8
        dctx_all = self._z(to8, b, E) if hoist else None
        d_pm = self._z(b * ti, A, dtype=f32)
        dv_acc = self._z(b, A, dtype=f32)                                # per-sample partial sums of dv (no atomics)
        dw_loc = self._z(A, self.KL * 8, dtype=f32)
        dq_all = self._e(to, b, A)
        # Per-step scratch, allocated ONCE: every sum of gradient pieces (autograd's accumulation into a tensor that feeds several
        # consumers) happens on load inside the kernel that consumes it -- the sweep holds no ATen arithmetic and no allocation.
        #   dxd = gradient wrt x_d[t] = [attention_hidden_t | context_t | decoder_hidden_{t-1}]  (decoder LSTM's gates, step t)
        #   dxa = gradient wrt x_a[t] = [context_{t-1} | attention_hidden_{t-1}]                  (attention LSTM's gates, step t)
        # both zero before the first (= last in time) step: nothing follows it
        dxd_all = self._e(to + 1, b, Ha + E + Hd, dtype=f32)              # [t] = dxd of step t; [to] = zeros (nothing follows the last step)
        dxd_all[to].zero_()
        dxa = self._z(b, E + Ha, dtype=f32)
        d_ah_q = self._e(b, Ha, dtype=f32)                               # through the query layer
        d_dc = [self._z(b, Hd, dtype=f32), self._e(b, Hd, dtype=f32)]
        d_ac = [self._z(b, Ha, dtype=f32), self._e(b, Ha, dtype=f32)]
        d_aw_loc = self._z(b, ti, dtype=f32)                             # wrt weights_t as "previous weights" of step t+1
        d_cum = self._z(b, ti, dtype=f32)                                # wrt cumulative weights_t (all later steps)
        dcol = self._e(b * ti, self.KL * 8) if not self.fuse_loc else None
        # the location layer's weight gradient contracts over (step, sample, position): d_pl is kept for CH steps at a time and
        # meets the gathered rows of the same steps in ONE split-K product per chunk
        ch_steps = max(1, min(to, 64))
        d_pl_ch = self._e(ch_steps, b * ti, A)
        pa, pd = cfg["p_attention_dropout"], cfg["p_decoder_dropout"]
        # The decoder LSTM's backward recurrence (cell of step t <- its own gates of step t+1) does not depend on the attention
        # chain: it runs AHEAD on a second stream (one 160-workgroup product + one cell per step) and leaves dxd for every step;
        # the attention chain of step t (128-workgroup attention kernel, query product, attention cell, its gate product) waits
        # for the event of step t.  Both chains are a few-workgroup latency-bound launches each: side by side they share the chip.
        main = torch.cuda.current_stream() if self.dev.type == "cuda" else None
        side = self._side_stream() if main is not None else None
        done = []
        if side is not None:
            side.wait_stream(main)
        with (torch.cuda.stream(side) if side is not None else _null_ctx()):
            for t in range(to - 1, -1, -1):
                cur, nxt = (to - 1 - t) & 1, ((to - 1 - t) & 1) ^ 1
                # decoder LSTM: dh = projection piece + the piece through step t+1's gates
                ops.lstm_bwd(dhc[:, t, :Hd], d_dc[cur], gd[t], dc[t], gd[t], d_dc[nxt], keep=sv["keep_d"], keep_index=t * b * Hd, p=pd,
                             dh_add=(dxd_all[t + 1][:, Ha + E:],))
                F.gemm(gd[t], w["d_catT"], b, Ha + E + Hd, 4 * Hd, True, True, out=dxd_all[t])
                if side is not None:
                    ev = torch.cuda.Event()
                    ev.record(side)
                    done.append(ev)
        for t in range(to - 1, -1, -1):
            cur, nxt = (to - 1 - t) & 1, ((to - 1 - t) & 1) ^ 1
            dxd = dxd_all[t]
            if side is not None:
                main.wait_event(done[to - 1 - t])
            # attention: d context_t = projection piece + decoder gates (step t) + attention gates (step t+1)
            slot = t % ch_steps
            # location term: pl = taps(awc[t]) x W_loc^T + pm; awc[t] = (weights_{t-1}, cumulative_{t-1}).  Fused: the kernel also
            # runs the transposed convolution and leaves d_aw_loc (for step t-1) / adds to d_cum in place
            fz = dict(wloc_t=w["loc2T"], kl=self.KL, d_prev=d_aw_loc, d_cum=d_cum) if self.fuse_loc else {}
            ops.attention_bwd(dhc[:, t, Hd:], d_aw_loc, aw[t], sv["tanh_all"][t], w["v"], memory, None if hoist else d_memory,
                              d_pl_ch[slot], None, dv_acc, None, d_ctx_add=(dxd[:, Ha:Ha + E], dxa[:, :E]), d_aw_add=d_cum,
                              dq16=dq_all[t], dctx16=dctx_all[t] if hoist else None, **fz)
            if not self.fuse_loc:
                F.gemm(d_pl_ch[slot], w["locT"], b * ti, self.KL * 8, A, True, True, out=dcol)
                ops.location_bwd(dcol, d_aw_loc, d_cum, b, ti, self.KL)
            if slot == 0:
                n = min(ch_steps, to - t)
                cols = wops.taps(awc[t:t + n].view(-1, 8), n * b, ti, self.KL, 1, self.KL // 2)
                rows = n * b * ti
                F.gemm(d_pl_ch[:n].view(rows, A), cols, A, self.KL * 8, rows, False, False, out=dw_loc, accumulate=True,
                       splitk=F.pick_splitk(A, self.KL * 8, rows))
                del cols
                ops.sum_steps(d_pl_ch[:n], d_pm.view(-1))                 # d processed memory = sum over the steps of d_pl
            # attention LSTM: d attention_hidden_t = decoder gates (step t) + query layer + attention gates (step t+1)
            F.gemm(dq_all[t], w["qT"], b, Ha, A, True, True, out=d_ah_q)
            ops.lstm_bwd(dxd[:, :Ha], d_ac[cur], ga[t], ac[t], ga[t], d_ac[nxt], keep=sv["keep_a"], keep_index=t * b * Ha, p=pa,
                         dh_add=(d_ah_q, dxa[:, E:]))
            F.gemm(ga[t], w["a_catT"], b, E + Ha, 4 * Ha, True, True, out=dxa)
        if side is not None:
            main.wait_stream(side)
        # ---- leaves of the backward graph on the second stream: the decoder's weight gradients (one GEMM each over all To*B rows,
        # ~8 ms of full-chip work) and the whole prenet backward run beside the memory gradient and the encoder's backward, whose
        # bi-LSTM sweep is 2 x Ti dependent few-row launches that leave the chip almost empty
        if side is not None:
            side.wait_stream(main)
        with (torch.cuda.stream(side) if side is not None else _null_ctx()):
            # ---- weight gradients of the decoder: one GEMM each over all steps
            rt = to * b
            ga2, gd2 = ga.view(rt, 4 * Ha), gd.view(rt, 4 * Hd)
            dwa = torch.empty((4 * Ha, E + Ha), dtype=f32, device=self.dev)
            self._wgrad(ga2, x_a.view(-1, E + Ha), dwa, rt)
            dwa_pre = torch.empty((4 * Ha, P), dtype=f32, device=self.dev)
            self._wgrad(ga2, sv["l2d"], dwa_pre, rt)
            gih = g["decoder.attention_rnn.weight_ih"]
            gih[:, :P].copy_(dwa_pre)
            gih[:, P:].copy_(dwa[:, :E])
            g["decoder.attention_rnn.weight_hh"].copy_(dwa[:, E:])
            F.colsum(ga2, out=g["decoder.attention_rnn.bias_ih"])
            g["decoder.attention_rnn.bias_hh"].copy_(g["decoder.attention_rnn.bias_ih"])
            dwd = torch.empty((4 * Hd, Ha + E + Hd), dtype=f32, device=self.dev)
            self._wgrad(gd2, x_d.view(-1, Ha + E + Hd), dwd, rt)
            g["decoder.decoder_rnn.weight_ih"].copy_(dwd[:, :Ha + E])
            g["decoder.decoder_rnn.weight_hh"].copy_(dwd[:, Ha + E:])
            F.colsum(gd2, out=g["decoder.decoder_rnn.bias_ih"])
            g["decoder.decoder_rnn.bias_hh"].copy_(g["decoder.decoder_rnn.bias_ih"])
            att = "decoder.attention_layer."
            self._wgrad(dq_all.view(rt, A), x_d.view(-1, Ha + E + Hd)[:, :Ha], g[att + "query_layer.linear_layer.weight"], rt)
            # ---- prenet
            r_all = (to + 1) * b
            d_l2d = self._z(r_all, P)
            F.gemm(ga2, w["a_pre"], rt, P, 4 * Ha, True, False, out=d_l2d[:rt])
            d_l2 = F.dropout_bwd(d_l2d, sv["m2"], 0.5)
            d_pre2 = self._relu_mask(d_l2, sv["l2"])
            self._wgrad(d_pre2, sv["l1d"], g["decoder.prenet.layers.1.linear_layer.weight"], r_all)
            d_l1d = F.gemm(d_pre2, w["pre1"], r_all, P, P, True, False)
            d_pre1 = self._relu_mask(F.dropout_bwd(d_l1d, sv["m1"], 0.5), sv["l1"])
            self._wgrad(d_pre1, sv["dec_in"].view(r_all, NM), g["decoder.prenet.layers.0.linear_layer.weight"], r_all)
        att = "decoder.attention_layer."
        F.colsum(dv_acc, out=g[att + "v.linear_layer.weight"].view(-1))
        dwl16 = self._cast(dw_loc)
        F.gemm(dwl16, w["loc_c"], A, self.NF, self.KL * 8, True, True, out=g[att + "location_layer.location_dense.linear_layer.weight"])
        dwc = torch.empty((self.NF, self.KL * 8), dtype=f32, device=self.dev)
        F.gemm(w["loc_d"], dwl16, self.NF, self.KL * 8, A, False, False, out=dwc)
        wops.weight_norm_bwd(dwc, p[att + "location_layer.location_conv.conv.weight"], None,
                             g[att + "location_layer.location_conv.conv.weight"], None, cip=8)
        if hoist:
            aw16 = self._z(to8, b, ti)
            F.cast(aw, dt, out=aw16[:to])
            # sample b: d_memory[b] [Ti, E] = aw16[:, b, :]^T [Ti, To] x dctx_all[:, b, :] [To, E]
            F.gemm_batched(aw16, dctx_all, d_memory, ti, E, to8, b * ti, b * E, E, False, False, b, 1, (ti, 0), (E, 0), (ti * E, 0))
        d_pm16 = self._cast(d_pm)
        self._wgrad(d_pm16, memory, g[att + "memory_layer.linear_layer.weight"], b * ti)
        F.gemm(d_pm16, w["mem"], b * ti, E, A, True, False, out=d_memory, accumulate=True)
        self._grads_final(("decoder.",))                                # reduced under the encoder's backward pass
        # ---- encoder: bi-LSTM BPTT, convolutions, embedding
        dm3 = d_memory.view(b, ti, E)
        x_enc = sv["enc_out"]
        dx_enc = None
        for d, sfx in enumerate(("", "_reverse")):
            s = sv["lstm"][sfx]
            gates, hprev, c_all, order = s["gates"], s["hprev"], s["c_all"], s["order"]
            dh_rec = [self._z(b, h, dtype=f32), self._e(b, h, dtype=f32)]
            d_c = [self._z(b, h, dtype=f32), self._e(b, h, dtype=f32)]
            for k in range(ti - 1, -1, -1):
                t = order[k]
                cur, nxt = (ti - 1 - k) & 1, ((ti - 1 - k) & 1) ^ 1
                ops.lstm_bwd(dm3[:, t, d * h:(d + 1) * h], d_c[cur], gates[:, t], c_all[k], gates[:, t], d_c[nxt], live=sv["live"][t],
                             dh_prev=dh_rec[nxt], dh_add=(dh_rec[cur],))
                F.gemm(gates[:, t], w["ehhT" + sfx], b, h, 4 * h, True, True, out=dh_rec[nxt], accumulate=True)
            g2 = gates.view(b * ti, 4 * h)
            self._wgrad(g2, x_enc, g["encoder.lstm.weight_ih_l0" + sfx], b * ti)
            self._wgrad(g2, hprev.view(b * ti, h), g["encoder.lstm.weight_hh_l0" + sfx], b * ti)
            F.colsum(g2, out=g["encoder.lstm.bias_ih_l0" + sfx])
            g["encoder.lstm.bias_hh_l0" + sfx].copy_(g["encoder.lstm.bias_ih_l0" + sfx])
            dx_d = F.gemm(g2, w["eih" + sfx], b * ti, E, 4 * h, True, False, out_dtype=f32)
            dx_enc = dx_d if dx_enc is None else dx_enc + dx_d
        dy = self._cast(dx_enc)
        for i in range(cfg["encoder_n_convolutions"] - 1, -1, -1):
            dy = self._conv_bn_bwd(dy, sv["enc"][i], b, ti, w["enc%d" % i])
        g["embedding.weight"].zero_()
        F.embed_scatter_add_(g["embedding.weight"], dy, sv["text"])
        if side is not None:
            main.wait_stream(side)
        self._grads_final(None)                                        # encoder + embedding: everything that is left
        if self._rng_calls:
            self._rng_base += self._rng_calls          # on the device: the next step (or graph replay) draws new masks

    def _grads_final(self, prefixes):
        """Gradients complete from the END of the flat buffer (postnet, projection, ... , embedding = reverse layout order).  Walk
        the layout backwards over every parameter whose name starts with one of `prefixes` (None: all that is left) and fire the
        buckets they close: their all-reduce runs on the communication stream under the rest of the backward pass."""
        if self.buckets is None:
            return
        names = self._rev_names
        while self._rev_pos < len(names) and (prefixes is None or names[self._rev_pos].startswith(prefixes)):
            self.buckets.grad_ready(names[self._rev_pos])
            self._rev_pos += 1

    def _side_stream(self):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.dev)
            if getattr(self, "buckets", None) is not None:             # gradients are written on it: the bucket hooks wait for it too
                self.buckets.extra_streams.append(self._side)
        return self._side

    def _relu_mask(self, g, y):
        out = torch.empty_like(g)
        return F.relu_bwd(g, y, out)

    # ------------------------------------------------------------------ optimizer (train.py:487-497)
    def optimizer_step(self):
        sc = self.scaler
        if self.buckets is not None:
            self.buckets.wait()
        t_g = self._tables.get("g", [[self.g.flat]])
        if sc.enabled:
            F.check_nonfinite_(self.g.flat, sc.found_inf)
        self.noop.copy_(sc.found_inf.to(torch.int32))
        self.step_t += (1 - self.noop)
        gnorm, _ = mt.l2norm(t_g)
        self.grad_norm = gnorm
        t_adam = self._tables.get("adam", [[self.g.flat], [self.p.flat], [self.m.flat], [self.v.flat]],
                                  chunk=mt.streaming_chunk([[self.g.flat]]))
        mt.adam(t_adam, self.lr_t, 0.9, 0.999, 1e-8, self.wd, self.step_t, skip_flag=sc.found_inf if sc.enabled else None,
                inv_scale=sc.inv_scale if sc.enabled else None, grad_norm=gnorm, max_grad_norm=self.clip)
        sc.update()

    def set_lr(self, lr):
        if lr != self.lr:
            self.lr = float(lr)
            self.lr_t.fill_(self.lr)

    def eval_loss(self, text, text_lengths, mel, gate_target, output_lengths=None):
        """The validation pass of train.py:273-318: model.eval() forward + criterion, nothing kept for a backward pass, BatchNorm
        buffers untouched.  The prenet's dropout stays on, as in the reference (F.dropout(..., training=True), model.py:133)."""
        self.training = False
        try:
            loss = self.forward(text, text_lengths, mel, gate_target, output_lengths)
        finally:
            self.training = True
            self.sv = None
            if self._rng_calls:
                # the prenet's dropout drew masks at counters base + 1 .. base + calls: move the base on, so that the next
                # validation batch and the next training step draw fresh ones (F.dropout(training=True) in the reference)
                self._rng_base += self._rng_calls
                self._rng_calls = 0
        return loss

    def train_step(self, text, text_lengths, mel, gate_target, output_lengths=None):
        loss = self.forward(text, text_lengths, mel, gate_target, output_lengths)
        self.backward()
        self.optimizer_step()
        return loss
