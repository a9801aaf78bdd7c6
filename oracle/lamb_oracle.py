"""LAMB / L2-norm / SGD oracle (numpy fp32).  TEST INFRASTRUCTURE ONLY.

Pinning: the reference ships no CPU implementation and no numeric test of
BERT/lamb_amp_opt/csrc/multi_tensor_lamb.cu, so the per-element arithmetic below restates the .cu line by line
(citations relative to /root/reference/PyTorch/LanguageModeling/BERT/lamb_amp_opt/) and is cross-checked against an
independent float64 closed form (tests/test_oracle_lamb.py).  The HOST sequence -- FusedLAMBAMP.step
(fused_lamb/fused_lamb.py:131-260), PolyWarmUpScheduler.step (schedulers.py:123-136) and torch's GradScaler around
them (run_pretraining.py:527-536) -- is pinned by tests/golden/lamb_ref_steps.npz, produced by the reference's own
unmodified Python classes stepping on CPU with `fused_lamb_CUDA` bound to these functions
(oracle/lamb_cpu_ext.py, oracle/make_golden.py gen_lamb): `fused_lamb_host_steps` below must reproduce it exactly.
"""
import numpy as np

f32 = np.float32


def l2norm(tensors):
    """csrc/multi_tensor_l2norm_kernel.cu:28-151 -- sqrt(sum x^2) in fp32 (global, per tensor)."""
    per = np.asarray([np.sqrt(np.sum(np.square(t.astype(np.float32)), dtype=np.float32)) for t in tensors], f32)
    tot = np.sqrt(np.sum(np.square(per.astype(np.float64)))).astype(f32)
    return tot, per


def lamb_step(g_list, p_list, m_list, v_list, lr, beta1, beta2, eps, step, bias_correction, weight_decay,
              grad_averaging, mode, global_grad_norm, max_grad_norm, use_nvlamb=False, inv_scale=1.0,
              grad_dtype=np.float32, model_copy_dtype=None, round_update=None):
    """csrc/multi_tensor_lamb.cu:371-500 host sequence for one param group:
       per-tensor ||p|| -> stage1 (:43-245) -> per-tensor ||update|| -> stage2 (:251-368).
    Returns (updates_in_grad_buffer, new_p, new_m, new_v, model_copies)."""
    beta3 = f32(1 - beta1) if grad_averaging else f32(1.0)                      # :405-407
    b1c = b2c = f32(1.0)
    if bias_correction:                                                         # :67-73 (pow in double)
        b1c = f32(1.0 - float(beta1) ** step)
        b2c = f32(1.0 - float(beta2) ** step)
    clip = f32(global_grad_norm / max_grad_norm) if global_grad_norm > max_grad_norm else f32(1.0)   # :79
    beta1, beta2, eps, decay = f32(beta1), f32(beta2), f32(eps), f32(weight_decay)
    _, p_norms = l2norm(p_list)                                                 # :411
    upd, new_m, new_v = [], [], []
    for g, p, m, v in zip(g_list, p_list, m_list, v_list):
        sg = (g.astype(f32) * f32(inv_scale)) / clip                            # :126,:131
        pp = p.astype(f32) if decay != 0 else np.zeros_like(p, f32)             # :118-123
        if mode == 0:                                                           # :130-139
            sg = sg + decay * pp
            m2 = m * beta1 + beta3 * sg
            v2 = v * beta2 + (f32(1) - beta2) * sg * sg
            u = (m2 / b1c) / (np.sqrt(v2 / b2c) + eps)
        else:                                                                   # :141-149
            m2 = m * beta1 + beta3 * sg
            v2 = v * beta2 + (f32(1) - beta2) * sg * sg
            u = (m2 / b1c) / (np.sqrt(v2 / b2c) + eps) + decay * pp
        # :163 the update is stored back INTO the gradient buffer, i.e. rounded to the gradient's dtype, and
        # stage 2 / the update norm read that rounded value (round_update emulates dtypes numpy lacks, bf16)
        upd.append(round_update(u) if round_update is not None else u.astype(grad_dtype))
        new_m.append(m2.astype(f32))
        new_v.append(v2.astype(f32))
    _, u_norms = l2norm([u.astype(f32) for u in upd])                           # :457
    new_p, copies = [], []
    for i, (u, p) in enumerate(zip(upd, p_list)):
        ratio = f32(lr)
        if use_nvlamb or decay != 0:                                            # :277-282
            pn, un = p_norms[i], u_norms[i]
            ratio = f32(lr) * (pn / un) if (un != 0 and pn != 0) else f32(lr)
        p2 = (p.astype(f32) - ratio * u.astype(f32)).astype(f32)                # :316
        new_p.append(p2)
        copies.append(p2.astype(model_copy_dtype) if model_copy_dtype is not None else None)
    return upd, new_p, new_m, new_v, copies


def sgd_step(g, p, buf, lr, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False, first=False,
             inv_scale=1.0):
    """torch.optim.SGD single-tensor math (what CN/image_classification/optimizers.py:34-56 configures)."""
    d = g.astype(f32) * f32(inv_scale) + f32(weight_decay) * p
    if momentum != 0:
        buf = d.copy() if first else f32(momentum) * buf + f32(1 - dampening) * d
        d = d + f32(momentum) * buf if nesterov else buf
    return (p - f32(lr) * d).astype(f32), buf


def poly_warmup_lr(step_after, base_lr, warmup, total_steps, degree=0.5):
    """PolyWarmUpScheduler.get_lr (schedulers.py:133-136), fp32 like the device tensors it is computed with."""
    progress = f32(f32(step_after) / f32(total_steps))
    if progress < f32(warmup):
        return f32(f32(base_lr) * progress / f32(warmup))
    return f32(f32(base_lr) * (f32(1.0) - progress) ** f32(degree))


class FusedLambHost:
    """Restatement of the reference's optimizer step as run_pretraining.py:527-536 drives it:
        lr_scheduler.step()            # lr from group['step'] + 1 (schedulers.py:123-131)
        grad_scaler.step(optimizer)    # FusedLAMBAMP.step(grad_scaler=...) (fused_lamb.py:131-260)
        grad_scaler.update()           # torch GradScaler: backoff 0.5 on overflow, x2 after growth_interval clean steps
    State per parameter: fp32 master `p`, `m`, `v`, and (fp16 parameters) the 16-bit model copy `p16`.
    `kernels` supplies (l2norm, lamb) with the fused_lamb_CUDA calling convention on numpy arrays; default = this
    module -- tests pass the HIP kernels through the same sequence."""

    def __init__(self, params, groups, lr, warmup, total_steps, max_grad_norm=1.0, init_scale=2.0 ** 16,
                 growth_interval=2000, betas=(0.9, 0.999), eps=1e-6, kernels=None):
        # params: name -> (array, is_half); groups: [(weight_decay, [names])] in the optimizer's group order
        self.names = list(params)
        self.half = {k: bool(v[1]) for k, v in params.items()}
        self.p = {k: np.asarray(v[0], np.float32).copy() for k, v in params.items()}
        self.p16 = {k: self.p[k].astype(np.float16) for k in self.names if self.half[k]}
        self.m = {k: np.zeros_like(self.p[k]) for k in self.names}
        self.v = {k: np.zeros_like(self.p[k]) for k in self.names}
        self.groups = groups
        self.base_lr, self.warmup, self.total = lr, warmup, total_steps
        self.max_grad_norm, self.betas, self.eps = max_grad_norm, betas, eps
        self.scale, self.growth_interval, self.growth_tracker = f32(init_scale), growth_interval, 0
        self.step = 0                       # group['step'] (one value: every group advances together)
        self.lr = None
        # (l2norm(list) -> (total, per_tensor), lamb(g, p, m, v, **kw) -> (upd, p, m, v, copies)); the -m gpu tests
        # substitute adapters around the HIP kernels (shims/fused_lamb_CUDA.py) to run them through this sequence
        self.k_l2norm, self.k_lamb = kernels if kernels is not None else (l2norm, lamb_step)

    def optimizer_step(self, grads):
        """grads: name -> SCALED gradient in the parameter's dtype (fp16 for half parameters)."""
        # PolyWarmUpScheduler.step: last_epoch = group['step'] + 1 (or 1 before the first optimizer step)
        self.lr = poly_warmup_lr(self.step + 1, self.base_lr, self.warmup, self.total)
        # GradScaler._check_inf_per_device: any non-finite gradient element
        found_inf = any(not np.all(np.isfinite(g.astype(np.float32))) for g in grads.values())
        inv_scale = f32(1.0 / np.float64(self.scale))                       # fused_lamb.py:155 (double reciprocal)
        max_norm = f32(f32(self.max_grad_norm) * self.scale)                # :163 norms are of SCALED gradients
        if not found_inf:
            g32 = [grads[k] for k in self.names if not self.half[k]]
            g16 = [grads[k] for k in self.names if self.half[k]]
            n32 = self.k_l2norm(g32)[0] if g32 else f32(0)                  # :164-183
            n16 = self.k_l2norm(g16)[0] if g16 else f32(0)
            gnorm = self.k_l2norm([np.asarray([n32], f32), np.asarray([n16], f32)])[0]      # :186-191 blend
            self.step += 1                                                  # :202-205 step += (noop != 1)
            for wd, names in self.groups:
                for half in (True, False):                                  # :240-258 fp16 list first, then fp32
                    sel = [k for k in names if self.half[k] == half]
                    if not sel:
                        continue
                    upd, p2, m2, v2, cp = self.k_lamb(
                        [grads[k] for k in sel], [self.p[k] for k in sel], [self.m[k] for k in sel],
                        [self.v[k] for k in sel], self.lr, self.betas[0], self.betas[1], self.eps, self.step, True, wd,
                        True, 1, gnorm, max_norm, False, inv_scale=inv_scale,
                        grad_dtype=np.float16 if half else np.float32, model_copy_dtype=np.float16 if half else None)
                    for k, a, b, c, d in zip(sel, p2, m2, v2, cp):
                        self.p[k], self.m[k], self.v[k] = a, b, c
                        if half:
                            self.p16[k] = d
        # GradScaler.update (torch/amp/grad_scaler.py, _amp_update_scale_): backoff / growth
        if found_inf:
            self.scale = f32(self.scale * f32(0.5))
            self.growth_tracker = 0
        else:
            self.growth_tracker += 1
            if self.growth_tracker == self.growth_interval:
                self.scale = f32(self.scale * f32(2.0))
                self.growth_tracker = 0
        return found_inf


LAMB_GOLDEN_CASE = dict(
    shapes={"w_a": ((33, 5), True), "w_b": ((64,), False), "w_c": ((1024,), True), "b_a": ((7,), False),
            "b_b": ((3, 3, 3), True)},
    groups=[(0.01, ["w_a", "w_b", "w_c"]), (0.0, ["b_a", "b_b"])],
    lr=6e-3, warmup=0.2843, total_steps=12, init_scale=1024.0, growth_interval=2, steps=7, overflow_at=(2,), seed=77)


def lamb_golden_inputs(case=LAMB_GOLDEN_CASE):
    """Seeded initial parameters and per-step UNSCALED gradients of the golden scenario."""
    rng = np.random.default_rng(case["seed"])
    params = {}
    for k, (shape, half) in case["shapes"].items():
        a = rng.standard_normal(shape).astype(np.float32)
        params[k] = (a.astype(np.float16).astype(np.float32) if half else a, half)
    grads = [{k: (rng.standard_normal(shape) * 0.3).astype(np.float32) for k, (shape, _) in case["shapes"].items()}
             for _ in range(case["steps"])]
    return params, grads


# ---- call trace of the reference's FusedLAMBAMP.step (oracle/make_golden.py gen_lamb_trace): a flat .npz of plain arrays ---------
_TRACE_SCALARS = ("chunk", "beta1", "beta2", "eps", "bias_correction", "weight_decay", "grad_averaging", "mode")
_TRACE_ARRAYS = ("noop_in", "noop_out", "total", "per", "lr", "step", "global_grad_norm", "max_grad_norm", "found_inf", "inv_scale")


def save_call_trace(path, calls, calls_after_step):
    out = {"n_calls": np.int64(len(calls)), "calls_after_step": np.asarray(calls_after_step, np.int64)}
    for i, c in enumerate(calls):
        pre = "c%d." % i
        out[pre + "fn"] = np.asarray(c["fn"])
        for k in ("per_tensor", "use_nvlamb"):                       # Optional[bool] arguments: -1 = None
            if k in c:
                out[pre + k] = np.int64(-1 if c[k] is None else int(bool(c[k])))
        for k in _TRACE_SCALARS:
            if k in c:
                out[pre + k] = np.float64(c[k])
        for k in _TRACE_ARRAYS:
            if k in c:
                out[pre + k] = np.asarray(c[k])
        for which in ("lists", "out"):
            if which in c:
                out[pre + which + ".n"] = np.asarray([len(l) for l in c[which]], np.int64)
                for li, l in enumerate(c[which]):
                    for ti, a in enumerate(l):
                        out["%s%s.%d.%d" % (pre, which, li, ti)] = a
    np.savez_compressed(path, **out)


def load_call_trace(path):
    z = np.load(path)
    calls = []
    for i in range(int(z["n_calls"])):
        pre = "c%d." % i
        c = {"fn": str(z[pre + "fn"])}
        for k in ("per_tensor", "use_nvlamb"):
            if pre + k in z.files:
                v = int(z[pre + k])
                c[k] = None if v < 0 else bool(v)
        for k in _TRACE_SCALARS + _TRACE_ARRAYS:
            if pre + k in z.files:
                c[k] = z[pre + k] if k in _TRACE_ARRAYS else float(z[pre + k])
        for which in ("lists", "out"):
            if pre + which + ".n" in z.files:
                c[which] = [[z["%s%s.%d.%d" % (pre, which, li, ti)] for ti in range(int(n))]
                            for li, n in enumerate(z[pre + which + ".n"])]
        calls.append(c)
    return calls, z["calls_after_step"].tolist()
