"""LAMB / L2-norm / SGD oracle (numpy fp32).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference ships no CPU implementation and no numeric test of
BERT/lamb_amp_opt/csrc/multi_tensor_lamb.cu; this file restates the .cu arithmetic line by line
(citations below, relative to /root/reference/PyTorch/LanguageModeling/BERT/lamb_amp_opt/) and is
cross-checked in tests/test_oracle_lamb.py against an independent float64 closed form.
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
