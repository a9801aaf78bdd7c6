"""CPU stand-in for the reference's compiled `fused_lamb_CUDA` module, backed by oracle/lamb_oracle.py.

TEST INFRASTRUCTURE ONLY (imported by oracle/make_golden.py and tests/): it lets the reference's UNMODIFIED
`FusedLAMBAMP.step` (LanguageModeling/BERT/lamb_amp_opt/fused_lamb/fused_lamb.py:131-260) run on CPU tensors, so the
host sequence of that class -- gradient lists per dtype, norm blending, max_grad_norm * scale, `step += (found_inf
!= 1)`, the argument order of both entry points -- is pinned by the reference's own Python; the per-element
arithmetic of multi_tensor_lamb.cu (which has no CPU implementation anywhere) stays the numpy restatement of
oracle/lamb_oracle.py.  Same signatures and in-place semantics as csrc/frontend.cpp:3-32.
"""
import numpy as np
import torch

from . import lamb_oracle as L


def _np(t):
    return t.detach().float().numpy() if t.dtype != torch.float32 else t.detach().numpy()


def multi_tensor_l2norm(chunk_size, noop_flag, tensor_lists, per_tensor=False):
    """multi_tensor_l2norm_kernel.cu:28-151,153-230: (sqrt(sum x^2) over all tensors, per-tensor norms or empty);
    a set noop flag leaves the zero-initialised outputs untouched (:40, :121); a non-finite element sets it (:104)."""
    ts = tensor_lists[0]
    total = torch.zeros(1, dtype=torch.float32)
    per = torch.zeros(len(ts) if per_tensor else 0, dtype=torch.float32)
    if int(noop_flag.item()) != 0:
        return total, per
    tot, pn = L.l2norm([_np(t) for t in ts])
    if not np.isfinite(tot):
        # :103-104 the functor raises the flag; `cleanup` (:121-123) then returns before it writes: the outputs stay at::zeros
        noop_flag.fill_(1)
        return total, per
    total[0] = float(tot)
    if per_tensor:
        per.copy_(torch.from_numpy(np.asarray(pn, np.float32)))
    return total, per


def multi_tensor_lamb(chunk_size, noop_flag, tensor_lists, lr, beta1, beta2, epsilon, step, bias_correction,
                      weight_decay, grad_averaging, mode, global_grad_norm, max_grad_norm, use_nvlamb, found_inf,
                      inv_scale):
    """multi_tensor_lamb.cu:371-500.  g <- update, m, v, p (and the fp16 model copy) in place; nothing happens when
    the noop flag is set (:63, :265)."""
    if int(noop_flag.item()) != 0:
        return
    g, p, m, v = tensor_lists[:4]
    copies = tensor_lists[4] if len(tensor_lists) == 5 else None
    gdt = np.float16 if g[0].dtype == torch.float16 else np.float32
    upd, p2, m2, v2, cp = L.lamb_step(
        [_np(t).astype(gdt) for t in g], [_np(t) for t in p], [_np(t) for t in m], [_np(t) for t in v],
        np.float32(lr.item()), beta1, beta2, epsilon, int(step.item()), bool(bias_correction), weight_decay,
        bool(grad_averaging), int(mode), np.float32(global_grad_norm.item()), np.float32(max_grad_norm.item()),
        bool(use_nvlamb), inv_scale=np.float32(inv_scale.item()), grad_dtype=gdt,
        model_copy_dtype=np.float16 if copies is not None else None)
    with torch.no_grad():
        for i in range(len(g)):
            g[i].copy_(torch.from_numpy(np.asarray(upd[i]).astype(np.float32)).to(g[i].dtype))
            p[i].copy_(torch.from_numpy(p2[i]))
            m[i].copy_(torch.from_numpy(m2[i]))
            v[i].copy_(torch.from_numpy(v2[i]))
            if copies is not None:
                copies[i].copy_(torch.from_numpy(cp[i].astype(np.float32)).to(copies[i].dtype))
