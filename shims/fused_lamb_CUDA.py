"""`fused_lamb_CUDA` -- the compiled module of LanguageModeling/BERT/lamb_amp_opt (csrc/frontend.cpp:3-32), on the
MI355X multi-tensor kernels.  Same argument order and in-place semantics as the reference:

    multi_tensor_l2norm(chunk_size, noop_flag, [[t...]], per_tensor) -> (norm[1], per_tensor_norms[n or 0])
    multi_tensor_lamb(chunk_size, noop_flag, [g, p, m, v(, p16)], lr, beta1, beta2, eps, step, bias_correction,
                      weight_decay, grad_averaging, mode, global_grad_norm, max_grad_norm, use_nvlamb, found_inf,
                      inv_scale) -> None        (g becomes the update; m, v, p, p16 updated in place)

Host sequence of multi_tensor_lamb.cu:371-500: l2norm(params, per tensor) -> stage 1 -> l2norm(updates, per tensor)
-> stage 2.  lr / step / norms / found_inf / inv_scale are device tensors (no host sync)."""
from deeplearningexamples_amd import multi_tensor as _mt

_cache = _mt.TableCache()


def multi_tensor_l2norm(chunk_size, noop_flag, tensor_lists, per_tensor=False):
    table = _cache.get("l2norm", [list(tensor_lists[0])], chunk_size)
    total, per = _mt.l2norm(table, noop_flag, bool(per_tensor))
    return total, per


def multi_tensor_lamb(chunk_size, noop_flag, tensor_lists, lr, beta1, beta2, epsilon, step, bias_correction,
                      weight_decay, grad_averaging, mode, global_grad_norm, max_grad_norm, use_nvlamb, found_inf,
                      inv_scale):
    g, p, m, v = (list(x) for x in tensor_lists[:4])
    copies = list(tensor_lists[4]) if len(tensor_lists) == 5 else None
    beta3 = 1.0 - beta1 if grad_averaging else 1.0
    _, p_norm = _mt.l2norm(_cache.get("lamb_p", [p], chunk_size), noop_flag, True)
    _mt.lamb_stage1(_cache.get("lamb_s1", [g, p, m, v], chunk_size), noop_flag, beta1, beta2, beta3, step,
                    bool(bias_correction), epsilon, int(mode), weight_decay, global_grad_norm, max_grad_norm, inv_scale)
    _, u_norm = _mt.l2norm(_cache.get("lamb_u", [g], chunk_size), noop_flag, True)
    lists2 = [g, p, copies] if copies is not None else [g, p]
    _mt.lamb_stage2(_cache.get("lamb_s2", lists2, chunk_size), noop_flag, p_norm, u_norm, lr, weight_decay,
                    bool(use_nvlamb))
