"""DLRM hot-path oracle (numpy / pure python).  TEST INFRASTRUCTURE ONLY.

Restates, on the CPU, the arithmetic of the reference's DLRM path.  Citations are
relative to /root/reference/PyTorch/Recommendation/DLRM/.
Pinned by tests/golden/dlrm_*.json|npz (generated from the reference itself by
oracle/make_golden.py).
"""
import itertools
import math

import numpy as np


# ---------------------------------------------------------------- placement (integer, bit-exact)
def gpu_batch_sizes(global_batch, num_gpus=4, batch_std=64, divisible_by=64):
    """dlrm/utils/distributed.py:102-114 -- per-GPU batch split.

    Enumerates non-decreasing tuples of multiples of ``divisible_by`` within
    +-batch_std of the mean that sum to the global batch, keeps the one with the
    largest product (first such tuple in lexicographic enumeration order on ties).
    """
    mean = global_batch // num_gpus
    cand = [x for x in range(mean - batch_std, mean + batch_std + 1) if x % divisible_by == 0]
    best, best_prod = None, None
    for combo in itertools.combinations_with_replacement(cand, num_gpus):
        if sum(combo) != global_batch:
            continue
        prod = 1
        for c in combo:
            prod *= c
        if best is None or prod > best_prod:      # strict: max() keeps the first maximum
            best, best_prod = combo, prod
    if best is None:
        raise RuntimeError("no per-GPU batch split for this configuration")
    return best


def stable_argsort(seq, reverse=False):
    """dlrm/utils/distributed.py:117-120.  Python's sort is stable, also with reverse=True
    (equal keys keep their original relative order)."""
    return [i for _, i in sorted(((x, i) for i, x in enumerate(seq)), key=lambda t: t[0], reverse=reverse)]


def buckets_greedy(sizes, n_buckets):
    """dlrm/utils/distributed.py:123-143 -- greedy 'largest table into the currently
    lightest open bucket' with a per-bucket table-count cap of ceil(T / n_buckets)."""
    cap = math.ceil(len(sizes) / n_buckets)
    todo = stable_argsort(sizes, reverse=True)
    open_buckets = [[] for _ in range(n_buckets)]
    closed = []
    pos = 0
    while pos < len(todo):
        b = open_buckets[0]
        b.append(todo[pos])
        pos += 1
        if len(b) == cap:
            closed.append(open_buckets.pop(0))
        # stable sort by current load (list.sort is stable: ties keep current order)
        open_buckets.sort(key=lambda idx: sum(sizes[i] for i in idx))
    return closed + open_buckets


def device_mapping(embedding_sizes, num_gpus=8):
    """dlrm/utils/distributed.py:146-176.  >4 GPUs: rank 0 keeps only the bottom MLP."""
    if num_gpus > 4:
        buckets = [[]] + buckets_greedy(embedding_sizes, num_gpus - 1)
    else:
        buckets = buckets_greedy(embedding_sizes, num_gpus)
    vectors = [len(b) for b in buckets]
    vectors[0] += 1
    return {"bottom_mlp": 0, "embedding": buckets, "vectors_per_gpu": vectors}


# ---------------------------------------------------------------- index arithmetic (integer, bit-exact)
def table_offsets(sizes):
    """dlrm/nn/embeddings.py:123 -- cumsum([0] + sizes) as int64."""
    return np.concatenate([[0], np.cumsum(np.asarray(sizes, dtype=np.int64))]).astype(np.int64)


def hash_indices(idx, sizes):
    """dlrm/nn/embeddings.py:132-134 -- idx[:, t] %= size_t (python/torch floor-mod on int64)."""
    return np.mod(idx.astype(np.int64), np.asarray(sizes, dtype=np.int64)[None, :])


def offset_indices(idx, offsets):
    """dlrm/nn/embeddings.py:137 / cuda_src/gather_gpu_fused.cu:161-175 -- row = idx + offsets[t]."""
    return idx.astype(np.int64) + np.asarray(offsets[:-1], dtype=np.int64)[None, :]


def tril_pairs(n_vec):
    """dlrm/nn/interactions.py:50-53 -- strict lower triangle in row-major order: (1,0),(2,0),(2,1),..."""
    rows = [i for i in range(n_vec) for _ in range(i)]
    cols = [j for i in range(n_vec) for j in range(i)]
    return np.asarray(rows, np.int64), np.asarray(cols, np.int64)


def padding_size(n):
    """dlrm/nn/interactions.py:20-22."""
    return ((n - 1) // 8 + 1) * 8 - n


def interact_out_width(n_vec, dim):
    raw = n_vec * (n_vec - 1) // 2 + dim
    return raw + padding_size(raw)


# ---------------------------------------------------------------- float ops
def embedding_gather(weight, rows):
    """dlrm/nn/embeddings.py:137, cuda_src/gather_gpu_fused.cu:107-159,
    cuda_src/sparse_gather/gather_gpu.cu:15-49 -- out[b, t, :] = W[rows[b, t], :]."""
    return weight[rows]


def sparse_sgd(weight, rows, grad_values, lr):
    """torch.optim.SGD on the sparse COO grad (scripts/main.py:482,605-606) ==
    W[row] -= lr * g summed over duplicate rows (cuda_src/sparse_gather/gather_gpu.cu:53-75).
    fp32; duplicates are accumulated in float64 here so the oracle is order-independent."""
    acc = np.zeros_like(weight, dtype=np.float64)
    np.add.at(acc, rows.reshape(-1), grad_values.reshape(-1, weight.shape[1]).astype(np.float64))
    return (weight.astype(np.float64) - lr * acc).astype(weight.dtype)


def dot_interact_fwd(x, out_dtype=None):
    """dlrm/nn/interactions.py:65-82 (and cuda_src/dot_based_interact/*_fwd.cu):
    out = [x[:,0,:] | (X X^T)[i,j] for i>j row-major | zero pad to a multiple of 8].
    Accumulates in float32 from the (possibly fp16/bf16-rounded) inputs, like the kernels
    (fp32 accumulators, dot_based_interact_fp16_fwd.cu:220-249)."""
    xf = x.astype(np.float32)
    b, r, c = xf.shape
    z = np.einsum("brc,bsc->brs", xf, xf, dtype=np.float32)
    ri, ci = tril_pairs(r)
    flat = z[:, ri, ci]
    pad = np.zeros((b, padding_size(r * (r - 1) // 2 + c)), np.float32)
    out = np.concatenate([xf[:, 0, :], flat, pad], axis=1)
    return out.astype(out_dtype or x.dtype)


def dot_interact_bwd(x, ugrad, out_dtype=None):
    """cuda_src/dot_based_interact/dot_based_interact_fp32_bwd.cu:93-169 (same contract as
    the fp16 WMMA kernel, dot_based_interact_fp16_bwd.cu:243-331):
      grad[b]      = U_sym[b] @ X[b]   with U_sym[i][j] = U_sym[j][i] = ugrad[b, C + tril(i,j)], diag 0
      mlp_grad[b]  = ugrad[b, :C]
    (autograd later adds mlp_grad onto row 0 because both came from the same tensor.)"""
    xf = x.astype(np.float32)
    uf = ugrad.astype(np.float32)
    b, r, c = xf.shape
    ri, ci = tril_pairs(r)
    u = np.zeros((b, r, r), np.float32)
    u[:, ri, ci] = uf[:, c:c + ri.size]
    u = u + np.transpose(u, (0, 2, 1))
    grad = np.einsum("brs,bsc->brc", u, xf, dtype=np.float32)
    mlp_grad = uf[:, :c].copy()
    dt = out_dtype or x.dtype
    return grad.astype(dt), mlp_grad.astype(dt)
