"""Device-resident tensor-list tables + wrappers for the multi-tensor optimizer kernels.

Host side of csrc/multi_tensor.hip.  Mirrors the call shapes of the reference's
fused_lamb_CUDA.multi_tensor_l2norm / multi_tensor_lamb
(BERT/lamb_amp_opt/csrc/frontend.cpp:3-32) used through apex's multi_tensor_applier
(BERT/lamb_amp_opt/fused_lamb/fused_lamb.py:167-191,240-258).
"""
import numpy as np
import torch

from . import _cabi as C

CHUNK = 2048 * 32   # apex multi_tensor_applier default chunk (65536 elements)


def streaming_chunk(lists, want_blocks=1024, lo=2048):
    """Chunk size for an elementwise multi-tensor pass (SGD) over `lists`: one workgroup per chunk, so a small parameter set
    cut into apex's 65536-element chunks is a few dozen workgroups each looping 32 dependent trips (DLRM's 2.2 M MLP
    parameters: 34 workgroups, 45 us per launch for 31 MB).  Power of two in [lo, CHUNK] giving ~want_blocks chunks."""
    total = sum(int(t.numel()) for t in lists[0])
    c = CHUNK
    while c > lo and total // c < want_blocks:
        c //= 2
    return c


class TensorTable:
    """int64 device table {size[n] | chunk_start[n+1] | ptr[list][n]} for `lists` (list of lists)."""

    def __init__(self, lists, chunk=CHUNK):
        n = len(lists[0])
        for lst in lists:
            if len(lst) != n:
                raise ValueError("tensor lists must have equal length")
        if any(t is None for t in lists[0]):
            raise ValueError("the first list holds every tensor")
        dev = lists[0][0].device if n else torch.device("cuda")
        for li, lst in enumerate(lists):
            for t, ref in zip(lst, lists[0]):
                if t is None:                    # an OPTIONAL list entry (SGD's 16-bit model copy of a tensor that has none): pointer 0
                    if li == 0:
                        raise ValueError("the first list holds every tensor")
                    continue
                if not t.is_contiguous():
                    raise ValueError("multi-tensor kernels need contiguous tensors")
                if t.numel() != ref.numel() or t.device != dev:
                    raise ValueError("tensors at the same list position must match in size and device")
        C.require_cuda(*[t for lst in lists for t in lst if t is not None])
        self.n, self.n_lists, self.chunk = n, len(lists), chunk
        sizes = np.asarray([t.numel() for t in lists[0]], dtype=np.int64)
        nchunks = (sizes + chunk - 1) // chunk
        start = np.concatenate([[0], np.cumsum(nchunks)]).astype(np.int64)
        ptrs = np.asarray([[t.data_ptr() if t is not None else 0 for t in lst] for lst in lists], dtype=np.int64).reshape(-1)
        host = np.concatenate([sizes, start, ptrs]).astype(np.int64)
        self.total_chunks = int(start[-1])
        self.total_elems = int(sizes.sum())
        self.key = host.tobytes()
        self.table = torch.from_numpy(host).to(dev)
        self.dtypes = [next((t.dtype for t in lst if t is not None), torch.float32) for lst in lists]
        self.device = dev
        self._keep = lists   # keep the tensors alive while the table exists

    @staticmethod
    def key_of(lists, chunk=CHUNK):
        return (chunk,) + tuple(t.data_ptr() if t is not None else 0 for lst in lists for t in lst) + tuple(t.numel() for t in lists[0])


def _note(table, bytes_per_elem):
    """Algorithmic bytes + a per-table tag for bench.py's kernel timer: launches over different tables (the decay / no-decay
    groups, the whole-model norm) are different shapes of one entry point and must not share a replay record."""
    C.annotate(bytes=float(table.total_elems) * bytes_per_elem, tag="%dt,%de" % (table.n, table.total_elems))


class TableCache:
    """Re-use a device table while the tensor addresses are unchanged (grads may be reallocated)."""

    def __init__(self):
        self._d = {}

    def get(self, tag, lists, chunk=CHUNK):
        k = TensorTable.key_of(lists, chunk)
        hit = self._d.get(tag)
        if hit is not None and hit[0] == k:
            return hit[1]
        tab = TensorTable(lists, chunk)
        self._d[tag] = (k, tab)
        return tab


def l2norm(table, noop_flag=None, per_tensor=False):
    """-> (norm[1], per_tensor_norms[n] or empty): fused_lamb_CUDA.multi_tensor_l2norm."""
    dev = table.device
    ret = torch.empty(1, dtype=torch.float32, device=dev)
    per = torch.empty(table.n if per_tensor else 0, dtype=torch.float32, device=dev)
    scratch = torch.empty(max(table.total_chunks, 1), dtype=torch.float32, device=dev)
    _note(table, table._keep[0][0].element_size() if table.n else 4)
    C.call("dle_mt_l2norm", C.ptr(table.table), table.n, table.total_chunks, table.chunk, C.dt(table.dtypes[0]),
           C.ptr(scratch), C.ptr(ret), C.ptr(per) if per_tensor else 0, int(per_tensor), C.ptr(noop_flag),
           C.stream())
    return ret, per


def lamb_stage1(table, noop_flag, beta1, beta2, beta3, step, bias_correction, eps, mode, weight_decay,
                global_grad_norm, max_grad_norm, inv_scale):
    _note(table, 7 * 4)                 # reads g, p, m, v; writes the update (over g), m, v
    C.call("dle_mt_lamb_stage1", C.ptr(table.table), table.n, table.total_chunks, table.chunk,
           C.dt(table.dtypes[0]), C.ptr(noop_flag), beta1, beta2, beta3, C.ptr(step), int(bias_correction),
           eps, int(mode), weight_decay, C.ptr(global_grad_norm), C.ptr(max_grad_norm), C.ptr(inv_scale),
           C.stream())


def lamb_stage1_norms(table, noop_flag, beta1, beta2, beta3, step, bias_correction, eps, mode, weight_decay,
                      global_grad_norm, max_grad_norm, inv_scale):
    """lamb_stage1 that also returns (param_norm, update_norm), fp32 [n] each: the per-tensor l2norm sweeps over p (before the
    step) and over the update (after stage 1) that multi_tensor_lamb_cuda runs around the stage, taken from per-chunk partial sums
    the stage leaves (include/dle_mi355x.h, dle_mt_lamb_stage1_norms).  weight_decay != 0."""
    dev = table.device
    pn = torch.empty(table.n, dtype=torch.float32, device=dev)
    un = torch.empty(table.n, dtype=torch.float32, device=dev)
    scratch = torch.empty(2 * max(table.total_chunks, 1), dtype=torch.float32, device=dev)
    _note(table, 7 * 4)
    C.call("dle_mt_lamb_stage1_norms", C.ptr(table.table), table.n, table.total_chunks, table.chunk,
           C.dt(table.dtypes[0]), C.ptr(noop_flag), beta1, beta2, beta3, C.ptr(step), int(bias_correction),
           eps, int(mode), weight_decay, C.ptr(global_grad_norm), C.ptr(max_grad_norm), C.ptr(inv_scale),
           C.ptr(scratch), C.ptr(pn), C.ptr(un), C.stream())
    return pn, un


def lamb_stage2(table, noop_flag, param_norm, update_norm, lr, weight_decay, use_nvlamb):
    _note(table, 3 * 4 + (2 if table.n_lists == 3 else 0))      # reads the update and p, writes p (+ the 16-bit model copy)
    C.call("dle_mt_lamb_stage2", C.ptr(table.table), table.n, table.total_chunks, table.chunk,
           C.dt(table.dtypes[0]), C.dt(table.dtypes[2]) if table.n_lists == 3 else -1, C.ptr(noop_flag), C.ptr(param_norm),
           C.ptr(update_norm), C.ptr(lr), weight_decay, int(bool(use_nvlamb)), C.stream())


def sgd(table, lr, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False, first_step=False,
        skip_flag=None, inv_scale=None, has_momentum=None, model_copy=False):
    """lists: g, p [, momentum buffer] [, 16-bit model copy (model_copy=True -> last list; None entries: tensors without one)]."""
    lr_dev = lr if isinstance(lr, torch.Tensor) else None
    if has_momentum is None:
        has_momentum = (table.n_lists - int(model_copy)) >= 3
    copy_dt = C.dt(table.dtypes[-1]) if model_copy else -1
    _note(table, 4 * (3 + 2 * int(bool(has_momentum))) + (2 if model_copy else 0))
    C.call("dle_mt_sgd", C.ptr(table.table), table.n, table.total_chunks, table.chunk, C.dt(table.dtypes[0]),
           int(has_momentum), C.ptr(skip_flag), C.ptr(lr_dev), 0.0 if lr_dev is not None else float(lr),
           momentum, dampening, weight_decay, int(nesterov), int(first_step), C.ptr(inv_scale), copy_dt,
           C.stream())


def adam(table, lr, beta1, beta2, eps, weight_decay, step, skip_flag=None, inv_scale=None, grad_norm=None,
         max_grad_norm=0.0):
    """lists: g, p, exp_avg, exp_avg_sq (fp32).  GradScaler.unscale_ + clip_grad_norm_ + torch.optim.Adam.step in one pass
    (SpeechSynthesis/Tacotron2/train.py:400-401,487-497); `step` = int32 device word holding THIS update's step number."""
    if table.n_lists != 4 or any(d != torch.float32 for d in table.dtypes):
        raise ValueError("adam expects four fp32 lists: g, p, exp_avg, exp_avg_sq")
    lr_dev = lr if isinstance(lr, torch.Tensor) else None
    _note(table, 7 * 4)
    C.call("dle_mt_adam", C.ptr(table.table), table.n, table.total_chunks, table.chunk, C.ptr(skip_flag), C.ptr(lr_dev),
           0.0 if lr_dev is not None else float(lr), beta1, beta2, eps, weight_decay, C.ptr(step), C.ptr(inv_scale),
           C.ptr(grad_norm), float(max_grad_norm if grad_norm is not None else 0.0), C.stream())
