"""One call that lets the reference's UNMODIFIED Python load against libdle_mi355x.so.

    import dle_reference_shims; dle_reference_shims.install()

* puts this directory on sys.path (dllogger, pynvml, apex, amp_C, fused_lamb_CUDA, absl, lddl, h5py, boto3, botocore);
* registers the four compiled modules `dlrm.cuda_ext` expects -- `fused_embedding`, `interaction_ampere`,
  `interaction_volta`, `sparse_gather` -- with the pybind names and argument order of
  dlrm/cuda_src/pytorch_embedding_ops.cpp:3-21, dot_based_interact_*/pytorch_ops.cpp:3-12 and
  sparse_gather/sparse_pytorch_ops.cpp:3-16 (including the (embedding, indices) positional order of gather_gpu_fwd);
* maps F.gelu(x, approximate=True) to "tanh" (BERT/modeling.py:122 passes a bool that torch >= 2 rejects).
Outputs are freshly allocated on the input's device, the stream is torch's current stream, errors raise
RuntimeError / ValueError from the C ABI's return code -- the conventions of the reference's boundary."""
import os
import sys
import types

_HERE = os.path.dirname(os.path.abspath(__file__))
_installed = False


def _cuda_ext_modules():
    import torch
    from deeplearningexamples_amd import functional as F

    def _sparse(rows, values, num_rows):
        return torch.sparse_coo_tensor(rows.reshape(1, -1), values.reshape(-1, values.shape[-1]),
                                       (num_rows, values.shape[-1]), check_invariants=False)

    fused = types.ModuleType("dlrm.cuda_ext.fused_embedding")

    def gather_gpu_fused_fwd(embedding, indices, offsets, amp_train):
        return F.emb_gather_fwd(embedding, indices, offsets, out_dtype=torch.float16 if amp_train else torch.float32)

    def gather_gpu_fused_bwd(embedding, indices, offsets, upstreamGrad):
        rows = F.emb_offset_indices(indices, offsets)
        return _sparse(rows, F.emb_grad_values(upstreamGrad.contiguous()), embedding.shape[0])

    fused.gather_gpu_fused_fwd, fused.gather_gpu_fused_bwd = gather_gpu_fused_fwd, gather_gpu_fused_bwd

    def _interaction(name):
        m = types.ModuleType(name)

        def dotBasedInteractFwd(input, bottom_mlp_output):
            return F.dot_interact_fwd(input.contiguous())      # bottom_mlp_output is input[:, 0, :]

        def dotBasedInteractBwd(input, upstreamGrad):
            grad, mlp_grad = F.dot_interact_bwd(input.contiguous(), upstreamGrad.contiguous())
            return [grad, mlp_grad]

        m.dotBasedInteractFwd, m.dotBasedInteractBwd = dotBasedInteractFwd, dotBasedInteractBwd
        return m

    sparse = types.ModuleType("dlrm.cuda_ext.sparse_gather")

    def gather_gpu_fwd(weight, indices):                         # called positionally as (embedding, indices)
        return F.emb_gather_fwd(weight, indices)

    def gather_gpu_bwd(grad, indices, num_features):
        return _sparse(indices, F.emb_grad_values(grad.contiguous()), num_features)

    def gather_gpu_bwd_fuse_sgd(grad, indices, lr, weight):
        F.emb_sparse_sgd_(weight, indices.reshape(-1), grad.reshape(-1, grad.shape[-1]).contiguous(), float(lr))

    sparse.gather_gpu_fwd, sparse.gather_gpu_bwd = gather_gpu_fwd, gather_gpu_bwd
    sparse.gather_gpu_bwd_fuse_sgd = gather_gpu_bwd_fuse_sgd
    return {"dlrm.cuda_ext.fused_embedding": fused,
            "dlrm.cuda_ext.interaction_ampere": _interaction("dlrm.cuda_ext.interaction_ampere"),
            "dlrm.cuda_ext.interaction_volta": _interaction("dlrm.cuda_ext.interaction_volta"),
            "dlrm.cuda_ext.sparse_gather": sparse}


def cuda_ext_modules():
    """The four module objects (also usable without the reference tree, e.g. from tests)."""
    return _cuda_ext_modules()


def install(patch_gelu=True):
    global _installed
    if _installed:
        return
    if _HERE not in sys.path:
        sys.path.insert(0, _HERE)
    for name, mod in _cuda_ext_modules().items():
        sys.modules.setdefault(name, mod)
    if patch_gelu:
        import torch.nn.functional as TF
        orig = TF.gelu

        def gelu(input, approximate="none"):
            if approximate is True:
                approximate = "tanh"
            elif approximate is False:
                approximate = "none"
            return orig(input, approximate=approximate)

        TF.gelu = gelu
    _installed = True
