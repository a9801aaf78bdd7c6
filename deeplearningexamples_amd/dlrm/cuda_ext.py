"""Autograd-Function form of the DLRM custom ops -- the reference's `dlrm.cuda_ext` plugin boundary.

Same names, argument order and outputs as (Recommendation/DLRM/dlrm/cuda_ext/):
    dot_based_interact.py:25-43        dotBasedInteract(input, bottom_mlp_output)
    fused_gather_embedding.py:26-45    buckle_embedding_fused_gather(embedding, indices, offsets, amp_train)
    sparse_embedding.py:24-67          embedding_gather(embedding, indices), JointSparseEmbedding
so `dlrm/nn/{interactions,embeddings}.py` of the reference can call them unchanged.  They run the HIP kernels of
libdle_mi355x.so; the sparse weight gradient is returned as an (uncoalesced) sparse COO tensor exactly like the
reference (gather_gpu_fused_pytorch_impl.cu:97-99, at::embedding_sparse_backward).
"""
import torch
from torch import nn
from torch.autograd import Function

from .. import functional as F


class DotBasedInteract(Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.half)
    def forward(ctx, input, bottom_mlp_output):
        output = F.dot_interact_fwd(input)          # bottom_mlp_output == input[:, 0, :] (ignored by the kernels)
        ctx.save_for_backward(input)
        return output

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_output):
        input, = ctx.saved_tensors
        grad, mlp_grad = F.dot_interact_bwd(input, grad_output.contiguous())
        return grad, mlp_grad


dotBasedInteract = DotBasedInteract.apply


def _sparse_grad(rows, values, num_rows):
    dim = values.shape[-1]
    return torch.sparse_coo_tensor(rows.reshape(1, -1), values.reshape(-1, dim), (num_rows, dim),
                                   check_invariants=False)


class BuckleEmbeddingFusedGatherFunction(Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, embedding, indices, offsets, amp_train):
        output = F.emb_gather_fwd(embedding, indices, offsets,
                                  out_dtype=torch.float16 if amp_train else torch.float32)
        ctx.save_for_backward(embedding, indices, offsets)
        return output

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_output):
        embedding, indices, offsets = ctx.saved_tensors
        rows = F.emb_offset_indices(indices, offsets)
        values = F.emb_grad_values(grad_output.contiguous())
        return _sparse_grad(rows, values, embedding.shape[0]), None, None, None


buckle_embedding_fused_gather = BuckleEmbeddingFusedGatherFunction.apply


class EmbeddingGatherFunction(Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, embedding, indices):
        output = F.emb_gather_fwd(embedding, indices)
        ctx.save_for_backward(indices)
        ctx.num_features = embedding.size(0)
        return output

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_output):
        indices = ctx.saved_tensors[0]
        values = F.emb_grad_values(grad_output.contiguous())
        return _sparse_grad(indices, values, ctx.num_features), None


embedding_gather = EmbeddingGatherFunction.apply


class JointSparseEmbedding(nn.Module):
    def __init__(self, categorical_feature_sizes, embedding_dim, device="cuda"):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.categorical_feature_sizes = list(categorical_feature_sizes)
        self.register_buffer("offsets", torch.tensor([0] + list(categorical_feature_sizes)).cumsum(0).to(device))
        self.weights = nn.Parameter(torch.rand((int(self.offsets[-1].item()), embedding_dim), device=device))

    def forward(self, categorical_inputs):
        assert categorical_inputs.shape[1] == len(self.categorical_feature_sizes)
        return embedding_gather(self.weights, categorical_inputs + self.offsets[:-1])
