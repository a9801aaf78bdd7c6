"""ctypes binding of libdle_mi355x.so (the C ABI declared in include/dle_mi355x.h).

The product path has NO fallback: if the library is missing or a call fails this raises.
(The reference's custom ops raise RuntimeError/ValueError through TORCH_CHECK /
std::invalid_argument, e.g. DLRM/dlrm/cuda_src/sparse_gather/gather_gpu.cu:80-100,
dot_based_interact_ampere/dot_based_interact_pytorch_types.cu:38,72 -- same behaviour here.)
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DLE_LIB_PATH: another build of the same library (tools/build_lib_at.sh <git revision>) -- same-box A/B of two source trees
LIB_PATH = os.environ.get("DLE_LIB_PATH") or os.path.join(_HERE, "lib", "libdle_mi355x.so")

F32, F16, BF16 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_GELU, ACT_RELU_BWD, ACT_ADD, ACT_GELU_BWD, ACT_TANH, ACT_TANH_BWD = 0, 1, 2, 3, 4, 5, 6, 7
ACT_ADD_MASKED, ACT_MUL, ACT_GELU_DAUX = 8, 9, 10

_DT = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}

c_void_p, c_int, c_i64, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
c_u64 = ctypes.c_uint64

# name -> (restype, argtypes); mirrors include/dle_mi355x.h one to one
_SIGS = {
    "dle_last_error": (ctypes.c_char_p, []),
    "dle_abi_version": (c_int, []),
    "dle_device_check": (c_int, [c_int, ctypes.c_char_p, c_int]),
    "dle_dot_interact_out_width": (c_int, [c_int, c_int]),
    "dle_dot_interact_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dle_dot_interact_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                     c_int, c_void_p]),
    "dle_dot_interact_bwd_checked": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                             c_int, c_void_p, c_void_p]),
    "dle_emb_gather_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int,
                                   c_int, c_i64, c_void_p]),
    "dle_emb_offset_indices": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_void_p]),
    "dle_emb_grad_values": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_void_p]),
    "dle_emb_sparse_sgd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                   c_i64, c_int, c_int, c_i64, c_int, c_void_p]),
    "dle_emb_small_table_mask": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "dle_emb_sgd_dedup": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_float, c_void_p, c_void_p, c_i64, c_int, c_int, c_i64, c_int,
                                  c_void_p]),
    "dle_emb_sgd_dedup_ws": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_float, c_void_p, c_void_p, c_i64, c_int, c_int, c_i64, c_int,
                                     c_void_p, c_i64, c_void_p]),
    "dle_emb_onehot_workspace_bytes": (c_i64, [c_int, c_i64]),
    "dle_emb_sgd_workspace_bytes": (c_i64, [c_void_p, c_int, c_int, c_i64]),
    "dle_emb_onehot_try": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_int, c_i64, c_int, c_int, c_i64, c_int, c_void_p, c_i64, c_void_p]),
    "dle_cast_rows": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_i64, c_i64, c_int, c_int, c_void_p]),
    "dle_bce_logits": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_int, c_void_p]),
    "dle_conv1x1_bnbwd_dgrad": (c_int, [c_void_p] * 16 + [c_i64, c_int, c_int, c_int, c_int, c_void_p]),
    "dle_conv1x1_bnbwd_groups": (c_int, [c_int]),
    "dle_gemm_expand_masked_bnred": (c_int, [c_void_p] * 10 + [c_i64, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_int, c_int, c_void_p]),
    "dle_bn_bwd_finish": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "dle_gemm_colsum": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_int,
                                c_int, c_int, c_void_p, c_i64, c_void_p]),
    "dle_head_bce_workspace_bytes": (c_i64, [c_i64, c_int]),
    "dle_head_bce_fwd_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_int, c_i64, c_i64, c_int, c_void_p]),
    "dle_amp_update_scale": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_int, c_int,
                                     c_void_p]),
    "dle_check_nonfinite": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_void_p]),
    "dle_axpby_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_float, c_i64, c_void_p]),
    "dle_act_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p]),
    "dle_relu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_i64, c_i64, c_i64, c_int, c_void_p]),
    "dle_gemm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                         c_i64, c_i64, c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                         c_void_p, c_i64, c_void_p]),
    "dle_conv2d_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 12 + [c_void_p]),
    "dle_conv2d_dgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 10 + [c_void_p]),
    "dle_conv2d_wgrad": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 12 + [c_void_p, c_i64, c_void_p]),
    "dle_nchw_to_nhwc": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_i64, c_int, c_int, c_void_p]),
    "dle_gemm_expand_add_up2": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_i64, c_int,
                                        c_int, c_int, c_int, c_void_p]),
    "dle_conv1x1_bnload_groups": (c_int, [c_int, c_int, c_int]),
    "dle_conv1x1_bnload_fwd": (c_int, [c_void_p] * 11 + [c_i64, c_int, c_int, c_int, c_int, c_void_p]),
    "dle_bn_relu_maxpool_fwd": (c_int, [c_void_p] * 8 + [c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dle_wgrad1x1_workspace": (c_i64, []),
    "dle_wgrad1x1_mode": (c_int, [c_int]),
    "dle_wgrad1x1_workspace_for": (c_i64, [c_int, c_int, c_int]),
    "dle_gemm8_mode": (c_int, [c_int]),
    "dle_gemm8_min_items": (c_int, [c_int]),
    "dle_gemm8_launch_count": (c_i64, []),
    "dle_wgrad1x1_try": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_i64, c_void_p]),
    "dle_conv3x3_wgrad_workspace": (c_i64, []),
    "dle_conv3x3_wgrad_mode": (c_int, [c_int]),
    "dle_stem_conv7_groups": (c_int, [c_int, c_int]),
    "dle_stem_conv7_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_int, c_void_p]),
    "dle_stem_conv7_wgrad_workspace": (c_i64, [c_int, c_int]),
    "dle_stem_conv7_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dle_stem_pack_weight": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "dle_bn_workspace_bytes": (c_i64, [c_i64, c_int]),
    "dle_bn_fwd_stats": (c_int, [c_void_p, c_i64, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_i64, c_int, c_void_p]),
    "dle_conv2d_fwd_colstats": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 10 + [c_void_p, c_i64, c_void_p, c_void_p]),
    "dle_bn_stats_from_partials": (c_int, [c_void_p, c_int, c_i64, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_i64, c_void_p]),
    "dle_bn_fwd_apply": (c_int, [c_void_p] * 8 + [c_i64, c_int, c_int, c_int, c_void_p]),
    "dle_rccl_available": (c_int, []),
    "dle_rccl_unique_id": (c_int, [c_void_p]),
    "dle_rccl_init": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "dle_rccl_count": (c_int, [c_void_p]),
    "dle_rccl_destroy": (c_int, [c_void_p]),
    "dle_rccl_allreduce": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p]),
    "dle_rccl_broadcast": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_void_p]),
    "dle_rccl_alltoallv": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "dle_gemm8_relu_bits_try": (c_int, [c_void_p] * 5 + [c_int, c_int, c_int, c_i64, c_i64, c_int, c_void_p]),
    "dle_gemm_colsum_bits": (c_int, [c_void_p] * 5 + [c_int, c_int, c_int, c_i64, c_i64, c_int, c_int, c_void_p, c_i64, c_void_p]),
    "dle_bn_bwd_reduce2": (c_int, [c_void_p] * 12 + [c_i64, c_int, c_void_p, c_i64, c_int, c_void_p]),
    "dle_bn_fwd_apply2": (c_int, [c_void_p] * 12 + [c_i64, c_int, c_int, c_void_p]),
    "dle_conv1x1_bnload_fwd2": (c_int, [c_void_p] * 15 + [c_i64, c_int, c_int, c_int, c_int, c_void_p]),
    "dle_bn_bwd_reduce": (c_int, [c_void_p] * 8 + [c_i64, c_int, c_int, c_void_p, c_i64, c_int, c_void_p]),
    "dle_bn_bwd_apply": (c_int, [c_void_p] * 11 + [c_i64, c_int, c_int, c_void_p]),
    "dle_maxpool_fwd": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 8 + [c_void_p]),
    "dle_maxpool_bwd": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 8 + [c_void_p]),
    "dle_pool_bn_bwd_workspace_bytes": (c_i64, [c_int] * 4),
    "dle_pool_bn_bwd": (c_int, [c_void_p] * 10 + [c_int] * 4 + [c_void_p, c_i64, c_int, c_void_p]),
    "dle_avgpool_fwd": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_void_p]),
    "dle_avgpool_bwd": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_void_p]),
    "dle_upsample_zero": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dle_softmax_xent": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_i64,
                                 c_i64, c_float, c_i64, c_int, c_void_p]),
    "dle_gemm_batched": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_int, c_int,
                                 c_int, c_int, c_float, c_int, c_int, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_void_p]),
    "dle_layernorm_fwd": (c_int, [c_void_p] * 8 + [c_i64, c_int, c_float, c_int, c_void_p]),
    "dle_layernorm_workspace_bytes": (c_i64, [c_int]),
    "dle_layernorm_bwd": (c_int, [c_void_p] * 8 + [c_i64, c_int, c_int, c_void_p, c_i64, c_int, c_void_p]),
    "dle_embed_sum": (c_int, [c_void_p] * 6 + [c_i64, c_int, c_int, c_int, c_void_p]),
    "dle_embed_scatter_add": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p]),
    "dle_rows_select_sum": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_void_p, c_i64, c_int,
                                    c_void_p]),
    "dle_rows_gather": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p]),
    "dle_rows_scatter": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_void_p]),
    "dle_softmax_fwd": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_float, c_int, c_void_p]),
    "dle_softmax_bwd": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_float, c_int, c_void_p]),
    "dle_dropout_add_layernorm_bwd": (c_int, [c_void_p] * 6 + [c_float] + [c_void_p] * 5 + [c_i64, c_int, c_int, c_void_p, c_i64, c_int, c_void_p]),
    "dle_dropout_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_float, c_u64, c_u64, c_void_p, c_int, c_void_p]),
    "dle_dropout_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_float, c_int, c_void_p]),
    "dle_dropout_add_layernorm_fwd": (c_int, [c_void_p] * 9 + [c_i64, c_int, c_float, c_float, c_u64, c_u64, c_void_p, c_int, c_void_p]),
    "dle_softmax_dropout_fwd": (c_int, [c_void_p] * 4 + [c_i64, c_int, c_int, c_float, c_float, c_u64, c_u64, c_void_p, c_int, c_void_p]),
    "dle_softmax_dropout_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_float, c_float, c_int, c_void_p]),
    "dle_u8_nchw_normalize_nhwc": (c_int, [c_void_p] * 4 + [c_i64, c_int, c_i64, c_int, c_int, c_void_p]),
    "dle_conv2d_dgrad_s2": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_void_p, c_i64, c_int, c_void_p]),
    "dle_attention_supported": (c_int, [c_int, c_int]),
    "dle_attention_stats_floats": (c_int, [c_int]),
    "dle_attention_fwd": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_float, c_float, c_u64, c_u64, c_void_p, c_int, c_void_p]),
    "dle_attention_bwd": (c_int, [c_void_p] * 6 + [c_int] * 4 + [c_float, c_float, c_u64, c_u64, c_void_p, c_int, c_void_p]),
    "dle_attention_bwd_keep": (c_int, [c_void_p] * 7 + [c_int] * 4 + [c_float, c_float, c_u64, c_u64, c_void_p, c_int, c_void_p]),
    "dle_colsum": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_i64, c_int, c_int, c_void_p, c_i64, c_void_p]),
    "dle_colsum_batched_workspace_bytes": (c_i64, [c_int, c_i64, c_int]),
    "dle_colsum_batched": (c_int, [c_void_p, c_int, c_i64, c_int, c_i64, c_int, c_void_p, c_i64, c_void_p]),
    "dle_mt_table_len": (c_i64, [c_int, c_int]),
    "dle_mt_table_fill": (c_i64, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int]),
    "dle_mt_l2norm": (c_int, [c_void_p, c_int, c_i64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                              c_void_p, c_void_p]),
    "dle_mt_lamb_stage1": (c_int, [c_void_p, c_int, c_i64, c_int, c_int, c_void_p, c_float, c_float, c_float,
                                   c_void_p, c_int, c_float, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                   c_void_p]),
    "dle_mt_lamb_stage1_norms": (c_int, [c_void_p, c_int, c_i64, c_int, c_int, c_void_p, c_float, c_float, c_float,
                                         c_void_p, c_int, c_float, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p]),
    "dle_mt_lamb_stage2": (c_int, [c_void_p, c_int, c_i64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_float, c_int, c_void_p]),
    "dle_mt_sgd": (c_int, [c_void_p, c_int, c_i64, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_float,
                           c_float, c_float, c_int, c_int, c_void_p, c_int, c_void_p]),
    "dle_mt_adam": (c_int, [c_void_p, c_int, c_i64, c_int, c_void_p, c_void_p, c_float, c_float, c_float, c_float,
                            c_float, c_void_p, c_void_p, c_void_p, c_float, c_void_p]),
    "dle_wg_taps": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_i64, c_int, c_void_p]),
    "dle_wg_taps_bwd": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_i64, c_i64, c_int, c_void_p]),
    "dle_wg_gate_fwd": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_i64, c_int, c_void_p]),
    "dle_wg_gate_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_i64, c_i64, c_int, c_void_p]),
    "dle_wg_invconv_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p]),
    "dle_wg_invconv_bwd_partials": (c_int, [c_i64]),
    "dle_wg_invconv_bwd": (c_int, [c_void_p] * 8 + [c_float, c_void_p, c_i64, c_int, c_void_p]),
    "dle_wg_logdet_inv": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "dle_wg_coupling_partials": (c_int, [c_i64]),
    "dle_wg_coupling_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_void_p]),
    "dle_wg_coupling_bwd": (c_int, [c_void_p] * 6 + [c_float, c_i64, c_int, c_int, c_void_p]),
    "dle_wg_loss": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_float, c_i64, c_void_p, c_void_p, c_void_p]),
    "dle_wg_dz_init": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_i64, c_void_p]),
    "dle_wg_weight_norm_fwd": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]),
    "dle_wg_weight_norm_bwd": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_void_p]),
    "dle_wg_upsample_weight": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p]),
    "dle_wg_upsample_weight_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dle_wg_weight_norm_fwd_batched": (c_int, [c_void_p, c_int, c_i64, c_int, c_void_p]),
    "dle_wg_weight_norm_bwd_batched": (c_int, [c_void_p, c_int, c_i64, c_void_p]),
    "dle_wg_logdet_inv_batched": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "dle_t2_tanh_fwd": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_void_p]),
    "dle_t2_lstm_fwd": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p,
                                c_i64, c_float, c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_int, c_int, c_int, c_void_p]),
    "dle_t2_lstm_bwd": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_void_p, c_void_p,
                                c_i64, c_void_p, c_void_p, c_i64, c_float, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dle_t2_attention_fwd": (c_int, [c_void_p] * 10 + [c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p] + [c_int] * 7 + [c_void_p]),
    "dle_t2_attention_bwd": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64] + [c_void_p] * 14 + [c_int, c_int, c_void_p, c_void_p] +
                             [c_int] * 5 + [c_void_p]),
    "dle_a2a_blocks": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "dle_t2_sum_steps": (c_int, [c_void_p, c_void_p, c_int, c_i64, c_int, c_void_p]),
    "dle_t2_location_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dle_transpose_cast": (c_int, [c_void_p, c_void_p, c_int, c_int, c_i64, c_i64, c_int, c_int, c_void_p]),
    "dle_t2_mel_loss": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_i64,
                                c_int, c_int, c_void_p]),
    "dle_t2_lstm_gemm_fwd": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p,
                                     c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_float, c_int, c_int, c_int, c_int,
                                     c_void_p]),
    "dle_t2_mask_rows": (c_int, [c_void_p, c_i64, c_int, c_void_p, c_i64, c_int, c_float, c_int, c_void_p]),
}

_lib = None


class DleError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle.  Raises loudly when the HIP library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DleError(
                "libdle_mi355x.so is not built (%s). Run `python -m deeplearningexamples_amd.build` "
                "or __graft_entry__.build(); there is no CPU/eager fallback." % LIB_PATH)
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(h, name)          # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def declared_symbols():
    return sorted(_SIGS)


def dt(t_or_dtype):
    d = t_or_dtype.dtype if isinstance(t_or_dtype, torch.Tensor) else t_or_dtype
    try:
        return _DT[d]
    except KeyError:
        raise ValueError("unsupported dtype %s (f32/f16/bf16 only)" % d)


def ptr(t):
    return 0 if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def check(rc, what):
    if rc == 0:
        return
    msg = lib().dle_last_error().decode("utf-8", "replace")
    if rc == -1:
        raise ValueError("%s: %s" % (what, msg))
    raise DleError("%s failed (hip error %d): %s" % (what, rc, msg))


class KernelTimer:
    """Optional per-entry-point HIP-event timing (bench.py's roofline leg).  Events are recorded on the
    stream the kernels are launched on (torch's current stream); nothing is synchronised until report()."""

    def __init__(self):
        self.records = []          # (name, start_event, end_event, meta)
        self.meta = None           # set by callers that know the algorithmic bytes/flops of the next call
        self.last = {}             # (name, tag) -> (fn, args) of the most recent such call, for replay()

    # entry points whose duration depends on state they consume (row lists, touched-row sets): never replayed
    # (the optimizer updates are not pure either: a replay would step the weights / moments again)
    _STATEFUL = ("emb_sgd", "emb_sparse", "emb_link", "emb_grad", "amp_update", "mt_lamb", "mt_sgd", "mt_adam")

    def replay(self, name, tag, iters=20, warmup=3, cold=False):
        """Average duration (ms) of the recorded call re-launched back to back: ONE event pair around `iters`
        launches on the launch stream, so the host-side cost of a per-call event pair (which leaves the queue idle
        between short kernels and inflates their event-to-event time) is out of the measurement.  Returns None for
        entry points that are not pure functions of their inputs.
        cold=True (launches whose operands fit in L2: a back-to-back replay would find them there, inside the step they come from
        the Infinity Cache / HBM): a 64 MB fill runs in front of every launch and each launch gets its own event pair (the fill
        keeps the queue busy up to the launch, so the pair measures the kernel, not an idle queue)."""
        if any(k in name for k in self._STATEFUL) or (name, tag) not in self.last:
            return None
        fn, args, st = self.last[(name, tag)]
        # the recorded arguments carry the stream the call was launched on (a side stream for leaves of the backward graph): the
        # timing events must sit on THAT stream, or they bracket nothing
        with torch.cuda.stream(st):
            return self._replay_on_current_stream(fn, args, iters, warmup, cold)

    def _replay_on_current_stream(self, fn, args, iters, warmup, cold):
        if cold:
            if getattr(self, "_flush", None) is None:
                self._flush = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
            pairs = []
            for i in range(warmup + iters):
                self._flush.fill_(i & 1)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                fn(*args)
                e.record()
                if i >= warmup:
                    pairs.append((s, e))
            torch.cuda.synchronize()
            return sum(s.elapsed_time(e) for s, e in pairs) / len(pairs)
        for _ in range(warmup):
            fn(*args)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn(*args)
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / iters

    def report(self):
        torch.cuda.synchronize()
        agg = {}
        for name, s, e, meta in self.records:
            key = (name, meta.get("tag") if meta else None)
            a = agg.setdefault(key, {"name": name, "tag": key[1], "calls": 0, "ms": 0.0, "bytes": 0.0, "flops": 0.0})
            a["calls"] += 1
            a["ms"] += s.elapsed_time(e)
            if meta:
                a["bytes"] += meta.get("bytes", 0.0)
                a["flops"] += meta.get("flops", 0.0)
        return sorted(agg.values(), key=lambda a: -a["ms"])


_timer = None


def set_timer(t):
    """Install (or remove with None) a KernelTimer; returns the previous one."""
    global _timer
    old, _timer = _timer, t
    return old


def annotate(**meta):
    """Attach algorithmic work (bytes=..., flops=..., tag=...) to the NEXT call() when a timer is active; replay=False marks a
    launch that is not a pure function of its inputs (it adds into its output): timed, never re-launched by KernelTimer.replay."""
    if _timer is not None:
        _timer.meta = meta


def call(name, *args):
    t = _timer
    if t is None:
        rc = getattr(lib(), name)(*args)
        check(rc, name)
        return
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn = getattr(lib(), name)
    s.record()
    rc = fn(*args)
    e.record()
    t.records.append((name, s, e, t.meta))
    if not (t.meta and t.meta.get("replay") is False):      # (replay=False: the launch accumulates into a live buffer)
        t.last[(name, t.meta.get("tag") if t.meta else None)] = (fn, args, torch.cuda.current_stream())
    t.meta = None
    check(rc, name)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise DleError("deeplearningexamples_amd ops run on the MI355X only (got a %s tensor); "
                           "there is no CPU fallback in the product path" % t.device)
