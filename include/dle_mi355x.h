/* dle_mi355x.h -- C ABI of libdle_mi355x.so, the MI355X (gfx950 / CDNA4) kernels behind the
 * NVIDIA/DeepLearningExamples AMP+DDP train-step hot path (RN50, BERT-L, DLRM).
 *
 * Conventions:
 *   - every entry point is extern "C", takes plain device pointers + sizes and a hipStream_t
 *     (pass torch.cuda.current_stream().cuda_stream), returns int: 0 = ok, -1 = invalid argument,
 *     >0 = hipError_t.  dle_last_error() returns the message (thread local).  Nothing exits the
 *     process (the reference's CHK_CUDA calls std::exit, gather_gpu_fused.cu:6-14).
 *   - inputs are borrowed, outputs are caller-allocated; no hidden allocation, no host sync,
 *     no global state -> every call is hipGraph-capturable.
 *   - dtype codes: DLE_F32 = 0, DLE_F16 = 1, DLE_BF16 = 2.
 * Paths are relative to /root/reference/PyTorch/.
 */
#ifndef DLE_MI355X_H
#define DLE_MI355X_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hipStream_t;

enum { DLE_F32 = 0, DLE_F16 = 1, DLE_BF16 = 2 };
enum { DLE_ACT_NONE = 0, DLE_ACT_RELU = 1, DLE_ACT_GELU = 2, DLE_ACT_RELU_BWD = 3, DLE_ACT_ADD = 4,
       DLE_ACT_GELU_BWD = 5, DLE_ACT_TANH = 6, DLE_ACT_TANH_BWD = 7,
       DLE_ACT_ADD_MASKED = 8, /* C = acc + (bit ? mask_src : 0): aux is an INPUT here, the bit-packed keep bits of the
                                  addend (bit k of byte i <-> element 8 i + k of the [M, ldc] addend): the residual-branch
                                  gradient dy * (y > 0) of a ResNet block without materialising it */
       DLE_ACT_MUL = 9,        /* C = acc * mask_src */
       DLE_ACT_GELU_DAUX = 10  /* C = gelu(v), aux = gelu'(v): the backward GEMM then multiplies (DLE_ACT_MUL) */ };

/* ---- library plumbing ---------------------------------------------------------------------- */
const char* dle_last_error(void);
int dle_abi_version(void);
int dle_device_check(int device, char* arch_name, int arch_name_len); /* 0 iff gfx950 */

/* ---- DLRM dot interaction -------------------------------------------------------------------
 * replaces dlrm.cuda_ext.interaction_{ampere,volta}.dotBasedInteractFwd / dotBasedInteractBwd
 *   Recommendation/DLRM/dlrm/cuda_src/dot_based_interact_ampere/pytorch_ops.cpp:3-12
 *   .../dot_based_interact_ampere/dot_based_interact_pytorch_types.cu:10-75
 * x[B,R,C] -> out[B, OW], OW = ceil8(R(R-1)/2 + C): [x[:,0,:] | strict lower tri of X X^T | 0 pad]. */
int dle_dot_interact_out_width(int rows, int cols);
int dle_dot_interact_fwd(const void* x, void* out, int batch, int rows, int cols, int dtype,
                         int force_generic, hipStream_t stream);
/* grad[B,R,C] = U_sym X, mlp_grad[B,C] = upstream[:, :C]  (dotBasedInteractBwd returns both).
 * mlp_grad == NULL: fused form, upstream[:, :C] is added onto grad[:, 0, :] (the sum autograd forms
 * because bottom_mlp_output and input[:,0] are the same tensor, dot_based_interact_fp16_bwd.cu:329-331). */
int dle_dot_interact_bwd(const void* x, const void* upstream, void* grad, void* mlp_grad,
                         int batch, int rows, int cols, int dtype, int force_generic,
                         hipStream_t stream);
/* The same, and *found_inf = 1 when any element of grad as stored is inf / nan (left untouched otherwise): the check
 * torch's GradScaler.unscale_ makes on this gradient (dlrm/scripts/main.py:585-608 scaler.step), without a pass over it. */
int dle_dot_interact_bwd_checked(const void* x, const void* upstream, void* grad, void* mlp_grad,
                                 int batch, int rows, int cols, int dtype, int force_generic,
                                 float* found_inf, hipStream_t stream);

/* ---- DLRM embeddings --------------------------------------------------------------------------
 * replaces dlrm.cuda_ext.fused_embedding.gather_gpu_fused_fwd / _bwd
 *   Recommendation/DLRM/dlrm/cuda_src/pytorch_embedding_ops.cpp:3-21, gather_gpu_fused.cu:107-220
 * and dlrm.cuda_ext.sparse_gather.gather_gpu_fwd / gather_gpu_bwd / gather_gpu_bwd_fuse_sgd
 *   Recommendation/DLRM/dlrm/cuda_src/sparse_gather/sparse_pytorch_ops.cpp:1-15, gather_gpu.cu:79-171
 * weight fp32 [sum rows, dim]; indices int64 [batch, tables]; offsets int64 [tables(+1)] or NULL when
 * indices already address the joint table; hash_sizes int64 [tables] or NULL (idx %= size).       */
/* out_batch_stride (elements, 0 = tables*dim): row (b,t) lands at out + b*stride + t*dim, so the gather
 * can write straight into rows 1.. of the [B, 1+tables, dim] interaction input (no torch.cat).        */
int dle_emb_gather_fwd(const float* weight, const int64_t* indices, const int64_t* offsets,
                       const int64_t* hash_sizes, void* out, int64_t batch, int tables, int dim,
                       int out_dtype, int64_t out_batch_stride, hipStream_t stream);
int dle_emb_offset_indices(const int64_t* indices, const int64_t* offsets, const int64_t* hash_sizes,
                           int64_t* rows_out, int64_t batch, int tables, hipStream_t stream);
/* fp32 COO values of the sparse weight gradient: values = (float)grad * (*scale_dev or 1) */
int dle_emb_grad_values(const void* grad, float* values, const float* scale_dev, int64_t n_elems,
                        int grad_dtype, hipStream_t stream);
/* W[rows[i],:] -= lr * scale * grad[i,:]; lr from lr_dev if non-NULL else lr_host; whole update is
 * skipped when *skip_flag_dev != 0 (GradScaler found_inf). */
int dle_emb_sparse_sgd(float* weight, const int64_t* rows, const void* grad, const float* lr_dev,
                       float lr_host, const float* scale_dev, const float* skip_flag_dev,
                       int64_t n_rows, int tables, int dim, int64_t grad_batch_stride, int grad_dtype,
                       hipStream_t stream);
/* Same update without float atomics (the train step's path): tiny tables are reduced in LDS, all other
 * rows through per-row lists (one 4-byte atomicExch per lookup) and ONE plain row read-modify-write.
 * head: int32[total rows] device workspace, all -1 on entry and restored on exit; next: int32[batch*tables]
 * scratch; is_small: uint8[tables] device copy of dle_emb_small_table_mask(); table_offsets_host: HOST
 * int64[tables+1].  grad row (b,t) at grad + b*grad_batch_stride + t*dim (stride 0 = tables*dim). */
int dle_emb_small_table_mask(const int64_t* table_offsets_host, int tables, int dim, unsigned char* mask_host);
int dle_emb_sgd_dedup(float* weight, const int64_t* rows, const void* grad, int32_t* head, int32_t* next,
                      const unsigned char* is_small_dev, const int64_t* table_offsets_host,
                      const float* lr_dev, float lr_host, const float* scale_dev,
                      const float* skip_flag_dev, int64_t batch, int tables, int dim,
                      int64_t grad_batch_stride, int grad_dtype, hipStream_t stream);
/* The same with a scratch buffer: tables of <= 128 rows (dim 128, 16-bit gradients) are summed as OneHot(ids)^T G on the matrix
 * pipe (csrc/emb_onehot.hip: the gradient rows stream HBM -> LDS once, the one-hot operand is built in registers; one fp32
 * partial block per (table, batch slice), folded in slice order -- bit-reproducible, no float atomics; replaces the per-lookup
 * atomicAdd of gather_gpu_fused.cu:161-202 on the tables every sample hits); tables of <= 4096 rows thread EIGHT lists per row
 * (one per residue of the sample index: the ~70-deep duplicate chains of 1-2 k-row tables become 8 chains walked side by side),
 * whose partial sums meet in a third small pass.  ws: dle_emb_sgd_workspace_bytes() bytes of scratch, 256-byte aligned, or NULL
 * (= dle_emb_sgd_dedup).  dle_emb_onehot_try is the one-hot kernel's own entry (1 launched, 0 outside its envelope; scratch
 * dle_emb_onehot_workspace_bytes): tab_t / tab_base / tab_rows = column index, first joint row, row count of each table (host).
 * Non-finite gradients: the one-hot product multiplies every gradient row into every row of a tiny table, so an inf / NaN gradient
 * row turns the WHOLE table NaN (row-local in the reference).  skip_flag_dev must therefore be final before the call (the fp16
 * path: dle_check_nonfinite on the gradient first); without a scaler a non-finite gradient loses the table instead of the row. */
int64_t dle_emb_sgd_workspace_bytes(const int64_t* table_offsets_host, int tables, int dim, int64_t batch);
int64_t dle_emb_onehot_workspace_bytes(int n_tables, int64_t batch);
int dle_emb_sgd_dedup_ws(float* weight, const int64_t* rows, const void* grad, int32_t* head, int32_t* next,
                         const unsigned char* is_small_dev, const int64_t* table_offsets_host,
                         const float* lr_dev, float lr_host, const float* scale_dev,
                         const float* skip_flag_dev, int64_t batch, int tables, int dim,
                         int64_t grad_batch_stride, int grad_dtype, void* ws, int64_t ws_bytes, hipStream_t stream);
int dle_emb_onehot_try(float* weight, const int64_t* rows, const void* grad, const float* lr_dev, float lr_host,
                       const float* scale_dev, const float* skip_flag_dev, const int* tab_t, const int64_t* tab_base,
                       const int* tab_rows, int n_tab, int64_t batch, int tables, int dim, int64_t grad_batch_stride,
                       int grad_dtype, void* ws, int64_t ws_bytes, hipStream_t stream);

/* ---- dense contraction with fused epilogue ---------------------------------------------------
 * replaces cuBLAS GEMM + NVFuser/apex epilogues: apex.mlp (Recommendation/DLRM/dlrm/nn/mlps.py:18-43),
 * F.linear + bias + gelu (LanguageModeling/BERT/modeling.py:130-165), RN50 fc.
 * C[M,N] = act(alpha * A(m,k) B(n,k) + bias[n]); a_kc/b_kc = operand stored contraction-contiguous
 * ([M][lda] / [N][ldb]) or not ([K][lda] / [K][ldb]).  aux = optional pre-activation output.
 * DLE_ACT_RELU_BWD: C = (mask_src > 0) ? acc : 0.  splitk > 1 requires a plain fp32 output; with a
 * caller-provided workspace of >= splitk*M*N*4 bytes the K slices are summed from fp32 slabs, without it by
 * fp32 atomics (10x slower on this part).                                                             */
int dle_gemm(const void* A, const void* B, void* C, void* aux, const float* bias,
             const void* mask_src, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
             int a_kc, int b_kc, int in_dtype, int out_dtype, int act, int splitk, int accumulate,
             float alpha, void* workspace, int64_t workspace_bytes, hipStream_t stream);
/* The big linear layers (M, N >= 256, K >= 128 and a multiple of 8, >= 128 (tile, K slice) items) run on the persistent
 * ping-pong 256 x 256 kernel of csrc/gemm8.hip; dle_gemm8_mode(0 / 1) switches it off / on for A/B measurements and parity
 * tests (-1: query), returns the previous setting; environment default DLE_GEMM_8PH (1).
 * dle_gemm8_min_items(n): the item count from which dle_gemm routes a shape to that kernel (n < 0: query; environment default
 * DLE_GEMM_8PH_MIN_ITEMS, 128) -- the parity tests set 1 so that small and ragged shapes reach it; returns the previous value.
 * dle_gemm8_launch_count(): launches of that kernel by this process so far (tests assert that a call was NOT declined).  */
int dle_gemm8_mode(int mode);
int dle_gemm8_min_items(int n);
int64_t dle_gemm8_launch_count(void);
/* Epilogues that read mask_src (same shape/ld as C): RELU_BWD (mask by mask_src > 0), ADD (C = acc + mask_src),
 * GELU_BWD (acc * gelu'(mask_src), mask_src = saved pre-activation), TANH_BWD (acc * (1 - mask_src^2)).
 * Batched form (attention contractions, torch.bmm at LanguageModeling/BERT/modeling.py:354,373):
 * slice z = (z / batch_inner, z % batch_inner) of X starts at X + zo*sx_o + zi*sx_i (elements).              */
int dle_gemm_batched(const void* A, const void* B, void* C, int M, int N, int K, int64_t lda, int64_t ldb,
                     int64_t ldc, int a_kc, int b_kc, int in_dtype, int out_dtype, float alpha, int batch,
                     int batch_inner, int64_t sa_o, int64_t sa_i, int64_t sb_o, int64_t sb_i, int64_t sc_o,
                     int64_t sc_i, hipStream_t stream);

/* ---- convolutions as implicit GEMM (csrc/gemm_dma.hip) ------------------------------------------------
 * replace cuDNN conv forward / backward-data / backward-filter behind nn.Conv2d(bias=False)
 *   Classification/ConvNets/image_classification/models/common.py:31-60, models/resnet.py:126-175
 * Layouts: activations NHWC, weights KRSC (= torch channels_last memory of an OIHW tensor), 16-bit in,
 * fp32 accumulate.  C and Ko multiples of 8; each tensor < 4 GiB.  P = (H+2*pad-R)/stride+1.
 *   fwd:   y[N,P,Q,Ko]  = conv(x[N,H,W,C], w[Ko,R,S,C]) (+bias, DLE_ACT_RELU optional)
 *   dgrad: dx[N,H,W,C]  = conv_transpose(dy[N,P,Q,Ko], w) (+ addend[N,H,W,C] if non-NULL: skip-branch sum)
 *   wgrad: dw[Ko,R,S,C] (fp32) (+)= dy^T * im2col(x); split-K slabs through the workspace                */
int dle_conv2d_fwd(const void* x, const void* w, void* y, const float* bias, int N, int H, int W, int C,
                   int Ko, int R, int S, int stride, int pad, int dtype, int out_dtype, int act,
                   hipStream_t stream);
int dle_conv2d_dgrad(const void* dy, const void* w, void* dx, const void* addend, int N, int H, int W, int C,
                     int Ko, int R, int S, int stride, int pad, int dtype, hipStream_t stream);
int dle_conv2d_wgrad(const void* dy, const void* x, float* dw, int N, int H, int W, int C, int Ko, int R, int S,
                     int stride, int pad, int dtype, int splitk, int accumulate, void* workspace,
                     int64_t workspace_bytes, hipStream_t stream);
/* dw [Ko, C] fp32 (+)= dy [M, Ko]^T x [M, C] for the 1x1 convolutions whose whole output fits one workgroup's accumulators
 * ((Ko, C) in {(256,64), (64,256), (64,64), (128,256), (256,128), (512,128), (128,512)}, M >= 8192): a streaming kernel that reads
 * every operand byte once (csrc/wgrad1x1.hip; replaces cuDNN's bwd-filter behind the 1x1 nn.Conv2d of models/resnet.py:148-175).
 * Returns 1 when launched, 0 outside the envelope (use dle_gemm with split-K); workspace >= dle_wgrad1x1_workspace() bytes.
 * dle_wgrad1x1_mode(0 / 1): off / on for A/B measurements. */
int64_t dle_wgrad1x1_workspace(void);
int dle_wgrad1x1_mode(int mode);
/* workspace of one shape in bytes; 0: (M, Ko, C) is outside the envelope of dle_wgrad1x1_try (do not allocate, do not call) */
int64_t dle_wgrad1x1_workspace_for(int M, int Ko, int C);
int dle_wgrad1x1_try(const void* dy, const void* x, float* dw, int M, int Ko, int C, int dtype, int accumulate, void* workspace,
                     int64_t workspace_bytes, hipStream_t stream);
/* 3x3 / stride 1 / pad 1 weight gradients with C, Ko multiples of 64 run on the halo-tile kernel of csrc/conv3x3_wgrad.hip when the
 * workspace holds dle_conv3x3_wgrad_workspace() bytes (256 partial blocks of 64 x 9 x 64 fp32, folded in a fixed order);
 * dle_conv3x3_wgrad_mode(0 / 1) switches it off / on for A/B measurements (returns the previous value). */
int64_t dle_conv3x3_wgrad_workspace(void);
int dle_conv3x3_wgrad_mode(int mode);

/* out[n] (+)= sum_m x[m][n]  (bias gradients).  workspace (optional, fp32, >= 2048*N*4 bytes is always
 * enough): row groups are combined through it instead of through same-address atomics. */
int dle_colsum(const void* x, float* out, int64_t M, int N, int64_t ld, int dtype, int accumulate,
               void* workspace, int64_t workspace_bytes, hipStream_t stream);
/* The same sums for n same-shaped 16-bit matrices in one launch pair (the bias gradients of a stack of layers whose output
 * gradients are kept until the end of backward: WaveGlow's 96 res_skip layers, waveglow/model.py:138-157).  table_dev: n x
 * { int64 address of the [M, N] matrix (row pitch ld), int64 address of its fp32 [N] destination } in device memory;
 * workspace >= dle_colsum_batched_workspace_bytes(n, M, N). */
int64_t dle_colsum_batched_workspace_bytes(int n, int64_t M, int N);
int dle_colsum_batched(const int64_t* table_dev, int n, int64_t M, int N, int64_t ld, int dtype, void* workspace,
                       int64_t workspace_bytes, hipStream_t stream);

/* ---- multi-tensor optimizer kernels -----------------------------------------------------------
 * replaces fused_lamb_CUDA.multi_tensor_l2norm / multi_tensor_lamb
 *   LanguageModeling/BERT/lamb_amp_opt/csrc/frontend.cpp:3-32
 *   .../csrc/multi_tensor_l2norm_kernel.cu:153-216, .../csrc/multi_tensor_lamb.cu:371-500
 * and the dense SGD steps (apex FusedSGD, Recommendation/DLRM/dlrm/scripts/main.py:468-471;
 * torch.optim.SGD, Classification/ConvNets/image_classification/optimizers.py:34-56).
 * Tensor lists are described by a device-resident int64 table:
 *   { size[n] | chunk_start[n+1] | ptr[list 0][n] | ptr[list 1][n] | ... }
 * built on the host with dle_mt_table_fill and copied to the device once per address set.      */
int64_t dle_mt_table_len(int n_tensors, int n_lists);
int64_t dle_mt_table_fill(int64_t* table_host, int n_tensors, int n_lists, const int64_t* sizes,
                          const void* const* ptrs, int chunk);
int dle_mt_l2norm(const int64_t* table_dev, int n_tensors, int64_t total_chunks, int chunk, int dtype,
                  float* partial_scratch, float* ret, float* ret_per_tensor, int per_tensor,
                  int* noop_flag, hipStream_t stream);
/* lists: g (grad_dtype; overwritten with the update), p, m, v (fp32) */
int dle_mt_lamb_stage1(const int64_t* table_dev, int n_tensors, int64_t total_chunks, int chunk,
                       int grad_dtype, const int* noop_flag, float beta1, float beta2, float beta3,
                       const int* step_dev, int bias_correction, float eps, int mode,
                       float weight_decay, const float* global_grad_norm, const float* max_grad_norm,
                       const float* inv_scale, hipStream_t stream);
/* lists: update (grad_dtype), p (fp32) [, model copy (copy_dtype; -1 = no copy list)] */
/* Stage 1 that also leaves the two per-tensor norms multi_tensor_lamb_cuda takes in l2norm sweeps around it
 * (lamb_amp_opt/csrc/multi_tensor_lamb.cu:380-420: ||p_t|| before the step, ||update_t|| after stage 1): per-chunk partial sums
 * leave with the pass (partial: fp32 scratch, >= 2 * total_chunks), a fold writes param_norm / update_norm (fp32 [n_tensors]) and
 * raises noop_flag on a non-finite sum -- two sweeps over 1.34 GB each less per BERT-Large step.  weight_decay != 0 (without decay
 * and NVLAMB stage 2 ignores the norms).  The update, m and v are bit-identical to dle_mt_lamb_stage1. */
int dle_mt_lamb_stage1_norms(const int64_t* table_dev, int n_tensors, int64_t total_chunks, int chunk, int grad_dtype,
                             int* noop_flag, float beta1, float beta2, float beta3, const int* step_dev, int bias_correction,
                             float eps, int mode, float weight_decay, const float* global_grad_norm, const float* max_grad_norm,
                             const float* inv_scale, float* partial, float* param_norm, float* update_norm, hipStream_t stream);
int dle_mt_lamb_stage2(const int64_t* table_dev, int n_tensors, int64_t total_chunks, int chunk,
                       int grad_dtype, int copy_dtype, const int* noop_flag,
                       const float* param_norm, const float* update_norm, const float* lr_dev,
                       float weight_decay, int use_nvlamb, hipStream_t stream);
/* lists: g (grad_dtype), p (fp32) [, momentum buffer (fp32)] [, low-precision model copy (copy_dtype)];
 * copy_dtype = -1: no copy list.  The copy is the 16-bit working weight the next forward reads, written
 * in the same pass (replaces autocast's per-forward weight cast). */
int dle_mt_sgd(const int64_t* table_dev, int n_tensors, int64_t total_chunks, int chunk, int grad_dtype,
               int has_momentum, const float* skip_flag_dev, const float* lr_dev, float lr_host,
               float momentum, float dampening, float weight_decay, int nesterov, int first_step,
               const float* inv_scale_dev, int copy_dtype, hipStream_t stream);

/* ---- small train-step kernels (csrc/elementwise.hip) -------------------------------------------
 * dle_cast_rows: out[r, c] = (out_dtype) in[r, c] for c < cols, 0 for cols <= c < cols_out; replaces
 *   autocast's activation/weight casts (Recommendation/DLRM/dlrm/scripts/main.py:588).
 * dle_bce_logits: torch.nn.BCEWithLogitsLoss(reduction="mean") forward + backward in one pass
 *   (main.py:556,589-592): loss_out[0] = mean loss (fp32), dlogits[i] = (sigmoid(x_i) - y_i) * (*grad_scale)/n
 *   in the logits' dtype (dlogits may be NULL); logits element i at logits[i*ld_logits].
 * dle_amp_update_scale: GradScaler.update() = torch._amp_update_scale_ on device scalars (main.py:497,608).
 * dle_check_nonfinite: found_inf = 1 if any element is inf/nan (GradScaler.unscale_'s check).        */
int dle_cast_rows(const void* in, void* out, int64_t rows, int cols, int cols_out, int64_t ld_in,
                  int64_t ld_out, int in_dtype, int out_dtype, hipStream_t stream);
/* y[c][r] = (16-bit) x[r][c]: transposed 16-bit working copy of a weight matrix (fp32 master or 16-bit copy), made once per
 * iteration for the products that contract over the OUTPUT dimension with k-contiguous operands (data gradients of the
 * recurrent cells, model.py:405-455 backward); torch.nn.LSTMCell's backward reads weight.t() the same way. */
int dle_transpose_cast(const void* x, void* y, int rows, int cols, int64_t ld_x, int64_t ld_y, int in_dtype, int out_dtype,
                       hipStream_t stream);
/* Exchange buffer <-> interaction input of the DLRM bottom -> top all-to-all (dlrm/model/distributed.py:32-98: torch.cat(dim=1) of
 * the received blocks forward, the split of the gradient backward).  blocks = concatenation over ranks s of contiguous
 * [rows, widths[s]] matrices; x = [rows, sum widths]; pack = 0: x <- blocks, 1: blocks <- x.  widths in elements (host array of
 * `world` <= 8 ints), widths[s] * elem_size a multiple of 16. */
int dle_a2a_blocks(void* blocks, void* x, int rows, int world, const int* widths, int elem_size, int pack, hipStream_t stream);
int dle_bce_logits(const void* logits, const float* target, float* loss_out, void* dlogits,
                   const float* grad_scale_dev, int64_t n, int64_t ld_logits, int dtype, hipStream_t stream);
/* Data gradient of a linear layer through the activation derivative of the layer below + that layer's bias gradient in one launch
 * (csrc/gemm_dma.hip): C[M, N] = f(A[M, K] B[K, N], src[M, N]) (16-bit, pitch ldc for C and src), colsum_out[n] (+)= sum_m C[m, n]
 * of the rounded output (fp32).  act = DLE_ACT_RELU_BWD: C = product where src > 0 else 0 (dlrm/nn/mlps.py:38-43 backward, apex.mlp);
 * DLE_ACT_MUL: C = product * src, src = the stored GELU derivative (BERT/modeling.py:130-160, bias_gelu backward).  One partial row
 * per tile row in `workspace` (>= ceil(M / 128) * N * 4 bytes), folded in a fixed order.  A k-contiguous, B n-contiguous.
 * 1: launched; 0: outside the envelope (run dle_gemm + dle_colsum). */
int dle_gemm_colsum(const void* A, const void* B, void* C, const void* src, float* colsum_out, int M, int N, int K, int64_t lda,
                    int64_t ldb, int64_t ldc, int dtype, int act, int accumulate_colsum, void* workspace,
                    int64_t workspace_bytes, hipStream_t stream);
/* The head of the DLRM top model in one pass (csrc/dlrm_head.hip): the last linear layer (out_features = 1,
 * dlrm/model/distributed.py top MLP, dlrm/nn/mlps.py:38-43), BCEWithLogitsLoss(mean) (scripts/main.py:556,589-592) and the
 * backward of both -- replaces dle_gemm (batch x 1 x K) + dle_bce_logits + dle_gemm (batch x K x 1, ReLU mask) + dle_gemm
 * (1 x K x batch) + dle_colsum x 2 with the same 16-bit rounding points:
 *   z = (16-bit) (h w + bias); loss_out[0] = mean BCE(z, target); dz = (16-bit) ((sigmoid(z) - target) * (*grad_scale) / M);
 *   dh[m, k] = h[m, k] > 0 ? (16-bit) (dz[m] w[k]) : 0; gw[k] = sum_m dz[m] h[m, k]; gb = sum_m dz[m];
 *   gprev_bias[k] (optional) = sum_m dh[m, k]  (bias gradient of the layer that produced h); logits_out (optional) = z.
 * K % 8 == 0, K <= 512; ws: dle_head_bce_workspace_bytes(M, K) bytes of scratch (one partial row per workgroup, folded in a
 * fixed order by a second small launch: bit-reproducible). */
int64_t dle_head_bce_workspace_bytes(int64_t M, int K);
int dle_head_bce_fwd_bwd(const void* h, const void* w16, const float* bias, const float* target,
                         const float* grad_scale_dev, float* loss_out, void* logits_out, void* dh, float* gw, float* gb,
                         float* gprev_bias, void* ws, int64_t ws_bytes, int64_t M, int K, int64_t ldh, int64_t ldd,
                         int dtype, hipStream_t stream);
int dle_amp_update_scale(float* scale, int* growth_tracker, float* found_inf, float* inv_scale,
                         float growth_factor, float backoff_factor, int growth_interval,
                         int clear_found_inf, hipStream_t stream);
int dle_check_nonfinite(const void* x, float* found_inf, int64_t n, int dtype, hipStream_t stream);
/* out = a * x + b * y on flat fp32 arrays (16-byte aligned; out may alias x or y; y is not read when b == 0): the
 * accumulation of micro-batch gradients behind --optimizer-batch-size (Classification/ConvNets/main.py:405-416,
 * image_classification/training.py:86-96,167-186: loss / divide_loss, autograd's += into .grad; apex amp_C.multi_tensor_axpby). */
int dle_axpby_f32(const float* x, const float* y, float* out, float a, float b, int64_t n, hipStream_t stream);
/* out = g * act'(src) on flat 16-bit arrays; act = DLE_ACT_GELU_BWD (src = pre-activation) or DLE_ACT_TANH_BWD */
int dle_act_bwd(const void* g, const void* src, void* out, int64_t n, int act, int dtype, hipStream_t stream);
/* out[r,c] = y[r,c] > 0 ? g[r,c] : 0 on 16-bit strided views (nn.ReLU backward, dlrm/nn/mlps.py:85-87) */
int dle_relu_bwd(const void* g, const void* y, void* out, int64_t rows, int cols, int64_t ld_g, int64_t ld_y,
                 int64_t ld_out, int dtype, hipStream_t stream);

/* ---- ResNet-50 HBM-bound kernels (csrc/convnet.hip), NHWC 16-bit activations, fp32 statistics ----------------
 * replace nn.BatchNorm2d(train) + nn.ReLU + residual add, nn.MaxPool2d(3,2,1), nn.AdaptiveAvgPool2d(1)
 *   Classification/ConvNets/image_classification/models/resnet.py:148-175,270,299, models/common.py:107-128
 * and LabelSmoothing / CrossEntropyLoss (smoothing.py:18-40, main.py:453-457; ignore_index as used by
 * LanguageModeling/BERT/run_pretraining.py:75-95).  M = N*H*W rows of C channels (C % 8 == 0).
 * BN workspace: fp32 scratch of dle_bn_workspace_bytes(M, C).                                              */
int dle_nchw_to_nhwc(const float* x, void* y, int64_t N, int C, int64_t HW, int C_padded, int out_dtype,
                     hipStream_t stream);
int64_t dle_bn_workspace_bytes(int64_t M, int C);
/* mean/rstd (biased var, eps) of x over M; running stats updated with `momentum` (unbiased var) when non-NULL */
int dle_bn_fwd_stats(const void* x, int64_t M, int C, float eps, float momentum, float* mean, float* rstd,
                     float* running_mean, float* running_var, void* workspace, int64_t workspace_bytes,
                     int dtype, hipStream_t stream);
/* Batch statistics without re-reading the activation: dle_conv2d_fwd_colstats is dle_conv2d_fwd (no bias / act) whose
 * epilogue also leaves column sums of the rounded output in col_partial[groups][2][Ko]: one row per 128-row tile, or -- the
 * channel-widening 1x1 convolutions on the streaming kernel (csrc/gemm_expand.hip), chosen when the buffer also holds
 * min(1032, ceil(M / 64) + 8) rows -- one row per workgroup group; the buffer holds >= ceil(N*P*Q / 128) * 2 * Ko floats and
 * *groups receives the number of rows written; dle_bn_stats_from_partials folds them (fixed order, deterministic) into
 * mean / rstd / running stats exactly like dle_bn_fwd_stats.  workspace: >= 64*C floats. */
int dle_conv2d_fwd_colstats(const void* x, const void* w, void* y, int N, int H, int W, int C, int Ko, int R, int S,
                            int stride, int pad, int dtype, float* col_partial, int64_t col_partial_bytes,
                            int* groups, hipStream_t stream);
int dle_bn_stats_from_partials(const float* partial, int groups, int64_t M, int C, float eps, float momentum,
                               float* mean, float* rstd, float* running_mean, float* running_var, void* workspace,
                               int64_t workspace_bytes, hipStream_t stream);
/* y = act((x - mean) * rstd * gamma + beta (+ residual)), act = ReLU when relu != 0.
 * relu_mask (optional, M*C/8 bytes): bit k of byte i = (y[8 i + k] > 0) -- the backward pass then reads 1 bit per
 * element instead of the 2-byte output to rebuild the ReLU mask.                                                */
int dle_bn_fwd_apply(const void* x, const void* residual, void* y, void* relu_mask, const float* mean,
                     const float* rstd, const float* gamma, const float* beta, int64_t M, int C, int relu,
                     int dtype, hipStream_t stream);
/* y = relu(bn(x) + round16(bn_r(xr))) + keep bits: the bn3 apply of a bottleneck whose residual comes from the downsample branch
 * (conv -> BatchNorm, no ReLU: models/resnet.py:166-173, `residual = self.downsample(x)`) with that branch's BatchNorm taken on
 * the residual's load -- its 16-bit output is never written.  Bit-identical to dle_bn_fwd_apply(xr -> res, relu = 0) followed by
 * dle_bn_fwd_apply(x, res, relu = 1).  mean_r / rstd_r / gamma_r / beta_r: the branch's BatchNorm, fp32 [C].                  */
int dle_bn_fwd_apply2(const void* x, const void* xr, void* y, void* relu_mask, const float* mean, const float* rstd,
                      const float* gamma, const float* beta, const float* mean_r, const float* rstd_r, const float* gamma_r,
                      const float* beta_r, int64_t M, int C, int dtype, hipStream_t stream);
/* g = dy * relu'(y): from relu_mask when given, else from y > 0 (both NULL: g = dy);
 * dgamma = sum g*xhat, dbeta = sum g */
int dle_bn_bwd_reduce(const void* dy, const void* y, const void* relu_mask, const void* x, const float* mean,
                      const float* rstd, float* dgamma, float* dbeta, int64_t M, int C, int accumulate,
                      void* workspace, int64_t workspace_bytes, int dtype, hipStream_t stream);
/* The same reduction for TWO BatchNorms that receive the same gradient (bn3 and the downsample branch's BatchNorm of a
 * bottleneck's first block, models/resnet.py:166-173: both see dy under the block's output keep bits): dy and relu_mask are read
 * ONCE.  Bit-identical to two dle_bn_bwd_reduce calls.  workspace >= 2 * dle_bn_workspace_bytes(M, C). */
int dle_bn_bwd_reduce2(const void* dy, const void* relu_mask, const void* x1, const float* mean1, const float* rstd1, float* dgamma1,
                       float* dbeta1, const void* x2, const float* mean2, const float* rstd2, float* dgamma2, float* dbeta2,
                       int64_t M, int C, void* workspace, int64_t workspace_bytes, int dtype, hipStream_t stream);
/* dx = gamma*rstd*(g - dbeta/M - xhat*dgamma/M); g_out (optional) = g, the skip-branch gradient */
int dle_bn_bwd_apply(const void* dy, const void* y, const void* relu_mask, const void* x, void* dx, void* g_out,
                     const float* mean, const float* rstd, const float* gamma, const float* dgamma,
                     const float* dbeta, int64_t M, int C, int dtype, hipStream_t stream);
/* ---- conv + BatchNorm + ReLU as ONE unit (csrc/conv_bnload.hip): the PRODUCER unit's BatchNorm-apply (+ residual) + ReLU runs on
 * the operand load of the CONSUMER 1x1 convolution (models/resnet.py:148-175: relu(bn2(.)) -> conv3, relu(bn3(.) + residual) ->
 * the next block's conv1; models/common.py:31-128).  out [M, N] = relu(t * sc + sh (+ res)) W^T with sc = rstd * gamma,
 * sh = beta - mean * sc; side outputs y [M, K] (16-bit), bits [M*K/8] (bit k of byte i = y[8 i + k] > 0) and, when stats != NULL,
 * the column sums / sums of squares of the rounded out ([dle_conv1x1_bnload_groups(M, N, K)][2][N], fold with
 * dle_bn_stats_from_partials).  Bit-identical to dle_bn_fwd_apply followed by dle_conv2d_fwd_colstats.  Returns 1 when launched,
 * 0 when the shape is outside the envelope (K in {64, 128, 256}, N % 64 == 0, M >= 4096), > 1 on error.                      */
int dle_conv1x1_bnload_groups(int M, int N, int K);
int dle_conv1x1_bnload_fwd(const void* t, const void* res, const void* w, void* out, void* y, void* bits, const float* mean,
                           const float* rstd, const float* gamma, const float* beta, float* stats, int64_t stats_bytes, int M,
                           int N, int K, int dtype, hipStream_t stream);
/* The same with the residual taken through its own BatchNorm on load (the downsample branch, as dle_bn_fwd_apply2):
 * y = relu(bn(t) + round16(bn_r(res))); bit-identical to dle_bn_fwd_apply(res) followed by dle_conv1x1_bnload_fwd.            */
int dle_conv1x1_bnload_fwd2(const void* t, const void* res, const void* w, void* out, void* y, void* bits, const float* mean,
                            const float* rstd, const float* gamma, const float* beta, const float* mean_r, const float* rstd_r,
                            const float* gamma_r, const float* beta_r, float* stats, int64_t stats_bytes, int M, int N, int K,
                            int dtype, hipStream_t stream);
/* The backward counterpart (csrc/conv_bnbwd.hip): BatchNorm backward (second pass) on the operand load of the 1x1 data gradient
 * that consumes it -- the conv3 / bn3 unit of a bottleneck (models/resnet.py:148-175 backward).  dt [M, K] = ka (g - dbeta / M -
 * xhat dgamma / M) with g = dy under relu_mask (bit-packed, may be NULL), dx [M, N] = dt W, W [K][N] n-contiguous; dgamma / dbeta
 * = the sums dle_bn_bwd_reduce left.  dt and dx are bit-identical to dle_bn_bwd_apply + dle_gemm; dt is never re-read.
 * t2 / bits2 / mean2 / rstd2 / partial (all or none, NULL: off): dx is itself the gradient that enters a second BatchNorm (bn2 of the
 * bottleneck, ReLU keep bits bits2); partial [dle_conv1x1_bnbwd_groups(M)][2][N] receives its backward reduction (sum g, sum g xhat
 * per workgroup; fold with dle_bn_bwd_finish), as dle_gemm_expand_masked_bnred does for bn3.
 * 1: launched; 0: outside the envelope (K = 256, N = 64, M >= 4096, 16-byte aligned dense operands). */
int dle_conv1x1_bnbwd_dgrad(const void* dy, const void* t, const void* relu_mask, const void* w, void* dt, void* dx,
                            const float* mean, const float* rstd, const float* gamma, const float* dgamma, const float* dbeta,
                            const void* t2, const void* bits2, const float* mean2, const float* rstd2, float* partial,
                            int64_t partial_bytes, int M, int N, int K, int dtype, hipStream_t stream);
int dle_conv1x1_bnbwd_groups(int M);
/* The backward reduction of a BatchNorm taken where its input gradient is PRODUCED (csrc/gemm_expand.hip, BRED): conv1's data
 * gradient of the next bottleneck, C = A B^T + src under `bits` (the DLE_ACT_ADD_MASKED form of dle_gemm), is the gradient of the
 * previous block's output; with g = C under bits2 and xhat = (t2 - mean2) rstd2 the kernel also leaves partial [groups][2][N] rows
 * of (sum g, sum g xhat) -- groups = dle_gemm_expand_groups(M, N, K) -- which dle_bn_bwd_finish folds into dgamma / dbeta (what
 * dle_bn_bwd_reduce computes from dy + t + mask in a pass of its own).  t2 [M, N] with C's pitch; bits2 indexed like bits.
 * 1: launched; 0: outside the envelope (K in {64, 128}, N >= 2 K, N % 128 == 0, M >= 4096, 16-byte aligned operands). */
int dle_gemm_expand_masked_bnred(const void* A, const void* B, void* C, const void* src, const void* bits, const void* t2,
                                 const void* bits2, const float* mean2, const float* rstd2, float* partial, int64_t partial_bytes,
                                 int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int b_kc, int dtype, hipStream_t stream);
int dle_bn_bwd_finish(const float* partial, int groups, int C, float* dgamma, float* dbeta, int accumulate, hipStream_t stream);
/* ---- the ResNet stem (csrc/stem.hip): conv7x7 / stride 2 / pad 3 of a 3-channel image, forward (+ BatchNorm partial sums) and
 * weight gradient, on a 4-channel NHWC image (dle_nchw_to_nhwc with C_padded = 4: 8 bytes per pixel, channel 3 zero).
 *   replaces cuDNN behind builder.conv7x7(3, 64, stride=2) + bn1's statistics: Classification/ConvNets/image_classification/
 *   models/resnet.py:262-268,318-322, models/common.py:31-60.
 * w2: packed 16-bit weights [64][7][8][4] (k = r*32 + s*4 + c, zero for s = 7 / c = 3) written by dle_stem_pack_weight from the
 * fp32 master in KRSC memory order [64][7][7][3] (a channels_last nn.Conv2d.weight).  Images up to 224 pixels wide.
 * fwd: y [N, P, Q, 64]; stats (optional): [dle_stem_conv7_groups(N, H)][2][64] column sums / sums of squares of the ROUNDED output
 *      (contract of dle_conv2d_fwd_colstats; fold with dle_bn_stats_from_partials).
 * wgrad: dw [64][7][7][3] fp32 in the master's memory order (+)= sum over pixels dy x im2col(x); workspace >=
 *      dle_stem_conv7_wgrad_workspace(N, H) bytes (one partial per persistent workgroup, folded in a fixed order).      */
int dle_stem_conv7_groups(int N, int H);
int dle_stem_conv7_fwd(const void* x4, const void* w2, void* y, float* stats, int64_t stats_bytes, int N, int H, int W,
                       int dtype, hipStream_t stream);
int64_t dle_stem_conv7_wgrad_workspace(int N, int H);
int dle_stem_conv7_wgrad(const void* dy, const void* x4, float* dw, void* workspace, int64_t workspace_bytes, int N, int H,
                         int W, int dtype, int accumulate, hipStream_t stream);
int dle_stem_pack_weight(const float* w_krsc, void* out, int dtype, hipStream_t stream);
/* argmax: uint8 [N,P,Q,C] window-scan index of the first maximum (ATen tie rule) */
int dle_maxpool_fwd(const void* x, void* y, void* argmax, int N, int H, int W, int C, int ksize, int stride,
                    int pad, int dtype, hipStream_t stream);
/* BatchNorm-apply + ReLU + MaxPool2d(3, 2, 1) in one pass (the stem: models/resnet.py:318-322 bn1 -> relu -> maxpool): y
 * [N, H/2, W/2, C], argmax and relu_mask exactly as dle_bn_fwd_apply + dle_maxpool_fwd would leave them (bit-identical); the
 * 16-bit activation between the two never exists.  H, W even. */
int dle_bn_relu_maxpool_fwd(const void* x, void* y, void* argmax, void* relu_mask, const float* mean, const float* rstd,
                            const float* gamma, const float* beta, int N, int H, int W, int C, int dtype, hipStream_t stream);
int dle_maxpool_bwd(const void* dy, const void* argmax, void* dx, int N, int H, int W, int C, int ksize,
                    int stride, int pad, int dtype, hipStream_t stream);
/* The stem's backward chain MaxPool2d(3, 2, 1) -> ReLU -> BatchNorm (models/resnet.py:318-322 backward; cuDNN pooling backward +
 * batch_norm_backward in the reference) without the pooling gradient's full-resolution tensor: dz [N, H, W, C] = BatchNorm backward
 * (keep bits relu_mask [N H W C / 8], x = the BatchNorm's input) of the gradient dle_maxpool_bwd would scatter from dy
 * [N, H/2, W/2, C] + argmax; dgamma / dbeta (fp32 [C]) are written.  dz equals dle_maxpool_bwd + dle_bn_bwd_apply bit for bit
 * given the same dgamma / dbeta; the reduction groups its fp32 partial sums differently.  workspace: fp32,
 * >= dle_pool_bn_bwd_workspace_bytes (0 = shape outside the envelope: H, W even, C / 8 a power of two <= 256). */
int64_t dle_pool_bn_bwd_workspace_bytes(int N, int H, int W, int C);
int dle_pool_bn_bwd(const void* dy, const void* argmax, const void* relu_mask, const void* x, void* dz, const float* mean,
                    const float* rstd, const float* gamma, float* dgamma, float* dbeta, int N, int H, int W, int C, void* workspace,
                    int64_t workspace_bytes, int dtype, hipStream_t stream);
int dle_avgpool_fwd(const void* x, void* y, int64_t N, int HW, int C, int dtype, hipStream_t stream);
int dle_avgpool_bwd(const void* dy, void* dx, int64_t N, int HW, int C, int dtype, hipStream_t stream);
/* C [M = n_img*H*W, N] = A [M, K] B^T + zero_stuffed(compact [n_img*(H/2)*(W/2), N]): the data gradient of a bottleneck's first 1x1
 * convolution plus the gradient of the stride-2 1x1 downsample branch (models/resnet.py:148-175,150-158), whose full-resolution
 * zero-stuffed form is never written (csrc/gemm_expand.hip).  Returns 1 when launched, 0 when the shape is outside the streaming
 * kernel's envelope (K in {64, 128, 256}, N % 128 == 0, N >= 2 K, H and W even): the caller then materialises dle_upsample_zero
 * and uses dle_gemm with DLE_ACT_ADD. */
int dle_gemm_expand_add_up2(const void* A, const void* B, void* C, const void* compact, int M, int N, int K, int64_t lda,
                            int64_t ldb, int64_t ldc, int64_t ld_compact, int b_kc, int H, int W, int dtype, hipStream_t stream);
/* y[n,h,w,:] = x[n,h/s,w/s,:] where h, w are multiples of s, else 0: with a plain GEMM on the P x Q grid this is the
 * data gradient of a 1x1 stride-s convolution (ResNet downsample branches, models/resnet.py:150-158)           */
int dle_upsample_zero(const void* x, void* y, int64_t N, int P, int Q, int H, int W, int C, int stride, int dtype,
                      hipStream_t stream);
/* loss_out[0] = mean over rows with target != ignore_index of (1-s)*nll + s*(lse - mean logits);
 * dlogits (optional, dlogits_dtype, row stride ld_out) = d loss / d logits * (*grad_scale_dev).
 * scratch: one int32 device word.                                                                           */
int dle_softmax_xent(const float* logits, const int64_t* target, float* loss_out, void* dlogits,
                     const float* grad_scale_dev, int* scratch, int64_t rows, int classes, int64_t ld,
                     int64_t ld_out, float smoothing, int64_t ignore_index, int dlogits_dtype,
                     hipStream_t stream);

/* ---- BERT HBM-bound kernels (csrc/transformer.hip): hidden states [tokens, H] 16-bit, statistics fp32 --------
 * replace LayerNorm(dense(x) + residual) (LanguageModeling/BERT/modeling.py:394-398,430-434), BertEmbeddings
 * (modeling.py:285-301), scores/sqrt(d) + mask -> softmax (modeling.py:354-366), the masked-row index_select of
 * the dense MLM head (modeling.py:587-595) and the pooler's token-0 slice (modeling.py:518-524).               */
int dle_layernorm_fwd(const void* x, const void* residual, void* z_out, void* y, const float* gamma,
                      const float* beta, float* mean, float* rstd, int64_t rows, int H, float eps, int dtype,
                      hipStream_t stream);
int64_t dle_layernorm_workspace_bytes(int H);
/* dz = d/dz of LN; dgamma/dbeta (fp32, (+)= when accumulate) */
int dle_layernorm_bwd(const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma,
                      void* dz, float* dgamma, float* dbeta, int64_t rows, int H, int accumulate, void* workspace,
                      int64_t workspace_bytes, int dtype, hipStream_t stream);
/* z[t] = word[ids[t]] + pos[t mod S] + type[token_type[t]]  (fp32 tables -> 16-bit) */
int dle_embed_sum(const float* word, const float* pos, const float* type, const int64_t* ids,
                  const int64_t* token_type, void* z, int64_t tokens, int S, int H, int dtype, hipStream_t stream);
/* grad_word[ids[t], :] += dz[t, :] (fp32) */
int dle_embed_scatter_add(const void* dz, const int64_t* ids, float* grad_word, int64_t tokens, int H, int dtype,
                          hipStream_t stream);
/* out[k, :] (+)= sum of rows t with sel[t] == k, k < K <= 4; workspace >= 128*K*H*4 bytes */
int dle_rows_select_sum(const void* x, const int64_t* sel, float* out, int64_t rows, int H, int K, int accumulate,
                        void* workspace, int64_t workspace_bytes, int dtype, hipStream_t stream);
int dle_rows_gather(const void* src, const int64_t* idx, void* dst, int64_t n, int H, int dtype, hipStream_t stream);
int dle_rows_scatter(const void* src, const int64_t* idx, void* dst, int64_t n, int H, int accumulate, int dtype,
                     hipStream_t stream);
/* in place: p = softmax(s * scale + mask_add[row / rows_per_batch][col]); rows of length L (power of two <= 512) */
int dle_softmax_fwd(void* scores, const float* mask_add, int64_t rows, int L, int rows_per_batch, float scale,
                    int dtype, hipStream_t stream);
/* in place over dprobs: dS = P * (dP - sum(dP * P)) * scale */
int dle_softmax_bwd(const void* probs, void* dprobs, int64_t rows, int L, float scale, int dtype,
                    hipStream_t stream);

/* ---- dropout (training-mode nn.Dropout of the BERT path: modeling.py:276,296 embeddings; :320,369 attention
 * probabilities; :392-396, :428-432 BertSelfOutput / BertOutput before the residual LayerNorm).
 * Counter-based Philox4x32-10: key = seed, counter = (8-element chunk index, offset); the caller advances `offset`
 * per call site and per step.  Masks are bit-packed (bit k of byte i <-> element 8 i + k, 1 = kept), n / 8 bytes;
 * the drop probability is quantised to round(p * 65536) / 65536 and kept values are scaled by its complement.
 * The reference draws from torch's CUDA Philox stream, so masks are NOT bit-identical to the reference's; parity of
 * the step is checked against the oracle under the masks these entry points produce.                            */
/* backward of dle_dropout_add_layernorm_fwd in ONE pass: dz (residual branch), dx = dz * keep / (1 - p) (dense branch),
 * dgamma / dbeta, and dbias (+)= column sums of dx when non-NULL -- the dropout-backward pass and the dense layer's bias
 * gradient pass (autograd of modeling.py:394-398,430-434) folded into the LayerNorm backward. */
int dle_dropout_add_layernorm_bwd(const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma,
                                  const void* keep_mask, float p, void* dz, void* dx, float* dgamma, float* dbeta,
                                  float* dbias, int64_t rows, int H, int accumulate, void* workspace,
                                  int64_t workspace_bytes, int dtype, hipStream_t stream);
int dle_dropout_fwd(const void* x, void* y, void* mask, int64_t n, float p, uint64_t seed, uint64_t offset,
                    const uint64_t* offset_base,
                    int dtype, hipStream_t stream);
int dle_dropout_bwd(const void* dy, const void* mask, void* dx, int64_t n, float p, int dtype, hipStream_t stream);
/* y = LayerNorm(dropout(x) + residual); z_out = dropout(x) + residual (16-bit, for the backward pass) */
int dle_dropout_add_layernorm_fwd(const void* x, const void* residual, void* z_out, void* y, void* mask,
                                  const float* gamma, const float* beta, float* mean, float* rstd, int64_t rows,
                                  int H, float eps, float p, uint64_t seed, uint64_t offset,
                                  const uint64_t* offset_base, int dtype,
                                  hipStream_t stream);
/* scores -> probs in place; dropped = dropout(probs) is the operand of the P V contraction */
int dle_softmax_dropout_fwd(void* scores, void* dropped, void* mask, const float* mask_add, int64_t rows, int L,
                            int rows_per_batch, float scale, float p, uint64_t seed, uint64_t offset,
                            const uint64_t* offset_base, int dtype,
                            hipStream_t stream);
/* in place over dprobs: g = dP * mask / (1 - p); dS = P * (g - sum(g * P)) * scale */
int dle_softmax_dropout_bwd(const void* probs, void* dprobs, const void* mask, int64_t rows, int L, float scale,
                            float p, int dtype, hipStream_t stream);

/* uint8 NCHW images -> (x - mean[c]) / std[c] as 16-bit NHWC (channels zero-padded to C_padded): the normalisation of
 * PrefetchedWrapper.prefetched_loader (Classification/ConvNets/image_classification/dataloaders.py:354-384) fused with the
 * layout change of the step's first kernel. */
int dle_u8_nchw_normalize_nhwc(const void* x, void* y, const float* mean, const float* std, int64_t N, int C, int64_t HW,
                               int C_padded, int out_dtype, hipStream_t stream);

/* 3x3 / stride 2 / pad 1 data gradient as four parity-class correlations (1 + 2 + 2 + 4 taps: no zero work), each written
 * into its strided sub-grid of dx -- cuDNN's bwd-data behind the strided 3x3 convolutions of the ResNet bottleneck
 * (Classification/ConvNets/image_classification/models/resnet.py:126,148-175).  H, W even; Ko % 64 == 0; workspace:
 * >= 9 * Ko * C * 2 bytes (tap-restricted weight matrices, rebuilt by every call). */
int dle_conv2d_dgrad_s2(const void* dy, const void* w, void* dx, int N, int H, int W, int C, int Ko, void* workspace,
                        int64_t workspace_bytes, int dtype, hipStream_t stream);

/* ---- fused self-attention (64-wide heads; S = 128, or a multiple of 128 up to 1024 -- phase 2 runs S = 512): replaces
 * BertSelfAttention.forward between the QKV and the output projection, LanguageModeling/BERT/modeling.py:340-384 (torch.bmm +
 * softmax + nn.Dropout + torch.bmm) and its autograd backward.  qkv [T = B*S, 3H] (q | k | v, head h at columns h*64 of each
 * third), ctx / dctx [T, H], dqkv [T, 3H]; mask_add fp32 [B, S] (0 / -10000) or NULL; stats fp32 [B*heads, S, W] with
 * W = dle_attention_stats_floats(S): (row max, 1 / row sum) saved for backward (+ for S > 128 a third word the backward pass
 * uses as scratch: the row's delta); keep_mask (optional, B*heads*S*S/8 bytes) = the dropout keep bits in the layout of
 * dle_softmax_dropout_fwd.  The backward regenerates probabilities and mask from (qkv, stats, seed, offset): nothing of shape
 * [B, heads, S, S] is stored -- for S > 128 K / V are streamed through LDS in 128-key blocks (forward: online max / sum, then a
 * second pass with the final statistics; backward: a per-query-block kernel for delta and dQ, a per-key-block kernel for dK / dV).
 * colsum_partial (optional, fp32 [B * S / 128, 3H]): column sums of dqkv per (sequence, 128-row block) -- their sum over the rows
 * is the bias gradient of the QKV projection.  dle_attention_supported: 1 when (S, head_dim) is inside the kernels' envelope. */
int dle_attention_supported(int S, int head_dim);
int dle_attention_stats_floats(int S);
int dle_attention_fwd(const void* qkv, const float* mask_add, void* ctx, float* stats, void* keep_mask, int B, int S,
                      int heads, int head_dim, float scale, float p, uint64_t seed, uint64_t offset,
                      const uint64_t* offset_base, int dtype,
                      hipStream_t stream);
int dle_attention_bwd(const void* qkv, const void* dctx, const float* mask_add, const float* stats, void* dqkv,
                      float* colsum_partial, int B, int S, int heads, int head_dim, float scale, float p, uint64_t seed,
                      uint64_t offset, const uint64_t* offset_base, int dtype, hipStream_t stream);
/* The same with the keep mask the forward pass wrote (keep_mask of dle_attention_fwd) READ instead of re-drawn from the Philox
 * counters (S = 128: 16 bytes per query row replace 8 generator calls per lane); identical results.  keep_mask may be NULL. */
int dle_attention_bwd_keep(const void* qkv, const void* dctx, const float* mask_add, const float* stats, const void* keep_mask,
                           void* dqkv, float* colsum_partial, int B, int S, int heads, int head_dim, float scale, float p,
                           uint64_t seed, uint64_t offset, const uint64_t* offset_base, int dtype, hipStream_t stream);

/* ---- torch.optim.Adam over a tensor table (csrc/multi_tensor.hip): GradScaler.unscale_ + clip_grad_norm_ + Adam.step of
 * SpeechSynthesis/Tacotron2/train.py:400-401,487-497 in one pass.  lists: g, p, exp_avg, exp_avg_sq (all fp32).
 * step_dev: int32 device word = the step number t of THIS update (the caller advances it when the step is not skipped);
 * grad_norm_dev: L2 norm of the (still scaled) gradients from dle_mt_l2norm, max_grad_norm <= 0 disables clipping. */
int dle_mt_adam(const int64_t* table_dev, int n_tensors, int64_t total_chunks, int chunk, const float* skip_flag_dev,
                const float* lr_dev, float lr_host, float beta1, float beta2, float eps, float weight_decay,
                const int* step_dev, const float* inv_scale_dev, const float* grad_norm_dev, float max_grad_norm,
                hipStream_t stream);

/* ---- WaveGlow training step (csrc/waveglow.hip): SpeechSynthesis/Tacotron2/waveglow/model.py + loss_function.py -------
 * Channels-last: a series [B, C, T] of the reference is the matrix [B*T, C]; Conv1d = dle_gemm over rows.  The flow state
 * is fp32 [M, 8] (M = B*T/8 groups of n_group = 8 samples); a flow with c remaining channels works on columns [8-c, 8).
 * dle_wg_taps: row gather of a k-tap dilated Conv1d, col[b,t,k*C+ch] = x[b, t+(k-left)*dilation, ch] (0 outside [0,T); x rows ld_x apart);
 *   WN in_layers (model.py:112-118, left = 1) and ConvTranspose1d(1024, stride 256) (model.py:165-167; dilation -1, left 0).
 * dle_wg_taps_bwd: its transpose, dx[b,t,ch] = sum_k dcol[b, t-(k-left)*dilation, k*C+ch] (+ addend; dx may alias addend).
 * dle_wg_gate_fwd/bwd: fused_add_tanh_sigmoid_multiply (model.py:34-41) on the summed pre-activation s [M, 2nc] (row stride ld).
 * dle_wg_invconv_fwd/bwd: Invertible1x1Conv.forward (model.py:62-85) on the active channels, and its backward incl. the
 *   d/dW of the log-determinant term; dle_wg_logdet_inv: log|det W|, sign and W^-T of the c x c matrix (torch.logdet).
 * dle_wg_coupling_fwd/bwd: audio_1 = exp(log_s) * audio_1 + b (model.py:217-222); o fp32 [M, 8] = (b | log_s | 0).
 * dle_wg_loss / dle_wg_dz_init: WaveGlowLoss.forward (loss_function.py:30-48) and its gradient wrt z.
 * dle_wg_weight_norm_fwd/bwd: torch.nn.utils.weight_norm(dim 0) of Conv1d weights (model.py:95-136) -> 16-bit GEMM operand
 *   w16[co, tap*Cip + ci]; g NULL = plain weight.  dle_wg_upsample_weight(_bwd): ConvTranspose1d weight [Cm, Cm, ksize] <->
 *   GEMM operand b16[(r*Cm+co), (j*Cm+ci)] = w[ci, co, r + stride*j] and the bias repeated per phase r. */
int dle_wg_taps(const void* x, void* col, int B, int T, int C, int ntaps, int dilation, int left, int64_t ld_x, int dtype,
                hipStream_t stream);
int dle_wg_taps_bwd(const void* dcol, const void* addend, void* dx, int B, int T, int C, int ntaps, int dilation, int left,
                    int64_t ld_add, int64_t ld_dx, int dtype, hipStream_t stream);
int dle_wg_gate_fwd(const void* s, void* acts, int64_t M, int nc, int64_t ld_s, int dtype, hipStream_t stream);
int dle_wg_gate_bwd(const void* dacts, const void* s, void* ds, int64_t M, int nc, int64_t ld_s, int64_t ld_ds, int dtype,
                    hipStream_t stream);
int dle_wg_invconv_fwd(const float* x, const float* W, float* y, void* a0_16, int64_t M, int c, int dtype,
                       hipStream_t stream);
int dle_wg_invconv_bwd_partials(int64_t M);
int dle_wg_invconv_bwd(const float* dy, const float* da0, const float* x, const float* W, const float* WinvT, float* dx,
                       float* dW, const float* scale_dev, float logdet_coef, float* workspace, int64_t M, int c,
                       hipStream_t stream);
int dle_wg_logdet_inv(const float* W, float* logdet, float* WinvT, float* sign, int c, hipStream_t stream);
int dle_wg_coupling_partials(int64_t M);
int dle_wg_coupling_fwd(const float* y, const float* o, float* z, float* logs_partial, int64_t M, int c, hipStream_t stream);
int dle_wg_coupling_bwd(const float* dz, const float* y, const float* o, float* dy, void* d_o16, const float* scale_dev,
                        float logs_coef, int64_t M, int c, int dtype, hipStream_t stream);
int dle_wg_loss(const float* z, const float* logs_partial, int n_logs, const float* logdets, int n_flows, float sigma,
                int64_t M, float* loss_out, float* workspace, hipStream_t stream);
int dle_wg_dz_init(const float* z, float* dz, const float* scale_dev, float coef, int64_t M, hipStream_t stream);
int dle_wg_weight_norm_fwd(const float* v, const float* g, void* w16, int Co, int Ci, int Kt, int Cip, int dtype,
                           hipStream_t stream);
int dle_wg_weight_norm_bwd(const float* dw, const float* v, const float* g, float* dv, float* dg, int Co, int Ci, int Kt,
                           int Cip, hipStream_t stream);
int dle_wg_upsample_weight(const float* w, const float* bias, void* b16, float* bias_rep, int Cm, int ksize, int stride,
                           int dtype, hipStream_t stream);
int dle_wg_upsample_weight_bwd(const float* db, float* dw, int Cm, int ksize, int stride, hipStream_t stream);
/* One launch for every tensor of the network.  Weight-norm table: n_entries x 11 int64 = { row_start, Co, Ci, Kt, Cip, v, g
 * (0: plain weight), w16, dw, dv, dg } (device addresses), sorted by row_start, total_rows = sum of Co.  Log-determinant
 * table: n_flows x 2 int64 = { element offset of the c x c matrix from base, c }; WinvT: n_flows x 64 floats. */
int dle_wg_weight_norm_fwd_batched(const int64_t* table_dev, int n_entries, int64_t total_rows, int dtype, hipStream_t stream);
int dle_wg_weight_norm_bwd_batched(const int64_t* table_dev, int n_entries, int64_t total_rows, hipStream_t stream);
int dle_wg_logdet_inv_batched(const float* base, const int64_t* table_dev, float* logdets, float* WinvT, float* signs,
                              int n_flows, hipStream_t stream);

/* ---- Tacotron2 training step (csrc/tacotron2.hip): SpeechSynthesis/Tacotron2/tacotron2/model.py + loss_function.py -----------
 * dle_t2_lstm_fwd/bwd: the pointwise part of nn.LSTM / nn.LSTMCell (model.py:205-214,425-444) on gates [B, 4H] (i, f, g, o, biases
 *   included; replaced in place by the gate activations, which the backward reads) + the F.dropout on the hidden state (bit-packed
 *   keep mask of dle_dropout_fwd, bit <-> element keep_index + b*H + j, scale inv_keep); h goes to up to three row-strided
 *   destinations (the operand buffers of its consumers).  live (fp32 [B], optional): rows with 0 keep (h_prev, c_prev) and write 0
 *   to out_dst -- pack_padded_sequence semantics of the encoder (model.py:205-214).
 * dle_t2_attention_fwd/bwd: Attention.forward of one decoder step (model.py:79-121): energies v . tanh(q + pl) (pl = processed
 *   memory + location term, [B*Ti, A]), softmax over the first lengths[b] text positions, context = weights x memory ([B*Ti, E]);
 *   awc rows = (weights, cumulative weights, 0 x 6) 16-bit = the next step's location-convolution input.  Backward accumulates
 *   d_memory (fp32; NULL: the caller sums weights_t (x) d_ctx_t over the steps itself) / d_pm (fp32; NULL: the caller folds the
 *   kept d_pl with dle_t2_sum_steps) and the per-sample partials
 *   of dv (dv_acc fp32 [B, A]: one owner per row, no atomics; the caller folds the rows) across steps, writes d_pl (16-bit), dq
 *   (fp32 [B, A] and / or 16-bit dq16 = the operand of the query layer's products) and, when dctx16 is given, the summed context
 *   gradient in 16 bits.  The context gradient is the sum of up to three fp32 row-strided pieces d_ctx0..2 [B, E] (NULL = absent),
 *   the gradient reaching the weights the sum of d_aw0 (+ d_aw1) fp32 [B, Ti]: autograd's accumulation of the gradients a tensor
 *   receives from its consumers (model.py:405-455), done on load.  dh1 / dh2 of dle_t2_lstm_bwd likewise.
 *   wloc (forward) / wlocT (backward) non-NULL: the location term of the layer (model.py:40-76: Conv1d(2 -> F, kernel KL, no bias)
 *   on (previous, cumulative) weights followed by Linear(F -> A)) is fused: `pl` is then the processed memory alone, wloc the
 *   pre-multiplied 16-bit [A, KK] matrix (k = tap * 2 + channel, zero padded to KK >= 2 KL, KK % 16 == 0), wlocT its transpose
 *   [KK, A] (KK % 32 == 0); the backward writes the gradient of the previous weights to d_prev (may be d_aw0's buffer) and
 *   accumulates that of the cumulative weights into d_cum (may be d_aw1's), fp32 [B, Ti].
 * dle_t2_location_bwd: transpose of the row gather of the 2-channel location convolution (model.py:40-76): dcol 16-bit
 *   [B*Ti, KL*8] -> d_prev (channel 0, written) and d_cum (channel 1, accumulated), fp32 [B, Ti].
 * dle_t2_tanh_fwd: torch.tanh of the postnet (model.py:170).  dle_t2_mel_loss: MSE(mel_out) + MSE(mel_out + postnet) of
 *   Tacotron2Loss (loss_function.py:42-44) and its gradients (scaled by *scale_dev); workspace >= 1024 floats. */
int dle_t2_tanh_fwd(const void* x, void* y, int64_t n, int dtype, hipStream_t stream);
int dle_t2_lstm_fwd(void* gates, int64_t ld_g, const float* c_prev, float* c_out, void* d0, int64_t ld0, void* d1, int64_t ld1,
                    void* d2, int64_t ld2, const void* keep, int64_t keep_index, float inv_keep, const float* live,
                    const void* h_prev, int64_t ld_hp, void* out_dst, int64_t ld_out, int B, int H, int dtype, hipStream_t stream);
/* One LSTMCell of the decoder in ONE launch (the few-row weight-streaming kernel of csrc/gemm_smallm.hip with the cell as its
 * epilogue): gates [B, 4H] = x [B, K] w [4H, K]^T (+ bias[4H]) (+ addend [B, 4H], 16-bit, row pitch ld_g), rounded to 16 bits as
 * the unfused dle_gemm output would be, then exactly dle_t2_lstm_fwd: activations to `gates`, c_out, dropout(h) to d0..d2.
 * H, K and every pitch multiples of 8, 16-byte aligned bases. */
int dle_t2_lstm_gemm_fwd(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* bias, const void* addend,
                         const float* c_prev, float* c_out, void* gates, int64_t ld_g, void* d0, int64_t ld0, void* d1, int64_t ld1,
                         void* d2, int64_t ld2, const void* keep, int64_t keep_index, float inv_keep, int B, int H, int K, int dtype,
                         hipStream_t stream);
int dle_t2_lstm_bwd(const float* dh, int64_t ld_dh, const float* dh1, int64_t ld_dh1, const float* dh2, int64_t ld_dh2,
                    const float* dc_next, const void* act, int64_t ld_act, const float* c_prev,
                    void* dgates, int64_t ld_dg, float* dc_prev, const void* keep, int64_t keep_index, float inv_keep,
                    const float* live, float* dh_prev, int B, int H, int dtype, hipStream_t stream);
int dle_t2_attention_fwd(const float* q, const void* pl, const float* v, const void* memory, const int64_t* lengths,
                         const void* awc_prev, void* tanh_out, float* aw_out, void* awc_next, void* d0, int64_t ld0, void* d1,
                         int64_t ld1, void* d2, int64_t ld2, const void* wloc, int KL, int KK, int B, int Ti, int A, int E, int dtype,
                         hipStream_t stream);
int dle_t2_attention_bwd(const float* d_ctx0, int64_t ld_c0, const float* d_ctx1, int64_t ld_c1, const float* d_ctx2, int64_t ld_c2,
                         const float* d_aw0, const float* d_aw1, const float* aw, const void* tanh_out, const float* v,
                         const void* memory, float* d_memory, void* d_pl, float* dq, void* dq16, void* dctx16, float* dv_acc,
                         float* d_pm_acc, const void* wlocT, int KL, int KK, float* d_prev, float* d_cum, int B, int Ti, int A, int E,
                         int dtype, hipStream_t stream);
/* out[r] += sum over the n_steps rows of x (16-bit [n_steps, R], R % 8 == 0), fp32 out: folds the kept per-step gradients of the
 * processed memory (d_pm_acc = NULL in dle_t2_attention_bwd). */
int dle_t2_sum_steps(const void* x, float* out, int n_steps, int64_t R, int dtype, hipStream_t stream);
int dle_t2_location_bwd(const void* dcol, float* d_prev, float* d_cum, int B, int Ti, int KL, int dtype, hipStream_t stream);
int dle_t2_mel_loss(const float* out_all, int64_t ld_out, const void* post, const float* target, const float* scale_dev,
                    void* d_out, int64_t ld_dout, void* d_post, float* loss, float* workspace, int64_t R, int n_mel, int dtype,
                    hipStream_t stream);
/* --mask-padding (Tacotron2.parse_output, model.py:648-655): rows (b, t) with t >= lengths[b] of x[B*To, cols] (row pitch ld,
 * dtype DLE_F32 / F16 / BF16) := value.  The trainer applies it to the mel outputs (0), the gate energies (1e3) and -- because
 * masked_fill_ cuts the graph at those positions -- to the gradients that flow back into them (0). */
int dle_t2_mask_rows(void* x, int64_t ld, int cols, const int64_t* lengths, int64_t B, int To, float value, int dtype,
                     hipStream_t stream);

/* ---- the ReLU of an MLP layer as one bit per element (csrc/gemm8_kernel.h, round 6; Recommendation/DLRM/dlrm/nn/mlps.py:38-43,
 * 106-114 = apex mlp_cuda forward / backward).  dle_gemm8_relu_bits_try: Y [M, N] = relu(X [M, K] W [N, K]^T + bias) AND the keep
 * bits of Y (bits [M N / 8]: bit (m N + n) & 7 of byte (m N + n) >> 3 = rounded Y > 0); dense Y, N % 16 == 0.
 * dle_gemm_colsum_bits: the masked data gradient C = (A B) under those bits + the column sums of the rounded C (the bias gradient
 * of the layer below), as dle_gemm_colsum but reading 1 bit per element instead of the 16-bit activation; M % 256 == 0,
 * workspace >= ceil(M / 128) * N floats.  Both: 1 = launched, 0 = outside the ping-pong kernel's envelope (use dle_gemm with
 * DLE_ACT_RELU / dle_gemm_colsum with the activation), > 1 = error. */
int dle_gemm8_relu_bits_try(const void* X, const void* W, void* Y, void* bits, const float* bias, int M, int N, int K, int64_t ldx,
                            int64_t ldw, int dtype, hipStream_t stream);
int dle_gemm_colsum_bits(const void* A, const void* B, void* C, const void* bits, float* colsum_out, int M, int N, int K, int64_t lda,
                         int64_t ldb, int dtype, int accumulate_colsum, void* workspace, int64_t workspace_bytes,
                         hipStream_t stream);

/* ---- collectives over librccl.so (csrc/rccl_comm.hip) -------------------------------------------------
 * What the reference reaches through torch.distributed's ProcessGroupNCCL: the gradient all-reduce of the DDP reducer
 * (Classification/ConvNets/image_classification/training.py:78-84), BERT's comm hook (LanguageModeling/BERT/run_pretraining.py:
 * 461-470: `dist.all_reduce(bucket, async_op=True)` behind a pre-division) and DLRM's `dist.all_to_all` of the bottom -> top
 * exchange and its backward (Recommendation/DLRM/dlrm/model/distributed.py:68,95).  RCCL is bound with dlopen at first use (the
 * copy the process already holds, else /opt/rocm/lib); every call is enqueued on the stream the caller names and returns at
 * once; 0 = ok.  The 128-byte unique id of rank 0 travels over the caller's rendezvous (utils/rccl.py: torch.distributed's
 * MASTER_ADDR / MASTER_PORT store).  dtype: DLE_F32 / DLE_F16 / DLE_BF16, 100 = int32, 101 = int64, 102 = uint8.            */
int dle_rccl_available(void);
int dle_rccl_unique_id(void* id128);
int dle_rccl_init(const void* id128, int rank, int world, void** comm_out);          /* ncclCommInitRank on the current device */
int dle_rccl_count(void* comm);                                                       /* ncclCommCount: ranks that joined      */
int dle_rccl_destroy(void* comm);
int dle_rccl_allreduce(void* comm, void* buf, int64_t count, int dtype, int op, hipStream_t stream);   /* op: 0 sum, 2 max, 4 avg */
int dle_rccl_broadcast(void* comm, void* buf, int64_t bytes, int root, hipStream_t stream);
/* all_to_all_single with split lists: peer p gets send_bytes[p] bytes from send + sum(send_bytes[:p]) and delivers recv_bytes[p]
 * bytes at recv + sum(recv_bytes[:p]); one grouped ncclSend / ncclRecv pair per peer; the size arrays are HOST memory.       */
int dle_rccl_alltoallv(void* comm, const void* send, const int64_t* send_bytes, void* recv, const int64_t* recv_bytes, int world,
                       hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DLE_MI355X_H */
