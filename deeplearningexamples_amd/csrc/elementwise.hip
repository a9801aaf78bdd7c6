// Small HBM-bound kernels around the train step for gfx950: dtype casts with row padding,
// BCE-with-logits loss forward+backward, GradScaler bookkeeping.
//
// Replaces (paths relative to /root/reference/PyTorch/):
//   * autocast's per-forward fp32->fp16 casts of activations/weights (torch.cuda.amp.autocast,
//     Recommendation/DLRM/dlrm/scripts/main.py:588; Classification/ConvNets/image_classification/training.py:91)
//   * torch.nn.BCEWithLogitsLoss(reduction="mean") + its backward
//     (Recommendation/DLRM/dlrm/scripts/main.py:556,589-592)
//   * torch.cuda.amp.GradScaler.update() == torch._amp_update_scale_ (main.py:497,608)
// All of them are 16 B/lane streaming kernels; the loss reduction is a wave64 shuffle reduction
// followed by one atomic per workgroup.
#include "common.h"

// ---- cast rows [rows, cols] (ld_in) -> [rows, cols_out >= cols] (ld_out), zero padded columns -------
template <int IDT, int ODT> struct CastIO;
template <int DT> struct LdF {   // load one element as f32
  static __device__ __forceinline__ float ld(const void* p, long long i) {
    if (DT == DLE_F32) return ((const float*)p)[i];
    if (DT == DLE_F16) return Elem<DLE_F16>::to_f32(((const unsigned short*)p)[i]);
    return Elem<DLE_BF16>::to_f32(((const unsigned short*)p)[i]);
  }
};
template <int DT> struct StF {
  static __device__ __forceinline__ void st(void* p, long long i, float v) {
    if (DT == DLE_F32) ((float*)p)[i] = v;
    else if (DT == DLE_F16) ((unsigned short*)p)[i] = Elem<DLE_F16>::from_f32(v);
    else ((unsigned short*)p)[i] = Elem<DLE_BF16>::from_f32(v);
  }
};

template <int IDT, int ODT>
__global__ __launch_bounds__(256) void cast_rows_kernel(const void* __restrict__ in, void* __restrict__ out,
                                                        long long rows, int cols, int cols_out,
                                                        long long ld_in, long long ld_out) {
  const long long total = rows * cols_out;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cols_out;
    const int c = (int)(i - r * cols_out);
    const float v = c < cols ? LdF<IDT>::ld(in, r * ld_in + c) : 0.f;
    StF<ODT>::st(out, r * ld_out + c, v);
  }
}

// contiguous fast path: 4 elements per lane
template <int IDT, int ODT>
__global__ __launch_bounds__(256) void cast_flat_kernel(const void* __restrict__ in, void* __restrict__ out,
                                                        long long n) {
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (long long)gridDim.x * blockDim.x) {
    float4_t v;
    if (IDT == DLE_F32) v = ((const float4_t*)in)[i];
    else {
      const ushort4_t u = ((const ushort4_t*)in)[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = IDT == DLE_F16 ? Elem<DLE_F16>::to_f32(u[k]) : Elem<DLE_BF16>::to_f32(u[k]);
    }
    if (ODT == DLE_F32) ((float4_t*)out)[i] = v;
    else {
      ushort4_t u;
#pragma unroll
      for (int k = 0; k < 4; ++k) u[k] = ODT == DLE_F16 ? Elem<DLE_F16>::from_f32(v[k]) : Elem<DLE_BF16>::from_f32(v[k]);
      ((ushort4_t*)out)[i] = u;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    StF<ODT>::st(out, i, LdF<IDT>::ld(in, i));
  }
}

static int ew_grid(long long items, int per_block) {
  long long g = (items + per_block - 1) / per_block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

#define DISPATCH2(IDT, ODT, CALL)                                                     \
  do {                                                                                \
    bool ok__ = true;                                                                 \
    if (IDT == DLE_F32 && ODT == DLE_F16) { CALL(DLE_F32, DLE_F16); }                 \
    else if (IDT == DLE_F32 && ODT == DLE_BF16) { CALL(DLE_F32, DLE_BF16); }          \
    else if (IDT == DLE_F16 && ODT == DLE_F32) { CALL(DLE_F16, DLE_F32); }            \
    else if (IDT == DLE_BF16 && ODT == DLE_F32) { CALL(DLE_BF16, DLE_F32); }          \
    else if (IDT == DLE_F32 && ODT == DLE_F32) { CALL(DLE_F32, DLE_F32); }            \
    else if (IDT == DLE_F16 && ODT == DLE_F16) { CALL(DLE_F16, DLE_F16); }            \
    else if (IDT == DLE_BF16 && ODT == DLE_BF16) { CALL(DLE_BF16, DLE_BF16); }        \
    else if (IDT == DLE_F16 && ODT == DLE_BF16) { CALL(DLE_F16, DLE_BF16); }          \
    else if (IDT == DLE_BF16 && ODT == DLE_F16) { CALL(DLE_BF16, DLE_F16); }          \
    else ok__ = false;                                                                \
    if (!ok__) { dle_set_error("cast: bad dtype pair %d -> %d", IDT, ODT); return -1; } \
  } while (0)

extern "C" int dle_cast_rows(const void* in, void* out, int64_t rows, int cols, int cols_out, int64_t ld_in,
                             int64_t ld_out, int in_dtype, int out_dtype, hipStream_t stream) {
  DLE_CHECK_ARG(rows >= 0 && cols >= 0 && cols_out >= cols, "cast_rows: bad shape");
  if (rows == 0 || cols_out == 0) return 0;
  DLE_CHECK_ARG(in && out, "cast_rows: null pointer");
  const int ein = in_dtype == DLE_F32 ? 4 : 2, eout = out_dtype == DLE_F32 ? 4 : 2;
  const bool flat = cols == cols_out && ld_in == cols && ld_out == cols_out &&
                    (((uintptr_t)in) % (4 * ein)) == 0 && (((uintptr_t)out) % (4 * eout)) == 0;
  if (flat) {
    const long long n = (long long)rows * cols;
    const int grid = ew_grid(n / 4 + 1, 256);
#define CALL(I, O) hipLaunchKernelGGL((cast_flat_kernel<I, O>), dim3(grid), dim3(256), 0, stream, in, out, n)
    DISPATCH2(in_dtype, out_dtype, CALL);
#undef CALL
  } else {
    const int grid = ew_grid((long long)rows * cols_out, 256);
#define CALL(I, O) hipLaunchKernelGGL((cast_rows_kernel<I, O>), dim3(grid), dim3(256), 0, stream, in, out, (long long)rows, cols, cols_out, (long long)ld_in, (long long)ld_out)
    DISPATCH2(in_dtype, out_dtype, CALL);
#undef CALL
  }
  DLE_LAUNCH_CHECK();
  return 0;
}

// ---- BCE with logits, mean reduction, forward + backward in one pass --------------------------------
// loss = mean( max(x,0) - x*y + log1p(exp(-|x|)) )         (torch.nn.BCEWithLogitsLoss)
// dx   = (sigmoid(x) - y) * (*grad_scale or 1) / n           (d mean / dx, times the AMP loss scale)
// loss_out[0] must be zeroed by the caller side of this entry point (done here with a memset node).
template <int DT>
__global__ __launch_bounds__(256) void bce_logits_kernel(const void* __restrict__ logits,
                                                         const float* __restrict__ target,
                                                         float* __restrict__ loss_out, void* __restrict__ dlogits,
                                                         const float* __restrict__ grad_scale, long long n,
                                                         long long ld_logits) {
  __shared__ float red[16];
  const float gs = (grad_scale ? *grad_scale : 1.0f) / (float)n;
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float x = LdF<DT>::ld(logits, i * ld_logits);
    const float y = target[i];
    const float ax = fabsf(x);
    const float e = __expf(-ax);
    acc += fmaxf(x, 0.f) - x * y + log1pf(e);
    if (dlogits) {
      const float s = x >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
      StF<DT>::st(dlogits, i, (s - y) * gs);
    }
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) unsafeAtomicAdd(loss_out, acc / (float)n);
}

extern "C" int dle_bce_logits(const void* logits, const float* target, float* loss_out, void* dlogits,
                              const float* grad_scale_dev, int64_t n, int64_t ld_logits, int dtype,
                              hipStream_t stream) {
  DLE_CHECK_ARG(loss_out, "bce_logits: null loss pointer");
  hipError_t e = hipMemsetAsync(loss_out, 0, 4, stream);
  if (e != hipSuccess) { dle_set_error("bce memset: %s", hipGetErrorString(e)); return (int)e; }
  if (n == 0) return 0;
  DLE_CHECK_ARG(logits && target, "bce_logits: null pointer");
  const int grid = ew_grid(n, 256 * 4);
  if (dtype == DLE_F32) hipLaunchKernelGGL(bce_logits_kernel<DLE_F32>, dim3(grid), dim3(256), 0, stream, logits, target, loss_out, dlogits, grad_scale_dev, (long long)n, (long long)ld_logits);
  else if (dtype == DLE_F16) hipLaunchKernelGGL(bce_logits_kernel<DLE_F16>, dim3(grid), dim3(256), 0, stream, logits, target, loss_out, dlogits, grad_scale_dev, (long long)n, (long long)ld_logits);
  else if (dtype == DLE_BF16) hipLaunchKernelGGL(bce_logits_kernel<DLE_BF16>, dim3(grid), dim3(256), 0, stream, logits, target, loss_out, dlogits, grad_scale_dev, (long long)n, (long long)ld_logits);
  else { dle_set_error("bce_logits: bad dtype %d", dtype); return -1; }
  DLE_LAUNCH_CHECK();
  return 0;
}

// ---- GradScaler.update(): torch._amp_update_scale_ on device scalars -----------------------------------
// found_inf > 0: scale *= backoff, tracker = 0; else tracker += 1 and, when it reaches the interval,
// scale *= growth (kept if the product overflows), tracker = 0.  inv_scale (optional) = 1/scale for the
// next step's unscale; found_inf is cleared for the next step when clear_found_inf != 0.
__global__ void amp_update_scale_kernel(float* scale, int* growth_tracker, float* found_inf, float* inv_scale,
                                        float growth, float backoff, int interval, int clear_found_inf) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float s = *scale;
  if (*found_inf > 0.f) {
    s *= backoff;
    *growth_tracker = 0;
  } else {
    const int t = *growth_tracker + 1;
    if (t == interval) {
      const float ns = s * growth;
      if (isfinite(ns)) s = ns;
      *growth_tracker = 0;
    } else {
      *growth_tracker = t;
    }
  }
  *scale = s;
  if (inv_scale) *inv_scale = 1.0f / s;
  if (clear_found_inf) *found_inf = 0.f;
}

extern "C" int dle_amp_update_scale(float* scale, int* growth_tracker, float* found_inf, float* inv_scale,
                                    float growth_factor, float backoff_factor, int growth_interval,
                                    int clear_found_inf, hipStream_t stream) {
  DLE_CHECK_ARG(scale && growth_tracker && found_inf, "amp_update_scale: null pointer");
  hipLaunchKernelGGL(amp_update_scale_kernel, dim3(1), dim3(64), 0, stream, scale, growth_tracker, found_inf,
                     inv_scale, growth_factor, backoff_factor, growth_interval, clear_found_inf);
  DLE_LAUNCH_CHECK();
  return 0;
}

// found_inf |= any non-finite in x (16-bit or fp32), 16 B per lane.  Used on the embedding gradient
// (GradScaler.unscale_ on the sparse grad values, main.py:605) without a separate multiply pass.
template <int DT>
__global__ __launch_bounds__(256) void nonfinite_kernel(const void* __restrict__ x, float* __restrict__ found_inf,
                                                        long long n) {
  bool bad = false;
  if (DT == DLE_F32) {
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
      const uint4_t u = ((const uint4_t*)x)[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) bad |= (u[k] & 0x7f800000u) == 0x7f800000u;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3))
      bad |= (((const unsigned int*)x)[(n4 << 2) + threadIdx.x] & 0x7f800000u) == 0x7f800000u;
  } else {
    const unsigned int em = DT == DLE_F16 ? 0x7c00u : 0x7f80u;
    const long long n8 = n >> 3;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
      const uint4_t u = ((const uint4_t*)x)[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        bad |= ((u[k] & em) == em);
        bad |= (((u[k] >> 16) & em) == em);
      }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7))
      bad |= (((const unsigned short*)x)[(n8 << 3) + threadIdx.x] & em) == em;
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) *found_inf = 1.0f;
}

extern "C" int dle_check_nonfinite(const void* x, float* found_inf, int64_t n, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(found_inf, "check_nonfinite: null flag");
  if (n == 0) return 0;
  DLE_CHECK_ARG(x && (((uintptr_t)x) & 15) == 0, "check_nonfinite: null or misaligned pointer");
  const int grid = ew_grid(n / 8 + 1, 256);
  if (dtype == DLE_F32) hipLaunchKernelGGL(nonfinite_kernel<DLE_F32>, dim3(grid), dim3(256), 0, stream, x, found_inf, (long long)n);
  else if (dtype == DLE_F16) hipLaunchKernelGGL(nonfinite_kernel<DLE_F16>, dim3(grid), dim3(256), 0, stream, x, found_inf, (long long)n);
  else if (dtype == DLE_BF16) hipLaunchKernelGGL(nonfinite_kernel<DLE_BF16>, dim3(grid), dim3(256), 0, stream, x, found_inf, (long long)n);
  else { dle_set_error("check_nonfinite: bad dtype %d", dtype); return -1; }
  DLE_LAUNCH_CHECK();
  return 0;
}

// ---- ReLU backward on strided 2-D views: out[r,c] = y[r,c] > 0 ? g[r,c] : 0  (16-bit, 8 elems / lane) ----
// (nn.ReLU(inplace=True) backward in TorchMlp, Recommendation/DLRM/dlrm/nn/mlps.py:85-87, where the
//  gradient arrives from a non-GEMM producer; GEMM producers fuse the mask in their epilogue.)
template <int DT>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const unsigned short* __restrict__ g,
                                                       const unsigned short* __restrict__ y,
                                                       unsigned short* __restrict__ out, long long rows, int cols8,
                                                       long long ld_g, long long ld_y, long long ld_o) {
  const long long total = rows * cols8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cols8;
    const int c = (int)(i - r * cols8) * 8;
    const ushort8_t gv = *(const ushort8_t*)(g + r * ld_g + c);
    const ushort8_t yv = *(const ushort8_t*)(y + r * ld_y + c);
    ushort8_t o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = Elem<DT>::to_f32(yv[k]) > 0.f ? gv[k] : (unsigned short)0;
    *(ushort8_t*)(out + r * ld_o + c) = o;
  }
}

extern "C" int dle_relu_bwd(const void* g, const void* y, void* out, int64_t rows, int cols, int64_t ld_g,
                            int64_t ld_y, int64_t ld_out, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "relu_bwd: 16-bit dtypes only (got %d)", dtype);
  DLE_CHECK_ARG(rows >= 0 && cols >= 0 && cols % 8 == 0, "relu_bwd: cols must be a multiple of 8");
  if (rows == 0 || cols == 0) return 0;
  DLE_CHECK_ARG(g && y && out, "relu_bwd: null pointer");
  DLE_CHECK_ARG(((((uintptr_t)g) | ((uintptr_t)y) | ((uintptr_t)out)) & 15) == 0 && ld_g % 8 == 0 && ld_y % 8 == 0 && ld_out % 8 == 0,
                "relu_bwd: rows must be 16-byte aligned");
  const int grid = ew_grid((long long)rows * (cols / 8), 256);
  if (dtype == DLE_F16) hipLaunchKernelGGL(relu_bwd_kernel<DLE_F16>, dim3(grid), dim3(256), 0, stream, (const unsigned short*)g, (const unsigned short*)y, (unsigned short*)out, (long long)rows, cols / 8, (long long)ld_g, (long long)ld_y, (long long)ld_out);
  else hipLaunchKernelGGL(relu_bwd_kernel<DLE_BF16>, dim3(grid), dim3(256), 0, stream, (const unsigned short*)g, (const unsigned short*)y, (unsigned short*)out, (long long)rows, cols / 8, (long long)ld_g, (long long)ld_y, (long long)ld_out);
  DLE_LAUNCH_CHECK();
  return 0;
}

// ---- activation backward on flat 16-bit arrays: out = g * act'(src) ---------------------------------------
// act = 5 (DLE_ACT_GELU_BWD): src = pre-activation of the tanh-GELU (LanguageModeling/BERT/modeling.py:121-122)
// act = 7 (DLE_ACT_TANH_BWD): src = tanh output.  Used where the incoming gradient is not produced by a GEMM
// (after a LayerNorm backward); GEMM producers fuse the same math in their epilogue.
template <int DT>
__global__ __launch_bounds__(256) void act_bwd_kernel(const unsigned short* __restrict__ g,
                                                      const unsigned short* __restrict__ src,
                                                      unsigned short* __restrict__ out, long long n8, int act) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    const ushort8_t gv = ((const ushort8_t*)g)[i], sv = ((const ushort8_t*)src)[i];
    ushort8_t o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float y = Elem<DT>::to_f32(sv[k]);
      float d;
      if (act == 7) d = 1.f - y * y;
      else {
        const float k0 = 0.7978845608028654f, k1 = 0.044715f;
        const float th = fast_tanh(k0 * (y + k1 * y * y * y));
        d = 0.5f * (1.f + th) + 0.5f * y * (1.f - th * th) * k0 * (1.f + 3.f * k1 * y * y);
      }
      o[k] = Elem<DT>::from_f32(Elem<DT>::to_f32(gv[k]) * d);
    }
    ((ushort8_t*)out)[i] = o;
  }
}

extern "C" int dle_act_bwd(const void* g, const void* src, void* out, int64_t n, int act, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "act_bwd: 16-bit dtypes only");
  DLE_CHECK_ARG(act == 5 || act == 7, "act_bwd: act must be DLE_ACT_GELU_BWD or DLE_ACT_TANH_BWD");
  DLE_CHECK_ARG(n >= 0 && n % 8 == 0, "act_bwd: element count must be a multiple of 8");
  if (n == 0) return 0;
  DLE_CHECK_ARG(g && src && out, "act_bwd: null pointer");
  const int grid = ew_grid(n / 8, 256);
  if (dtype == DLE_F16) hipLaunchKernelGGL(act_bwd_kernel<DLE_F16>, dim3(grid), dim3(256), 0, stream, (const unsigned short*)g, (const unsigned short*)src, (unsigned short*)out, (long long)(n / 8), act);
  else hipLaunchKernelGGL(act_bwd_kernel<DLE_BF16>, dim3(grid), dim3(256), 0, stream, (const unsigned short*)g, (const unsigned short*)src, (unsigned short*)out, (long long)(n / 8), act);
  DLE_LAUNCH_CHECK();
  return 0;
}

// out = a * x + b * y on flat fp32 arrays (out may alias x or y).  Gradient accumulation over micro-batches:
// Classification/ConvNets/main.py:405-416 (batch_size_multiplier), image_classification/training.py:86-96,167-186 -- autograd's
// += into .grad with the loss pre-divided by the number of micro-batches; apex's amp_C.multi_tensor_axpby is the same pass.
__global__ __launch_bounds__(256) void axpby_f32_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                        float* __restrict__ out, float a, float b, long long n) {
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4_t xv = ((const float4_t*)x)[i];
    float4_t o = xv * a;
    if (b != 0.f) o += ((const float4_t*)y)[i] * b;      // (b == 0: y is not read -- it may be uninitialised)
    ((float4_t*)out)[i] = o;
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = a * x[i] + (b != 0.f ? b * y[i] : 0.f);
}

extern "C" int dle_axpby_f32(const float* x, const float* y, float* out, float a, float b, int64_t n, hipStream_t stream) {
  if (n == 0) return 0;
  DLE_CHECK_ARG(x && out && (y || b == 0.f), "axpby_f32: null pointer");
  DLE_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)out) | ((uintptr_t)y)) & 15) == 0, "axpby_f32: arrays must be 16-byte aligned");
  hipLaunchKernelGGL(axpby_f32_kernel, dim3(ew_grid(n / 4 + 1, 256)), dim3(256), 0, stream, x, y ? y : x, out, a, b, (long long)n);
  DLE_LAUNCH_CHECK();
  return 0;
}

// y[c][r] = (16-bit) x[r][c] of a row-major [rows, cols] matrix (fp32, or the output's own 16-bit type); row strides ld_x / ld_y.
// The recurrent steps read their weights as the k-contiguous operand of BOTH the forward and the data-gradient product
// (gemm_smallm.hip streams [N, K] rows): the transposed working copy is made once per iteration next to the plain one, the way
// autocast's per-forward weight casts are (torch.nn.LSTMCell / cuDNN keep transposed packed weights for the same reason).
template <int DT, bool SRC_F32>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const void* __restrict__ x, unsigned short* __restrict__ y, int rows,
                                                             int cols, long long ld_x, long long ld_y) {
  __shared__ unsigned short tile[64][66];
  const int tiles_c = (cols + 63) / 64;
  const int tr = blockIdx.x / tiles_c, tc = blockIdx.x - tr * tiles_c;
  const int r0 = tr * 64, c0 = tc * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    unsigned short v = 0;
    if (r < rows && c < cols)
      v = SRC_F32 ? Elem<DT>::from_f32(((const float*)x)[(long long)r * ld_x + c]) : ((const unsigned short*)x)[(long long)r * ld_x + c];
    tile[i][tx] = v;
  }
  __syncthreads();
#pragma unroll 4
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) y[(long long)c * ld_y + r] = tile[tx][i];
  }
}

extern "C" int dle_transpose_cast(const void* x, void* y, int rows, int cols, int64_t ld_x, int64_t ld_y, int in_dtype, int out_dtype,
                                  hipStream_t stream) {
  DLE_CHECK_ARG(x && y && rows > 0 && cols > 0 && ld_x >= cols && ld_y >= rows, "transpose_cast: bad args");
  DLE_CHECK_ARG((out_dtype == DLE_F16 || out_dtype == DLE_BF16) && (in_dtype == DLE_F32 || in_dtype == out_dtype),
                "transpose_cast: 16-bit output from fp32 or from the same 16-bit type");
  const int grid = ((rows + 63) / 64) * ((cols + 63) / 64);
#define GO(DT, F) hipLaunchKernelGGL((transpose_cast_kernel<DT, F>), dim3(grid), dim3(256), 0, stream, x, (unsigned short*)y, rows, cols, (long long)ld_x, (long long)ld_y)
  if (out_dtype == DLE_F16) { if (in_dtype == DLE_F32) GO(DLE_F16, true); else GO(DLE_F16, false); }
  else { if (in_dtype == DLE_F32) GO(DLE_BF16, true); else GO(DLE_BF16, false); }
#undef GO
  DLE_LAUNCH_CHECK();
  return 0;
}

// Exchange buffer <-> interaction input of the bottom -> top all-to-all (Recommendation/DLRM/dlrm/model/distributed.py:32-98: the
// `torch.cat(dim=1)` of the received blocks in forward, the `split` of the gradient in backward).  `blocks` is the concatenation
// over the ranks s of contiguous [rows, w_s] matrices (what all_to_all_single receives / sends), `x` is [rows, sum_s w_s] with
// block s in columns [base_s, base_s + w_s).  ONE launch per direction (the per-peer row copies were up to 8 + 8 launches per
// step); 16 bytes per lane, a row of x is read / written contiguously.
struct A2ABlocksArgs {
  int world, rows, row_chunks;        // 16-byte chunks per row of x
  int base[9];                        // first chunk of block s inside a row (base[world] = row_chunks)
  long long start[8];                 // first chunk of block s inside `blocks`
};
template <bool PACK>
__global__ __launch_bounds__(256) void a2a_blocks_kernel(uint4_t* __restrict__ blocks, uint4_t* __restrict__ x, A2ABlocksArgs a) {
  const long long total = (long long)a.rows * a.row_chunks;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / a.row_chunks), c = (int)(i - (long long)r * a.row_chunks);
    int b0 = a.base[0], b1 = a.base[1];             // (constant indices only: a run-time index would move the struct to scratch)
    long long st = a.start[0];
#pragma unroll
    for (int q = 1; q < 8; ++q)
      if (q < a.world && c >= a.base[q]) { b0 = a.base[q]; b1 = a.base[q + 1]; st = a.start[q]; }
    const long long j = st + (long long)r * (b1 - b0) + (c - b0);
    if (PACK) blocks[j] = x[i];
    else x[i] = blocks[j];
  }
}

// widths: elements per row of each rank's block; elem_size in bytes; every width * elem_size must be a multiple of 16.
extern "C" int dle_a2a_blocks(void* blocks, void* x, int rows, int world, const int* widths, int elem_size, int pack,
                              hipStream_t stream) {
  DLE_CHECK_ARG(blocks && x && rows > 0 && world >= 1 && world <= 8 && widths && (elem_size == 2 || elem_size == 4),
                "a2a_blocks: bad args (at most 8 ranks)");
  DLE_CHECK_ARG(((((uintptr_t)blocks) | ((uintptr_t)x)) & 15) == 0, "a2a_blocks: 16-byte aligned buffers");
  A2ABlocksArgs a = {};
  a.world = world; a.rows = rows;
  int c = 0;
  long long st = 0;
  for (int s = 0; s < world; ++s) {
    DLE_CHECK_ARG(widths[s] >= 0 && (widths[s] * elem_size) % 16 == 0, "a2a_blocks: block widths must be multiples of 16 bytes");
    a.base[s] = c;
    a.start[s] = st;
    c += widths[s] * elem_size / 16;
    st += (long long)rows * (widths[s] * elem_size / 16);
  }
  for (int s = world; s <= 8; ++s) a.base[s] = c;
  a.row_chunks = c;
  if (c == 0) return 0;
  const int grid = ew_grid((long long)rows * c, 256);
  if (pack) hipLaunchKernelGGL(a2a_blocks_kernel<true>, dim3(grid), dim3(256), 0, stream, (uint4_t*)blocks, (uint4_t*)x, a);
  else hipLaunchKernelGGL(a2a_blocks_kernel<false>, dim3(grid), dim3(256), 0, stream, (uint4_t*)blocks, (uint4_t*)x, a);
  DLE_LAUNCH_CHECK();
  return 0;
}
