// HBM-bound kernels of the ResNet-50 train step for gfx950 (NHWC, 16-bit activations, fp32 statistics):
// input layout conversion, BatchNorm (training mode) forward/backward fused with ReLU and the residual add,
// 3x3/2 max pooling, global average pooling, softmax cross entropy with label smoothing.
//
// Replace (paths relative to /root/reference/PyTorch/Classification/ConvNets/image_classification/):
//   nn.BatchNorm2d + nn.ReLU(inplace) + `out += residual`   models/resnet.py:148-175, models/common.py:107-128
//   nn.MaxPool2d(3, 2, 1), nn.AdaptiveAvgPool2d(1)           models/resnet.py:270,299
//   LabelSmoothing / nn.CrossEntropyLoss                      smoothing.py:18-40, main.py:453-457
//   channels_last conversion of the input batch               training.py:60-61 (memory_format)
// (cuDNN / ATen kernels in the reference).  All kernels move 16 B per lane (8 channels), reductions over the
// N*H*W axis keep per-lane fp32 partial sums, meet in LDS, and are combined across workgroups through a
// caller-provided workspace (plain stores, no same-address atomics) and finished in fp64.
#include "common.h"

template <int DT> __device__ __forceinline__ float up16(unsigned short u) { return Elem<DT>::to_f32(u); }
template <int DT> __device__ __forceinline__ unsigned short dn16(float f) { return Elem<DT>::from_f32(f); }

// Streaming reads on this chip peak with FEW loads in flight per lane once ~1000 workgroups are resident (a read-only
// sweep: 6.4 TB/s with 1-2 loads per lane, 4.8-5.3 with 8: tools/probes/read_bw.hip); tools/kbench/bn_bench on the
// batch-256 ResNet-50 shapes: forward apply 2.83 -> 2.58 ms / step, backward apply 3.84 -> 3.62 with one trip, 1024 workgroups.
static int g_bn_apply_trips = 1;       // grid-stride trips in flight of the apply kernels (1 / 2 / 4 forward, 1 / 2 / 3 backward)
static int g_bn_apply_cap = 1024;      // workgroup cap of the apply kernels

static int cn_grid(long long items, int per_block, int cap = 2048) {
  long long g = (items + per_block - 1) / per_block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// ---------------------------------------------------------------- NCHW fp32 -> NHWC 16-bit (channels padded)
template <int DT>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, unsigned short* __restrict__ y,
                                                           long long N, int C, long long HW, int Cp) {
  const long long total = N * HW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / HW, hw = i - n * HW;
    for (int c8 = 0; c8 < Cp; c8 += 8) {
      ushort8_t o;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = c8 + k;
        o[k] = c < C ? dn16<DT>(x[(n * C + c) * HW + hw]) : (unsigned short)0;
      }
      *(ushort8_t*)(y + i * Cp + c8) = o;
    }
  }
}

// C_padded = 4 (the stem kernels' 8-byte pixels): one 8-byte store per pixel; u8 != 0: uint8 source normalised by mean / std
template <int DT, bool U8>
__global__ __launch_bounds__(256) void nchw_to_nhwc4_kernel(const void* __restrict__ xv, unsigned short* __restrict__ y,
                                                            long long N, int C, long long HW, const float* __restrict__ mean,
                                                            const float* __restrict__ std) {
  const long long total = N * HW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / HW, hw = i - n * HW;
    ushort4_t o = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c < C) {
        if (U8) o[c] = dn16<DT>(((float)((const unsigned char*)xv)[(n * C + c) * HW + hw] - mean[c]) / std[c]);
        else o[c] = dn16<DT>(((const float*)xv)[(n * C + c) * HW + hw]);
      }
    }
    *(ushort4_t*)(y + i * 4) = o;
  }
}

extern "C" int dle_nchw_to_nhwc(const float* x, void* y, int64_t N, int C, int64_t HW, int C_padded, int out_dtype,
                                hipStream_t stream) {
  DLE_CHECK_ARG(out_dtype == DLE_F16 || out_dtype == DLE_BF16, "nchw_to_nhwc: 16-bit output only");
  DLE_CHECK_ARG(C > 0 && C_padded >= C && (C_padded % 8 == 0 || C_padded == 4),
                "nchw_to_nhwc: padded channel count must be 4 or a multiple of 8");
  if (N * HW == 0) return 0;
  DLE_CHECK_ARG(x && y, "nchw_to_nhwc: null pointer");
  const int grid = cn_grid(N * HW, 256);
  if (C_padded == 4) {
    if (out_dtype == DLE_F16) hipLaunchKernelGGL((nchw_to_nhwc4_kernel<DLE_F16, false>), dim3(grid), dim3(256), 0, stream, (const void*)x, (unsigned short*)y, (long long)N, C, (long long)HW, nullptr, nullptr);
    else hipLaunchKernelGGL((nchw_to_nhwc4_kernel<DLE_BF16, false>), dim3(grid), dim3(256), 0, stream, (const void*)x, (unsigned short*)y, (long long)N, C, (long long)HW, nullptr, nullptr);
    DLE_LAUNCH_CHECK();
    return 0;
  }
  if (out_dtype == DLE_F16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<DLE_F16>, dim3(grid), dim3(256), 0, stream, x, (unsigned short*)y, (long long)N, C, (long long)HW, C_padded);
  else hipLaunchKernelGGL(nchw_to_nhwc_kernel<DLE_BF16>, dim3(grid), dim3(256), 0, stream, x, (unsigned short*)y, (long long)N, C, (long long)HW, C_padded);
  DLE_LAUNCH_CHECK();
  return 0;
}

// uint8 NCHW (the decoded images a torch DataLoader collates) -> (x - mean[c]) / std[c] as 16-bit NHWC, padding channels
// zero: PrefetchedWrapper's `input.float().sub_(mean).div_(std)` (image_classification/dataloaders.py:354-384) fused with
// the layout change the step needs.
template <int DT>
__global__ __launch_bounds__(256) void u8_nchw_norm_nhwc_kernel(const unsigned char* __restrict__ x, unsigned short* __restrict__ y,
                                                                long long N, int C, long long HW, int Cp,
                                                                const float* __restrict__ mean, const float* __restrict__ std) {
  const long long total = N * HW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / HW, hw = i - n * HW;
    for (int c8 = 0; c8 < Cp; c8 += 8) {
      ushort8_t o;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = c8 + k;
        o[k] = c < C ? dn16<DT>(((float)x[(n * C + c) * HW + hw] - mean[c]) / std[c]) : (unsigned short)0;
      }
      *(ushort8_t*)(y + i * Cp + c8) = o;
    }
  }
}

extern "C" int dle_u8_nchw_normalize_nhwc(const void* x, void* y, const float* mean, const float* std, int64_t N, int C,
                                          int64_t HW, int C_padded, int out_dtype, hipStream_t stream) {
  DLE_CHECK_ARG(out_dtype == DLE_F16 || out_dtype == DLE_BF16, "u8_nchw_normalize_nhwc: 16-bit output only");
  DLE_CHECK_ARG(C > 0 && C_padded >= C && (C_padded % 8 == 0 || C_padded == 4),
                "u8_nchw_normalize_nhwc: padded channel count must be 4 or a multiple of 8");
  if (N * HW == 0) return 0;
  DLE_CHECK_ARG(x && y && mean && std, "u8_nchw_normalize_nhwc: null pointer");
  const int grid = cn_grid(N * HW, 256);
  if (C_padded == 4) {
    if (out_dtype == DLE_F16) hipLaunchKernelGGL((nchw_to_nhwc4_kernel<DLE_F16, true>), dim3(grid), dim3(256), 0, stream, x, (unsigned short*)y, (long long)N, C, (long long)HW, mean, std);
    else hipLaunchKernelGGL((nchw_to_nhwc4_kernel<DLE_BF16, true>), dim3(grid), dim3(256), 0, stream, x, (unsigned short*)y, (long long)N, C, (long long)HW, mean, std);
    DLE_LAUNCH_CHECK();
    return 0;
  }
  if (out_dtype == DLE_F16) hipLaunchKernelGGL(u8_nchw_norm_nhwc_kernel<DLE_F16>, dim3(grid), dim3(256), 0, stream, (const unsigned char*)x, (unsigned short*)y, (long long)N, C, (long long)HW, C_padded, mean, std);
  else hipLaunchKernelGGL(u8_nchw_norm_nhwc_kernel<DLE_BF16>, dim3(grid), dim3(256), 0, stream, (const unsigned char*)x, (unsigned short*)y, (long long)N, C, (long long)HW, C_padded, mean, std);
  DLE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- column reductions over [M][C]
// MODE 0: s0 = sum x, s1 = sum x^2                                  (BN forward statistics)
// MODE 1: g = dy * (y > 0 if y) ; s0 = sum g, s1 = sum g * xhat     (BN backward reductions)
struct BnRedArgs {
  const unsigned short* x;
  const unsigned short* dy;
  const unsigned short* y;      // post-activation output for the ReLU mask (NULL: no ReLU, or mask given)
  const unsigned char* mask;    // bit-packed ReLU mask written by bn_fwd_apply (1 bit / element instead of 2 bytes)
  const float* mean;
  const float* rstd;
  float* partial;               // [groups][2][C]
  long long M;
  int C;
  long long rows_per_block;
  int lpr;                      // lanes per row = pow2 >= C/8 (<= 256)
};

// MM (MODE 1): 0 = no ReLU mask, 1 = bit mask (a.mask), 2 = saved output (a.y), -1 = decided at run time (legacy form).
// With the mask kind a run-time pointer test, the loads of the U rows sat in conditional blocks and hipcc put s_waitcnt
// vmcnt(0) between them (a possibly-pending load into the same registers on another path): ONE row in flight per lane.
// The compile-time forms issue all loads of a batch back to back and request the next batch before reducing the current.
template <int DT, int MODE, int UU = 0, int MM = -1>
__global__ __launch_bounds__(256) void bn_reduce_kernel(BnRedArgs a) {
  __shared__ float red[2][256 * 8];
  const int cl = threadIdx.x % a.lpr, rl = threadIdx.x / a.lpr, rstep = 256 / a.lpr;
  const int c0 = (blockIdx.x * a.lpr + cl) * 8;
  const long long r0 = (long long)blockIdx.y * a.rows_per_block;
  long long r1 = r0 + a.rows_per_block;
  if (r1 > a.M) r1 = a.M;
  float s0[8], s1[8], mu[8], rs[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { s0[k] = 0.f; s1[k] = 0.f; mu[k] = 0.f; rs[k] = 1.f; }
  if (c0 < a.C) {
    if (MODE == 1) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { mu[k] = a.mean[c0 + k]; rs[k] = a.rstd[c0 + k]; }
    }
    auto accum = [&](ushort8_t xv, ushort8_t gv, ushort8_t yv, unsigned bits) {
      float xf[8];
      unpack8<DT>(xv, xf);
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { s0[k] += xf[k]; s1[k] += xf[k] * xf[k]; }
      } else {
        float gf[8], yf[8];
        unpack8<DT>(gv, gf);
        if (a.y) unpack8<DT>(yv, yf);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float g = gf[k];
          if (a.mask ? !((bits >> k) & 1u) : (a.y && !(yf[k] > 0.f))) g = 0.f;
          s0[k] += g;
          s1[k] += g * (xf[k] - mu[k]) * rs[k];
        }
      }
    };
    constexpr int U = UU ? UU : (MODE == 0 ? 8 : 4);   // rows in flight per lane (U x 1..3 independent 16-byte loads)
    long long r = r0 + rl;
    if constexpr (MODE == 1 && MM >= 0) {
      auto load_batch = [&](long long rb, ushort8_t (&xv)[U], ushort8_t (&gv)[U], ushort8_t (&yv)[U], unsigned (&mb)[U])
          __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const long long o = (rb + (long long)u * rstep) * a.C + c0;
          xv[u] = *(const ushort8_t*)(a.x + o);
          gv[u] = *(const ushort8_t*)(a.dy + o);
          mb[u] = 0;
          if constexpr (MM == 1) mb[u] = a.mask[o >> 3];
          if constexpr (MM == 2) yv[u] = *(const ushort8_t*)(a.y + o);
        }
      };
      auto accum_mm = [&](ushort8_t xv, ushort8_t gv, ushort8_t yv, unsigned bits) __attribute__((always_inline)) {
        float xf[8], gf[8], yf[8];
        unpack8<DT>(xv, xf);
        unpack8<DT>(gv, gf);
        if constexpr (MM == 2) unpack8<DT>(yv, yf);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float g = gf[k];
          if constexpr (MM == 1) { if (!((bits >> k) & 1u)) g = 0.f; }
          if constexpr (MM == 2) { if (!(yf[k] > 0.f)) g = 0.f; }
          s0[k] += g;
          s1[k] += g * (xf[k] - mu[k]) * rs[k];
        }
      };
      const long long bstep = (long long)U * rstep;
      if (r + (long long)(U - 1) * rstep < r1) {
        ushort8_t xv[U], gv[U], yv[U], xn[U], gn[U], yn[U];
        unsigned mb[U], mn[U];
        load_batch(r, xv, gv, yv, mb);
        for (; r + bstep + (long long)(U - 1) * rstep < r1; r += bstep) {
          load_batch(r + bstep, xn, gn, yn, mn);
#pragma unroll
          for (int u = 0; u < U; ++u) accum_mm(xv[u], gv[u], yv[u], mb[u]);
#pragma unroll
          for (int u = 0; u < U; ++u) { xv[u] = xn[u]; gv[u] = gn[u]; yv[u] = yn[u]; mb[u] = mn[u]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) accum_mm(xv[u], gv[u], yv[u], mb[u]);
        r += bstep;
      }
    } else {
    for (; r + (long long)(U - 1) * rstep < r1; r += (long long)U * rstep) {
      ushort8_t xv[U], gv[U], yv[U];
      unsigned mb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long o = (r + (long long)u * rstep) * a.C + c0;
        xv[u] = *(const ushort8_t*)(a.x + o);
        mb[u] = 0;
        if (MODE == 1) {
          gv[u] = *(const ushort8_t*)(a.dy + o);
          if (a.mask) mb[u] = a.mask[o >> 3];
          else if (a.y) yv[u] = *(const ushort8_t*)(a.y + o);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) accum(xv[u], gv[u], yv[u], mb[u]);
    }
    }
    for (; r < r1; r += rstep) {
      const long long o = r * a.C + c0;
      ushort8_t gv = {}, yv = {};
      unsigned mb = 0;
      if (MODE == 1) {
        gv = *(const ushort8_t*)(a.dy + o);
        if (a.mask) mb = a.mask[o >> 3];
        else if (a.y) yv = *(const ushort8_t*)(a.y + o);
      }
      accum(*(const ushort8_t*)(a.x + o), gv, yv, mb);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[0][threadIdx.x * 8 + k] = s0[k]; red[1][threadIdx.x * 8 + k] = s1[k]; }
  __syncthreads();
  if (rl == 0 && c0 < a.C) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float t0 = 0.f, t1 = 0.f;
      for (int q = 0; q < rstep; ++q) { t0 += red[0][(q * a.lpr + cl) * 8 + k]; t1 += red[1][(q * a.lpr + cl) * 8 + k]; }
      a.partial[((long long)blockIdx.y * 2 + 0) * a.C + c0 + k] = t0;
      a.partial[((long long)blockIdx.y * 2 + 1) * a.C + c0 + k] = t1;
    }
  }
}

// finish of MODE 0: mean, biased var -> rstd; running stats (momentum, unbiased var) like nn.BatchNorm2d
__global__ __launch_bounds__(256) void bn_stats_finish_kernel(const float* __restrict__ partial, int groups, int C,
                                                              long long M, float eps, float momentum,
                                                              float* __restrict__ mean, float* __restrict__ rstd,
                                                              float* __restrict__ running_mean,
                                                              float* __restrict__ running_var) {
  __shared__ double red[2][256];
  const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double s0 = 0.0, s1 = 0.0;
  if (c < C) {
    int g = sl;
    for (; g + 28 < groups; g += 32) {          // 8 independent loads per stream in flight
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { a[u] = partial[((long long)(g + 4 * u) * 2) * C + c]; b[u] = partial[((long long)(g + 4 * u) * 2 + 1) * C + c]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s0 += a[u]; s1 += b[u]; }
    }
    for (; g < groups; g += 4) { s0 += partial[((long long)g * 2) * C + c]; s1 += partial[((long long)g * 2 + 1) * C + c]; }
  }
  red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1;
  __syncthreads();
  if (sl == 0 && c < C) {
    const double t0 = red[0][cl] + red[0][64 + cl] + red[0][128 + cl] + red[0][192 + cl];
    const double t1 = red[1][cl] + red[1][64 + cl] + red[1][128 + cl] + red[1][192 + cl];
    const double m = t0 / (double)M;
    double var = t1 / (double)M - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
      const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
  }
}

// finish of MODE 1: dgamma = sum g xhat, dbeta = sum g.  A latency chain, not a bandwidth problem (<= 1024 groups x 2 x C
// floats): 8 columns x 32 group slices per workgroup, 8 loads per stream in flight -> <= 4 dependent batches.
__global__ __launch_bounds__(256) void bn_bwd_finish_kernel(const float* __restrict__ partial, int groups, int C,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int accumulate) {
  __shared__ double red[2][256];
  const int cl = threadIdx.x & 7, sl = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  double s0 = 0.0, s1 = 0.0;
  if (c < C) {
    int g = sl;
    for (; g + 7 * 32 < groups; g += 8 * 32) {
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { a[u] = partial[((long long)(g + 32 * u) * 2) * C + c]; b[u] = partial[((long long)(g + 32 * u) * 2 + 1) * C + c]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s0 += a[u]; s1 += b[u]; }
    }
    for (; g < groups; g += 32) { s0 += partial[((long long)g * 2) * C + c]; s1 += partial[((long long)g * 2 + 1) * C + c]; }
  }
  red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1;
  __syncthreads();
  if (sl == 0 && c < C) {
    double t0 = 0.0, t1 = 0.0;
#pragma unroll
    for (int q = 0; q < 32; ++q) { t0 += red[0][q * 8 + cl]; t1 += red[1][q * 8 + cl]; }
    dbeta[c] = accumulate ? dbeta[c] + (float)t0 : (float)t0;
    dgamma[c] = accumulate ? dgamma[c] + (float)t1 : (float)t1;
  }
}

// level 1 of the statistics finish when the partials come from a convolution epilogue (one group per 128-row tile:
// up to 25088 groups): 32 x (C/16) workgroups fold the groups g = y, y + 32, ... into out[y][2][C]; the existing
// finishing kernel then folds those 32.  Fixed summation order: deterministic.
__global__ __launch_bounds__(256) void bn_partial_fold_kernel(const float* __restrict__ partial, int groups, int C,
                                                              float* __restrict__ out) {
  __shared__ float red[2][256];
  const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float s0 = 0.f, s1 = 0.f;
  if (c < C) {
    int g = blockIdx.y + 32 * sl;
    for (; g + 3 * 512 < groups; g += 4 * 512) {
      float a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = partial[((long long)(g + 512 * u) * 2) * C + c];
        b[u] = partial[((long long)(g + 512 * u) * 2 + 1) * C + c];
      }
      s0 += (a[0] + a[1]) + (a[2] + a[3]);
      s1 += (b[0] + b[1]) + (b[2] + b[3]);
    }
    for (; g < groups; g += 512) { s0 += partial[((long long)g * 2) * C + c]; s1 += partial[((long long)g * 2 + 1) * C + c]; }
  }
  red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1;
  __syncthreads();
  if (sl == 0 && c < C) {
    float t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) { t0 += red[0][q * 16 + cl]; t1 += red[1][q * 16 + cl]; }
    out[((long long)blockIdx.y * 2) * C + c] = t0;
    out[((long long)blockIdx.y * 2 + 1) * C + c] = t1;
  }
}

// One-launch finish for up to ~1100 partial rows (the streaming 1x1 kernel's <= 1032 rows, every tile-kernel producer from the
// 28x28 layers down): 8 columns x 32 group slices per workgroup, 8 loads per stream in flight (<= 5 dependent batches), sums in
// fp64, fixed order.  It replaces the fold + finish PAIR (two 5 us launches on the forward chain of each of the 53 BatchNorms)
// wherever the second level of parallelism is not needed.
__global__ __launch_bounds__(256) void bn_stats_finish_wide_kernel(const float* __restrict__ partial, int groups, int C,
                                                                   long long M, float eps, float momentum,
                                                                   float* __restrict__ mean, float* __restrict__ rstd,
                                                                   float* __restrict__ running_mean,
                                                                   float* __restrict__ running_var) {
  __shared__ double red[2][256];
  const int cl = threadIdx.x & 7, sl = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  double s0 = 0.0, s1 = 0.0;
  if (c < C) {
    int g = sl;
    for (; g + 7 * 32 < groups; g += 8 * 32) {
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { a[u] = partial[((long long)(g + 32 * u) * 2) * C + c]; b[u] = partial[((long long)(g + 32 * u) * 2 + 1) * C + c]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s0 += a[u]; s1 += b[u]; }
    }
    for (; g < groups; g += 32) { s0 += partial[((long long)g * 2) * C + c]; s1 += partial[((long long)g * 2 + 1) * C + c]; }
  }
  red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1;
  __syncthreads();
  if (sl == 0 && c < C) {
    double t0 = 0.0, t1 = 0.0;
#pragma unroll
    for (int q = 0; q < 32; ++q) { t0 += red[0][q * 8 + cl]; t1 += red[1][q * 8 + cl]; }
    const double m = t0 / (double)M;
    double var = t1 / (double)M - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
      const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
  }
}

// mean / rstd / running statistics from per-group column sums [groups][2][C] (dle_conv2d_fwd_colstats);
// workspace: >= 32 * 2 * C floats.
extern "C" int dle_bn_stats_from_partials(const float* partial, int groups, int64_t M, int C, float eps, float momentum,
                                          float* mean, float* rstd, float* running_mean, float* running_var,
                                          void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  DLE_CHECK_ARG(partial && mean && rstd && workspace && groups > 0 && M > 0 && C > 0, "bn_stats_from_partials: bad arguments");
  DLE_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "bn_stats_from_partials: running stats come in pairs");
  DLE_CHECK_ARG(workspace_bytes >= 32LL * 2 * C * 4, "bn_stats_from_partials: workspace too small");
  const float* src = partial;
  int g = groups;
  static const int wide_max = getenv("DLE_BN_FINISH_WIDE") ? atoi(getenv("DLE_BN_FINISH_WIDE")) : 1100;   // 0: always fold + finish
  if (groups > 32 && groups <= wide_max) {
    hipLaunchKernelGGL(bn_stats_finish_wide_kernel, dim3((C + 7) / 8), dim3(256), 0, stream, partial, groups, C, (long long)M, eps,
                       momentum, mean, rstd, running_mean, running_var);
    DLE_LAUNCH_CHECK();
    return 0;
  }
  if (groups > 32) {
    hipLaunchKernelGGL(bn_partial_fold_kernel, dim3((C + 15) / 16, 32), dim3(256), 0, stream, partial, groups, C, (float*)workspace);
    DLE_LAUNCH_CHECK();
    src = (const float*)workspace;
    g = 32;
  }
  hipLaunchKernelGGL(bn_stats_finish_kernel, dim3((C + 63) / 64), dim3(256), 0, stream, src, g, C, (long long)M, eps, momentum,
                     mean, rstd, running_mean, running_var);
  DLE_LAUNCH_CHECK();
  return 0;
}

static const int g_bn_pf = getenv("DLE_BN_PF") ? atoi(getenv("DLE_BN_PF")) : 1;   // prefetched / compile-time-mode streaming kernels
static int g_bn_want_blocks = 1024;    // ~4 workgroups per CU
static int g_bn_bwd_u = 2;             // rows in flight per lane of the backward reduction (2 / 4 / 8); measured on the
                                       // batch-256 ResNet-50 shapes (tools/kbench/bn_bench): 2 -> 2.5 ms / step, 4 -> 3.1, 8 -> 5.5
static int g_bn_lpr_cap = 32;          // lanes per row: wide layers split their columns over blockIdx.x instead of
                                       // multiplying the row groups (the partials are groups x 2 x C floats)

// Tuning knobs for A/B measurements (tools/kbench/bn_bench): target workgroup count and loads in flight of the
// reduction kernels; values <= 0 keep the current setting.
extern "C" int dle_bn_tune_apply(int trips, int grid_cap) {
  if (trips > 0) g_bn_apply_trips = trips;
  if (grid_cap > 0) g_bn_apply_cap = grid_cap;
  return 0;
}

extern "C" int dle_bn_tune(int want_blocks, int bwd_rows_in_flight) {
  if (want_blocks >= 100000) { g_bn_lpr_cap = want_blocks / 100000; want_blocks %= 100000; }   // cap * 100000 + blocks
  if (want_blocks > 0) g_bn_want_blocks = want_blocks;
  if (bwd_rows_in_flight == 2 || bwd_rows_in_flight == 4 || bwd_rows_in_flight == 8) g_bn_bwd_u = bwd_rows_in_flight;
  return 0;
}

static int bn_reduce_geometry(long long M, int C, int& lpr, int& gx, long long& rpb, long long& gy) {
  const int cols_v = C / 8;
  lpr = 1;
  while (lpr < cols_v && lpr < g_bn_lpr_cap) lpr <<= 1;
  gx = (cols_v + lpr - 1) / lpr;
  long long want = g_bn_want_blocks / gx;
  if (want < 1) want = 1;
  rpb = (M + want - 1) / want;
  const long long min_rows = 8LL * (256 / lpr);
  if (rpb < min_rows) rpb = min_rows;
  gy = (M + rpb - 1) / rpb;
  return 0;
}

// workspace: fp32, >= dle_bn_workspace_bytes(M, C)
extern "C" int64_t dle_bn_workspace_bytes(int64_t M, int C) {
  int lpr, gx; long long rpb, gy;
  if (C <= 0 || C % 8 != 0 || M <= 0) return 0;
  bn_reduce_geometry(M, C, lpr, gx, rpb, gy);
  return gy * 2 * (int64_t)C * 4;
}

extern "C" int dle_bn_fwd_stats(const void* x, int64_t M, int C, float eps, float momentum, float* mean, float* rstd,
                                float* running_mean, float* running_var, void* workspace, int64_t workspace_bytes,
                                int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "bn_fwd_stats: 16-bit activations only");
  DLE_CHECK_ARG(M > 0 && C > 0 && C % 8 == 0, "bn_fwd_stats: bad shape (C must be a multiple of 8)");
  DLE_CHECK_ARG(x && mean && rstd && workspace, "bn_fwd_stats: null pointer");
  DLE_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "bn_fwd_stats: running stats come in pairs");
  int lpr, gx; long long rpb, gy;
  bn_reduce_geometry(M, C, lpr, gx, rpb, gy);
  DLE_CHECK_ARG(workspace_bytes >= gy * 2 * (long long)C * 4, "bn_fwd_stats: workspace too small");
  BnRedArgs a = {(const unsigned short*)x, nullptr, nullptr, nullptr, nullptr, nullptr, (float*)workspace, (long long)M, C, rpb, lpr};
  dim3 grid(gx, (unsigned)gy), block(256);
  if (dtype == DLE_F16) hipLaunchKernelGGL((bn_reduce_kernel<DLE_F16, 0>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((bn_reduce_kernel<DLE_BF16, 0>), grid, block, 0, stream, a);
  DLE_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_stats_finish_kernel, dim3((C + 63) / 64), dim3(256), 0, stream, (const float*)workspace, (int)gy, C,
                     (long long)M, eps, momentum, mean, rstd, running_mean, running_var);
  DLE_LAUNCH_CHECK();
  return 0;
}

// y = act( (x - mean) * rstd * gamma + beta (+ residual) ),  act = ReLU when relu != 0
// The grid stride (gridDim * 256 lanes) is a multiple of C/8 (a power of two <= 256 ... or any divisor of the
// stride), so a lane keeps the SAME 8 channels for its whole sweep: scale/shift live in registers.
template <int DT, int TR>
__global__ __launch_bounds__(256) void bn_apply_kernel(const unsigned short* __restrict__ x,
                                                       const unsigned short* __restrict__ res,
                                                       unsigned short* __restrict__ y, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, long long total8, int C8,
                                                       int relu, unsigned char* __restrict__ mask_out) {
  const long long first = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const bool invariant = (stride % C8) == 0;
  float sc[8], sh[8];
  int c0 = (int)(first % C8) * 8;
#pragma unroll
  for (int k = 0; k < 8; ++k) { sc[k] = rstd[c0 + k] * gamma[c0 + k]; sh[k] = beta[c0 + k] - mean[c0 + k] * sc[k]; }
  auto one = [&](long long i, ushort8_t xv, ushort8_t rv) {
    float xf[8], rf[8], of[8];
    unpack8<DT>(xv, xf);
    if (res) unpack8<DT>(rv, rf);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float v = xf[k] * sc[k] + sh[k];
      if (res) v += rf[k];
      if (relu) v = v > 0.f ? v : 0.f;
      of[k] = v;
    }
    ((ushort8_t*)y)[i] = pack8<DT>(of);
    if (mask_out) {                      // ReLU mask for the backward pass: 1 bit per element instead of re-reading y
      unsigned bits = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) bits |= (of[k] > 0.f ? 1u : 0u) << k;
      mask_out[i] = (unsigned char)bits;
    }
  };
  long long i = first;
  if (invariant) {
    // 4 trips in flight per lane (4-8 independent 16-byte loads): a lane alone does not cover the HBM latency
    for (; i + (TR - 1) * stride < total8; i += TR * stride) {
      ushort8_t xv[TR], rv[TR];
#pragma unroll
      for (int u = 0; u < TR; ++u) {
        xv[u] = ((const ushort8_t*)x)[i + u * stride];
        if (res) rv[u] = ((const ushort8_t*)res)[i + u * stride];
      }
#pragma unroll
      for (int u = 0; u < TR; ++u) one(i + u * stride, xv[u], rv[u]);
    }
  }
  for (; i < total8; i += stride) {
    if (!invariant) {
      c0 = (int)(i % C8) * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) { sc[k] = rstd[c0 + k] * gamma[c0 + k]; sh[k] = beta[c0 + k] - mean[c0 + k] * sc[k]; }
    }
    ushort8_t rv = {};
    if (res) rv = ((const ushort8_t*)res)[i];
    one(i, ((const ushort8_t*)x)[i], rv);
  }
}

// The same pass with the NEXT trip's loads requested before the current trip's stores (vmcnt retires in order: a load issued
// after a store cannot be consumed before that store is acknowledged), the residual / mask choices compile-time (no loads
// inside run-time conditionals) and the grid stride a multiple of C/8 (the launcher checks it).
template <int DT, bool HAS_RES, int ACT>      // ACT: 0 = none, 1 = ReLU, 2 = ReLU + bit mask
__global__ __launch_bounds__(256) void bn_apply_pf_kernel(const unsigned short* __restrict__ x,
                                                          const unsigned short* __restrict__ res,
                                                          unsigned short* __restrict__ y, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, long long total8, int C8,
                                                          unsigned char* __restrict__ mask_out) {
  const long long first = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  if (first >= total8) return;
  float sc[8], sh[8];
  const int c0 = (int)(first % C8) * 8;
#pragma unroll
  for (int k = 0; k < 8; ++k) { sc[k] = rstd[c0 + k] * gamma[c0 + k]; sh[k] = beta[c0 + k] - mean[c0 + k] * sc[k]; }
  auto one = [&](long long i, ushort8_t xv, ushort8_t rv) __attribute__((always_inline)) {
    float xf[8], rf[8], of[8];
    unpack8<DT>(xv, xf);
    if constexpr (HAS_RES) unpack8<DT>(rv, rf);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float v = xf[k] * sc[k] + sh[k];
      if constexpr (HAS_RES) v += rf[k];
      if constexpr (ACT >= 1) v = v > 0.f ? v : 0.f;
      of[k] = v;
    }
    ((ushort8_t*)y)[i] = pack8<DT>(of);
    if constexpr (ACT == 2) {
      unsigned bits = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) bits |= (of[k] > 0.f ? 1u : 0u) << k;
      mask_out[i] = (unsigned char)bits;
    }
  };
  long long i = first;
  ushort8_t xc = ((const ushort8_t*)x)[i], rc = {};
  if constexpr (HAS_RES) rc = ((const ushort8_t*)res)[i];
  // (first trip peeled: at the head of the loop the pending queue then has the same shape on both incoming edges -- next
  //  trip's loads, then this trip's stores -- and the compiler can wait with vmcnt(#stores) instead of vmcnt(0))
  auto trip = [&]() __attribute__((always_inline)) {
    const ushort8_t xn = ((const ushort8_t*)x)[i + stride];
    ushort8_t rn = {};
    if constexpr (HAS_RES) rn = ((const ushort8_t*)res)[i + stride];
    __builtin_amdgcn_sched_barrier(0);
    one(i, xc, rc);
    xc = xn; rc = rn;
    i += stride;
  };
  if (i + stride < total8) {
    trip();
    while (i + stride < total8) trip();
  }
  one(i, xc, rc);
}

extern "C" int dle_bn_fwd_apply(const void* x, const void* residual, void* y, void* relu_mask, const float* mean, const float* rstd,
                                const float* gamma, const float* beta, int64_t M, int C, int relu, int dtype,
                                hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "bn_fwd_apply: 16-bit activations only");
  DLE_CHECK_ARG(M >= 0 && C > 0 && C % 8 == 0, "bn_fwd_apply: bad shape");
  if (M == 0) return 0;
  DLE_CHECK_ARG(x && y && mean && rstd && gamma && beta, "bn_fwd_apply: null pointer");
  const long long total8 = (long long)M * (C / 8);
  const int grid = cn_grid(total8, 256, g_bn_apply_cap);
#define BN_APP(DT, TR) hipLaunchKernelGGL((bn_apply_kernel<DT, TR>), dim3(grid), dim3(256), 0, stream, (const unsigned short*)x, (const unsigned short*)residual, (unsigned short*)y, mean, rstd, gamma, beta, total8, C / 8, relu, (unsigned char*)relu_mask)
#define BN_APP_T(DT) do { if (g_bn_apply_trips == 1) BN_APP(DT, 1); else if (g_bn_apply_trips == 2) BN_APP(DT, 2); else BN_APP(DT, 4); } while (0)
  if (g_bn_pf && ((long long)grid * 256) % (C / 8) == 0) {
    const int act = !relu ? 0 : (relu_mask ? 2 : 1);
#define BN_PF(DT, HR, ACT) hipLaunchKernelGGL((bn_apply_pf_kernel<DT, HR, ACT>), dim3(grid), dim3(256), 0, stream, (const unsigned short*)x, (const unsigned short*)residual, (unsigned short*)y, mean, rstd, gamma, beta, total8, C / 8, (unsigned char*)relu_mask)
#define BN_PF_A(DT, HR) do { if (act == 0) BN_PF(DT, HR, 0); else if (act == 1) BN_PF(DT, HR, 1); else BN_PF(DT, HR, 2); } while (0)
#define BN_PF_R(DT) do { if (residual) BN_PF_A(DT, true); else BN_PF_A(DT, false); } while (0)
    if (dtype == DLE_F16) BN_PF_R(DLE_F16); else BN_PF_R(DLE_BF16);
#undef BN_PF
#undef BN_PF_A
#undef BN_PF_R
  } else {
    if (dtype == DLE_F16) BN_APP_T(DLE_F16); else BN_APP_T(DLE_BF16);
  }
#undef BN_APP
#undef BN_APP_T
  DLE_LAUNCH_CHECK();
  return 0;
}

// y = relu(bn(x) + bn_r(xr)) with the keep bits: a bottleneck's bn3 apply with the downsample branch's BatchNorm (no ReLU of its
// own, models/resnet.py:166-173) taken on the residual's load -- the branch's 16-bit output is never written or re-read.  The
// residual term is rounded to the activation dtype exactly where the stand-alone pass stored it, so y and the bits are
// bit-identical to dle_bn_fwd_apply(xr -> res) followed by dle_bn_fwd_apply(x, res).  Same trip structure as bn_apply_pf_kernel.
template <int DT>
__global__ __launch_bounds__(256) void bn_apply2_pf_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ xr,
                                                           unsigned short* __restrict__ y, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ mean_r,
                                                           const float* __restrict__ rstd_r, const float* __restrict__ gamma_r,
                                                           const float* __restrict__ beta_r, long long total8, int C8,
                                                           unsigned char* __restrict__ mask_out) {
  const long long first = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  if (first >= total8) return;
  float sc[8], sh[8], scr[8], shr[8];
  const int c0 = (int)(first % C8) * 8;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    sc[k] = rstd[c0 + k] * gamma[c0 + k]; sh[k] = beta[c0 + k] - mean[c0 + k] * sc[k];
    scr[k] = rstd_r[c0 + k] * gamma_r[c0 + k]; shr[k] = beta_r[c0 + k] - mean_r[c0 + k] * scr[k];
  }
  auto one = [&](long long i, ushort8_t xv, ushort8_t rv) __attribute__((always_inline)) {
    float xf[8], rf[8], of[8];
    unpack8<DT>(xv, xf);
    unpack8<DT>(rv, rf);
#pragma unroll
    for (int k = 0; k < 8; ++k) rf[k] = rf[k] * scr[k] + shr[k];
    unpack8<DT>(pack8<DT>(rf), rf);                       // (the rounding point of the stand-alone branch output)
    unsigned bits = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float v = xf[k] * sc[k] + sh[k];
      v += rf[k];
      v = v > 0.f ? v : 0.f;
      of[k] = v;
      bits |= (v > 0.f ? 1u : 0u) << k;
    }
    ((ushort8_t*)y)[i] = pack8<DT>(of);
    mask_out[i] = (unsigned char)bits;
  };
  long long i = first;
  ushort8_t xc = ((const ushort8_t*)x)[i], rc = ((const ushort8_t*)xr)[i];
  auto trip = [&]() __attribute__((always_inline)) {
    const ushort8_t xn = ((const ushort8_t*)x)[i + stride];
    const ushort8_t rn = ((const ushort8_t*)xr)[i + stride];
    __builtin_amdgcn_sched_barrier(0);
    one(i, xc, rc);
    xc = xn; rc = rn;
    i += stride;
  };
  if (i + stride < total8) {
    trip();
    while (i + stride < total8) trip();
  }
  one(i, xc, rc);
}

// y = relu(bn(x) + bn_r(xr)), relu_mask = keep bits: see bn_apply2_pf_kernel.  All four statistics / affine vectors of both
// BatchNorms are fp32 [C]; x, xr, y 16-bit [M, C]; relu_mask [M C / 8].
extern "C" int dle_bn_fwd_apply2(const void* x, const void* xr, void* y, void* relu_mask, const float* mean, const float* rstd,
                                 const float* gamma, const float* beta, const float* mean_r, const float* rstd_r, const float* gamma_r,
                                 const float* beta_r, int64_t M, int C, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "bn_fwd_apply2: 16-bit activations only");
  DLE_CHECK_ARG(M >= 0 && C > 0 && C % 8 == 0, "bn_fwd_apply2: bad shape");
  if (M == 0) return 0;
  DLE_CHECK_ARG(x && xr && y && relu_mask && mean && rstd && gamma && beta && mean_r && rstd_r && gamma_r && beta_r,
                "bn_fwd_apply2: null pointer");
  const long long total8 = (long long)M * (C / 8);
  int grid = cn_grid(total8, 256, g_bn_apply_cap);
  while (((long long)grid * 256) % (C / 8) != 0) ++grid;   // a lane keeps its 8 channels for the whole sweep
  if (dtype == DLE_F16)
    hipLaunchKernelGGL((bn_apply2_pf_kernel<DLE_F16>), dim3(grid), dim3(256), 0, stream, (const unsigned short*)x, (const unsigned short*)xr,
                       (unsigned short*)y, mean, rstd, gamma, beta, mean_r, rstd_r, gamma_r, beta_r, total8, C / 8, (unsigned char*)relu_mask);
  else
    hipLaunchKernelGGL((bn_apply2_pf_kernel<DLE_BF16>), dim3(grid), dim3(256), 0, stream, (const unsigned short*)x, (const unsigned short*)xr,
                       (unsigned short*)y, mean, rstd, gamma, beta, mean_r, rstd_r, gamma_r, beta_r, total8, C / 8, (unsigned char*)relu_mask);
  DLE_LAUNCH_CHECK();
  return 0;
}

// backward pass 1: dgamma / dbeta (fp32) with the ReLU mask taken from the saved output y (NULL: no ReLU)
extern "C" int dle_bn_bwd_reduce(const void* dy, const void* y, const void* relu_mask, const void* x, const float* mean, const float* rstd,
                                 float* dgamma, float* dbeta, int64_t M, int C, int accumulate, void* workspace,
                                 int64_t workspace_bytes, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "bn_bwd_reduce: 16-bit activations only");
  DLE_CHECK_ARG(M > 0 && C > 0 && C % 8 == 0, "bn_bwd_reduce: bad shape");
  DLE_CHECK_ARG(dy && x && mean && rstd && dgamma && dbeta && workspace, "bn_bwd_reduce: null pointer");
  int lpr, gx; long long rpb, gy;
  bn_reduce_geometry(M, C, lpr, gx, rpb, gy);
  DLE_CHECK_ARG(workspace_bytes >= gy * 2 * (long long)C * 4, "bn_bwd_reduce: workspace too small");
  BnRedArgs a = {(const unsigned short*)x, (const unsigned short*)dy, (const unsigned short*)y,
                 (const unsigned char*)relu_mask, mean, rstd, (float*)workspace, (long long)M, C, rpb, lpr};
  dim3 grid(gx, (unsigned)gy), block(256);
  const int mm = g_bn_pf ? (relu_mask ? 1 : (y ? 2 : 0)) : -1;
#define BN_RED(DT) do { if (mm == 1) hipLaunchKernelGGL((bn_reduce_kernel<DT, 1, 2, 1>), grid, block, 0, stream, a); \
    else if (mm == 0) hipLaunchKernelGGL((bn_reduce_kernel<DT, 1, 2, 0>), grid, block, 0, stream, a); \
    else if (mm == 2) hipLaunchKernelGGL((bn_reduce_kernel<DT, 1, 2, 2>), grid, block, 0, stream, a); \
    else if (g_bn_bwd_u == 2) hipLaunchKernelGGL((bn_reduce_kernel<DT, 1, 2>), grid, block, 0, stream, a); \
    else if (g_bn_bwd_u == 8) hipLaunchKernelGGL((bn_reduce_kernel<DT, 1, 8>), grid, block, 0, stream, a); \
    else hipLaunchKernelGGL((bn_reduce_kernel<DT, 1, 4>), grid, block, 0, stream, a); } while (0)
  if (dtype == DLE_F16) BN_RED(DLE_F16); else BN_RED(DLE_BF16);
#undef BN_RED
  DLE_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((C + 7) / 8), dim3(256), 0, stream, (const float*)workspace, (int)gy, C,
                     dgamma, dbeta, accumulate);
  DLE_LAUNCH_CHECK();
  return 0;
}

// The backward reduction of TWO BatchNorms that receive the same gradient: bn3 and the downsample branch's BatchNorm of a
// bottleneck's first block both see g = dy under the block's output keep bits (out = relu(bn3(t3) + bn_ds(t_ds)),
// models/resnet.py:166-173).  Two dle_bn_bwd_reduce launches read dy and the mask twice; this one reads them once:
// sum g (shared), sum g xhat_1, sum g xhat_2.  Per BatchNorm the partial rows and the fold are those of bn_reduce_kernel<.., 1, 2, 1>
// + bn_bwd_finish_kernel: the same products in the same order, results bit-identical to the two launches.
struct BnRed2Args {
  const unsigned short* x1; const unsigned short* x2; const unsigned short* dy; const unsigned char* mask;
  const float* mean1; const float* rstd1; const float* mean2; const float* rstd2;
  float* partial1; float* partial2;   // [groups][2][C] each
  long long M; int C; long long rows_per_block; int lpr;
};
template <int DT>
__global__ __launch_bounds__(256) void bn_reduce2_kernel(BnRed2Args a) {
  __shared__ float red[3][256 * 8];
  const int cl = threadIdx.x % a.lpr, rl = threadIdx.x / a.lpr, rstep = 256 / a.lpr;
  const int c0 = (blockIdx.x * a.lpr + cl) * 8;
  const long long r0 = (long long)blockIdx.y * a.rows_per_block;
  long long r1 = r0 + a.rows_per_block;
  if (r1 > a.M) r1 = a.M;
  float s0[8], s1[8], s2[8], mu1[8], rs1[8], mu2[8], rs2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { s0[k] = 0.f; s1[k] = 0.f; s2[k] = 0.f; mu1[k] = 0.f; rs1[k] = 1.f; mu2[k] = 0.f; rs2[k] = 1.f; }
  if (c0 < a.C) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { mu1[k] = a.mean1[c0 + k]; rs1[k] = a.rstd1[c0 + k]; mu2[k] = a.mean2[c0 + k]; rs2[k] = a.rstd2[c0 + k]; }
    constexpr int U = 2;
    auto load_batch = [&](long long rb, ushort8_t (&xv)[U], ushort8_t (&zv)[U], ushort8_t (&gv)[U], unsigned (&mb)[U])
        __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long o = (rb + (long long)u * rstep) * a.C + c0;
        xv[u] = *(const ushort8_t*)(a.x1 + o);
        zv[u] = *(const ushort8_t*)(a.x2 + o);
        gv[u] = *(const ushort8_t*)(a.dy + o);
        mb[u] = a.mask[o >> 3];
      }
    };
    auto accum = [&](ushort8_t xv, ushort8_t zv, ushort8_t gv, unsigned bits) __attribute__((always_inline)) {
      float xf[8], zf[8], gf[8];
      unpack8<DT>(xv, xf);
      unpack8<DT>(zv, zf);
      unpack8<DT>(gv, gf);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float g = gf[k];
        if (!((bits >> k) & 1u)) g = 0.f;
        s0[k] += g;
        s1[k] += g * (xf[k] - mu1[k]) * rs1[k];
        s2[k] += g * (zf[k] - mu2[k]) * rs2[k];
      }
    };
    long long r = r0 + rl;
    const long long bstep = (long long)U * rstep;
    if (r + (long long)(U - 1) * rstep < r1) {
      ushort8_t xv[U], zv[U], gv[U], xn[U], zn[U], gn[U];
      unsigned mb[U], mn[U];
      load_batch(r, xv, zv, gv, mb);
      for (; r + bstep + (long long)(U - 1) * rstep < r1; r += bstep) {
        load_batch(r + bstep, xn, zn, gn, mn);
#pragma unroll
        for (int u = 0; u < U; ++u) accum(xv[u], zv[u], gv[u], mb[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) { xv[u] = xn[u]; zv[u] = zn[u]; gv[u] = gn[u]; mb[u] = mn[u]; }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) accum(xv[u], zv[u], gv[u], mb[u]);
      r += bstep;
    }
    for (; r < r1; r += rstep) {
      const long long o = r * a.C + c0;
      accum(*(const ushort8_t*)(a.x1 + o), *(const ushort8_t*)(a.x2 + o), *(const ushort8_t*)(a.dy + o), a.mask[o >> 3]);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[0][threadIdx.x * 8 + k] = s0[k]; red[1][threadIdx.x * 8 + k] = s1[k]; red[2][threadIdx.x * 8 + k] = s2[k]; }
  __syncthreads();
  if (rl == 0 && c0 < a.C) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float t0 = 0.f, t1 = 0.f, t2 = 0.f;
      for (int q = 0; q < rstep; ++q) {
        t0 += red[0][(q * a.lpr + cl) * 8 + k]; t1 += red[1][(q * a.lpr + cl) * 8 + k]; t2 += red[2][(q * a.lpr + cl) * 8 + k];
      }
      a.partial1[((long long)blockIdx.y * 2 + 0) * a.C + c0 + k] = t0;
      a.partial1[((long long)blockIdx.y * 2 + 1) * a.C + c0 + k] = t1;
      a.partial2[((long long)blockIdx.y * 2 + 0) * a.C + c0 + k] = t0;
      a.partial2[((long long)blockIdx.y * 2 + 1) * a.C + c0 + k] = t2;
    }
  }
}

// dgamma1 / dbeta1 of (x1, mean1, rstd1) and dgamma2 / dbeta2 of (x2, mean2, rstd2) from ONE pass over dy and relu_mask (bit-packed,
// required).  workspace: fp32, >= 2 * dle_bn_workspace_bytes(M, C).  Bit-identical to two dle_bn_bwd_reduce calls.
extern "C" int dle_bn_bwd_reduce2(const void* dy, const void* relu_mask, const void* x1, const float* mean1, const float* rstd1,
                                  float* dgamma1, float* dbeta1, const void* x2, const float* mean2, const float* rstd2,
                                  float* dgamma2, float* dbeta2, int64_t M, int C, void* workspace, int64_t workspace_bytes,
                                  int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "bn_bwd_reduce2: 16-bit activations only");
  DLE_CHECK_ARG(M > 0 && C > 0 && C % 8 == 0, "bn_bwd_reduce2: bad shape");
  DLE_CHECK_ARG(dy && relu_mask && x1 && x2 && mean1 && rstd1 && mean2 && rstd2 && dgamma1 && dbeta1 && dgamma2 && dbeta2 && workspace,
                "bn_bwd_reduce2: null pointer");
  int lpr, gx; long long rpb, gy;
  bn_reduce_geometry(M, C, lpr, gx, rpb, gy);
  const long long one = gy * 2 * (long long)C;
  DLE_CHECK_ARG(workspace_bytes >= 2 * one * 4, "bn_bwd_reduce2: workspace too small");
  BnRed2Args a = {(const unsigned short*)x1, (const unsigned short*)x2, (const unsigned short*)dy, (const unsigned char*)relu_mask,
                  mean1, rstd1, mean2, rstd2, (float*)workspace, (float*)workspace + one, (long long)M, C, rpb, lpr};
  dim3 grid(gx, (unsigned)gy), block(256);
  if (dtype == DLE_F16) hipLaunchKernelGGL(bn_reduce2_kernel<DLE_F16>, grid, block, 0, stream, a);
  else hipLaunchKernelGGL(bn_reduce2_kernel<DLE_BF16>, grid, block, 0, stream, a);
  DLE_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((C + 7) / 8), dim3(256), 0, stream, (const float*)workspace, (int)gy, C, dgamma1, dbeta1, 0);
  hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((C + 7) / 8), dim3(256), 0, stream, (const float*)workspace + one, (int)gy, C, dgamma2, dbeta2, 0);
  DLE_LAUNCH_CHECK();
  return 0;
}

// The fold half of dle_bn_bwd_reduce on its own: dgamma / dbeta from `groups` partial rows [groups][2][C] of (sum g, sum g xhat)
// that another kernel left (dle_gemm_expand_masked_bnred takes the reduction in the epilogue that PRODUCES the gradient).
extern "C" int dle_bn_bwd_finish(const float* partial, int groups, int C, float* dgamma, float* dbeta, int accumulate,
                                 hipStream_t stream) {
  DLE_CHECK_ARG(partial && dgamma && dbeta && groups > 0 && C > 0, "bn_bwd_finish: bad args");
  hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((C + 7) / 8), dim3(256), 0, stream, partial, groups, C, dgamma, dbeta, accumulate);
  DLE_LAUNCH_CHECK();
  return 0;
}

// backward pass 2: dx = gamma * rstd * (g - dbeta/M - xhat * dgamma/M), g = dy * (y > 0);
// g_out (optional) receives g: the gradient that flows into the residual branch.
template <int DT, int TR>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const unsigned short* __restrict__ dy,
                                                           const unsigned short* __restrict__ y,
                                                           const unsigned char* __restrict__ mask,
                                                           const unsigned short* __restrict__ x,
                                                           unsigned short* __restrict__ dx,
                                                           unsigned short* __restrict__ g_out,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                                           const float* __restrict__ dbeta, long long total8, int C8,
                                                           float inv_m) {
  const long long first = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const bool invariant = (stride % C8) == 0;
  // dx = a * g + b * x + c   with  a = gamma*rstd,  b = -a * rstd^2 * dgamma/M ... written per channel:
  //   xhat = (x - mean) * rstd ;  dx = a * (g - dbeta/M - xhat * dgamma/M)
  float ka[8], kmu[8], krs[8], kb[8], kg[8];
  auto load = [&](int c0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      krs[k] = rstd[c0 + k];
      kmu[k] = mean[c0 + k];
      ka[k] = gamma[c0 + k] * krs[k];
      kb[k] = dbeta[c0 + k] * inv_m;
      kg[k] = dgamma[c0 + k] * inv_m;
    }
  };
  load((int)(first % C8) * 8);
  auto one = [&](long long i, ushort8_t gv, ushort8_t xv, ushort8_t yv, unsigned bits) {
    float gf[8], xf[8], yf[8], of[8];
    unpack8<DT>(gv, gf);
    unpack8<DT>(xv, xf);
    if (y) unpack8<DT>(yv, yf);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (mask ? !((bits >> k) & 1u) : (y && !(yf[k] > 0.f))) gf[k] = 0.f;
      const float xh = (xf[k] - kmu[k]) * krs[k];
      of[k] = ka[k] * (gf[k] - kb[k] - xh * kg[k]);
    }
    ((ushort8_t*)dx)[i] = pack8<DT>(of);
    if (g_out) ((ushort8_t*)g_out)[i] = pack8<DT>(gf);
  };
  long long i = first;
  if (invariant) {
    for (; i + (TR - 1) * stride < total8; i += TR * stride) {        // TR trips = 2-3 TR independent 16-byte loads in flight
      ushort8_t gv[TR], xv[TR], yv[TR];
      unsigned mb[TR];
#pragma unroll
      for (int u = 0; u < TR; ++u) {
        gv[u] = ((const ushort8_t*)dy)[i + u * stride];
        xv[u] = ((const ushort8_t*)x)[i + u * stride];
        mb[u] = 0;
        if (mask) mb[u] = mask[i + u * stride];
        else if (y) yv[u] = ((const ushort8_t*)y)[i + u * stride];
      }
#pragma unroll
      for (int u = 0; u < TR; ++u) one(i + u * stride, gv[u], xv[u], yv[u], mb[u]);
    }
  }
  for (; i < total8; i += stride) {
    if (!invariant) load((int)(i % C8) * 8);
    ushort8_t yv = {};
    unsigned mb = 0;
    if (mask) mb = mask[i];
    else if (y) yv = ((const ushort8_t*)y)[i];
    one(i, ((const ushort8_t*)dy)[i], ((const ushort8_t*)x)[i], yv, mb);
  }
}

// Prefetched form of the pass above (see bn_apply_pf_kernel): MM = 0 no ReLU mask, 1 bit mask, 2 saved output.
template <int DT, int MM, bool HAS_GOUT>
__global__ __launch_bounds__(256) void bn_bwd_apply_pf_kernel(const unsigned short* __restrict__ dy,
                                                              const unsigned short* __restrict__ y,
                                                              const unsigned char* __restrict__ mask,
                                                              const unsigned short* __restrict__ x,
                                                              unsigned short* __restrict__ dx,
                                                              unsigned short* __restrict__ g_out,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                                              const float* __restrict__ dbeta, long long total8, int C8,
                                                              float inv_m) {
  const long long first = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  if (first >= total8) return;
  float ka[8], kmu[8], krs[8], kb[8], kg[8];
  const int c0 = (int)(first % C8) * 8;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    krs[k] = rstd[c0 + k];
    kmu[k] = mean[c0 + k];
    ka[k] = gamma[c0 + k] * krs[k];
    kb[k] = dbeta[c0 + k] * inv_m;
    kg[k] = dgamma[c0 + k] * inv_m;
  }
  auto one = [&](long long i, ushort8_t gv, ushort8_t xv, ushort8_t yv, unsigned bits) __attribute__((always_inline)) {
    float gf[8], xf[8], yf[8], of[8];
    unpack8<DT>(gv, gf);
    unpack8<DT>(xv, xf);
    if constexpr (MM == 2) unpack8<DT>(yv, yf);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if constexpr (MM == 1) { if (!((bits >> k) & 1u)) gf[k] = 0.f; }
      if constexpr (MM == 2) { if (!(yf[k] > 0.f)) gf[k] = 0.f; }
      const float xh = (xf[k] - kmu[k]) * krs[k];
      of[k] = ka[k] * (gf[k] - kb[k] - xh * kg[k]);
    }
    ((ushort8_t*)dx)[i] = pack8<DT>(of);
    if constexpr (HAS_GOUT) ((ushort8_t*)g_out)[i] = pack8<DT>(gf);
  };
  auto load = [&](long long i, ushort8_t& gv, ushort8_t& xv, ushort8_t& yv, unsigned& mb) __attribute__((always_inline)) {
    gv = ((const ushort8_t*)dy)[i];
    xv = ((const ushort8_t*)x)[i];
    mb = 0;
    if constexpr (MM == 1) mb = mask[i];
    if constexpr (MM == 2) yv = ((const ushort8_t*)y)[i];
  };
  long long i = first;
  ushort8_t gc, xc, yc = {}, gn, xn, yn = {};
  unsigned mc, mn;
  load(i, gc, xc, yc, mc);
  auto trip = [&]() __attribute__((always_inline)) {      // (first trip peeled, see bn_apply_pf_kernel)
    load(i + stride, gn, xn, yn, mn);
    __builtin_amdgcn_sched_barrier(0);
    one(i, gc, xc, yc, mc);
    gc = gn; xc = xn; yc = yn; mc = mn;
    i += stride;
  };
  if (i + stride < total8) {
    trip();
    while (i + stride < total8) trip();
  }
  one(i, gc, xc, yc, mc);
}

extern "C" int dle_bn_bwd_apply(const void* dy, const void* y, const void* relu_mask, const void* x, void* dx, void* g_out, const float* mean,
                                const float* rstd, const float* gamma, const float* dgamma, const float* dbeta,
                                int64_t M, int C, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "bn_bwd_apply: 16-bit activations only");
  DLE_CHECK_ARG(M > 0 && C > 0 && C % 8 == 0, "bn_bwd_apply: bad shape");
  DLE_CHECK_ARG(dy && x && dx && mean && rstd && gamma && dgamma && dbeta, "bn_bwd_apply: null pointer");
  const long long total8 = (long long)M * (C / 8);
  const int grid = cn_grid(total8, 256, g_bn_apply_cap);
#define BN_BAPP(DT, TR) hipLaunchKernelGGL((bn_bwd_apply_kernel<DT, TR>), dim3(grid), dim3(256), 0, stream, (const unsigned short*)dy, (const unsigned short*)y, (const unsigned char*)relu_mask, (const unsigned short*)x, (unsigned short*)dx, (unsigned short*)g_out, mean, rstd, gamma, dgamma, dbeta, total8, C / 8, 1.0f / (float)M)
#define BN_BAPP_T(DT) do { if (g_bn_apply_trips == 1) BN_BAPP(DT, 1); else if (g_bn_apply_trips == 2) BN_BAPP(DT, 2); else BN_BAPP(DT, 3); } while (0)
  if (g_bn_pf && ((long long)grid * 256) % (C / 8) == 0) {
    const int mm = relu_mask ? 1 : (y ? 2 : 0);
#define BN_PF(DT, MMV, GO) hipLaunchKernelGGL((bn_bwd_apply_pf_kernel<DT, MMV, GO>), dim3(grid), dim3(256), 0, stream, (const unsigned short*)dy, (const unsigned short*)y, (const unsigned char*)relu_mask, (const unsigned short*)x, (unsigned short*)dx, (unsigned short*)g_out, mean, rstd, gamma, dgamma, dbeta, total8, C / 8, 1.0f / (float)M)
#define BN_PF_M(DT, GO) do { if (mm == 0) BN_PF(DT, 0, GO); else if (mm == 1) BN_PF(DT, 1, GO); else BN_PF(DT, 2, GO); } while (0)
#define BN_PF_G(DT) do { if (g_out) BN_PF_M(DT, true); else BN_PF_M(DT, false); } while (0)
    if (dtype == DLE_F16) BN_PF_G(DLE_F16); else BN_PF_G(DLE_BF16);
#undef BN_PF
#undef BN_PF_M
#undef BN_PF_G
  } else {
    if (dtype == DLE_F16) BN_BAPP_T(DLE_F16); else BN_BAPP_T(DLE_BF16);
  }
#undef BN_BAPP
#undef BN_BAPP_T
  DLE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- zero-stuffed upsampling (1x1 stride-s data gradient)
// y[n, h, w, :] = x[n, h/s, w/s, :] when h, w are multiples of s (and in range), else 0.  The data gradient of a
// 1x1 stride-s convolution (the ResNet downsample branches) is a plain GEMM on the P x Q grid followed by this pass --
// the gather-form implicit GEMM spends 3/4 of its tiles on rows that are identically zero.
template <int DT>
__global__ __launch_bounds__(256) void upsample_zero_kernel(const unsigned short* __restrict__ x,
                                                            unsigned short* __restrict__ y, long long total8, int H,
                                                            int W, int P, int Q, int C8, int stride) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (long long)gridDim.x * blockDim.x) {
    const long long pix = i / C8;
    const int c = (int)(i - pix * C8);
    const int w = (int)(pix % W);
    const long long t = pix / W;
    const int h = (int)(t % H);
    const long long n = t / H;
    ushort8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
    const int p = h / stride, q = w / stride;
    if (p * stride == h && q * stride == w && p < P && q < Q) v = ((const ushort8_t*)x)[((n * P + p) * Q + q) * C8 + c];
    ((ushort8_t*)y)[i] = v;
  }
}

extern "C" int dle_upsample_zero(const void* x, void* y, int64_t N, int P, int Q, int H, int W, int C, int stride,
                                 int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "upsample_zero: 16-bit activations only");
  DLE_CHECK_ARG(N >= 0 && P > 0 && Q > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && stride >= 1, "upsample_zero: bad shape");
  if (N == 0) return 0;
  DLE_CHECK_ARG(x && y, "upsample_zero: null pointer");
  const long long total8 = (long long)N * H * W * (C / 8);
  const int grid = cn_grid(total8, 256, 4096);
  if (dtype == DLE_F16) hipLaunchKernelGGL(upsample_zero_kernel<DLE_F16>, dim3(grid), dim3(256), 0, stream, (const unsigned short*)x, (unsigned short*)y, total8, H, W, P, Q, C / 8, stride);
  else hipLaunchKernelGGL(upsample_zero_kernel<DLE_BF16>, dim3(grid), dim3(256), 0, stream, (const unsigned short*)x, (unsigned short*)y, total8, H, W, P, Q, C / 8, stride);
  DLE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- max pooling (NHWC, kernel k, stride s, pad p)
// argmax[n,p,q,c] = window-scan index (r * k + s) of the FIRST maximum (ATen's tie rule), uint8.
template <int DT>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const unsigned short* __restrict__ x,
                                                          unsigned short* __restrict__ y,
                                                          unsigned char* __restrict__ amax, int N, int H, int W, int C8,
                                                          int P, int Q, int ks, int st, int pad) {
  const long long total = (long long)N * P * Q * C8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % C8);
    long long t = i / C8;
    const int q = (int)(t % Q); t /= Q;
    const int p = (int)(t % P);
    const int n = (int)(t / P);
    float best[8];
    unsigned char bi[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { best[k] = -INFINITY; bi[k] = 0; }
    for (int r = 0; r < ks; ++r) {
      const int h = p * st - pad + r;
      if (h < 0 || h >= H) continue;
      for (int s = 0; s < ks; ++s) {
        const int w = q * st - pad + s;
        if (w < 0 || w >= W) continue;
        const ushort8_t v = *(const ushort8_t*)(x + ((((long long)n * H + h) * W + w) * C8 + c8) * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float f = up16<DT>(v[k]);
          if (f > best[k] || f != f) { best[k] = f; bi[k] = (unsigned char)(r * ks + s); }
        }
      }
    }
    ushort8_t o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = dn16<DT>(best[k]);
    ((ushort8_t*)y)[i] = o;
    uint2_t packed;
    packed[0] = bi[0] | (bi[1] << 8) | (bi[2] << 16) | ((unsigned)bi[3] << 24);
    packed[1] = bi[4] | (bi[5] << 8) | (bi[6] << 16) | ((unsigned)bi[7] << 24);
    ((uint2_t*)amax)[i] = packed;
  }
}

// gather form (no atomics): each input pixel visits the <= ceil(k/s)^2 windows that cover it
template <int DT>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const unsigned short* __restrict__ dy,
                                                          const unsigned char* __restrict__ amax,
                                                          unsigned short* __restrict__ dx, int N, int H, int W, int C8,
                                                          int P, int Q, int ks, int st, int pad) {
  const long long total = (long long)N * H * W * C8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % C8);
    long long t = i / C8;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    for (int r = 0; r < ks; ++r) {
      const int hp = h + pad - r;
      if (hp < 0 || hp % st != 0) continue;
      const int p = hp / st;
      if (p >= P) continue;
      for (int s = 0; s < ks; ++s) {
        const int wp = w + pad - s;
        if (wp < 0 || wp % st != 0) continue;
        const int q = wp / st;
        if (q >= Q) continue;
        const long long o = (((long long)n * P + p) * Q + q) * C8 + c8;
        const ushort8_t g = ((const ushort8_t*)dy)[o];
        const uint2_t am = ((const uint2_t*)amax)[o];
        const unsigned char want = (unsigned char)(r * ks + s);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const unsigned char a = (unsigned char)((am[k >> 2] >> ((k & 3) * 8)) & 0xff);
          if (a == want) acc[k] += up16<DT>(g[k]);
        }
      }
    }
    ushort8_t o8;
#pragma unroll
    for (int k = 0; k < 8; ++k) o8[k] = dn16<DT>(acc[k]);
    ((ushort8_t*)dx)[i] = o8;
  }
}

extern "C" int dle_maxpool_fwd(const void* x, void* y, void* argmax, int N, int H, int W, int C, int ksize, int stride,
                               int pad, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "maxpool_fwd: 16-bit activations only");
  DLE_CHECK_ARG(C % 8 == 0 && ksize > 0 && ksize * ksize <= 255 && stride > 0, "maxpool_fwd: bad geometry");
  if (N == 0) return 0;
  DLE_CHECK_ARG(x && y && argmax, "maxpool_fwd: null pointer");
  const int P = (H + 2 * pad - ksize) / stride + 1, Q = (W + 2 * pad - ksize) / stride + 1;
  const int grid = cn_grid((long long)N * P * Q * (C / 8), 256, 8192);
  if (dtype == DLE_F16) hipLaunchKernelGGL(maxpool_fwd_kernel<DLE_F16>, dim3(grid), dim3(256), 0, stream, (const unsigned short*)x, (unsigned short*)y, (unsigned char*)argmax, N, H, W, C / 8, P, Q, ksize, stride, pad);
  else hipLaunchKernelGGL(maxpool_fwd_kernel<DLE_BF16>, dim3(grid), dim3(256), 0, stream, (const unsigned short*)x, (unsigned short*)y, (unsigned char*)argmax, N, H, W, C / 8, P, Q, ksize, stride, pad);
  DLE_LAUNCH_CHECK();
  return 0;
}

// MaxPool2d(3, 2, 1) -- the ResNet stem: an input pixel lies in ONE window per axis when its coordinate is even (tap 1) and in
// two when it is odd (taps 0 and 2): at most four candidate windows, requested up front from unconditional (clamped)
// addresses and masked.  The generic gather above walks 3 x 3 taps through `continue`s (loads inside conditionals: one in
// flight per lane) with 64-bit div / mod per element: 303 us for the 205 MB gradient of a batch-256 stem.
template <int DT>
__global__ __launch_bounds__(256) void maxpool_bwd_k3s2_kernel(const unsigned short* __restrict__ dy,
                                                               const unsigned char* __restrict__ amax,
                                                               unsigned short* __restrict__ dx, int N, int H, int W, int C8,
                                                               int P, int Q) {
  const unsigned total = (unsigned)N * H * W * C8;                 // (the launcher checks < 2^31)
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned c8 = i % (unsigned)C8;
    unsigned t = i / (unsigned)C8;
    const int w = (int)(t % (unsigned)W); t /= (unsigned)W;
    const int h = (int)(t % (unsigned)H);
    const int n = (int)(t / (unsigned)H);
    // candidate windows per axis: (index, tap); the second exists only for odd coordinates
    const int p0 = (h + 1) >> 1, r0 = (h & 1) ? 0 : 1, p1 = (h - 1) >> 1;      // p1: tap 2, odd h only
    const int q0 = (w + 1) >> 1, s0 = (w & 1) ? 0 : 1, q1 = (w - 1) >> 1;
    const bool vp[2] = {p0 < P, (h & 1) != 0};
    const bool vq[2] = {q0 < Q, (w & 1) != 0};
    const int pp[2] = {p0 < P ? p0 : P - 1, (h & 1) ? p1 : 0}, rr[2] = {r0, 2};
    const int qq[2] = {q0 < Q ? q0 : Q - 1, (w & 1) ? q1 : 0}, ss[2] = {s0, 2};
    ushort8_t g[4];
    uint2_t am[4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const unsigned o = (((unsigned)n * P + pp[a]) * Q + qq[b]) * C8 + c8;
        g[a * 2 + b] = ((const ushort8_t*)dy)[o];
        am[a * 2 + b] = ((const uint2_t*)amax)[o];
      }
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const bool ok = vp[a] && vq[b];
        const unsigned want = ok ? (unsigned)(rr[a] * 3 + ss[b]) : 0xFFu;       // argmax codes are < 9
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const unsigned code = (am[a * 2 + b][k >> 2] >> ((k & 3) * 8)) & 0xffu;
          if (code == want) acc[k] += up16<DT>(g[a * 2 + b][k]);
        }
      }
    ushort8_t o8;
#pragma unroll
    for (int k = 0; k < 8; ++k) o8[k] = dn16<DT>(acc[k]);
    ((ushort8_t*)dx)[i] = o8;
  }
}

// The same pooling backward with one thread per 2 x 2 PATCH of input pixels (2 p', 2 q') .. (2 p' + 1, 2 q' + 1): the patch is
// covered by the four windows (p', q'), (p', q' + 1), (p' + 1, q'), (p' + 1, q' + 1) and by no other, so their gradient / argmax
// vectors are loaded ONCE for four outputs (the per-pixel form above requests 16 window vectors per patch, 12 of them twice).
// Summation order per pixel = the per-pixel form's (bit-identical results).
template <int DT>
__global__ __launch_bounds__(256) void maxpool_bwd_k3s2_patch_kernel(const unsigned short* __restrict__ dy,
                                                                     const unsigned char* __restrict__ amax,
                                                                     unsigned short* __restrict__ dx, int N, int H, int W, int C8,
                                                                     int P, int Q) {
  const unsigned total = (unsigned)N * P * Q * C8;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned c8 = i % (unsigned)C8;
    unsigned t = i / (unsigned)C8;
    const int q = (int)(t % (unsigned)Q); t /= (unsigned)Q;
    const int p = (int)(t % (unsigned)P);
    const int n = (int)(t / (unsigned)P);
    const bool vp1 = p + 1 < P, vq1 = q + 1 < Q;
    const int p1 = vp1 ? p + 1 : p, q1 = vq1 ? q + 1 : q;
    ushort8_t g[4];
    uint2_t am[4];
    const int pp[2] = {p, p1}, qq[2] = {q, q1};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const unsigned o = (((unsigned)n * P + pp[a]) * Q + qq[b]) * C8 + c8;
        g[a * 2 + b] = ((const ushort8_t*)dy)[o];
        am[a * 2 + b] = ((const uint2_t*)amax)[o];
      }
    // window (a, b) -> index a * 2 + b; term(win, code): the window's gradient where its argmax is `code`, else 0
    auto term = [&](int win, unsigned code, bool valid, float* acc) __attribute__((always_inline)) {
      const unsigned want = valid ? code : 0xFFu;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const unsigned c = (am[win][k >> 2] >> ((k & 3) * 8)) & 0xffu;
        if (c == want) acc[k] += up16<DT>(g[win][k]);
      }
    };
    auto store = [&](int dh, int dw, const float* acc) __attribute__((always_inline)) {
      ushort8_t o8;
#pragma unroll
      for (int k = 0; k < 8; ++k) o8[k] = dn16<DT>(acc[k]);
      ((ushort8_t*)dx)[((((unsigned)n * H + 2 * p + dh) * W) + 2 * q + dw) * C8 + c8] = o8;
    };
    float a00[8], a01[8], a10[8], a11[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a00[k] = 0.f; a01[k] = 0.f; a10[k] = 0.f; a11[k] = 0.f; }
    // (even, even): window (p, q) tap (1, 1)
    term(0, 4, true, a00);
    // (even, odd): windows (p, q + 1) tap (1, 0), then (p, q) tap (1, 2)
    term(1, 3, vq1, a01); term(0, 5, true, a01);
    // (odd, even): windows (p + 1, q) tap (0, 1), then (p, q) tap (2, 1)
    term(2, 1, vp1, a10); term(0, 7, true, a10);
    // (odd, odd): (p + 1, q + 1) tap (0, 0), (p + 1, q) tap (0, 2), (p, q + 1) tap (2, 0), (p, q) tap (2, 2)
    term(3, 0, vp1 && vq1, a11); term(2, 2, vp1, a11); term(1, 6, vq1, a11); term(0, 8, true, a11);
    store(0, 0, a00); store(0, 1, a01); store(1, 0, a10); store(1, 1, a11);
  }
}

// The stem's backward chain maxpool -> ReLU -> BatchNorm (models/resnet.py:318-322 backward) WITHOUT the pooling gradient's
// full-resolution tensor: both BatchNorm passes gather it from the pooled gradient + argmax themselves.  Stand-alone, the
// pooling backward writes dx (411 MB at batch 256), the reduction reads it and the apply pass reads it again; here each pass
// reads the pooled gradient (1/4 of the pixels) and the argmax bytes instead.  One thread per 2 x 2 patch of input pixels x 8
// channels (the patch is covered by four windows and no other: maxpool_bwd_k3s2_patch_kernel's walk, same summation order per
// pixel); the gathered value is rounded to the activation dtype where the stand-alone pass stored it, so the apply pass's
// output is bit-identical given the same dgamma / dbeta (the reduction's partial sums are grouped differently: fp32 rounding).
//   APPLY = false: partial[block][2][C] = (sum g, sum g xhat) over the block's patches, g = dx under the ReLU keep bits
//   APPLY = true:  dz = gamma rstd (g - dbeta / M - xhat dgamma / M)
template <int DT, bool APPLY>
__global__ __launch_bounds__(256) void pool_bn_bwd_kernel(const unsigned short* __restrict__ dy, const unsigned char* __restrict__ amax,
                                                          const unsigned char* __restrict__ mask, const unsigned short* __restrict__ x,
                                                          unsigned short* __restrict__ dz, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                          float* __restrict__ partial, int N, int H, int W, int C8, int P, int Q,
                                                          float inv_m) {
  __shared__ float red[APPLY ? 1 : 2][APPLY ? 1 : 256 * 8];
  const unsigned total = (unsigned)N * P * Q * C8;
  const unsigned first = blockIdx.x * blockDim.x + threadIdx.x;
  const int c0 = (int)(first % (unsigned)C8) * 8;              // (256 % C8 == 0: a thread keeps its channel group)
  float kmu[8], krs[8], ka[8], kb[8], kg[8], s0[8], s1[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    kmu[k] = mean[c0 + k]; krs[k] = rstd[c0 + k];
    s0[k] = 0.f; s1[k] = 0.f;
    if constexpr (APPLY) { ka[k] = gamma[c0 + k] * krs[k]; kb[k] = dbeta[c0 + k] * inv_m; kg[k] = dgamma[c0 + k] * inv_m; }
  }
  for (unsigned i = first; i < total; i += gridDim.x * blockDim.x) {
    const unsigned c8 = i % (unsigned)C8;
    unsigned t = i / (unsigned)C8;
    const int q = (int)(t % (unsigned)Q); t /= (unsigned)Q;
    const int p = (int)(t % (unsigned)P);
    const int n = (int)(t / (unsigned)P);
    const bool vp1 = p + 1 < P, vq1 = q + 1 < Q;
    const int pp[2] = {p, vp1 ? p + 1 : p}, qq[2] = {q, vq1 ? q + 1 : q};
    ushort8_t g[4], xv[4];
    uint2_t am[4];
    unsigned mb[4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const unsigned o = (((unsigned)n * P + pp[a]) * Q + qq[b]) * C8 + c8;
        g[a * 2 + b] = ((const ushort8_t*)dy)[o];
        am[a * 2 + b] = ((const uint2_t*)amax)[o];
        const unsigned ox = ((((unsigned)n * H + 2 * p + a) * W) + 2 * q + b) * C8 + c8;
        xv[a * 2 + b] = ((const ushort8_t*)x)[ox];
        mb[a * 2 + b] = mask[ox];
      }
    auto term = [&](int win, unsigned code, bool valid, float* acc) __attribute__((always_inline)) {
      const unsigned want = valid ? code : 0xFFu;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const unsigned c = (am[win][k >> 2] >> ((k & 3) * 8)) & 0xffu;
        if (c == want) acc[k] += up16<DT>(g[win][k]);
      }
    };
    float a00[8], a01[8], a10[8], a11[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a00[k] = 0.f; a01[k] = 0.f; a10[k] = 0.f; a11[k] = 0.f; }
    term(0, 4, true, a00);
    term(1, 3, vq1, a01); term(0, 5, true, a01);
    term(2, 1, vp1, a10); term(0, 7, true, a10);
    term(3, 0, vp1 && vq1, a11); term(2, 2, vp1, a11); term(1, 6, vq1, a11); term(0, 8, true, a11);
    auto pixel = [&](int px, const float* acc) __attribute__((always_inline)) {
      float xf[8], of[8];
      unpack8<DT>(xv[px], xf);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float gv = up16<DT>(dn16<DT>(acc[k]));                  // the value the stand-alone pooling backward stores
        if (!((mb[px] >> k) & 1u)) gv = 0.f;
        const float xh = (xf[k] - kmu[k]) * krs[k];
        if constexpr (APPLY) of[k] = ka[k] * (gv - kb[k] - xh * kg[k]);
        else { s0[k] += gv; s1[k] += gv * xh; }
      }
      if constexpr (APPLY)
        ((ushort8_t*)dz)[((((unsigned)n * H + 2 * p + (px >> 1)) * W) + 2 * q + (px & 1)) * C8 + c8] = pack8<DT>(of);
    };
    pixel(0, a00); pixel(1, a01); pixel(2, a10); pixel(3, a11);
  }
  if constexpr (!APPLY) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { red[0][threadIdx.x * 8 + k] = s0[k]; red[1][threadIdx.x * 8 + k] = s1[k]; }
    __syncthreads();
    if ((int)threadIdx.x < C8) {
      const int C = C8 * 8, rows = 256 / C8;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float t0 = 0.f, t1 = 0.f;
        for (int r = 0; r < rows; ++r) { t0 += red[0][(r * C8 + threadIdx.x) * 8 + k]; t1 += red[1][(r * C8 + threadIdx.x) * 8 + k]; }
        partial[((long long)blockIdx.x * 2 + 0) * C + threadIdx.x * 8 + k] = t0;
        partial[((long long)blockIdx.x * 2 + 1) * C + threadIdx.x * 8 + k] = t1;
      }
    }
  }
}

// Workgroups (= partial rows) of dle_pool_bn_bwd's reduction; 0 when the shape is outside its envelope.
static int pool_bn_grid(int N, int H, int W, int C) {
  if (N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || (C & 7)) return 0;
  const int C8 = C / 8;
  if (C8 > 256 || (C8 & (C8 - 1))) return 0;
  const long long total = (long long)N * (H / 2) * (W / 2) * C8;
  if ((long long)N * H * W * C8 >= 0x7FFFFFFFLL) return 0;
  long long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  return (int)g;
}

// fp32 workspace bytes of dle_pool_bn_bwd (0: shape outside the envelope -- run dle_maxpool_bwd + dle_bn_bwd_reduce / _apply).
extern "C" int64_t dle_pool_bn_bwd_workspace_bytes(int N, int H, int W, int C) {
  return (int64_t)pool_bn_grid(N, H, W, C) * 2 * C * 4;
}

// dz [N, H, W, C] = the BatchNorm (+ ReLU, keep bits relu_mask [N H W C / 8]) backward of the gradient that MaxPool2d(3, 2, 1)'s
// backward would scatter from dy [N, H/2, W/2, C] + argmax (dle_maxpool_fwd's encoding) -- dgamma / dbeta (fp32 [C], written)
// included; x = the BatchNorm's input (the stem convolution's output).  Replaces dle_maxpool_bwd + dle_bn_bwd_reduce +
// dle_bn_bwd_apply for the stem (models/resnet.py:318-322 backward).
extern "C" int dle_pool_bn_bwd(const void* dy, const void* argmax, const void* relu_mask, const void* x, void* dz, const float* mean,
                               const float* rstd, const float* gamma, float* dgamma, float* dbeta, int N, int H, int W, int C,
                               void* workspace, int64_t workspace_bytes, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "pool_bn_bwd: 16-bit activations only");
  const int grid = pool_bn_grid(N, H, W, C);
  DLE_CHECK_ARG(grid > 0, "pool_bn_bwd: shape outside the envelope (H, W even, C / 8 a power of two <= 256)");
  DLE_CHECK_ARG(dy && argmax && relu_mask && x && dz && mean && rstd && gamma && dgamma && dbeta && workspace, "pool_bn_bwd: null pointer");
  DLE_CHECK_ARG(workspace_bytes >= (int64_t)grid * 2 * C * 4, "pool_bn_bwd: workspace too small");
  const int P = H / 2, Q = W / 2, C8 = C / 8;
  const float inv_m = 1.0f / (float)((long long)N * H * W);
#define POOL_BN(DT)                                                                                                              \
  do {                                                                                                                           \
    hipLaunchKernelGGL((pool_bn_bwd_kernel<DT, false>), dim3(grid), dim3(256), 0, stream, (const unsigned short*)dy,             \
                       (const unsigned char*)argmax, (const unsigned char*)relu_mask, (const unsigned short*)x,                  \
                       (unsigned short*)nullptr, mean, rstd, gamma, (const float*)nullptr, (const float*)nullptr,               \
                       (float*)workspace, N, H, W, C8, P, Q, inv_m);                                                             \
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((C + 7) / 8), dim3(256), 0, stream, (const float*)workspace, grid, C, dgamma,  \
                       dbeta, 0);                                                                                                \
    hipLaunchKernelGGL((pool_bn_bwd_kernel<DT, true>), dim3(grid), dim3(256), 0, stream, (const unsigned short*)dy,              \
                       (const unsigned char*)argmax, (const unsigned char*)relu_mask, (const unsigned short*)x,                  \
                       (unsigned short*)dz, mean, rstd, gamma, (const float*)dgamma, (const float*)dbeta, (float*)nullptr, N,    \
                       H, W, C8, P, Q, inv_m);                                                                                   \
  } while (0)
  if (dtype == DLE_F16) POOL_BN(DLE_F16); else POOL_BN(DLE_BF16);
#undef POOL_BN
  DLE_LAUNCH_CHECK();
  return 0;
}

// BatchNorm-apply + ReLU + MaxPool2d(3, 2, 1) in ONE pass (the ResNet stem: bn1 -> relu -> maxpool, models/resnet.py:318-322):
// the 16-bit activation between them (411 MB at batch 256, written once and read once) never exists.  One thread per pooled
// pixel x 8 channels: nine (clamped, masked) loads of t, y = relu(t * sc + sh) rounded to 16 bits exactly as dle_bn_fwd_apply stores
// it, first-maximum argmax in window-scan order (dle_maxpool_fwd's rule), and the ReLU keep bits of the 2 x 2 input pixels
// (2 p, 2 q) .. (2 p + 1, 2 q + 1) = taps (1..2, 1..2), which this window alone owns.  Outputs are bit-identical to the two launches.
template <int DT>
__global__ __launch_bounds__(256) void bn_relu_maxpool_k3s2_kernel(const unsigned short* __restrict__ x,
                                                                   unsigned short* __restrict__ y, unsigned char* __restrict__ amax,
                                                                   unsigned char* __restrict__ mask, const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, int N, int H, int W, int C8, int P,
                                                                   int Q) {
  const unsigned total = (unsigned)N * P * Q * C8;
  const unsigned first = blockIdx.x * blockDim.x + threadIdx.x;
  if (first >= total) return;
  float sc[8], sh[8];
  {
    const int c0 = (int)(first % (unsigned)C8) * 8;            // (the launcher makes the grid stride a multiple of C8)
#pragma unroll
    for (int k = 0; k < 8; ++k) { sc[k] = rstd[c0 + k] * gamma[c0 + k]; sh[k] = beta[c0 + k] - mean[c0 + k] * sc[k]; }
  }
  for (unsigned i = first; i < total; i += gridDim.x * blockDim.x) {
    const unsigned c8 = i % (unsigned)C8;
    unsigned t = i / (unsigned)C8;
    const int q = (int)(t % (unsigned)Q); t /= (unsigned)Q;
    const int p = (int)(t % (unsigned)P);
    const int n = (int)(t / (unsigned)P);
    ushort8_t v[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s2 = 0; s2 < 3; ++s2) {
        int h = 2 * p - 1 + r, w = 2 * q - 1 + s2;
        h = h < 0 ? 0 : (h >= H ? H - 1 : h);
        w = w < 0 ? 0 : (w >= W ? W - 1 : w);
        v[r * 3 + s2] = ((const ushort8_t*)x)[(((unsigned)n * H + h) * W + w) * C8 + c8];
      }
    float best[8];
    unsigned bi[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { best[k] = -INFINITY; bi[k] = 0; }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s2 = 0; s2 < 3; ++s2) {
        const int h = 2 * p - 1 + r, w = 2 * q - 1 + s2;
        const bool valid = h >= 0 && h < H && w >= 0 && w < W;
        float xf[8];
        unpack8<DT>(v[r * 3 + s2], xf);
        unsigned bits = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float f = xf[k] * sc[k] + sh[k];
          f = f > 0.f ? f : 0.f;
          bits |= (f > 0.f ? 1u : 0u) << k;
          const float fr = up16<DT>(dn16<DT>(f));              // the value the stand-alone apply pass stores
          if (valid && (fr > best[k] || fr != fr)) { best[k] = fr; bi[k] = (unsigned)(r * 3 + s2); }
        }
        if (r >= 1 && s2 >= 1 && h < H && w < W) mask[(((unsigned)n * H + h) * W + w) * C8 + c8] = (unsigned char)bits;
      }
    ushort8_t o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = dn16<DT>(best[k]);
    ((ushort8_t*)y)[i] = o;
    uint2_t packed;
    packed[0] = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
    packed[1] = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
    ((uint2_t*)amax)[i] = packed;
  }
}

// y [N, H/2, W/2, C] = maxpool3x3/2/pad1(relu(bn(x))), argmax (dle_maxpool_fwd's encoding), relu_mask [N*H*W*C/8] (dle_bn_fwd_apply's
// encoding).  H, W even.
extern "C" int dle_bn_relu_maxpool_fwd(const void* x, void* y, void* argmax, void* relu_mask, const float* mean, const float* rstd,
                                       const float* gamma, const float* beta, int N, int H, int W, int C, int dtype,
                                       hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "bn_relu_maxpool_fwd: 16-bit activations only");
  DLE_CHECK_ARG(N >= 0 && H > 0 && W > 0 && (H & 1) == 0 && (W & 1) == 0 && C > 0 && C % 8 == 0, "bn_relu_maxpool_fwd: bad shape (H, W even, C % 8 == 0)");
  if (N == 0) return 0;
  DLE_CHECK_ARG(x && y && argmax && relu_mask && mean && rstd && gamma && beta, "bn_relu_maxpool_fwd: null pointer");
  DLE_CHECK_ARG((long long)N * H * W * (C / 8) < 0x7FFFFFFFLL, "bn_relu_maxpool_fwd: tensor too large");
  const int P = H / 2, Q = W / 2, C8 = C / 8;
  long long g = ((long long)N * P * Q * C8 + 255) / 256;
  if (g > 8192) g = 8192;
  while (((g * 256) % C8) != 0) --g;                          // a thread keeps its channel group over its whole walk
  if (g < 1) g = 1;
  DLE_CHECK_ARG(((g * 256) % C8) == 0, "bn_relu_maxpool_fwd: channel count %d does not divide the grid stride", C);
  if (dtype == DLE_F16) hipLaunchKernelGGL(bn_relu_maxpool_k3s2_kernel<DLE_F16>, dim3((unsigned)g), dim3(256), 0, stream, (const unsigned short*)x, (unsigned short*)y, (unsigned char*)argmax, (unsigned char*)relu_mask, mean, rstd, gamma, beta, N, H, W, C8, P, Q);
  else hipLaunchKernelGGL(bn_relu_maxpool_k3s2_kernel<DLE_BF16>, dim3((unsigned)g), dim3(256), 0, stream, (const unsigned short*)x, (unsigned short*)y, (unsigned char*)argmax, (unsigned char*)relu_mask, mean, rstd, gamma, beta, N, H, W, C8, P, Q);
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_maxpool_bwd(const void* dy, const void* argmax, void* dx, int N, int H, int W, int C, int ksize,
                               int stride, int pad, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "maxpool_bwd: 16-bit activations only");
  DLE_CHECK_ARG(C % 8 == 0 && ksize > 0 && stride > 0, "maxpool_bwd: bad geometry");
  if (N == 0) return 0;
  DLE_CHECK_ARG(dy && dx && argmax, "maxpool_bwd: null pointer");
  const int P = (H + 2 * pad - ksize) / stride + 1, Q = (W + 2 * pad - ksize) / stride + 1;
  const int grid = cn_grid((long long)N * H * W * (C / 8), 256, 8192);
  static const int patch_mode = getenv("DLE_MAXPOOL_BWD_PATCH") ? atoi(getenv("DLE_MAXPOOL_BWD_PATCH")) : 1;
  if (patch_mode && ksize == 3 && stride == 2 && pad == 1 && (H & 1) == 0 && (W & 1) == 0 && (long long)N * H * W * (C / 8) < 0x7FFFFFFFLL) {
    const int gp = cn_grid((long long)N * P * Q * (C / 8), 256, 8192);
    if (dtype == DLE_F16) hipLaunchKernelGGL(maxpool_bwd_k3s2_patch_kernel<DLE_F16>, dim3(gp), dim3(256), 0, stream, (const unsigned short*)dy, (const unsigned char*)argmax, (unsigned short*)dx, N, H, W, C / 8, P, Q);
    else hipLaunchKernelGGL(maxpool_bwd_k3s2_patch_kernel<DLE_BF16>, dim3(gp), dim3(256), 0, stream, (const unsigned short*)dy, (const unsigned char*)argmax, (unsigned short*)dx, N, H, W, C / 8, P, Q);
    DLE_LAUNCH_CHECK();
    return 0;
  }
  if (ksize == 3 && stride == 2 && pad == 1 && (long long)N * H * W * (C / 8) < 0x7FFFFFFFLL && P >= 1 && Q >= 1) {
    if (dtype == DLE_F16) hipLaunchKernelGGL(maxpool_bwd_k3s2_kernel<DLE_F16>, dim3(grid), dim3(256), 0, stream, (const unsigned short*)dy, (const unsigned char*)argmax, (unsigned short*)dx, N, H, W, C / 8, P, Q);
    else hipLaunchKernelGGL(maxpool_bwd_k3s2_kernel<DLE_BF16>, dim3(grid), dim3(256), 0, stream, (const unsigned short*)dy, (const unsigned char*)argmax, (unsigned short*)dx, N, H, W, C / 8, P, Q);
    DLE_LAUNCH_CHECK();
    return 0;
  }
  if (dtype == DLE_F16) hipLaunchKernelGGL(maxpool_bwd_kernel<DLE_F16>, dim3(grid), dim3(256), 0, stream, (const unsigned short*)dy, (const unsigned char*)argmax, (unsigned short*)dx, N, H, W, C / 8, P, Q, ksize, stride, pad);
  else hipLaunchKernelGGL(maxpool_bwd_kernel<DLE_BF16>, dim3(grid), dim3(256), 0, stream, (const unsigned short*)dy, (const unsigned char*)argmax, (unsigned short*)dx, N, H, W, C / 8, P, Q, ksize, stride, pad);
  DLE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- global average pooling [N, HW, C] <-> [N, C]
template <int DT>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const unsigned short* __restrict__ x,
                                                          unsigned short* __restrict__ y, long long N, int HW, int C8) {
  const long long total = N * C8;
  const float inv = 1.0f / (float)HW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / C8;
    const int c8 = (int)(i - n * C8);
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    for (int p = 0; p < HW; ++p) {
      const ushort8_t v = ((const ushort8_t*)x)[(n * HW + p) * C8 + c8];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += up16<DT>(v[k]);
    }
    ushort8_t o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = dn16<DT>(acc[k] * inv);
    ((ushort8_t*)y)[i] = o;
  }
}

template <int DT>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const unsigned short* __restrict__ dy,
                                                          unsigned short* __restrict__ dx, long long N, int HW, int C8) {
  const long long total = N * HW * C8;
  const float inv = 1.0f / (float)HW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % C8);
    const long long n = i / ((long long)HW * C8);
    const ushort8_t g = ((const ushort8_t*)dy)[n * C8 + c8];
    ushort8_t o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = dn16<DT>(up16<DT>(g[k]) * inv);
    ((ushort8_t*)dx)[i] = o;
  }
}

extern "C" int dle_avgpool_fwd(const void* x, void* y, int64_t N, int HW, int C, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "avgpool_fwd: 16-bit activations only");
  DLE_CHECK_ARG(C % 8 == 0 && HW > 0, "avgpool_fwd: bad shape");
  if (N == 0) return 0;
  DLE_CHECK_ARG(x && y, "avgpool_fwd: null pointer");
  const int grid = cn_grid(N * (C / 8), 256);
  if (dtype == DLE_F16) hipLaunchKernelGGL(avgpool_fwd_kernel<DLE_F16>, dim3(grid), dim3(256), 0, stream, (const unsigned short*)x, (unsigned short*)y, (long long)N, HW, C / 8);
  else hipLaunchKernelGGL(avgpool_fwd_kernel<DLE_BF16>, dim3(grid), dim3(256), 0, stream, (const unsigned short*)x, (unsigned short*)y, (long long)N, HW, C / 8);
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_avgpool_bwd(const void* dy, void* dx, int64_t N, int HW, int C, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "avgpool_bwd: 16-bit activations only");
  DLE_CHECK_ARG(C % 8 == 0 && HW > 0, "avgpool_bwd: bad shape");
  if (N == 0) return 0;
  DLE_CHECK_ARG(dy && dx, "avgpool_bwd: null pointer");
  const int grid = cn_grid(N * HW * (C / 8), 256);
  if (dtype == DLE_F16) hipLaunchKernelGGL(avgpool_bwd_kernel<DLE_F16>, dim3(grid), dim3(256), 0, stream, (const unsigned short*)dy, (unsigned short*)dx, (long long)N, HW, C / 8);
  else hipLaunchKernelGGL(avgpool_bwd_kernel<DLE_BF16>, dim3(grid), dim3(256), 0, stream, (const unsigned short*)dy, (unsigned short*)dx, (long long)N, HW, C / 8);
  DLE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- softmax cross entropy with label smoothing
// logits fp32 [N, ld] (first `classes` columns valid); one wavefront per row.
//   loss_row = (1 - s) * (lse - x[t]) + s * (lse - mean(x));   loss = mean over rows with t != ignore_index
//   dlogits  = (softmax - (1 - s) * onehot(t) - s / classes) * (*grad_scale) / n_valid   (16-bit or fp32)
// (LabelSmoothing.forward smoothing.py:33-40; s = 0 gives nn.CrossEntropyLoss, ignore_index as in
//  BertPretrainingCriterion, LanguageModeling/BERT/run_pretraining.py:75-95.)
template <int ODT>
__global__ __launch_bounds__(256) void softmax_xent_kernel(const float* __restrict__ logits,
                                                           const long long* __restrict__ target,
                                                           float* __restrict__ loss_sum, void* __restrict__ dlogits,
                                                           const float* __restrict__ grad_scale,
                                                           const int* __restrict__ n_valid_dev, long long rows,
                                                           int classes, long long ld, long long ld_out, float smoothing,
                                                           long long ignore_index) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* x = logits + row * ld;
  const long long t = target[row];
  const bool valid = t != ignore_index;
  float mx = -INFINITY;
  for (int c = lane; c < classes; c += 64) mx = fmaxf(mx, x[c]);
  mx = wave_max(mx);
  float se = 0.f, sx = 0.f;
  for (int c = lane; c < classes; c += 64) { se += __expf(x[c] - mx); sx += x[c]; }
  se = wave_sum(se);
  sx = wave_sum(sx);
  const float lse = mx + __logf(se);
  if (lane == 0 && valid) {
    const float nll = lse - x[t];
    const float smooth = lse - sx / (float)classes;
    unsafeAtomicAdd(loss_sum, (1.f - smoothing) * nll + smoothing * smooth);
  }
  if (dlogits) {
    const float nv = (float)(*n_valid_dev > 0 ? *n_valid_dev : 1);
    const float gs = valid ? (grad_scale ? *grad_scale : 1.0f) / nv : 0.f;
    const float inv_se = 1.0f / se, sm = smoothing / (float)classes;
    for (int c = lane; c < ld_out; c += 64) {
      float g = 0.f;
      if (c < classes) g = (__expf(x[c] - mx) * inv_se - (c == t ? 1.f - smoothing : 0.f) - sm) * gs;
      if (ODT == DLE_F32) ((float*)dlogits)[row * ld_out + c] = g;
      else ((unsigned short*)dlogits)[row * ld_out + c] = ODT == DLE_F16 ? Elem<DLE_F16>::from_f32(g) : Elem<DLE_BF16>::from_f32(g);
    }
  }
}

// Wide rows (the MLM head: 30522 classes, 122 KB of fp32 logits per row): ONE workgroup per row, 16-byte loads, four of them
// in flight per lane from unconditional (clamped) addresses, max and sum of exponentials in ONE pass (running maximum with
// rescaling), the gradient pass re-reads the row from L2.  The wave-per-row form above reads a row three times with 4-byte
// loads, one dependent load at a time: 774 us for the 5120 x 30528 logits of a BERT-Large batch (0.8 TB/s).
template <int ODT>
__global__ __launch_bounds__(256) void softmax_xent_wide_kernel(const float* __restrict__ logits,
                                                                const long long* __restrict__ target,
                                                                float* __restrict__ loss_sum, void* __restrict__ dlogits,
                                                                const float* __restrict__ grad_scale,
                                                                const int* __restrict__ n_valid_dev, long long rows,
                                                                int classes, long long ld, long long ld_out, float smoothing,
                                                                long long ignore_index) {
  __shared__ float red[3][4];
  const long long row = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float4_t* x4 = (const float4_t*)(logits + row * ld);
  const int n4 = (classes + 3) >> 2;
  const long long t = target[row];
  const bool valid = t != ignore_index;
  float m = -INFINITY, se = 0.f, sx = 0.f;
  for (int base = tid; base < n4; base += 1024) {
    float4_t v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = base + 256 * u; v[u] = x4[i < n4 ? i : n4 - 1]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + 256 * u;
      if (i < n4) {
        float xv[4];
        float mx = m;
#pragma unroll
        for (int k = 0; k < 4; ++k) { xv[k] = (4 * i + k < classes) ? v[u][k] : -INFINITY; mx = fmaxf(mx, xv[k]); }
        se *= __expf(m - mx);              // (m = -inf, se = 0 on the first group: exp(-inf) = 0)
#pragma unroll
        for (int k = 0; k < 4; ++k) { se += __expf(xv[k] - mx); sx += (4 * i + k < classes) ? xv[k] : 0.f; }
        m = mx;
      }
    }
  }
  const float wm = wave_max(m);
  se = wave_sum(se * __expf(m - wm));
  sx = wave_sum(sx);
  if (lane == 0) { red[0][w] = wm; red[1][w] = se; red[2][w] = sx; }
  __syncthreads();
  const float mx = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
  float tot = 0.f, totx = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) { tot += red[1][q] * __expf(red[0][q] - mx); totx += red[2][q]; }
  const float lse = mx + __logf(tot);
  if (tid == 0 && valid) {
    const float nll = lse - logits[row * ld + t];
    const float smooth = lse - totx / (float)classes;
    unsafeAtomicAdd(loss_sum, (1.f - smoothing) * nll + smoothing * smooth);
  }
  if (dlogits) {
    const float nv = (float)(*n_valid_dev > 0 ? *n_valid_dev : 1);
    const float gs = valid ? (grad_scale ? *grad_scale : 1.0f) / nv : 0.f;
    const float inv_se = 1.0f / tot, sm = smoothing / (float)classes;
    const int n4o = (int)(ld_out >> 2);
    for (int i = tid; i < n4o; i += 256) {
      const float4_t v = x4[i < n4 ? i : n4 - 1];
      float g[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = 4 * i + k;
        g[k] = c < classes ? (__expf(v[k] - mx) * inv_se - (c == t ? 1.f - smoothing : 0.f) - sm) * gs : 0.f;
      }
      if (ODT == DLE_F32) {
        const float4_t o = {g[0], g[1], g[2], g[3]};
        *(float4_t*)((float*)dlogits + row * ld_out + 4 * i) = o;
      } else {
        typedef __attribute__((ext_vector_type(4))) unsigned short ushort4v_t;
        ushort4v_t o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = ODT == DLE_F16 ? Elem<DLE_F16>::from_f32(g[k]) : Elem<DLE_BF16>::from_f32(g[k]);
        *(ushort4v_t*)((unsigned short*)dlogits + row * ld_out + 4 * i) = o;
      }
    }
  }
}

__global__ void count_valid_kernel(const long long* __restrict__ target, long long rows, long long ignore_index,
                                   int* __restrict__ n_valid) {
  __shared__ float red[16];
  float c = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (long long)gridDim.x * blockDim.x)
    c += target[i] != ignore_index ? 1.f : 0.f;
  c = block_sum(c, red);
  if (threadIdx.x == 0) atomicAdd(n_valid, (int)(c + 0.5f));
}

__global__ void xent_finish_kernel(float* loss_sum, const int* n_valid) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *loss_sum = *loss_sum / (float)(*n_valid > 0 ? *n_valid : 1);
}

// scratch: int32[1] device word (valid-row counter).  loss_out[0] = mean loss.
extern "C" int dle_softmax_xent(const float* logits, const int64_t* target, float* loss_out, void* dlogits,
                                const float* grad_scale_dev, int* scratch, int64_t rows, int classes, int64_t ld,
                                int64_t ld_out, float smoothing, int64_t ignore_index, int dlogits_dtype,
                                hipStream_t stream) {
  DLE_CHECK_ARG(loss_out && scratch, "softmax_xent: null loss/scratch pointer");
  hipError_t e = hipMemsetAsync(loss_out, 0, 4, stream);
  if (e == hipSuccess) e = hipMemsetAsync(scratch, 0, 4, stream);
  if (e != hipSuccess) { dle_set_error("softmax_xent memset: %s", hipGetErrorString(e)); return (int)e; }
  if (rows == 0) return 0;
  DLE_CHECK_ARG(logits && target && classes > 0 && ld >= classes, "softmax_xent: bad arguments");
  DLE_CHECK_ARG(!dlogits || ld_out >= classes, "softmax_xent: dlogits row too short");
  hipLaunchKernelGGL(count_valid_kernel, dim3(cn_grid(rows, 256, 256)), dim3(256), 0, stream, (const long long*)target,
                     (long long)rows, (long long)ignore_index, scratch);
  DLE_LAUNCH_CHECK();
  // wide rows: a workgroup per row with 16-byte accesses (rows, output rows and bases 16-byte aligned)
  const bool wide = classes >= 4096 && (ld & 3) == 0 && (((uintptr_t)logits) & 15) == 0 &&
                    (!dlogits || ((ld_out & 3) == 0 && (((uintptr_t)dlogits) & 15) == 0));
  dim3 grid(wide ? (unsigned)rows : (unsigned)((rows + 3) / 4)), block(256);
#define GO(ODT) do { if (wide) hipLaunchKernelGGL(softmax_xent_wide_kernel<ODT>, grid, block, 0, stream, logits, (const long long*)target, loss_out, dlogits, grad_scale_dev, (const int*)scratch, (long long)rows, classes, (long long)ld, (long long)ld_out, smoothing, (long long)ignore_index); \
  else hipLaunchKernelGGL(softmax_xent_kernel<ODT>, grid, block, 0, stream, logits, (const long long*)target, loss_out, dlogits, grad_scale_dev, (const int*)scratch, (long long)rows, classes, (long long)ld, (long long)ld_out, smoothing, (long long)ignore_index); } while (0)
  if (dlogits_dtype == DLE_F32) GO(DLE_F32);
  else if (dlogits_dtype == DLE_F16) GO(DLE_F16);
  else if (dlogits_dtype == DLE_BF16) GO(DLE_BF16);
  else { dle_set_error("softmax_xent: bad dlogits dtype %d", dlogits_dtype); return -1; }
#undef GO
  DLE_LAUNCH_CHECK();
  hipLaunchKernelGGL(xent_finish_kernel, dim3(1), dim3(64), 0, stream, loss_out, (const int*)scratch);
  DLE_LAUNCH_CHECK();
  return 0;
}
