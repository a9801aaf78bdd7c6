// WaveGlow training-step kernels for gfx950 (first correct path).
//
// Replace, around the dense contractions that go through dle_gemm (paths relative to
// /root/reference/PyTorch/SpeechSynthesis/Tacotron2/):
//   waveglow/model.py:34-41    fused_add_tanh_sigmoid_multiply          -> wg_gate_{fwd,bwd}
//   waveglow/model.py:44-85    Invertible1x1Conv (c <= 8 channels)       -> wg_invconv_{fwd,bwd}, wg_logdet_inv
//   waveglow/model.py:87-157   WN: weight_norm'd Conv1d (dilated k = 3)  -> wg_weight_norm_{fwd,bwd}, wg_taps / wg_taps_bwd
//   waveglow/model.py:160-231  WaveGlow.forward: ConvTranspose1d(1024, stride 256), grouping, affine coupling
//                                                                        -> wg_upsample_weight*, wg_taps, wg_coupling_{fwd,bwd}
//   waveglow/loss_function.py:30-48  WaveGlowLoss                        -> wg_loss, wg_dz_init
//
// Layout: everything is channels-last.  A time series [B, C, T] of the reference is the matrix [B*T, C] here (row = one
// time step), so a Conv1d is a GEMM over rows: a 1x1 convolution directly, a dilated k-tap convolution over the row-gathered
// matrix [B*T, k*C] that wg_taps builds (zero rows outside a sample).  The flow state ("audio" grouped by n_group = 8) is
// an fp32 matrix [M, 8], M = B * T / 8; channels that were emitted early stay where they are, so the last state IS z.
// The upsampled, grouped spectrogram [B, 640, T/8] of the reference is the time-major upsampling output [B, T, 80] read as
// [M, 640] with its channels in (g, mel) order instead of (mel, g) -- the cond-layer weights are laid out to match -- and one
// GEMM over (frame tap, mel) x (phase, mel) produces it (see wg_upsample_weight).
// These are small HBM-streaming kernels (M ~ 10^4 rows at the reference's batch 10 x 8000 samples): 16 B per lane where the
// row width allows, one thread per row for the 8-wide flow state.
#include "common.h"

#define WG_BLOCK 256

static int wg_grid(long long items, int per_block = WG_BLOCK) {
  long long g = (items + per_block - 1) / per_block;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

// ---------------------------------------------------------------------------------------------------------------
// Row gather for k-tap (dilated) 1-D convolutions: col[b, t, k*C + c] = x[b, t + (k - left) * dil, c], zero outside
// [0, T).  16 bytes (8 channels) per lane; C % 8 == 0; x may be a column slice of a wider matrix (row stride ld_x).
__global__ __launch_bounds__(WG_BLOCK) void wg_taps_kernel(const uint4_t* __restrict__ x, uint4_t* __restrict__ col,
                                                           int B, int T, int C8, int ntaps, int dil, int left, long long ldx8) {
  const long long total = (long long)B * T * ntaps * C8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C8);
    long long r = i / C8;
    const int k = (int)(r % ntaps);
    r /= ntaps;                                   // r = b * T + t
    const int t = (int)(r % T);
    const long long b = r / T;
    const long long ts = (long long)t + (long long)(k - left) * dil;
    uint4_t v = {0u, 0u, 0u, 0u};
    if (ts >= 0 && ts < T) v = x[(b * T + ts) * ldx8 + c];
    col[i] = v;
  }
}

// Transpose of the gather: dx[b, t, c] = sum_k dcol[b, t - (k - left) * dil, k*C + c] (+ addend[b, t, c]).
// A gather again (no atomics): every output element owns its <= ntaps sources.  dx may alias addend.
template <int DT>
__global__ __launch_bounds__(WG_BLOCK) void wg_taps_bwd_kernel(const uint4_t* __restrict__ dcol, const unsigned short* addend,
                                                               unsigned short* dx, int B, int T, int C8, int ntaps,
                                                               int dil, int left, long long ld_add, long long ld_dx) {
  const long long total = (long long)B * T * C8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C8);
    const long long r = i / C8;                   // b * T + t
    const int t = (int)(r % T);
    const long long b = r / T;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (addend) {
      float a[8];
      unpack8<DT>(*(const ushort8_t*)(addend + r * ld_add + (long long)c * 8), a);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = a[e];
    }
    for (int k = 0; k < ntaps; ++k) {
      const long long ts = (long long)t - (long long)(k - left) * dil;
      if (ts < 0 || ts >= T) continue;
      float a[8];
      unpack8<DT>(__builtin_bit_cast(ushort8_t, dcol[((b * T + ts) * ntaps + k) * C8 + c]), a);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += a[e];
    }
    *(ushort8_t*)(dx + r * ld_dx + (long long)c * 8) = pack8<DT>(acc);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// acts = tanh(s[:, :nc]) * sigmoid(s[:, nc:2nc])   (s = in_layer(audio) + cond_layer(spect), already summed by the GEMM
// epilogue); 8 channels per lane.
template <int DT>
__global__ __launch_bounds__(WG_BLOCK) void wg_gate_fwd_kernel(const unsigned short* __restrict__ s,
                                                               unsigned short* __restrict__ acts, long long M, int nc8,
                                                               long long ld_s) {
  const long long total = M * nc8;
  const int nc = nc8 * 8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % nc8);
    const long long m = i / nc8;
    float a[8], b[8], o[8];
    unpack8<DT>(*(const ushort8_t*)(s + m * ld_s + (long long)c * 8), a);
    unpack8<DT>(*(const ushort8_t*)(s + m * ld_s + nc + (long long)c * 8), b);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = tanhf(a[e]) * (1.0f / (1.0f + __expf(-b[e])));
    *(ushort8_t*)(acts + m * nc + (long long)c * 8) = pack8<DT>(o);
  }
}

// ds[:, :nc] = dacts * sig * (1 - tanh^2);  ds[:, nc:] = dacts * tanh * sig * (1 - sig)
template <int DT>
__global__ __launch_bounds__(WG_BLOCK) void wg_gate_bwd_kernel(const unsigned short* __restrict__ dacts,
                                                               const unsigned short* __restrict__ s,
                                                               unsigned short* __restrict__ ds, long long M, int nc8,
                                                               long long ld_s, long long ld_ds) {
  const long long total = M * nc8;
  const int nc = nc8 * 8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % nc8);
    const long long m = i / nc8;
    float a[8], b[8], g[8], da[8], db[8];
    unpack8<DT>(*(const ushort8_t*)(s + m * ld_s + (long long)c * 8), a);
    unpack8<DT>(*(const ushort8_t*)(s + m * ld_s + nc + (long long)c * 8), b);
    unpack8<DT>(*(const ushort8_t*)(dacts + m * nc + (long long)c * 8), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float th = tanhf(a[e]);
      const float sg = 1.0f / (1.0f + __expf(-b[e]));
      da[e] = g[e] * sg * (1.0f - th * th);
      db[e] = g[e] * th * sg * (1.0f - sg);
    }
    *(ushort8_t*)(ds + m * ld_ds + (long long)c * 8) = pack8<DT>(da);
    *(ushort8_t*)(ds + m * ld_ds + nc + (long long)c * 8) = pack8<DT>(db);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Invertible 1x1 convolution on the active channels [off, 8), off = 8 - c, of the fp32 flow state (one thread per row).
// The c x c matrix is embedded in an 8 x 8 one, W8 = diag(I_off, W), so that every row is one fully unrolled 8 x 8 product
// (no run-time register indexing):  y = W8 x  leaves the early-output channels [0, off) untouched.
// Also writes the first c/2 mixed channels as a zero-padded 16-bit [M, 8] operand for the WN `start` GEMM.
__device__ __forceinline__ void wg_embed_w8(const float* __restrict__ W, float* w8, int c) {
  const int off = 8 - c;
  if (threadIdx.x < 64) {
    const int j = threadIdx.x >> 3, i = threadIdx.x & 7;
    float v = (i == j) ? 1.0f : 0.0f;
    if (j >= off && i >= off) v = W[(j - off) * c + (i - off)];
    w8[threadIdx.x] = v;
  }
  __syncthreads();
}
__device__ __forceinline__ void wg_ld8(const float* p, float* v) {
  const float4_t lo = *(const float4_t*)p, hi = *(const float4_t*)(p + 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
}
__device__ __forceinline__ void wg_st8(float* p, const float* v) {
  float4_t lo, hi;
#pragma unroll
  for (int e = 0; e < 4; ++e) { lo[e] = v[e]; hi[e] = v[4 + e]; }
  *(float4_t*)p = lo;
  *(float4_t*)(p + 4) = hi;
}

template <int DT>
__global__ __launch_bounds__(WG_BLOCK) void wg_invconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                  float* __restrict__ y, unsigned short* __restrict__ a0,
                                                                  long long M, int c) {
  __shared__ float w8[64];
  wg_embed_w8(W, w8, c);
  const int off = 8 - c, nh = c >> 1;
  for (long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long long)gridDim.x * blockDim.x) {
    float xi[8], yo[8];
    wg_ld8(x + m * 8, xi);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += w8[j * 8 + i] * xi[i];
      yo[j] = acc;
    }
    wg_st8(y + m * 8, yo);
    if (a0) {
      float h[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) if (e < nh && q == off + e) v = yo[q];
        h[e] = v;
      }
      *(ushort8_t*)(a0 + m * 8) = pack8<DT>(h);
    }
  }
}

// Backward of the mixing: g = dy (+ da0 on the first c/2 active channels: the gradient that came back through WN);
//   dx = W8^T g;   dW8_partial[block][j*8 + i] = sum over the block's rows of g[j] * x[i]   (the finishing pass reads the
//   active c x c corner)
__global__ __launch_bounds__(WG_BLOCK) void wg_invconv_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ da0,
                                                                  const float* __restrict__ x, const float* __restrict__ W,
                                                                  float* __restrict__ dx, float* __restrict__ dW_partial,
                                                                  long long M, int c) {
  __shared__ float w8[64];
  __shared__ float red[WG_BLOCK / 64][64];
  wg_embed_w8(W, w8, c);
  const int off = 8 - c, nh = c >> 1;
  float acc[64];
#pragma unroll
  for (int e = 0; e < 64; ++e) acc[e] = 0.f;
  for (long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long long)gridDim.x * blockDim.x) {
    float g[8], xi[8], o[8];
    wg_ld8(dy + m * 8, g);
    wg_ld8(x + m * 8, xi);
    if (da0) {
      float d[8];
      wg_ld8(da0 + m * 8, d);
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e < nh && q == off + e) g[q] += d[e];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) a += w8[j * 8 + i] * g[j];
      o[i] = a;
    }
    wg_st8(dx + m * 8, o);
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[j * 8 + i] += g[j] * xi[i];
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int e = 0; e < 64; ++e) {
    const float s = wave_sum(acc[e]);
    if (lane == 0) red[wave][e] = s;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    float s = 0.f;
    for (int q = 0; q < WG_BLOCK / 64; ++q) s += red[q][threadIdx.x];
    dW_partial[(long long)blockIdx.x * 64 + threadIdx.x] = s;
  }
}

// dW[j, i] = sum_blocks partial[block][(off + j)*8 + off + i] - scale * coef * WinvT[j, i]   (d/dW of -M log det W / (M * 8))
__global__ void wg_invconv_wfinish_kernel(const float* __restrict__ partial, int G, const float* __restrict__ WinvT,
                                          float* __restrict__ dW, const float* __restrict__ scale, float coef, int c) {
  const int t = threadIdx.x;
  if (t >= c * c) return;
  const int off = 8 - c;
  const int j = t / c, i = t - j * c;
  float s = 0.f;
  for (int g = 0; g < G; ++g) s += partial[(long long)g * 64 + (off + j) * 8 + off + i];
  const float sc = scale ? *scale : 1.0f;
  dW[t] = s - sc * coef * WinvT[t];
}

// log |det W| and W^{-T} of one c x c matrix (c <= 8), Gauss-Jordan with partial pivoting in fp64 on one lane.
// (torch.logdet(W) returns nan for det < 0; the sign is reported so the host side can refuse such a matrix.)
__device__ __forceinline__ void wg_logdet_inv_one(const float* __restrict__ W, float* __restrict__ logdet,
                                                  float* __restrict__ WinvT, float* __restrict__ sign_out, int c) {
  double a[8][16];
  for (int i = 0; i < c; ++i)
    for (int j = 0; j < c; ++j) { a[i][j] = (double)W[i * c + j]; a[i][c + j] = (i == j) ? 1.0 : 0.0; }
  double ld = 0.0, sign = 1.0;
  for (int p = 0; p < c; ++p) {
    int best = p;
    double bv = fabs(a[p][p]);
    for (int r = p + 1; r < c; ++r) if (fabs(a[r][p]) > bv) { bv = fabs(a[r][p]); best = r; }
    if (best != p) {
      for (int q = 0; q < 2 * c; ++q) { const double tmp = a[p][q]; a[p][q] = a[best][q]; a[best][q] = tmp; }
      sign = -sign;
    }
    const double piv = a[p][p];
    if (piv < 0.0) sign = -sign;
    ld += log(fabs(piv));
    const double ip = 1.0 / piv;
    for (int q = 0; q < 2 * c; ++q) a[p][q] *= ip;
    for (int r = 0; r < c; ++r) {
      if (r == p) continue;
      const double f = a[r][p];
      if (f == 0.0) continue;
      for (int q = 0; q < 2 * c; ++q) a[r][q] -= f * a[p][q];
    }
  }
  *logdet = (float)ld;
  if (sign_out) *sign_out = (float)sign;
  for (int i = 0; i < c; ++i)
    for (int j = 0; j < c; ++j) WinvT[i * c + j] = (float)a[j][c + i];     // (W^-1)^T [i, j] = W^-1 [j, i]
}

__global__ void wg_logdet_inv_kernel(const float* __restrict__ W, float* __restrict__ logdet, float* __restrict__ WinvT,
                                     float* __restrict__ sign_out, int c) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  wg_logdet_inv_one(W, logdet, WinvT, sign_out, c);
}

// every flow of the network in one launch: workgroup f handles the matrix at base + table[2f] (c = table[2f + 1]),
// WinvT slot f = 64 floats
__global__ void wg_logdet_inv_batched_kernel(const float* __restrict__ base, const long long* __restrict__ table,
                                             float* __restrict__ logdets, float* __restrict__ WinvT,
                                             float* __restrict__ signs) {
  if (threadIdx.x != 0) return;
  const int f = blockIdx.x;
  wg_logdet_inv_one(base + table[2 * f], logdets + f, WinvT + (long long)f * 64, signs ? signs + f : nullptr, (int)table[2 * f + 1]);
}

// ---------------------------------------------------------------------------------------------------------------
// Affine coupling (model.py:217-222): o = WN(a0, spect) as an fp32 [M, 8] matrix, b = o[:, :nh], log_s = o[:, nh:2nh];
//   z[:, :off + nh] = y[:, :off + nh];  z[:, off + nh + i] = exp(log_s_i) * y[:, off + nh + i] + b_i
// and one partial sum of log_s per workgroup (summed by wg_loss).
__global__ __launch_bounds__(WG_BLOCK) void wg_coupling_fwd_kernel(const float* __restrict__ y, const float* __restrict__ o,
                                                                   float* __restrict__ z, float* __restrict__ logs_partial,
                                                                   long long M, int c) {
  __shared__ float red[16];
  const int off = 8 - c, nh = c >> 1;
  float ls_sum = 0.f;
  for (long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long long)gridDim.x * blockDim.x) {
    float yi[8], oi[8];
    {
      const float4_t lo = *(const float4_t*)(y + m * 8), hi = *(const float4_t*)(y + m * 8 + 4);
      const float4_t ol = *(const float4_t*)(o + m * 8), oh = *(const float4_t*)(o + m * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { yi[e] = lo[e]; yi[4 + e] = hi[e]; oi[e] = ol[e]; oi[4 + e] = oh[e]; }
    }
    for (int i = 0; i < nh; ++i) {
      const float ls = oi[nh + i];
      ls_sum += ls;
      yi[off + nh + i] = __expf(ls) * yi[off + nh + i] + oi[i];
    }
    float4_t olo, ohi;
#pragma unroll
    for (int e = 0; e < 4; ++e) { olo[e] = yi[e]; ohi[e] = yi[4 + e]; }
    *(float4_t*)(z + m * 8) = olo;
    *(float4_t*)(z + m * 8 + 4) = ohi;
  }
  const float t = block_sum(ls_sum, red);
  if (threadIdx.x == 0) logs_partial[blockIdx.x] = t;
}

// dz = grad wrt z (scaled by the loss scale).  dy[:, :off + nh] = dz;  dy[:, off + nh + i] = dz1_i * exp(log_s_i);
// d_o (16-bit [M, 8], the operand of the `end` backward GEMMs): d_b_i = dz1_i, d_log_s_i = dz1_i * y1_i * exp(log_s_i)
// - scale * logs_coef (the -sum(log_s) term of the loss), zero padding above 2 nh.
template <int DT>
__global__ __launch_bounds__(WG_BLOCK) void wg_coupling_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ y,
                                                                   const float* __restrict__ o, float* __restrict__ dy,
                                                                   unsigned short* __restrict__ d_o,
                                                                   const float* __restrict__ scale, float logs_coef,
                                                                   long long M, int c) {
  const int off = 8 - c, nh = c >> 1;
  const float lsg = (scale ? *scale : 1.0f) * logs_coef;
  for (long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long long)gridDim.x * blockDim.x) {
    float g[8], yi[8], oi[8], dd[8];
    {
      const float4_t gl = *(const float4_t*)(dz + m * 8), gh = *(const float4_t*)(dz + m * 8 + 4);
      const float4_t lo = *(const float4_t*)(y + m * 8), hi = *(const float4_t*)(y + m * 8 + 4);
      const float4_t ol = *(const float4_t*)(o + m * 8), oh = *(const float4_t*)(o + m * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        g[e] = gl[e]; g[4 + e] = gh[e]; yi[e] = lo[e]; yi[4 + e] = hi[e]; oi[e] = ol[e]; oi[4 + e] = oh[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) dd[e] = 0.f;
    for (int i = 0; i < nh; ++i) {
      const float es = __expf(oi[nh + i]);
      const float gz = g[off + nh + i];
      dd[i] = gz;
      dd[nh + i] = gz * yi[off + nh + i] * es - lsg;
      g[off + nh + i] = gz * es;
    }
    float4_t olo, ohi;
#pragma unroll
    for (int e = 0; e < 4; ++e) { olo[e] = g[e]; ohi[e] = g[4 + e]; }
    *(float4_t*)(dy + m * 8) = olo;
    *(float4_t*)(dy + m * 8 + 4) = ohi;
    *(ushort8_t*)(d_o + m * 8) = pack8<DT>(dd);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// WaveGlowLoss (loss_function.py:30-48): (sum z^2 / (2 sigma^2) - sum log_s - M * sum_k log det W_k) / (M * 8)
__global__ __launch_bounds__(WG_BLOCK) void wg_sumsq_kernel(const float* __restrict__ z, long long n4,
                                                            float* __restrict__ partial) {
  __shared__ float red[16];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4_t v = ((const float4_t*)z)[i];
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  const float t = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(WG_BLOCK) void wg_loss_finish_kernel(const float* __restrict__ sq_partial, int n_sq,
                                                                  const float* __restrict__ logs_partial, int n_logs,
                                                                  const float* __restrict__ logdets, int n_flows,
                                                                  float inv_two_sigma2, float rows, float inv_count,
                                                                  float* __restrict__ loss) {
  __shared__ float red[16];
  float a = 0.f, b = 0.f, d = 0.f;
  for (int i = threadIdx.x; i < n_sq; i += blockDim.x) a += sq_partial[i];
  for (int i = threadIdx.x; i < n_logs; i += blockDim.x) b += logs_partial[i];
  for (int i = threadIdx.x; i < n_flows; i += blockDim.x) d += logdets[i];
  a = block_sum(a, red);
  b = block_sum(b, red);
  d = block_sum(d, red);
  if (threadIdx.x == 0) *loss = (a * inv_two_sigma2 - b - rows * d) * inv_count;
}

// dz = z * scale * coef   (coef = 1 / (sigma^2 * M * 8)): the gradient of the loss with respect to every output channel
__global__ __launch_bounds__(WG_BLOCK) void wg_dz_init_kernel(const float* __restrict__ z, float* __restrict__ dz,
                                                              const float* __restrict__ scale, float coef, long long n4) {
  const float f = (scale ? *scale : 1.0f) * coef;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4_t v = ((const float4_t*)z)[i];
    ((float4_t*)dz)[i] = v * f;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// torch.nn.utils.weight_norm(dim = 0) of a Conv1d weight v [Co, Ci, Kt] (model.py:95-136): w = g * v / ||v||_row, written as
// the 16-bit GEMM operand w16[co, tap * Cip + ci] (Cip >= Ci: zero-padded input channels).  g == NULL: plain weight.
// One wavefront per output channel.
template <int DT>
__device__ __forceinline__ void wg_weight_norm_fwd_row(const float* __restrict__ v, const float* __restrict__ g,
                                                       unsigned short* __restrict__ w16, int co, int Ci, int Kt, int Cip) {
  const int n = Ci * Kt;
  const float* vr = v + (long long)co * n;
  float f = 1.0f;
  if (g) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) s += vr[i] * vr[i];
    s = wave_sum(s);
    f = g[co] / sqrtf(s);
  }
  unsigned short* wr = w16 + (long long)co * Kt * Cip;
  for (int i = threadIdx.x; i < Kt * Cip; i += 64) {
    const int tap = i / Cip, ci = i - tap * Cip;
    const float val = ci < Ci ? vr[ci * Kt + tap] * f : 0.f;
    wr[i] = Elem<DT>::from_f32(val);
  }
}

template <int DT>
__global__ __launch_bounds__(64) void wg_weight_norm_fwd_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                                unsigned short* __restrict__ w16, int Co, int Ci, int Kt,
                                                                int Cip) {
  wg_weight_norm_fwd_row<DT>(v, g, w16, blockIdx.x, Ci, Kt, Cip);
}

// dw fp32 [Co, Kt * Cip] (the wgrad GEMM's layout) -> dv [Co, Ci, Kt], dg [Co]:
//   dg = <dw, v> / ||v||;  dv = (g / ||v||) * (dw - v * <dw, v> / ||v||^2).   g == NULL: dv = dw (re-laid out).
__device__ __forceinline__ void wg_weight_norm_bwd_row(const float* __restrict__ dw, const float* __restrict__ v,
                                                       const float* __restrict__ g, float* __restrict__ dv,
                                                       float* __restrict__ dg, int co, int Ci, int Kt, int Cip) {
  const int n = Ci * Kt;
  const float* vr = v + (long long)co * n;
  const float* dr = dw + (long long)co * Kt * Cip;
  float* dvr = dv + (long long)co * n;
  if (!g) {
    for (int i = threadIdx.x; i < n; i += 64) {
      const int ci = i / Kt, tap = i - ci * Kt;
      dvr[i] = dr[tap * Cip + ci];
    }
    return;
  }
  float s = 0.f, d = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) {
    const int ci = i / Kt, tap = i - ci * Kt;
    const float x = vr[i];
    s += x * x;
    d += dr[tap * Cip + ci] * x;
  }
  s = wave_sum(s);
  d = wave_sum(d);
  const float inv = 1.0f / sqrtf(s);
  const float gg = g[co];
  if (threadIdx.x == 0) dg[co] = d * inv;
  for (int i = threadIdx.x; i < n; i += 64) {
    const int ci = i / Kt, tap = i - ci * Kt;
    dvr[i] = gg * inv * (dr[tap * Cip + ci] - vr[i] * d * inv * inv);
  }
}

__global__ __launch_bounds__(64) void wg_weight_norm_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ v,
                                                                const float* __restrict__ g, float* __restrict__ dv,
                                                                float* __restrict__ dg, int Co, int Ci, int Kt, int Cip) {
  wg_weight_norm_bwd_row(dw, v, g, dv, dg, blockIdx.x, Ci, Kt, Cip);
}

// Every weight of the network in ONE launch (the per-tensor form is ~600 launches of ~10 us per step at the reference's
// size): a device table of WG_WN_FIELDS int64 per tensor
//   { row_start, Co, Ci, Kt, Cip, v, g (0: plain), w16, dw, dv, dg }
// sorted by row_start; workgroup r (one wavefront) finds its tensor by binary search and handles row r - row_start.
#define WG_WN_FIELDS 11
__device__ __forceinline__ int wg_wn_find(const long long* __restrict__ table, int n, long long row) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[(long long)mid * WG_WN_FIELDS] <= row) lo = mid; else hi = mid - 1;
  }
  return lo;
}

template <int DT>
__global__ __launch_bounds__(64) void wg_weight_norm_fwd_batched_kernel(const long long* __restrict__ table, int n) {
  const long long row = blockIdx.x;
  const long long* e = table + (long long)wg_wn_find(table, n, row) * WG_WN_FIELDS;
  wg_weight_norm_fwd_row<DT>((const float*)e[5], (const float*)e[6], (unsigned short*)e[7], (int)(row - e[0]), (int)e[2],
                             (int)e[3], (int)e[4]);
}

__global__ __launch_bounds__(64) void wg_weight_norm_bwd_batched_kernel(const long long* __restrict__ table, int n) {
  const long long row = blockIdx.x;
  const long long* e = table + (long long)wg_wn_find(table, n, row) * WG_WN_FIELDS;
  wg_weight_norm_bwd_row((const float*)e[8], (const float*)e[5], (const float*)e[6], (float*)e[9], (float*)e[10],
                         (int)(row - e[0]), (int)e[2], (int)e[3], (int)e[4]);
}

// ConvTranspose1d(Cm, Cm, ksize, stride) with ksize = ntap * stride as ONE GEMM (model.py:165-167, 197):
//   S[b, q * stride + r, co] = bias[co] + sum_{j < ntap} sum_ci mel[b, q - j, ci] * w[ci, co, r + stride * j]
// A = wg_taps(mel channels-last, dilation -1) [B * Fq, ntap * Cm], B operand b16[(r * Cm + co), (j * Cm + ci)], bias repeated
// per phase.  The output rows [B * Fq, stride * Cm] ARE the time-major tensor [B, Fq * stride, Cm].
template <int DT>
__global__ __launch_bounds__(WG_BLOCK) void wg_upsample_weight_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                                      unsigned short* __restrict__ b16,
                                                                      float* __restrict__ bias_rep, int Cm, int ksize,
                                                                      int stride) {
  const int ntap = ksize / stride;
  const long long total = (long long)stride * Cm * ntap * Cm;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cm);
    long long r0 = i / Cm;
    const int j = (int)(r0 % ntap);
    r0 /= ntap;                          // r0 = r * Cm + co
    const int co = (int)(r0 % Cm);
    const int r = (int)(r0 / Cm);
    b16[i] = Elem<DT>::from_f32(w[((long long)ci * Cm + co) * ksize + r + stride * j]);
    if (j == 0 && ci == 0) bias_rep[r0] = bias[co];
  }
}

// db fp32 [stride * Cm, ntap * Cm] -> dw [Cm, Cm, ksize]
__global__ __launch_bounds__(WG_BLOCK) void wg_upsample_weight_bwd_kernel(const float* __restrict__ db, float* __restrict__ dw,
                                                                          int Cm, int ksize, int stride) {
  const int ntap = ksize / stride;
  const long long total = (long long)Cm * Cm * ksize;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % ksize);
    const long long r0 = i / ksize;      // ci * Cm + co
    const int co = (int)(r0 % Cm), ci = (int)(r0 / Cm);
    const int j = k / stride, r = k - j * stride;
    dw[i] = db[((long long)r * Cm + co) * (ntap * Cm) + j * Cm + ci];
  }
}

// =============================================================================================== C ABI
#define WG_DT_CHECK(what) DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, what ": 16-bit dtypes only (got %d)", dtype)
#define WG_AL16(p) ((((uintptr_t)(p)) & 15) == 0)

extern "C" int dle_wg_taps(const void* x, void* col, int B, int T, int C, int ntaps, int dilation, int left,
                           int64_t ld_x, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(x && col && B > 0 && T > 0 && C > 0 && ntaps > 0, "wg_taps: bad args");
  WG_DT_CHECK("wg_taps");
  DLE_CHECK_ARG(C % 8 == 0 && ld_x % 8 == 0 && ld_x >= C && WG_AL16(x) && WG_AL16(col),
                "wg_taps: C, ld_x %% 8 == 0, ld_x >= C and 16-byte aligned tensors");
  const long long total = (long long)B * T * ntaps * (C / 8);
  hipLaunchKernelGGL(wg_taps_kernel, dim3(wg_grid(total)), dim3(WG_BLOCK), 0, stream, (const uint4_t*)x, (uint4_t*)col, B, T,
                     C / 8, ntaps, dilation, left, (long long)(ld_x / 8));
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_wg_taps_bwd(const void* dcol, const void* addend, void* dx, int B, int T, int C, int ntaps,
                               int dilation, int left, int64_t ld_add, int64_t ld_dx, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dcol && dx && B > 0 && T > 0 && C > 0 && ntaps > 0, "wg_taps_bwd: bad args");
  WG_DT_CHECK("wg_taps_bwd");
  DLE_CHECK_ARG(C % 8 == 0 && ld_dx % 8 == 0 && (!addend || ld_add % 8 == 0) && WG_AL16(dcol) && WG_AL16(dx) &&
                WG_AL16(addend), "wg_taps_bwd: C, ld %% 8 == 0 and 16-byte aligned tensors");
  const long long total = (long long)B * T * (C / 8);
  if (dtype == DLE_F16)
    hipLaunchKernelGGL(wg_taps_bwd_kernel<DLE_F16>, dim3(wg_grid(total)), dim3(WG_BLOCK), 0, stream, (const uint4_t*)dcol,
                       (const unsigned short*)addend, (unsigned short*)dx, B, T, C / 8, ntaps, dilation, left,
                       (long long)ld_add, (long long)ld_dx);
  else
    hipLaunchKernelGGL(wg_taps_bwd_kernel<DLE_BF16>, dim3(wg_grid(total)), dim3(WG_BLOCK), 0, stream, (const uint4_t*)dcol,
                       (const unsigned short*)addend, (unsigned short*)dx, B, T, C / 8, ntaps, dilation, left,
                       (long long)ld_add, (long long)ld_dx);
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_wg_gate_fwd(const void* s, void* acts, int64_t M, int nc, int64_t ld_s, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(s && acts && M > 0 && nc > 0, "wg_gate_fwd: bad args");
  WG_DT_CHECK("wg_gate_fwd");
  DLE_CHECK_ARG(nc % 8 == 0 && ld_s % 8 == 0 && ld_s >= 2 * nc && WG_AL16(s) && WG_AL16(acts),
                "wg_gate_fwd: nc, ld_s %% 8 == 0, ld_s >= 2 nc, 16-byte aligned tensors");
  const long long total = (long long)M * (nc / 8);
  if (dtype == DLE_F16)
    hipLaunchKernelGGL(wg_gate_fwd_kernel<DLE_F16>, dim3(wg_grid(total)), dim3(WG_BLOCK), 0, stream,
                       (const unsigned short*)s, (unsigned short*)acts, (long long)M, nc / 8, (long long)ld_s);
  else
    hipLaunchKernelGGL(wg_gate_fwd_kernel<DLE_BF16>, dim3(wg_grid(total)), dim3(WG_BLOCK), 0, stream,
                       (const unsigned short*)s, (unsigned short*)acts, (long long)M, nc / 8, (long long)ld_s);
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_wg_gate_bwd(const void* dacts, const void* s, void* ds, int64_t M, int nc, int64_t ld_s, int64_t ld_ds,
                               int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dacts && s && ds && M > 0 && nc > 0, "wg_gate_bwd: bad args");
  WG_DT_CHECK("wg_gate_bwd");
  DLE_CHECK_ARG(nc % 8 == 0 && ld_s % 8 == 0 && ld_ds % 8 == 0 && ld_s >= 2 * nc && ld_ds >= 2 * nc && WG_AL16(s) &&
                WG_AL16(ds) && WG_AL16(dacts), "wg_gate_bwd: nc, ld %% 8 == 0, ld >= 2 nc, 16-byte aligned tensors");
  const long long total = (long long)M * (nc / 8);
  if (dtype == DLE_F16)
    hipLaunchKernelGGL(wg_gate_bwd_kernel<DLE_F16>, dim3(wg_grid(total)), dim3(WG_BLOCK), 0, stream,
                       (const unsigned short*)dacts, (const unsigned short*)s, (unsigned short*)ds, (long long)M, nc / 8,
                       (long long)ld_s, (long long)ld_ds);
  else
    hipLaunchKernelGGL(wg_gate_bwd_kernel<DLE_BF16>, dim3(wg_grid(total)), dim3(WG_BLOCK), 0, stream,
                       (const unsigned short*)dacts, (const unsigned short*)s, (unsigned short*)ds, (long long)M, nc / 8,
                       (long long)ld_s, (long long)ld_ds);
  DLE_LAUNCH_CHECK();
  return 0;
}

#define WG_C_CHECK(what) DLE_CHECK_ARG(c >= 2 && c <= 8 && (c & 1) == 0, what ": 2 <= c <= 8, c even (got %d)", c)

extern "C" int dle_wg_invconv_fwd(const float* x, const float* W, float* y, void* a0_16, int64_t M, int c, int dtype,
                                  hipStream_t stream) {
  DLE_CHECK_ARG(x && W && y && M > 0, "wg_invconv_fwd: bad args");
  WG_C_CHECK("wg_invconv_fwd");
  DLE_CHECK_ARG(WG_AL16(x) && WG_AL16(y) && WG_AL16(a0_16), "wg_invconv_fwd: 16-byte aligned tensors");
  if (a0_16) WG_DT_CHECK("wg_invconv_fwd");
  if (dtype == DLE_BF16)
    hipLaunchKernelGGL(wg_invconv_fwd_kernel<DLE_BF16>, dim3(wg_grid(M)), dim3(WG_BLOCK), 0, stream, x, W, y,
                       (unsigned short*)a0_16, (long long)M, c);
  else
    hipLaunchKernelGGL(wg_invconv_fwd_kernel<DLE_F16>, dim3(wg_grid(M)), dim3(WG_BLOCK), 0, stream, x, W, y,
                       (unsigned short*)a0_16, (long long)M, c);
  DLE_LAUNCH_CHECK();
  return 0;
}

// workspace: >= dle_wg_invconv_bwd_partials(M) * 64 floats.  dW = the full gradient of the flow's weight (c x c), including
// the log-determinant term  -scale * logdet_coef * W^{-T}  (logdet_coef = M / (M * 8) for WaveGlowLoss).
extern "C" int dle_wg_invconv_bwd_partials(int64_t M) {
  long long g = (M + WG_BLOCK - 1) / WG_BLOCK;
  if (g > 128) g = 128;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int dle_wg_invconv_bwd(const float* dy, const float* da0, const float* x, const float* W, const float* WinvT,
                                  float* dx, float* dW, const float* scale_dev, float logdet_coef, float* workspace,
                                  int64_t M, int c, hipStream_t stream) {
  DLE_CHECK_ARG(dy && x && W && WinvT && dx && dW && workspace && M > 0, "wg_invconv_bwd: bad args");
  WG_C_CHECK("wg_invconv_bwd");
  DLE_CHECK_ARG(WG_AL16(dy) && WG_AL16(da0) && WG_AL16(x) && WG_AL16(dx), "wg_invconv_bwd: 16-byte aligned tensors");
  const int G = dle_wg_invconv_bwd_partials(M);
  hipLaunchKernelGGL(wg_invconv_bwd_kernel, dim3(G), dim3(WG_BLOCK), 0, stream, dy, da0, x, W, dx, workspace, (long long)M, c);
  DLE_LAUNCH_CHECK();
  hipLaunchKernelGGL(wg_invconv_wfinish_kernel, dim3(1), dim3(64), 0, stream, (const float*)workspace, G, WinvT, dW,
                     scale_dev, logdet_coef, c);
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_wg_logdet_inv(const float* W, float* logdet, float* WinvT, float* sign, int c, hipStream_t stream) {
  DLE_CHECK_ARG(W && logdet && WinvT, "wg_logdet_inv: bad args");
  DLE_CHECK_ARG(c >= 1 && c <= 8, "wg_logdet_inv: 1 <= c <= 8 (got %d)", c);
  hipLaunchKernelGGL(wg_logdet_inv_kernel, dim3(1), dim3(64), 0, stream, W, logdet, WinvT, sign, c);
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_wg_coupling_partials(int64_t M) { return wg_grid(M); }

extern "C" int dle_wg_coupling_fwd(const float* y, const float* o, float* z, float* logs_partial, int64_t M, int c,
                                   hipStream_t stream) {
  DLE_CHECK_ARG(y && o && z && logs_partial && M > 0, "wg_coupling_fwd: bad args");
  WG_C_CHECK("wg_coupling_fwd");
  DLE_CHECK_ARG(WG_AL16(y) && WG_AL16(o) && WG_AL16(z), "wg_coupling_fwd: 16-byte aligned tensors");
  hipLaunchKernelGGL(wg_coupling_fwd_kernel, dim3(wg_grid(M)), dim3(WG_BLOCK), 0, stream, y, o, z, logs_partial, (long long)M, c);
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_wg_coupling_bwd(const float* dz, const float* y, const float* o, float* dy, void* d_o16,
                                   const float* scale_dev, float logs_coef, int64_t M, int c, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dz && y && o && dy && d_o16 && M > 0, "wg_coupling_bwd: bad args");
  WG_C_CHECK("wg_coupling_bwd");
  WG_DT_CHECK("wg_coupling_bwd");
  DLE_CHECK_ARG(WG_AL16(dz) && WG_AL16(y) && WG_AL16(o) && WG_AL16(dy) && WG_AL16(d_o16), "wg_coupling_bwd: 16-byte aligned tensors");
  if (dtype == DLE_F16)
    hipLaunchKernelGGL(wg_coupling_bwd_kernel<DLE_F16>, dim3(wg_grid(M)), dim3(WG_BLOCK), 0, stream, dz, y, o, dy,
                       (unsigned short*)d_o16, scale_dev, logs_coef, (long long)M, c);
  else
    hipLaunchKernelGGL(wg_coupling_bwd_kernel<DLE_BF16>, dim3(wg_grid(M)), dim3(WG_BLOCK), 0, stream, dz, y, o, dy,
                       (unsigned short*)d_o16, scale_dev, logs_coef, (long long)M, c);
  DLE_LAUNCH_CHECK();
  return 0;
}

// workspace: >= dle_wg_coupling_partials(M) floats
extern "C" int dle_wg_loss(const float* z, const float* logs_partial, int n_logs, const float* logdets, int n_flows,
                           float sigma, int64_t M, float* loss_out, float* workspace, hipStream_t stream) {
  DLE_CHECK_ARG(z && logs_partial && logdets && loss_out && workspace && M > 0 && n_logs >= 0 && n_flows >= 0 && sigma > 0.f,
                "wg_loss: bad args");
  DLE_CHECK_ARG(WG_AL16(z), "wg_loss: 16-byte aligned z");
  const int G = wg_grid(M * 2);
  hipLaunchKernelGGL(wg_sumsq_kernel, dim3(G), dim3(WG_BLOCK), 0, stream, z, (long long)M * 2, workspace);
  DLE_LAUNCH_CHECK();
  hipLaunchKernelGGL(wg_loss_finish_kernel, dim3(1), dim3(WG_BLOCK), 0, stream, (const float*)workspace, G, logs_partial,
                     n_logs, logdets, n_flows, 1.0f / (2.0f * sigma * sigma), (float)M, 1.0f / ((float)M * 8.0f), loss_out);
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_wg_dz_init(const float* z, float* dz, const float* scale_dev, float coef, int64_t M, hipStream_t stream) {
  DLE_CHECK_ARG(z && dz && M > 0, "wg_dz_init: bad args");
  DLE_CHECK_ARG(WG_AL16(z) && WG_AL16(dz), "wg_dz_init: 16-byte aligned tensors");
  hipLaunchKernelGGL(wg_dz_init_kernel, dim3(wg_grid(M * 2)), dim3(WG_BLOCK), 0, stream, z, dz, scale_dev, coef, (long long)M * 2);
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_wg_weight_norm_fwd(const float* v, const float* g, void* w16, int Co, int Ci, int Kt, int Cip, int dtype,
                                      hipStream_t stream) {
  DLE_CHECK_ARG(v && w16 && Co > 0 && Ci > 0 && Kt > 0 && Cip >= Ci, "wg_weight_norm_fwd: bad args");
  WG_DT_CHECK("wg_weight_norm_fwd");
  if (dtype == DLE_F16)
    hipLaunchKernelGGL(wg_weight_norm_fwd_kernel<DLE_F16>, dim3(Co), dim3(64), 0, stream, v, g, (unsigned short*)w16, Co, Ci, Kt, Cip);
  else
    hipLaunchKernelGGL(wg_weight_norm_fwd_kernel<DLE_BF16>, dim3(Co), dim3(64), 0, stream, v, g, (unsigned short*)w16, Co, Ci, Kt, Cip);
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_wg_weight_norm_bwd(const float* dw, const float* v, const float* g, float* dv, float* dg, int Co, int Ci,
                                      int Kt, int Cip, hipStream_t stream) {
  DLE_CHECK_ARG(dw && v && dv && (!g || dg) && Co > 0 && Ci > 0 && Kt > 0 && Cip >= Ci, "wg_weight_norm_bwd: bad args");
  hipLaunchKernelGGL(wg_weight_norm_bwd_kernel, dim3(Co), dim3(64), 0, stream, dw, v, g, dv, dg, Co, Ci, Kt, Cip);
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_wg_upsample_weight(const float* w, const float* bias, void* b16, float* bias_rep, int Cm, int ksize,
                                      int stride, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(w && bias && b16 && bias_rep && Cm > 0 && stride > 0 && ksize > 0 && ksize % stride == 0,
                "wg_upsample_weight: bad args (ksize must be a multiple of stride)");
  WG_DT_CHECK("wg_upsample_weight");
  const long long total = (long long)ksize * Cm * Cm;
  if (dtype == DLE_F16)
    hipLaunchKernelGGL(wg_upsample_weight_kernel<DLE_F16>, dim3(wg_grid(total)), dim3(WG_BLOCK), 0, stream, w, bias,
                       (unsigned short*)b16, bias_rep, Cm, ksize, stride);
  else
    hipLaunchKernelGGL(wg_upsample_weight_kernel<DLE_BF16>, dim3(wg_grid(total)), dim3(WG_BLOCK), 0, stream, w, bias,
                       (unsigned short*)b16, bias_rep, Cm, ksize, stride);
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_wg_upsample_weight_bwd(const float* db, float* dw, int Cm, int ksize, int stride, hipStream_t stream) {
  DLE_CHECK_ARG(db && dw && Cm > 0 && stride > 0 && ksize > 0 && ksize % stride == 0, "wg_upsample_weight_bwd: bad args");
  const long long total = (long long)ksize * Cm * Cm;
  hipLaunchKernelGGL(wg_upsample_weight_bwd_kernel, dim3(wg_grid(total)), dim3(WG_BLOCK), 0, stream, db, dw, Cm, ksize, stride);
  DLE_LAUNCH_CHECK();
  return 0;
}

// Table-driven forms (one launch for the whole network).  table_dev: n_entries x 11 int64 = { row_start, Co, Ci, Kt, Cip,
// v, g (0: plain weight), w16, dw, dv, dg } (device pointers as integers), sorted by row_start, total_rows = sum of Co.
extern "C" int dle_wg_weight_norm_fwd_batched(const int64_t* table_dev, int n_entries, int64_t total_rows, int dtype,
                                              hipStream_t stream) {
  DLE_CHECK_ARG(table_dev && n_entries > 0 && total_rows > 0 && total_rows < 0x7fffffffLL, "wg_weight_norm_fwd_batched: bad args");
  WG_DT_CHECK("wg_weight_norm_fwd_batched");
  if (dtype == DLE_F16)
    hipLaunchKernelGGL(wg_weight_norm_fwd_batched_kernel<DLE_F16>, dim3((unsigned)total_rows), dim3(64), 0, stream,
                       (const long long*)table_dev, n_entries);
  else
    hipLaunchKernelGGL(wg_weight_norm_fwd_batched_kernel<DLE_BF16>, dim3((unsigned)total_rows), dim3(64), 0, stream,
                       (const long long*)table_dev, n_entries);
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_wg_weight_norm_bwd_batched(const int64_t* table_dev, int n_entries, int64_t total_rows, hipStream_t stream) {
  DLE_CHECK_ARG(table_dev && n_entries > 0 && total_rows > 0 && total_rows < 0x7fffffffLL, "wg_weight_norm_bwd_batched: bad args");
  hipLaunchKernelGGL(wg_weight_norm_bwd_batched_kernel, dim3((unsigned)total_rows), dim3(64), 0, stream,
                     (const long long*)table_dev, n_entries);
  DLE_LAUNCH_CHECK();
  return 0;
}

// table_dev: n_flows x 2 int64 = { element offset of the c x c matrix from `base`, c }; WinvT: n_flows x 64 floats.
extern "C" int dle_wg_logdet_inv_batched(const float* base, const int64_t* table_dev, float* logdets, float* WinvT,
                                         float* signs, int n_flows, hipStream_t stream) {
  DLE_CHECK_ARG(base && table_dev && logdets && WinvT && n_flows > 0, "wg_logdet_inv_batched: bad args");
  hipLaunchKernelGGL(wg_logdet_inv_batched_kernel, dim3(n_flows), dim3(64), 0, stream, base, (const long long*)table_dev,
                     logdets, WinvT, signs);
  DLE_LAUNCH_CHECK();
  return 0;
}
