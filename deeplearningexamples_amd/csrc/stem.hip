// The ResNet stem for gfx950: 7x7 / stride 2 / pad 3 convolution of a 3-channel image (forward, + BatchNorm partial sums) and its
// weight gradient, as halo-tile MFMA kernels on a 4-channel (8 bytes / pixel) NHWC input.
//
// Replaces cuDNN's conv fwd / bwd-filter behind `self.conv1 = builder.conv7x7(3, 64, stride=2)` + bn1's batch statistics
// (Classification/ConvNets/image_classification/models/resnet.py:262-268,318-322, models/common.py:31-60,84-105).
//
// Why not the im2col GEMM of gemm_dma.hip: with C = 3 (padded to 8) a 64-deep K tile spans 8 filter taps, every 16-byte DMA chunk
// is one (pixel, tap) with its own address arithmetic, and 5 of 8 channels are zeros: 716 us forward / 503 us weight gradient at
// batch 256 against 616 MB = ~110 us of traffic.  Here
//  * the image is [N, H, W, 4] (3 real channels + one zero): a filter ROW is 8 taps (7 real + one zero) x 4 channels = 32
//    contraction elements = ONE v_mfma_f32_16x16x32 k step, and a lane's 8 consecutive k are TWO ADJACENT PIXELS = one 16-byte LDS
//    read at (2 q - 3 + 2 kq) -- no im2col, no gather;
//  * forward: a workgroup owns 2 output rows x all columns x 64 channels; its 9 x 232-pixel input patch (zero halo) sits in LDS,
//    each wavefront keeps its 32 output channels' weights (7 rows x 2 blocks) in registers; the weight rows of the two blocks are
//    interleaved so that a lane leaves with 8 consecutive output channels of one pixel (16-byte stores);
//  * weight gradient: contraction over pixels.  Both operands are pixel-major in LDS and read with the LDS transpose read
//    (ds_read_b64_tr_b16): dy rows as [channel block][pixel slot][16 channels] (pixel bits 2 / 3 swapped: conflict free), the image
//    straight from the patch -- a lane's 8-byte piece is one pixel's 4 channels, the 4 pieces of a 16-lane row group are 4 adjacent
//    taps.  Persistent workgroups accumulate 64 x 224 partial gradients in registers over their output rows; one partial per
//    workgroup, folded in a fixed order (deterministic, no atomics).
// Packed weight layout (both kernels): w2[ko][r][s8][c4], k = r * 32 + s * 4 + c, zero where s = 7 or c = 3.
#include "gemm_tiles.h"

#define ST_PW 232                 // forward patch pixels per input row: x = -3 .. 228 (W <= 224)
#define ST_PWG 264                // weight-gradient patch: its reads reach pixel 2 * 127 + 7 (dy pixels past Q are zero rows, but
                                  // the image side of the product must be FINITE there: 0 * garbage = NaN)
#define ST_K2 224                 // 7 filter rows x 8 taps x 4 channels

template <int DT> struct StMfma;
template <> struct StMfma<DLE_F16> {
  static __device__ __forceinline__ float4_t run(ushort8_t a, ushort8_t b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
  }
};
template <> struct StMfma<DLE_BF16> {
  static __device__ __forceinline__ float4_t run(ushort8_t a, ushort8_t b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};

struct StemArgs {
  const unsigned short* x;     // [N, H, W, 4]
  const unsigned short* w;     // [64][224] packed
  unsigned short* y;           // [N, P, Q, 64]
  float* stats;                // [N * PP][2][64] or NULL
  int N, H, W, P, Q, PP;       // PP = ceil(P / 2) workgroups per image
};

// Input patch rows [h0, h0 + ROWS) x columns [-3, PW - 3) of image n -> LDS, zero outside the image, in two halves so that the
// NEXT tile's loads fly under the current tile's MFMAs (registers, not a second LDS buffer).  All loads are unconditional
// (clamped address, masked value): loads under a run-time condition are serialised by hipcc's wait-count pass.
template <int ROWS, int PW>
struct StemPatch {
  static constexpr int TOTAL = ROWS * PW, ITERS = (TOTAL + 255) / 256;
  unsigned long long v[ITERS];
  unsigned okbits;
  __device__ __forceinline__ void issue(const unsigned short* x, int n, int h0, int H, int W, int tid) {
    const unsigned long long* xg = (const unsigned long long*)x;
    okbits = 0;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int idx = tid + it * 256;
      const int lr = idx / PW, lc = idx - lr * PW;
      const int h = h0 + lr, xw = lc - 3;
      const bool ok = idx < TOTAL && h >= 0 && h < H && xw >= 0 && xw < W;
      okbits |= (ok ? 1u : 0u) << it;
      const long long o = ok ? ((long long)n * H + h) * W + xw : 0;
      v[it] = xg[o];
    }
  }
  __device__ __forceinline__ void commit(unsigned long long* patch, int tid) const {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int idx = tid + it * 256;
      if (idx < TOTAL) patch[idx] = ((okbits >> it) & 1u) ? v[it] : 0ull;
    }
  }
};

// Persistent over row pairs (work item = (image, pair of output rows), dealt round-robin): the weights are loaded once, the next
// item's patch is requested before the MFMAs of the current one (and before its stores: vmcnt retires in order), the BatchNorm
// partial sums stay in registers across the walk -- ONE partial row per workgroup (<= 512 rows: a single finishing launch).
template <int DT>
__global__ __launch_bounds__(256) void stem7_fwd_kernel(StemArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned long long patch[9 * ST_PW];
  __shared__ float red[4][2][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int prow = wave >> 1, kh = wave & 1;              // output row of the pair, output-channel half
  const int i = lane & 15, kq = lane >> 4;
  const int items = a.N * a.PP;
  // this wavefront's weights: rows (first MFMA operand) = output channels, block 0 <-> channels 8 t + {0..3}, block 1 <-> 8 t + {4..7}
  ushort8_t wf[7][2];
#pragma unroll
  for (int r = 0; r < 7; ++r)
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const int ko = kh * 32 + 8 * (i >> 2) + 4 * blk + (i & 3);
      wf[r][blk] = *(const ushort8_t*)(a.w + ko * ST_K2 + r * 32 + 8 * kq);
    }
  float s1[8], s2[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
  StemPatch<9, ST_PW> pf;
  int item = blockIdx.x;
  {
    const int n = item / a.PP, pp = item - n * a.PP;
    pf.issue(a.x, n, 4 * pp - 3, a.H, a.W, tid);
  }
  const unsigned char* pb = (const unsigned char*)patch;
  for (; item < items; item += gridDim.x) {
    const int n = item / a.PP, pp = item - n * a.PP;
    const int p0 = pp * 2;
    pf.commit(patch, tid);
    __syncthreads();
    {
      const int nx = item + gridDim.x < items ? item + gridDim.x : item;      // past the end: re-read the current one (unused)
      const int n2 = nx / a.PP, pp2 = nx - n2 * a.PP;
      pf.issue(a.x, n2, 4 * pp2 - 3, a.H, a.W, tid);
    }
    float4_t acc[7][2];
#pragma unroll
    for (int b = 0; b < 7; ++b) { acc[b][0] = (float4_t){0.f, 0.f, 0.f, 0.f}; acc[b][1] = (float4_t){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      // pixel pair (2 q - 3 + 2 kq, + 1) of input row 2 p - 3 + r = patch (2 prow + r, 2 q + 2 kq); q = 16 b + i
      const unsigned char* rb = pb + (((2 * prow + r) * ST_PW + 2 * i + 2 * kq) << 3);
#pragma unroll
      for (int b = 0; b < 7; ++b) {
        const ushort8_t xf = *(const ushort8_t*)(rb + b * 256);
        acc[b][0] = StMfma<DT>::run(wf[r][0], xf, acc[b][0]);
        acc[b][1] = StMfma<DT>::run(wf[r][1], xf, acc[b][1]);
      }
    }
    // D[row = channel 4 kq + reg][col = pixel i]: the lane holds channels kh * 32 + 8 kq + {0..7} of pixel 16 b + i
    const int p = p0 + prow;
#pragma unroll
    for (int b = 0; b < 7; ++b) {
      const int q = 16 * b + i;
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = acc[b][0][r]; v[4 + r] = acc[b][1][r]; }
      const ushort8_t ov = pack8<DT>(v);
      if (p < a.P && q < a.Q) {
        *(ushort8_t*)(a.y + ((((long long)n * a.P + p) * a.Q + q) << 6) + kh * 32 + 8 * kq) = ov;
        float z[8];
        unpack8<DT>(ov, z);
#pragma unroll
        for (int r = 0; r < 8; ++r) { s1[r] += z[r]; s2[r] += z[r] * z[r]; }
      }
    }
    __syncthreads();                                       // every wave is done reading the patch
  }
  if (a.stats) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { s1[r] += __shfl_xor(s1[r], o, 64); s2[r] += __shfl_xor(s2[r], o, 64); }
    if (i == 0) {
#pragma unroll
      for (int r = 0; r < 8; ++r) { red[wave][0][8 * kq + r] = s1[r]; red[wave][1][8 * kq + r] = s2[r]; }
    }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, col = tid & 63, h2 = col >> 5, c = col & 31;
      a.stats[((long long)blockIdx.x * 2 + which) * 64 + col] = red[h2][which][c] + red[2 + h2][which][c];
    }
  }
}

static int stem_fwd_workgroups(int items) {
  static const int per_cu = getenv("DLE_STEM_FWD_WG_PER_CU") ? atoi(getenv("DLE_STEM_FWD_WG_PER_CU")) : 2;
  const int cap = 256 * (per_cu > 0 ? per_cu : 2);          // 238 registers: two workgroups per CU
  return items < cap ? items : cap;
}

// Number of statistics rows dle_stem_conv7_fwd writes (one per persistent workgroup).
extern "C" int dle_stem_conv7_groups(int N, int H) {
  const int P = (H - 1) / 2 + 1;
  return stem_fwd_workgroups(N * ((P + 1) / 2));
}

// y [N, P, Q, 64] = conv7x7/2 pad 3 (x4 [N, H, W, 4], w2 [64][224] packed); stats (optional): per-group column sums / sums of
// squares of the ROUNDED output, [groups][2][64] (same contract as dle_conv2d_fwd_colstats).
extern "C" int dle_stem_conv7_fwd(const void* x4, const void* w2, void* y, float* stats, int64_t stats_bytes, int N, int H, int W,
                                  int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "stem_conv7_fwd: 16-bit dtypes only");
  DLE_CHECK_ARG(x4 && w2 && y && N > 0 && H > 0 && W > 0, "stem_conv7_fwd: bad arguments");
  DLE_CHECK_ARG(W <= 224, "stem_conv7_fwd: images up to 224 pixels wide (got %d)", W);
  DLE_CHECK_ARG(((((uintptr_t)x4) & 7) | (((uintptr_t)w2) & 15) | (((uintptr_t)y) & 15)) == 0, "stem_conv7_fwd: misaligned pointer");
  StemArgs a;
  a.x = (const unsigned short*)x4; a.w = (const unsigned short*)w2; a.y = (unsigned short*)y; a.stats = stats;
  a.N = N; a.H = H; a.W = W; a.P = (H - 1) / 2 + 1; a.Q = (W - 1) / 2 + 1; a.PP = (a.P + 1) / 2;
  DLE_CHECK_ARG((long long)N * a.PP < 0x7FFFFFFFLL, "stem_conv7_fwd: too many rows");
  const int G = stem_fwd_workgroups(N * a.PP);
  if (stats) DLE_CHECK_ARG(stats_bytes >= (long long)G * 2 * 64 * 4, "stem_conv7_fwd: statistics buffer too small");
  const dim3 grid((unsigned)G), block(256);
  if (dtype == DLE_F16) hipLaunchKernelGGL(stem7_fwd_kernel<DLE_F16>, grid, block, 0, stream, a);
  else hipLaunchKernelGGL(stem7_fwd_kernel<DLE_BF16>, grid, block, 0, stream, a);
  DLE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- weight gradient
struct StemWgArgs {
  const unsigned short* gt;    // [N, P, Q, 64]
  const unsigned short* x;     // [N, H, W, 4]
  float* ws;                   // [workgroups][64][224]
  int N, H, W, P, Q;
};

// LDS slot of dy pixel px (bits 2 and 3 swapped): the 32 lanes of a transpose-read group touch pixels {0..3, 8..11} (+ 4 for the
// second read) of a 16-pixel span -> 8 consecutive 32-byte slots = every bank once.
__device__ __forceinline__ int stem_slot(int px) { return (px & ~0xC) | ((px & 8) >> 1) | ((px & 4) << 1); }

template <int DT>
__global__ __launch_bounds__(256, 2) void stem7_wgrad_kernel(StemWgArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char gtl[4 * 128 * 32];      // [channel block][pixel slot][16 channels]
  __shared__ __attribute__((aligned(16))) unsigned long long patch[7 * ST_PWG];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  float4_t acc[4][4];                                     // [k2 block wave + 4 t][channel block]
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = (float4_t){0.f, 0.f, 0.f, 0.f};
  const int rows = a.N * a.P;
  // the next row's operands (dy row: 4 x 16 bytes per thread, image patch: 8 x 8 bytes) are requested before the products of the
  // current one and parked in registers
  StemPatch<7, ST_PWG> pf;
  ushort8_t gv[4];
  auto issue_row = [&](int row) __attribute__((always_inline)) {
    const int n = row / a.P, p = row - n * a.P;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + it * 256, px = idx >> 3, c8 = idx & 7;
      const long long o = ((((long long)n * a.P + p) * a.Q + (px < a.Q ? px : 0)) << 6) + c8 * 8;
      gv[it] = *(const ushort8_t*)(a.gt + o);
    }
    pf.issue(a.x, n, 2 * p - 3, a.H, a.W, tid);
  };
  if ((int)blockIdx.x < rows) issue_row(blockIdx.x);
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    __syncthreads();                                      // the previous row's fragments have been read
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      // dy row -> gtl: 128 pixel slots x 8 chunks of 16 bytes, zero past Q
      const int idx = tid + it * 256, px = idx >> 3, c8 = idx & 7;
      const ushort8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
      *(ushort8_t*)(gtl + (c8 >> 1) * 4096 + stem_slot(px) * 32 + (c8 & 1) * 16) = px < a.Q ? gv[it] : z;
    }
    pf.commit(patch, tid);
    __syncthreads();
    issue_row(row + gridDim.x < rows ? row + gridDim.x : row);
    const unsigned char* pb = (const unsigned char*)patch;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      // contraction pixels of this lane's two reads: lo = ks * 32 + 8 kq + (i >> 2), hi = lo + 4; 8-byte piece (i & 3)
      const int plo = ks * 32 + 8 * kq + (i >> 2), phi = plo + 4;
      TrPair fa[4], fb[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        fa[c].lo = ds_read_tr16_asm((const unsigned short*)(gtl + c * 4096 + stem_slot(plo) * 32 + 8 * (i & 3)));
        fa[c].hi = ds_read_tr16_asm((const unsigned short*)(gtl + c * 4096 + stem_slot(phi) * 32 + 8 * (i & 3)));
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int kb = wave + 4 * t;                      // k2 block = (filter row kb >> 1, tap half kb & 1)
        if (kb < 14) {
          const int r = kb >> 1, tap = 4 * (kb & 1) + (i & 3);
          fb[t].lo = ds_read_tr16_asm((const unsigned short*)(pb + ((r * ST_PWG + 2 * plo + tap) << 3)));
          fb[t].hi = ds_read_tr16_asm((const unsigned short*)(pb + ((r * ST_PWG + 2 * phi + tap) << 3)));
        }
      }
      frag_wait<true>();
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (wave + 4 * t < 14) {
          const ushort8_t vb = frag_value(fb[t]);
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[t][c] = StMfma<DT>::run(frag_value(fa[c]), vb, acc[t][c]);
        }
      }
    }
  }
  // partial gradient of this workgroup: D[row = channel 16 c + 4 kq + reg][col = k2 = 16 kb + i]
  float* out = a.ws + (long long)blockIdx.x * 64 * ST_K2;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int kb = wave + 4 * t;
    if (kb < 14) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(16 * c + 4 * kq + r) * ST_K2 + 16 * kb + i] = acc[t][c][r];
    }
  }
}

// dw[ko][r][s][c] (fp32, the master's KRSC memory order, 7 x 7 x 3) (+)= sum over the G partials; fixed order.
__global__ __launch_bounds__(256) void stem7_wgrad_reduce_kernel(const float* __restrict__ ws, int G, float* __restrict__ dw,
                                                                 int accumulate) {
  __shared__ float red[256];
  const int e = blockIdx.x * 64 + (threadIdx.x & 63), gs = threadIdx.x >> 6;
  float s = 0.f;
  if (e < 64 * 147) {
    const int ko = e / 147, rem = e - ko * 147, r = rem / 21, sc = rem - r * 21, sx = sc / 3, c = sc - sx * 3;
    const float* src = ws + (long long)ko * ST_K2 + r * 32 + sx * 4 + c;
    int g = gs;
    for (; g + 28 < G; g += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(long long)(g + 4 * u) * 64 * ST_K2];
      s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; g < G; g += 4) s += src[(long long)g * 64 * ST_K2];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (gs == 0 && e < 64 * 147) {
    const float t = (red[threadIdx.x] + red[64 + threadIdx.x]) + (red[128 + threadIdx.x] + red[192 + threadIdx.x]);
    dw[e] = accumulate ? dw[e] + t : t;
  }
}

static int stem_wgrad_workgroups(int rows) { return rows < 512 ? rows : 512; }

// Workspace bytes dle_stem_conv7_wgrad needs for (N, H).
extern "C" int64_t dle_stem_conv7_wgrad_workspace(int N, int H) {
  const int P = (H - 1) / 2 + 1;
  return (int64_t)stem_wgrad_workgroups(N * P) * 64 * ST_K2 * 4;
}

// dw [64, 7, 7, 3] fp32 (KRSC memory order of the channels_last master) (+)= conv7x7/2 weight gradient of dy [N, P, Q, 64] against
// x4 [N, H, W, 4].
extern "C" int dle_stem_conv7_wgrad(const void* dy, const void* x4, float* dw, void* workspace, int64_t workspace_bytes, int N, int H,
                                    int W, int dtype, int accumulate, hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "stem_conv7_wgrad: 16-bit dtypes only");
  DLE_CHECK_ARG(dy && x4 && dw && workspace && N > 0 && H > 0 && W > 0, "stem_conv7_wgrad: bad arguments");
  DLE_CHECK_ARG(W <= 224, "stem_conv7_wgrad: images up to 224 pixels wide (got %d)", W);
  DLE_CHECK_ARG(((((uintptr_t)x4) & 7) | (((uintptr_t)dy) & 15) | (((uintptr_t)workspace) & 15)) == 0, "stem_conv7_wgrad: misaligned pointer");
  StemWgArgs a;
  a.gt = (const unsigned short*)dy; a.x = (const unsigned short*)x4; a.ws = (float*)workspace;
  a.N = N; a.H = H; a.W = W; a.P = (H - 1) / 2 + 1; a.Q = (W - 1) / 2 + 1;
  DLE_CHECK_ARG(a.Q <= 128 && (long long)N * a.P < 0x7FFFFFFFLL, "stem_conv7_wgrad: shape out of range");
  const int G = stem_wgrad_workgroups(N * a.P);
  DLE_CHECK_ARG(workspace_bytes >= (long long)G * 64 * ST_K2 * 4, "stem_conv7_wgrad: workspace too small");
  if (dtype == DLE_F16) hipLaunchKernelGGL(stem7_wgrad_kernel<DLE_F16>, dim3(G), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(stem7_wgrad_kernel<DLE_BF16>, dim3(G), dim3(256), 0, stream, a);
  DLE_LAUNCH_CHECK();
  hipLaunchKernelGGL(stem7_wgrad_reduce_kernel, dim3((64 * 147 + 63) / 64), dim3(256), 0, stream, (const float*)workspace, G, dw, accumulate);
  DLE_LAUNCH_CHECK();
  return 0;
}

// fp32 master weight in KRSC memory order [64][7][7][3] -> packed 16-bit [64][7][8][4] (zero tap 7 / channel 3).
template <int DT>
__global__ __launch_bounds__(256) void stem7_pack_weight_kernel(const float* __restrict__ w, unsigned short* __restrict__ out) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 64 * ST_K2) return;
  const int ko = e / ST_K2, k = e - ko * ST_K2, r = k >> 5, sx = (k >> 2) & 7, c = k & 3;
  out[e] = (sx < 7 && c < 3) ? Elem<DT>::from_f32(w[((ko * 7 + r) * 7 + sx) * 3 + c]) : (unsigned short)0;
}

extern "C" int dle_stem_pack_weight(const float* w_krsc, void* out, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "stem_pack_weight: 16-bit output only");
  DLE_CHECK_ARG(w_krsc && out, "stem_pack_weight: null pointer");
  if (dtype == DLE_F16) hipLaunchKernelGGL(stem7_pack_weight_kernel<DLE_F16>, dim3(56), dim3(256), 0, stream, w_krsc, (unsigned short*)out);
  else hipLaunchKernelGGL(stem7_pack_weight_kernel<DLE_BF16>, dim3(56), dim3(256), 0, stream, w_krsc, (unsigned short*)out);
  DLE_LAUNCH_CHECK();
  return 0;
}
