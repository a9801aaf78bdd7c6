// The head of the DLRM top model in ONE pass over its input: the last linear layer (out_features = 1), BCEWithLogitsLoss
// (mean) and the backward of both.  Reference: Recommendation/DLRM/dlrm/model/distributed.py (top MLP + `out` layer),
// dlrm/scripts/main.py:560-600 (loss = BCEWithLogitsLoss(mean); scaler.scale(loss).backward()).
//
// As separate launches this is five kernels that each sweep a [batch, K] or [batch] operand for a few flops per byte -- a
// batch x 1 x K GEMV, the loss, a batch x K x 1 outer product, a 1 x K x batch weight gradient and a column sum: 160 us of the
// 2.85 ms step at batch 65536, K = 256, against 67 MB of unavoidable traffic (read h, write dh).  Here a half-wave owns a row:
// 16 bytes of h per lane, a 5-step butterfly for the dot product, then every lane has the logit and finishes its 8 columns of
// dh = dlogit * w under the ReLU mask of h, and accumulates dw += dlogit * h, the bias gradient of the PREVIOUS layer
// (column sums of the rounded dh) and, on lane 0, d bias / the loss.  Every 16-bit rounding point of the separate launches is
// kept (logit, dlogit and dh are rounded to the storage type before they are used), so the results differ from them only in
// fp32 summation order.  One partial row per workgroup, folded in index order by a second small launch: deterministic.
#include "common.h"

struct HeadArgs {
  const unsigned short* h;       // [M, K] 16-bit, row pitch ldh
  const unsigned short* w;       // [K] 16-bit (the working copy of out.weight)
  const float* bias;             // [1] fp32 or NULL
  const float* target;           // [M] fp32
  const float* grad_scale;       // device scalar or NULL
  float* loss;                   // [1]
  unsigned short* logits;        // [M] 16-bit or NULL
  unsigned short* dh;            // [M, K] 16-bit, row pitch ldd
  float* gw;                     // [K]
  float* gb;                     // [1]
  float* gprev;                  // [K] column sums of dh (bias gradient of the layer that produced h) or NULL
  float* ws;                     // [grid][2 K + 2] partial rows
  long long M, ldh, ldd;
  int K;
};

template <int DT, int NCH>       // NCH 16-byte chunks per lane: K <= 256 NCH (1 or 2)
__global__ __launch_bounds__(256) void head_bce_kernel(HeadArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* red = (float*)smem_raw;                       // [8 half-waves][2 K + 2]
  const int hw = threadIdx.x >> 5, l = threadIdx.x & 31;
  const int K = p.K, KC = K >> 3;                       // 16-byte chunks per row
  const float gs = (p.grad_scale ? *p.grad_scale : 1.0f) / (float)p.M;
  const float b0 = p.bias ? *p.bias : 0.f;
  float wv[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = l + 32 * c;
    ushort8_t u = {0, 0, 0, 0, 0, 0, 0, 0};
    if (ch < KC) u = *(const ushort8_t*)(p.w + ch * 8);
    unpack8<DT>(u, wv[c]);
  }
  float aw[NCH][8], ap[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) { aw[c][j] = 0.f; ap[c][j] = 0.f; }
  float a_b = 0.f, a_loss = 0.f;

  constexpr int U = NCH == 1 ? 4 : 2;                   // rows in flight per half-wave
  const long long nhw = (long long)gridDim.x * 8;
  for (long long r0 = ((long long)blockIdx.x * 8 + hw) * U; r0 < p.M; r0 += nhw * U) {
    ushort8_t hv[U][NCH];
    float y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long r = r0 + u < p.M ? r0 + u : p.M - 1;            // clamped: the loads are unconditional
      y[u] = p.target[r];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = l + 32 * c;
        hv[u][c] = *(const ushort8_t*)(p.h + r * p.ldh + (ch < KC ? ch : 0) * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float hf[NCH][8];
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        unpack8<DT>(hv[u][c], hf[c]);
        if (l + 32 * c < KC) {
#pragma unroll
          for (int j = 0; j < 8; ++j) dot = fmaf(hf[c][j], wv[c][j], dot);
        }
      }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);   // butterfly inside the half-wave
      const bool live = r0 + u < p.M;
      const unsigned short z16 = Elem<DT>::from_f32(dot + b0);
      const float x = Elem<DT>::to_f32(z16);
      const float e = __expf(-fabsf(x));
      const float s = x >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
      const float d = live ? Elem<DT>::to_f32(Elem<DT>::from_f32((s - y[u]) * gs)) : 0.f;
      if (l == 0 && live) {
        a_loss += fmaxf(x, 0.f) - x * y[u] + log1pf(e);
        a_b += d;
        if (p.logits) p.logits[r0 + u] = z16;
      }
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = l + 32 * c;
        float g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          aw[c][j] = fmaf(d, hf[c][j], aw[c][j]);
          g[j] = hf[c][j] > 0.f ? d * wv[c][j] : 0.f;
        }
        const ushort8_t gp = pack8<DT>(g);
        if (p.gprev) {
          float gr[8];
          unpack8<DT>(gp, gr);
#pragma unroll
          for (int j = 0; j < 8; ++j) ap[c][j] += gr[j];
        }
        if (live && ch < KC) *(ushort8_t*)(p.dh + (r0 + u) * p.ldd + ch * 8) = gp;
      }
    }
  }
  // the 8 half-waves of the workgroup meet in LDS, then one partial row [dw | d bias_prev | d bias, loss] per workgroup
  const int RW = 2 * K + 2;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = l + 32 * c;
    if (ch < KC) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[hw * RW + ch * 8 + j] = aw[c][j];
        red[hw * RW + K + ch * 8 + j] = ap[c][j];
      }
    }
  }
  if (l == 0) { red[hw * RW + 2 * K] = a_b; red[hw * RW + 2 * K + 1] = a_loss; }
  __syncthreads();
  float* row = p.ws + (long long)blockIdx.x * RW;
  for (int q = threadIdx.x; q < RW; q += 256) {
    float t = 0.f;
#pragma unroll
    for (int w8 = 0; w8 < 8; ++w8) t += red[w8 * RW + q];
    row[q] = t;
  }
}

// Fold of the partial rows, every sum in index order (bit-reproducible).  A workgroup owns 16 columns; its 256 threads are 16
// row phases x 16 columns, each with 16 independent loads in flight per trip, then the phases meet in LDS in phase order.
// (The fold used to be done by the last workgroup of the main kernel to arrive.  On this chip a device-scope release / acquire
// pair is an L2 write-back + invalidate per workgroup -- 1024 of them stretched the 20 us pass to 169 us; a second launch costs
// 3 us.)
__global__ __launch_bounds__(256) void head_fold_kernel(HeadArgs p, int G) {
  __shared__ float red[16][17];
  const int K = p.K, RW = 2 * K + 2;
  const int cl = threadIdx.x & 15, ph = threadIdx.x >> 4;
  const int q = (int)blockIdx.x * 16 + cl;
  const int qc = q < RW ? q : RW - 1;
  float t = 0.f;
  for (int g0 = ph; g0 < G; g0 += 16 * 16) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int g = g0 + 16 * i;
      v[i] = p.ws[(long long)(g < G ? g : 0) * RW + qc];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) t += g0 + 16 * i < G ? v[i] : 0.f;
  }
  red[ph][cl] = t;
  __syncthreads();
  if (ph == 0 && q < RW) {
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += red[i][cl];
    if (q < K) p.gw[q] = tot;
    else if (q < 2 * K) { if (p.gprev) p.gprev[q - K] = tot; }
    else if (q == 2 * K) p.gb[0] = tot;
    else p.loss[0] = tot / (float)p.M;
  }
}

static int head_grid(long long M) {
  long long g = (M + 8 * 8 - 1) / (8 * 8);               // >= 8 rows per half-wave
  return (int)(g < 1 ? 1 : g > 1024 ? 1024 : g);
}

// workspace bytes for (M, K): the partial rows
extern "C" int64_t dle_head_bce_workspace_bytes(int64_t M, int K) {
  return (int64_t)head_grid(M) * (2 * K + 2) * 4;
}

extern "C" int dle_head_bce_fwd_bwd(const void* h, const void* w16, const float* bias, const float* target,
                                    const float* grad_scale_dev, float* loss_out, void* logits_out, void* dh, float* gw,
                                    float* gb, float* gprev_bias, void* ws, int64_t ws_bytes, int64_t M, int K, int64_t ldh,
                                    int64_t ldd, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(h && w16 && target && loss_out && dh && gw && gb && ws, "head_bce: null pointer");
  DLE_CHECK_ARG(M > 0 && K > 0 && (K % 8) == 0 && K <= 512, "head_bce: K must be a multiple of 8, <= 512 (got %d)", K);
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "head_bce: 16-bit activations only (dtype %d)", dtype);
  DLE_CHECK_ARG((ldh % 8) == 0 && (ldd % 8) == 0 && ((((uintptr_t)h) | ((uintptr_t)dh) | ((uintptr_t)w16)) & 15) == 0,
                "head_bce: 16-byte aligned rows");
  DLE_CHECK_ARG(ws_bytes >= dle_head_bce_workspace_bytes(M, K), "head_bce: workspace too small");
  const int grid = head_grid(M);
  HeadArgs p = {(const unsigned short*)h, (const unsigned short*)w16, bias, target, grad_scale_dev, loss_out,
                (unsigned short*)logits_out, (unsigned short*)dh, gw, gb, gprev_bias, (float*)ws,
                (long long)M, (long long)ldh, (long long)ldd, K};
  const size_t lds = (size_t)8 * (2 * K + 2) * 4;
#define GO(DT, NCH) hipLaunchKernelGGL((head_bce_kernel<DT, NCH>), dim3(grid), dim3(256), lds, stream, p)
#define PICK(DT) do { if (K <= 256) GO(DT, 1); else GO(DT, 2); } while (0)
  if (dtype == DLE_F16) PICK(DLE_F16); else PICK(DLE_BF16);
#undef PICK
#undef GO
  DLE_LAUNCH_CHECK();
  hipLaunchKernelGGL(head_fold_kernel, dim3((2 * K + 2 + 15) / 16), dim3(256), 0, stream, p, grid);
  DLE_LAUNCH_CHECK();
  return 0;
}
