// HBM-bound kernels of the BERT pre-training step for gfx950: fused embedding-sum + LayerNorm, LayerNorm
// (+ residual) forward/backward, masked softmax forward/backward for the attention scores, row gather/scatter
// (dense MLM head, pooler), embedding-gradient scatter.
//
// Replace (paths relative to /root/reference/PyTorch/LanguageModeling/BERT/):
//   modeling.py:285-301   BertEmbeddings.forward   (3 gathers + sum + LayerNorm(eps 1e-12))
//   modeling.py:394-398, 430-434  BertSelfOutput / BertOutput: LayerNorm(dense(x) + residual)
//   modeling.py:354-373   scores / sqrt(d) + mask -> softmax(dim=-1)      (torch.bmm + F.softmax in the reference)
//   modeling.py:587-595   index_select of the masked rows before the MLM head (sequence_output_is_dense)
//   modeling.py:518-524   pooler input = hidden state of token 0
// (ATen / NVFuser kernels in the reference, run_pretraining.py:56-60,419-420).
// Layout: hidden states are [tokens = B*S, H] row-major 16-bit; statistics fp32.  One wavefront per row with
// 16-byte lanes; row reductions are wave64 shuffles; column reductions (gamma/beta gradients) accumulate per
// lane over the rows a workgroup sweeps, meet in LDS and leave through a workspace (no same-address atomics).
#include "common.h"
#include "dropout.h"

template <int DT> __device__ __forceinline__ float tf_up(unsigned short u) { return Elem<DT>::to_f32(u); }
template <int DT> __device__ __forceinline__ unsigned short tf_dn(float f) { return Elem<DT>::from_f32(f); }

static int tf_grid(long long items, int per_block, int cap = 2048) {
  long long g = (items + per_block - 1) / per_block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

#define LN_MAX_CHUNKS 8      // H <= 64 lanes * 8 chunks * 8 elements = 4096

// ------------------------------------------------------------------ LayerNorm forward (+ residual)
// z = x + res (res optional);  y = (z - mean) * rstd * gamma + beta.   z_out (optional) receives z in 16 bits
// (the tensor the backward pass needs); mean / rstd fp32 per row.
template <int DT, int CH>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const unsigned short* __restrict__ x,
                                                     const unsigned short* __restrict__ res,
                                                     unsigned short* __restrict__ z_out, unsigned short* __restrict__ y,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ mean, float* __restrict__ rstd, long long rows,
                                                     int H, float eps, DropArgs drop) {
  drop_resolve(drop);
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
  const int nch = H >> 3;                              // 16-byte chunks per row
  for (long long r = wave; r < rows; r += nwaves) {
    float v[CH][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = lane + i * 64;
      if (c < nch) {
        float xf[8], rf[8];
        unpack8<DT>(*(const ushort8_t*)(x + r * H + c * 8), xf);
        if (drop.mask) {                                             // y = LN(dropout(x) + res)
          const long long chunk = (r * H >> 3) + c;
          const unsigned bits = drop_bits(drop, chunk);
          drop.mask[chunk] = (unsigned char)bits;
#pragma unroll
          for (int k = 0; k < 8; ++k) xf[k] = ((bits >> k) & 1u) ? xf[k] * drop.inv_keep : 0.f;
          unpack8<DT>(pack8<DT>(xf), xf);                            // the dropped tensor is a 16-bit tensor in the reference
        }
        if (res) {
          unpack8<DT>(*(const ushort8_t*)(res + r * H + c * 8), rf);
#pragma unroll
          for (int k = 0; k < 8; ++k) xf[k] += rf[k];
          const ushort8_t zo = pack8<DT>(xf);                      // z is what backward will see (16-bit)
          if (z_out) *(ushort8_t*)(z_out + r * H + c * 8) = zo;
          unpack8<DT>(zo, xf);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[i][k] = xf[k]; s += xf[k]; }
      }
    }
    s = wave_sum(s);
    const float mu = s / (float)H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i)
      if (lane + i * 64 < nch)
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float d = v[i][k] - mu; q += d * d; }
    q = wave_sum(q);
    const float rs = rsqrtf(q / (float)H + eps);
    if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = lane + i * 64;
      if (c < nch) {
        const float4_t g0 = *(const float4_t*)(gamma + c * 8), g1 = *(const float4_t*)(gamma + c * 8 + 4);
        const float4_t b0 = *(const float4_t*)(beta + c * 8), b1 = *(const float4_t*)(beta + c * 8 + 4);
        float of[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float g = k < 4 ? g0[k & 3] : g1[k & 3], b = k < 4 ? b0[k & 3] : b1[k & 3];
          of[k] = (v[i][k] - mu) * rs * g + b;
        }
        *(ushort8_t*)(y + r * H + c * 8) = pack8<DT>(of);
      }
    }
  }
}

static int ln_fwd_launch(const void* x, const void* residual, void* z_out, void* y, const float* gamma, const float* beta,
                         float* mean, float* rstd, int64_t rows, int H, float eps, DropArgs drop, int dtype,
                         hipStream_t stream) {
  const int grid = tf_grid(rows, 4, 4096);
  const int ch = (H / 8 + 63) / 64;           // 16-byte chunks per lane: register arrays are sized for exactly this
#define GO(DT, CH) hipLaunchKernelGGL((ln_fwd_kernel<DT, CH>), dim3(grid), dim3(256), 0, stream, (const unsigned short*)x, (const unsigned short*)residual, (unsigned short*)z_out, (unsigned short*)y, gamma, beta, mean, rstd, (long long)rows, H, eps, drop)
#define PICK(DT) do { if (ch <= 1) GO(DT, 1); else if (ch <= 2) GO(DT, 2); else if (ch <= 4) GO(DT, 4); else GO(DT, 8); } while (0)
  if (dtype == DLE_F16) PICK(DLE_F16); else PICK(DLE_BF16);
#undef GO
#undef PICK
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_layernorm_fwd(const void* x, const void* residual, void* z_out, void* y, const float* gamma,
                                 const float* beta, float* mean, float* rstd, int64_t rows, int H, float eps,
                                 int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "layernorm_fwd: 16-bit activations only");
  DLE_CHECK_ARG(H > 0 && H % 8 == 0 && H <= 64 * LN_MAX_CHUNKS * 8, "layernorm_fwd: H must be a multiple of 8, <= 4096");
  if (rows == 0) return 0;
  DLE_CHECK_ARG(x && y && gamma && beta && mean && rstd, "layernorm_fwd: null pointer");
  return ln_fwd_launch(x, residual, z_out, y, gamma, beta, mean, rstd, rows, H, eps, make_drop(nullptr, 0.f, 0, 0), dtype, stream);
}

// y = LayerNorm(dropout(x) + residual): BertSelfOutput / BertOutput in training mode (modeling.py:394-398,430-434).
// mask receives rows * H / 8 bytes.  p == 0 degenerates to dle_layernorm_fwd (mask all ones).
extern "C" int dle_dropout_add_layernorm_fwd(const void* x, const void* residual, void* z_out, void* y, void* mask,
                                             const float* gamma, const float* beta, float* mean, float* rstd,
                                             int64_t rows, int H, float eps, float p, uint64_t seed, uint64_t offset,
                                             const uint64_t* offset_base,
                                             int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "dropout_add_layernorm_fwd: 16-bit activations only");
  DLE_CHECK_ARG(H > 0 && H % 8 == 0 && H <= 64 * LN_MAX_CHUNKS * 8, "dropout_add_layernorm_fwd: H must be a multiple of 8, <= 4096");
  DLE_CHECK_ARG(p >= 0.f && p < 1.f, "dropout_add_layernorm_fwd: p must be in [0, 1)");
  if (rows == 0) return 0;
  DLE_CHECK_ARG(x && y && gamma && beta && mean && rstd && mask, "dropout_add_layernorm_fwd: null pointer");
  return ln_fwd_launch(x, residual, z_out, y, gamma, beta, mean, rstd, rows, H, eps, make_drop(mask, p, seed, offset, offset_base), dtype, stream);
}

// ------------------------------------------------------------------ LayerNorm backward
// xhat = (z - mean) * rstd;  dz = rstd * (dy*gamma - mean_H(dy*gamma) - xhat * mean_H(dy*gamma*xhat));
// partial[block][0][c] = sum_rows dy (dbeta), partial[block][1][c] = sum_rows dy * xhat (dgamma).
// ZT = dtype of z (16-bit) or fp32 reconstruction is done by the embedding variant below.
// DROP: the LayerNorm input was dense(x) -> dropout -> + residual (modeling.py:394-398, 430-434): the kernel also writes
// dzd = dz * keep / (1 - p), the gradient of the dense output (the separate dropout-backward pass is gone), and a third
// column sum, sum_rows dzd = the dense layer's bias gradient (the separate column-sum pass is gone).
// FULL: H == CH * 512, every chunk of every lane is inside the row -- no conditional stores, so the compiler can COUNT the
// stores issued after the prefetch and wait for the prefetched row alone (vmcnt(n), not vmcnt(0)).
template <int DT, int CH, bool DROP, bool FULL = false>
__global__ __launch_bounds__(512) void ln_bwd_kernel(const unsigned short* __restrict__ dy,
                                                     const unsigned short* __restrict__ z,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, unsigned short* __restrict__ dz,
                                                     float* __restrict__ partial, long long rows, int H,
                                                     const unsigned char* __restrict__ keep, float inv_keep,
                                                     unsigned short* __restrict__ dzd) {
  constexpr int NS = DROP ? 3 : 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* red = (float*)smem_raw;                       // [waves][H]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nwv = blockDim.x >> 6;
  const long long wave = (long long)blockIdx.x * nwv + w, nwaves = (long long)gridDim.x * nwv;
  const int nch = H >> 3;
  float ag[CH][8], ab[CH][8], ad[DROP ? CH : 1][8];
#pragma unroll
  for (int i = 0; i < CH; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) { ag[i][k] = 0.f; ab[i][k] = 0.f; if (DROP) ad[i][k] = 0.f; }
  if constexpr (CH <= 2) {
    // Every load of a row is issued UNCONDITIONALLY (out-of-range chunks / the row past the end read a clamped, valid address and
    // are masked where they are used) and one row AHEAD of its use: with the loads inside `if (c < nch)` blocks hipcc issued
    // chunk i + 1's loads only after chunk i had been reduced, and the first wait of a row also waited for the previous row's
    // stores to be acknowledged (vmcnt retires in order) -- two to three exposed memory round trips per 4 KB row, 3.2 TB/s.
    constexpr bool HOIST = CH <= 2;                 // gamma in registers for the whole sweep (16 VGPRs at H = 1024)
    int cc[CH];
    bool valid[CH];
  #pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = lane + i * 64;
      valid[i] = FULL || c < nch;
      cc[i] = valid[i] ? c : nch - 1;
    }
    float gmh[HOIST ? CH : 1][8];
    if (HOIST) {
  #pragma unroll
      for (int i = 0; i < CH; ++i) {
        const float4_t g0 = *(const float4_t*)(gamma + cc[i] * 8), g1 = *(const float4_t*)(gamma + cc[i] * 8 + 4);
  #pragma unroll
        for (int k = 0; k < 4; ++k) { gmh[HOIST ? i : 0][k] = g0[k]; gmh[HOIST ? i : 0][4 + k] = g1[k]; }
      }
    }
    ushort8_t dyv[CH], zv[CH], dyn[CH], zn[CH];
    unsigned kb[CH], kbn[CH];
    float mu, rs, mun, rsn;
    auto load_row = [&](long long r, ushort8_t (&a)[CH], ushort8_t (&b)[CH], unsigned (&kk)[CH], float& m_, float& r_)
        __attribute__((always_inline)) {
      const long long rr = r < rows ? r : rows - 1;
      m_ = mean[rr]; r_ = rstd[rr];
  #pragma unroll
      for (int i = 0; i < CH; ++i) {
        a[i] = *(const ushort8_t*)(dy + rr * H + cc[i] * 8);
        b[i] = *(const ushort8_t*)(z + rr * H + cc[i] * 8);
        kk[i] = DROP ? (unsigned)keep[rr * nch + cc[i]] : 0u;
      }
    };
    if (wave < rows) load_row(wave, dyv, zv, kb, mu, rs);
    for (long long r = wave; r < rows; r += nwaves) {
      load_row(r + nwaves, dyn, zn, kbn, mun, rsn);          // the next row of this wave (clamped at the end of the sweep)
      __builtin_amdgcn_sched_barrier(0);                     // (the loads stay AHEAD of this row's stores: in-order vmcnt)
      float g[CH][8], xh[CH][8];
      float s1 = 0.f, s2 = 0.f;
  #pragma unroll
      for (int i = 0; i < CH; ++i) {
        float df[8], zf[8], gm[8];
        unpack8<DT>(dyv[i], df);
        unpack8<DT>(zv[i], zf);
        if (HOIST) {
  #pragma unroll
          for (int k = 0; k < 8; ++k) gm[k] = gmh[HOIST ? i : 0][k];
        } else {
          const float4_t g0 = *(const float4_t*)(gamma + cc[i] * 8), g1 = *(const float4_t*)(gamma + cc[i] * 8 + 4);
  #pragma unroll
          for (int k = 0; k < 4; ++k) { gm[k] = g0[k]; gm[4 + k] = g1[k]; }
        }
        const float vm = valid[i] ? 1.f : 0.f;
  #pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float d = df[k] * vm;
          const float xx = (zf[k] - mu) * rs;
          ab[i][k] += d;
          ag[i][k] += d * xx;
          g[i][k] = d * gm[k];
          xh[i][k] = xx;
          s1 += g[i][k];
          s2 += g[i][k] * xx;
        }
      }
      s1 = wave_sum(s1) / (float)H;
      s2 = wave_sum(s2) / (float)H;
  #pragma unroll
      for (int i = 0; i < CH; ++i) {
        if (valid[i]) {
          const int c = lane + i * 64;
          float of[8];
  #pragma unroll
          for (int k = 0; k < 8; ++k) of[k] = rs * (g[i][k] - s1 - xh[i][k] * s2);
          const ushort8_t ov = pack8<DT>(of);
          *(ushort8_t*)(dz + r * H + c * 8) = ov;
          if (DROP) {
            // what dle_dropout_bwd computed from the ROUNDED dz: dz16 * keep / (1 - p), rounded; its column sum = bias gradient
            const unsigned bits = kb[i];
            float rf[8], df[8];
            unpack8<DT>(ov, rf);
  #pragma unroll
            for (int k = 0; k < 8; ++k) df[k] = ((bits >> k) & 1u) ? rf[k] * inv_keep : 0.f;
            const ushort8_t dv = pack8<DT>(df);
            *(ushort8_t*)(dzd + r * H + c * 8) = dv;
            unpack8<DT>(dv, df);
  #pragma unroll
            for (int k = 0; k < 8; ++k) ad[i][k] += df[k];
          }
        }
      }
      mu = mun; rs = rsn;
  #pragma unroll
      for (int i = 0; i < CH; ++i) { dyv[i] = dyn[i]; zv[i] = zn[i]; kb[i] = kbn[i]; }
    }
  } else {
  // wide rows (H > 1024): one row at a time (the prefetched form needs 2 x CH x 9 more registers and spills)
    for (long long r = wave; r < rows; r += nwaves) {
      const float mu = mean[r], rs = rstd[r];
      float g[CH][8], xh[CH][8];
      unsigned kb[CH];                  // keep bits, requested WITH the row's other loads: fetched after the row reductions
                                        // they exposed a second memory latency per row (the fused pass ran at 3.2 TB/s)
      float s1 = 0.f, s2 = 0.f;
  #pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int c = lane + i * 64;
        kb[i] = 0;
        if (DROP && c < nch) kb[i] = keep[r * nch + c];
        if (c < nch) {
          float df[8], zf[8];
          unpack8<DT>(*(const ushort8_t*)(dy + r * H + c * 8), df);
          unpack8<DT>(*(const ushort8_t*)(z + r * H + c * 8), zf);
          const float4_t g0 = *(const float4_t*)(gamma + c * 8), g1 = *(const float4_t*)(gamma + c * 8 + 4);
  #pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float d = df[k];
            const float xx = (zf[k] - mu) * rs;
            const float gm = k < 4 ? g0[k & 3] : g1[k & 3];
            ab[i][k] += d;
            ag[i][k] += d * xx;
            g[i][k] = d * gm;
            xh[i][k] = xx;
            s1 += g[i][k];
            s2 += g[i][k] * xx;
          }
        }
      }
      s1 = wave_sum(s1) / (float)H;
      s2 = wave_sum(s2) / (float)H;
  #pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
          float of[8];
  #pragma unroll
          for (int k = 0; k < 8; ++k) of[k] = rs * (g[i][k] - s1 - xh[i][k] * s2);
          const ushort8_t ov = pack8<DT>(of);
          *(ushort8_t*)(dz + r * H + c * 8) = ov;
          if (DROP) {
            // what dle_dropout_bwd computed from the ROUNDED dz: dz16 * keep / (1 - p), rounded; its column sum = bias gradient
            const unsigned bits = kb[i];
            float rf[8], df[8];
            unpack8<DT>(ov, rf);
  #pragma unroll
            for (int k = 0; k < 8; ++k) df[k] = ((bits >> k) & 1u) ? rf[k] * inv_keep : 0.f;
            const ushort8_t dv = pack8<DT>(df);
            *(ushort8_t*)(dzd + r * H + c * 8) = dv;
            unpack8<DT>(dv, df);
  #pragma unroll
            for (int k = 0; k < 8; ++k) ad[i][k] += df[k];
          }
        }
      }
    }
  }
  // column partials: waves of the block meet in LDS, one statistic at a time ([waves][H] fp32: 32 KiB at H = 1024 -- a
  // buffer for all statistics at once halved the resident workgroups and the kernel fell from 5.1 to 3.2 TB/s)
#pragma unroll
  for (int which = 0; which < NS; ++which) {
    if (which) __syncthreads();
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = lane + i * 64;
      if (c < nch)
#pragma unroll
        for (int k = 0; k < 8; ++k)
          red[w * H + c * 8 + k] = which == 0 ? ab[i][k] : which == 1 ? ag[i][k] : ad[DROP ? i : 0][k];
    }
    __syncthreads();
    for (int col = threadIdx.x; col < H; col += blockDim.x) {
      float t = 0.f;
      for (int q = 0; q < nwv; ++q) t += red[q * H + col];
      partial[((long long)blockIdx.x * NS + which) * H + col] = t;
    }
  }
}

// dgamma / dbeta = sum over blocks of the partials (fp64 accumulation).  16 columns x 16 group slices per
// workgroup: the pass is latency-bound, so it is spread over C/16 workgroups with 4 loads in flight per lane.
__global__ __launch_bounds__(256) void colpair_finish_kernel(const float* __restrict__ partial, int groups, int C,
                                                             float* __restrict__ out1, float* __restrict__ out0,
                                                             int accumulate, int NS = 2, float* __restrict__ out2 = nullptr) {
  // out2 (NS == 3): the third statistic of the fused dropout + LayerNorm backward (the dense layer's bias gradient) in the
  // same launch -- its own finishing kernel was 48 launches x 12 us per BERT step
  __shared__ double red[3][256];
  const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const bool third = out2 != nullptr && NS == 3;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  if (c < C) {
    int g = sl;
    for (; g + 48 < groups; g += 64) {
      float a[4], b[4], d[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = partial[((long long)(g + 16 * u) * NS) * C + c];
        b[u] = partial[((long long)(g + 16 * u) * NS + 1) * C + c];
        d[u] = partial[((long long)(g + 16 * u) * NS + (third ? 2 : 0)) * C + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { s0 += a[u]; s1 += b[u]; s2 += d[u]; }
    }
    for (; g < groups; g += 16) {
      s0 += partial[((long long)g * NS) * C + c];
      s1 += partial[((long long)g * NS + 1) * C + c];
      s2 += partial[((long long)g * NS + (third ? 2 : 0)) * C + c];
    }
  }
  red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = s2;
  __syncthreads();
  if (sl == 0 && c < C) {
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) { t0 += red[0][q * 16 + cl]; t1 += red[1][q * 16 + cl]; t2 += red[2][q * 16 + cl]; }
    out0[c] = accumulate ? out0[c] + (float)t0 : (float)t0;
    out1[c] = accumulate ? out1[c] + (float)t1 : (float)t1;
    if (third) out2[c] = accumulate ? out2[c] + (float)t2 : (float)t2;
  }
}

#define LN_BWD_BLOCKS 512
extern "C" int64_t dle_layernorm_workspace_bytes(int H) { return (int64_t)LN_BWD_BLOCKS * 3 * H * 4; }

static int ln_bwd_launch(const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma, void* dz,
                         float* dgamma, float* dbeta, int64_t rows, int H, int accumulate, void* workspace,
                         int64_t workspace_bytes, const void* keep, float p, void* dzd, float* dbias, int dtype,
                         hipStream_t stream) {
  const bool drop = keep != nullptr;
  const int ns = drop ? 3 : 2;
  const int nwv = H <= 1024 ? 8 : H <= 2048 ? 4 : 2;          // waves per workgroup: [waves][H] fp32 <= 32 KiB of LDS
  // (4-wave workgroups for more resident waves measured SLOWER: 4.2 -> 4.8 ms / step)
  int blocks = (int)((rows + nwv - 1) / nwv);
  if (blocks > LN_BWD_BLOCKS) blocks = LN_BWD_BLOCKS;
  DLE_CHECK_ARG(workspace_bytes >= (long long)blocks * ns * H * 4, "layernorm_bwd: workspace too small");
  const size_t lds = (size_t)nwv * H * 4;
  const int ch = (H / 8 + 63) / 64;
  const DropArgs d = make_drop(nullptr, p, 0, 0);
#define GO(DT, CH, DR, FL) hipLaunchKernelGGL((ln_bwd_kernel<DT, CH, DR, FL>), dim3(blocks), dim3(64 * nwv), lds, stream, (const unsigned short*)dy, (const unsigned short*)z, mean, rstd, gamma, (unsigned short*)dz, (float*)workspace, (long long)rows, H, (const unsigned char*)keep, d.inv_keep, (unsigned short*)dzd)
#define PICK(DT, DR) do { if (H == 1024) GO(DT, 2, DR, true); else if (H == 512) GO(DT, 1, DR, true); else if (ch <= 1) GO(DT, 1, DR, false); \
    else if (ch <= 2) GO(DT, 2, DR, false); else if (ch <= 4) GO(DT, 4, DR, false); else GO(DT, 8, DR, false); } while (0)
  if (drop) { if (dtype == DLE_F16) PICK(DLE_F16, true); else PICK(DLE_BF16, true); }
  else { if (dtype == DLE_F16) PICK(DLE_F16, false); else PICK(DLE_BF16, false); }
#undef GO
#undef PICK
  DLE_LAUNCH_CHECK();
  hipLaunchKernelGGL(colpair_finish_kernel, dim3((H + 15) / 16), dim3(256), 0, stream, (const float*)workspace, blocks, H,
                     dgamma, dbeta, accumulate, ns, (drop && dbias) ? dbias : (float*)nullptr);
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_layernorm_bwd(const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma,
                                 void* dz, float* dgamma, float* dbeta, int64_t rows, int H, int accumulate,
                                 void* workspace, int64_t workspace_bytes, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "layernorm_bwd: 16-bit activations only");
  DLE_CHECK_ARG(H > 0 && H % 8 == 0 && H <= 64 * LN_MAX_CHUNKS * 8, "layernorm_bwd: H must be a multiple of 8, <= 4096");
  DLE_CHECK_ARG(rows > 0 && dy && z && mean && rstd && gamma && dz && dgamma && dbeta && workspace, "layernorm_bwd: null pointer / empty");
  return ln_bwd_launch(dy, z, mean, rstd, gamma, dz, dgamma, dbeta, rows, H, accumulate, workspace, workspace_bytes, nullptr,
                       0.f, nullptr, nullptr, dtype, stream);
}

// Backward of y = LayerNorm(dropout(x) + residual) (dle_dropout_add_layernorm_fwd): dz (the gradient of the residual
// branch), dx = dz * keep / (1 - p) (the gradient of the dense layer's output) and, when dbias != NULL, dbias (+)= column
// sums of dx -- the dropout-backward pass and the bias-gradient pass of the dense layer folded into this one.
extern "C" int dle_dropout_add_layernorm_bwd(const void* dy, const void* z, const float* mean, const float* rstd,
                                             const float* gamma, const void* keep_mask, float p, void* dz, void* dx,
                                             float* dgamma, float* dbeta, float* dbias, int64_t rows, int H,
                                             int accumulate, void* workspace, int64_t workspace_bytes, int dtype,
                                             hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "dropout_add_layernorm_bwd: 16-bit activations only");
  DLE_CHECK_ARG(H > 0 && H % 8 == 0 && H <= 64 * LN_MAX_CHUNKS * 8, "dropout_add_layernorm_bwd: H must be a multiple of 8, <= 4096");
  DLE_CHECK_ARG(rows > 0 && dy && z && mean && rstd && gamma && keep_mask && dz && dx && dgamma && dbeta && workspace,
                "dropout_add_layernorm_bwd: null pointer / empty");
  DLE_CHECK_ARG(p >= 0.f && p < 1.f, "dropout_add_layernorm_bwd: p must be in [0, 1)");
  return ln_bwd_launch(dy, z, mean, rstd, gamma, dz, dgamma, dbeta, rows, H, accumulate, workspace, workspace_bytes, keep_mask,
                       p, dx, dbias, dtype, stream);
}

// ------------------------------------------------------------------ embeddings: gather-sum (fp32) -> 16-bit z
// z[t] = word[ids[t]] + pos[t mod S] + type[tt[t]]   (fp32 tables; the LayerNorm that follows reads z)
template <int DT>
__global__ __launch_bounds__(256) void embed_sum_kernel(const float* __restrict__ word, const float* __restrict__ pos,
                                                        const float* __restrict__ type, const long long* __restrict__ ids,
                                                        const long long* __restrict__ tt, unsigned short* __restrict__ z,
                                                        long long tokens, int S, int H) {
  const int nch = H >> 3;
  const long long total = tokens * nch;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long t = i / nch;
    const int c = (int)(i - t * nch) * 8;
    const float* w = word + ids[t] * H + c;
    const float* p = pos + (t % S) * H + c;
    const float* y = type + tt[t] * H + c;
    ushort8_t o;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float4_t a = *(const float4_t*)(w + 4 * h), b = *(const float4_t*)(p + 4 * h), d = *(const float4_t*)(y + 4 * h);
#pragma unroll
      for (int k = 0; k < 4; ++k) o[4 * h + k] = tf_dn<DT>(a[k] + b[k] + d[k]);
    }
    *(ushort8_t*)(z + t * H + c) = o;
  }
}

extern "C" int dle_embed_sum(const float* word, const float* pos, const float* type, const int64_t* ids,
                             const int64_t* token_type, void* z, int64_t tokens, int S, int H, int dtype,
                             hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "embed_sum: 16-bit output only");
  DLE_CHECK_ARG(H % 8 == 0 && S > 0, "embed_sum: bad shape");
  if (tokens == 0) return 0;
  DLE_CHECK_ARG(word && pos && type && ids && token_type && z, "embed_sum: null pointer");
  const int grid = tf_grid(tokens * (H / 8), 256);
  if (dtype == DLE_F16) hipLaunchKernelGGL(embed_sum_kernel<DLE_F16>, dim3(grid), dim3(256), 0, stream, word, pos, type, (const long long*)ids, (const long long*)token_type, (unsigned short*)z, (long long)tokens, S, H);
  else hipLaunchKernelGGL(embed_sum_kernel<DLE_BF16>, dim3(grid), dim3(256), 0, stream, word, pos, type, (const long long*)ids, (const long long*)token_type, (unsigned short*)z, (long long)tokens, S, H);
  DLE_LAUNCH_CHECK();
  return 0;
}

// word-embedding gradient: gw[ids[t], :] += dz[t, :]  (fp32 atomics, once per step);
// type gradient selector sums are done with dle_rows_select_sum below, position gradient with dle_colsum.
template <int DT>
__global__ __launch_bounds__(256) void embed_scatter_kernel(const unsigned short* __restrict__ dz,
                                                            const long long* __restrict__ ids, float* __restrict__ gw,
                                                            long long tokens, int H) {
  const int nch = H >> 3;
  const long long total = tokens * nch;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long t = i / nch;
    const int c = (int)(i - t * nch) * 8;
    const ushort8_t v = *(const ushort8_t*)(dz + t * H + c);
    float* dst = gw + ids[t] * H + c;
#pragma unroll
    for (int k = 0; k < 8; ++k) unsafeAtomicAdd(dst + k, tf_up<DT>(v[k]));
  }
}

extern "C" int dle_embed_scatter_add(const void* dz, const int64_t* ids, float* grad_word, int64_t tokens, int H,
                                     int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "embed_scatter_add: 16-bit gradients only");
  DLE_CHECK_ARG(H % 8 == 0, "embed_scatter_add: H must be a multiple of 8");
  if (tokens == 0) return 0;
  DLE_CHECK_ARG(dz && ids && grad_word, "embed_scatter_add: null pointer");
  const int grid = tf_grid(tokens * (H / 8), 256);
  if (dtype == DLE_F16) hipLaunchKernelGGL(embed_scatter_kernel<DLE_F16>, dim3(grid), dim3(256), 0, stream, (const unsigned short*)dz, (const long long*)ids, grad_word, (long long)tokens, H);
  else hipLaunchKernelGGL(embed_scatter_kernel<DLE_BF16>, dim3(grid), dim3(256), 0, stream, (const unsigned short*)dz, (const long long*)ids, grad_word, (long long)tokens, H);
  DLE_LAUNCH_CHECK();
  return 0;
}

// out[k][c] = sum over rows t with sel[t] == k of x[t][c], k in [0, K) (K small: token types).  Partials via workspace.
template <int DT, int KMAX>
__global__ __launch_bounds__(256) void rows_select_sum_kernel(const unsigned short* __restrict__ x,
                                                              const long long* __restrict__ sel, float* __restrict__ partial,
                                                              long long rows, int H, int K, long long rows_per_block) {
  const int nch = H >> 3;
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  long long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= nch) return;
  float acc[KMAX][8];
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[k][e] = 0.f;
  for (long long r = r0; r < r1; ++r) {
    const int s = (int)sel[r];
    const ushort8_t v = *(const ushort8_t*)(x + r * H + c * 8);
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k == s)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[k][e] += tf_up<DT>(v[e]);
  }
  for (int k = 0; k < K && k < KMAX; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) partial[((long long)blockIdx.y * K + k) * H + c * 8 + e] = acc[k][e];
}

__global__ __launch_bounds__(256) void select_sum_finish_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                                int groups, int KH, int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= KH) return;
  float s = accumulate ? out[i] : 0.f;
  for (int g = 0; g < groups; ++g) s += partial[(long long)g * KH + i];
  out[i] = s;
}

#define SELSUM_GROUPS 128
extern "C" int dle_rows_select_sum(const void* x, const int64_t* sel, float* out, int64_t rows, int H, int K,
                                   int accumulate, void* workspace, int64_t workspace_bytes, int dtype,
                                   hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "rows_select_sum: 16-bit input only");
  DLE_CHECK_ARG(H % 8 == 0 && K >= 1 && K <= 4, "rows_select_sum: K must be 1..4, H a multiple of 8");
  DLE_CHECK_ARG(rows > 0 && x && sel && out && workspace, "rows_select_sum: null pointer / empty");
  long long rpb = (rows + SELSUM_GROUPS - 1) / SELSUM_GROUPS;
  const int gy = (int)((rows + rpb - 1) / rpb);
  DLE_CHECK_ARG(workspace_bytes >= (long long)gy * K * H * 4, "rows_select_sum: workspace too small");
  dim3 grid((H / 8 + 255) / 256, gy), block(256);
  if (dtype == DLE_F16) hipLaunchKernelGGL((rows_select_sum_kernel<DLE_F16, 4>), grid, block, 0, stream, (const unsigned short*)x, (const long long*)sel, (float*)workspace, (long long)rows, H, K, rpb);
  else hipLaunchKernelGGL((rows_select_sum_kernel<DLE_BF16, 4>), grid, block, 0, stream, (const unsigned short*)x, (const long long*)sel, (float*)workspace, (long long)rows, H, K, rpb);
  DLE_LAUNCH_CHECK();
  hipLaunchKernelGGL(select_sum_finish_kernel, dim3((K * H + 255) / 256), dim3(256), 0, stream, (const float*)workspace, out, gy, K * H, accumulate);
  DLE_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ row gather / scatter (16-bit rows)
// gather:  dst[i, :] = src[idx[i], :]          scatter: dst[idx[i], :] (+)= src[i, :]   (idx unique per call)
template <int ADD, int DT>
__global__ __launch_bounds__(256) void rows_move_kernel(const unsigned short* __restrict__ src,
                                                        const long long* __restrict__ idx, unsigned short* __restrict__ dst,
                                                        long long n, int H, int gather) {
  const int nch = H >> 3;
  const long long total = n * nch;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / nch;
    const int c = (int)(i - r * nch) * 8;
    const long long far = idx[r];
    const unsigned short* s = gather ? src + far * H + c : src + r * H + c;
    unsigned short* d = gather ? dst + r * H + c : dst + far * H + c;
    ushort8_t v = *(const ushort8_t*)s;
    if (ADD) {
      const ushort8_t o = *(const ushort8_t*)d;
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = tf_dn<DT>(tf_up<DT>(v[k]) + tf_up<DT>(o[k]));
    }
    *(ushort8_t*)d = v;
  }
}

extern "C" int dle_rows_gather(const void* src, const int64_t* idx, void* dst, int64_t n, int H, int dtype,
                               hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "rows_gather: 16-bit rows only");
  DLE_CHECK_ARG(H % 8 == 0, "rows_gather: H must be a multiple of 8");
  if (n == 0) return 0;
  DLE_CHECK_ARG(src && idx && dst, "rows_gather: null pointer");
  hipLaunchKernelGGL((rows_move_kernel<0, DLE_F16>), dim3(tf_grid(n * (H / 8), 256)), dim3(256), 0, stream,
                     (const unsigned short*)src, (const long long*)idx, (unsigned short*)dst, (long long)n, H, 1);
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_rows_scatter(const void* src, const int64_t* idx, void* dst, int64_t n, int H, int accumulate,
                                int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "rows_scatter: 16-bit rows only");
  DLE_CHECK_ARG(H % 8 == 0, "rows_scatter: H must be a multiple of 8");
  if (n == 0) return 0;
  DLE_CHECK_ARG(src && idx && dst, "rows_scatter: null pointer");
  const int grid = tf_grid(n * (H / 8), 256);
  if (!accumulate) hipLaunchKernelGGL((rows_move_kernel<0, DLE_F16>), dim3(grid), dim3(256), 0, stream, (const unsigned short*)src, (const long long*)idx, (unsigned short*)dst, (long long)n, H, 0);
  else if (dtype == DLE_F16) hipLaunchKernelGGL((rows_move_kernel<1, DLE_F16>), dim3(grid), dim3(256), 0, stream, (const unsigned short*)src, (const long long*)idx, (unsigned short*)dst, (long long)n, H, 0);
  else hipLaunchKernelGGL((rows_move_kernel<1, DLE_BF16>), dim3(grid), dim3(256), 0, stream, (const unsigned short*)src, (const long long*)idx, (unsigned short*)dst, (long long)n, H, 0);
  DLE_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ masked softmax over attention scores
// scores [batch*heads][S][L] 16-bit in place:  p = softmax(s * scale + mask_add[b][col]) along L.
// L/8 lanes per row (L = 128 -> 16 lanes, 4 rows per wavefront); mask_add fp32 [batch][L] (0 / -10000).
template <int DT>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(unsigned short* __restrict__ s, const float* __restrict__ mask_add,
                                                          long long rows, int L, int rows_per_batch, float scale,
                                                          unsigned short* __restrict__ dropped, DropArgs drop) {
  drop_resolve(drop);
  const int lpr = L >> 3;
  const int rpw = 64 / lpr;
  const int lane = threadIdx.x & 63, sub = lane / lpr, cl = lane % lpr;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
  for (long long r0 = wave * rpw; r0 < rows; r0 += nwaves * rpw) {
    const long long r = r0 + sub;
    const bool ok = r < rows;
    float v[8];
    float mx = -INFINITY;
    if (ok) {
      unpack8<DT>(*(const ushort8_t*)(s + r * L + cl * 8), v);
      const float* m = mask_add ? mask_add + (r / rows_per_batch) * L + cl * 8 : nullptr;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v[k] = v[k] * scale + (m ? m[k] : 0.f);
        mx = fmaxf(mx, v[k]);
      }
    }
    for (int o = lpr >> 1; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    if (ok)
#pragma unroll
      for (int k = 0; k < 8; ++k) { v[k] = __expf(v[k] - mx); sum += v[k]; }
    for (int o = lpr >> 1; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (ok) {
      const float inv = 1.0f / sum;
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] *= inv;
      const ushort8_t pv = pack8<DT>(v);
      *(ushort8_t*)(s + r * L + cl * 8) = pv;
      if (drop.mask) {                                   // second output: dropout(probs) for the P V contraction
        const long long chunk = (r * L >> 3) + cl;
        const unsigned bits = drop_bits(drop, chunk);
        drop.mask[chunk] = (unsigned char)bits;
        unpack8<DT>(pv, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = ((bits >> k) & 1u) ? v[k] * drop.inv_keep : 0.f;
        *(ushort8_t*)(dropped + r * L + cl * 8) = pack8<DT>(v);
      }
    }
  }
}

// dS = P * (dP - sum_j dP_j P_j) * scale, written over dP
template <int DT>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const unsigned short* __restrict__ p, unsigned short* __restrict__ dp,
                                                          long long rows, int L, float scale,
                                                          const unsigned char* __restrict__ mask, float inv_keep) {
  const int lpr = L >> 3;
  const int rpw = 64 / lpr;
  const int lane = threadIdx.x & 63, sub = lane / lpr, cl = lane % lpr;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
  for (long long r0 = wave * rpw; r0 < rows; r0 += nwaves * rpw) {
    const long long r = r0 + sub;
    const bool ok = r < rows;
    float pv[8], gv[8];
    float dot = 0.f;
    if (ok) {
      unpack8<DT>(*(const ushort8_t*)(p + r * L + cl * 8), pv);
      unpack8<DT>(*(const ushort8_t*)(dp + r * L + cl * 8), gv);
      if (mask) {                                        // gradient through dropout(probs) first
        const unsigned bits = mask[(r * L >> 3) + cl];
#pragma unroll
        for (int k = 0; k < 8; ++k) gv[k] = ((bits >> k) & 1u) ? gv[k] * inv_keep : 0.f;
        unpack8<DT>(pack8<DT>(gv), gv);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) dot += pv[k] * gv[k];
    }
    for (int o = lpr >> 1; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    if (ok) {
      float of[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) of[k] = pv[k] * (gv[k] - dot) * scale;
      *(ushort8_t*)(dp + r * L + cl * 8) = pack8<DT>(of);
    }
  }
}

static bool softmax_len_ok(int L) { return L == 8 || L == 16 || L == 32 || L == 64 || L == 128 || L == 256 || L == 512; }

extern "C" int dle_softmax_fwd(void* scores, const float* mask_add, int64_t rows, int L, int rows_per_batch, float scale,
                               int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "softmax_fwd: 16-bit scores only");
  DLE_CHECK_ARG(softmax_len_ok(L), "softmax_fwd: row length %d must be a power of two in [8, 512]", L);
  if (rows == 0) return 0;
  DLE_CHECK_ARG(scores && rows_per_batch > 0, "softmax_fwd: bad arguments");
  const int grid = tf_grid(rows, 4 * (64 / (L / 8)), 4096);
  const DropArgs nodrop = make_drop(nullptr, 0.f, 0, 0);
  if (dtype == DLE_F16) hipLaunchKernelGGL(softmax_fwd_kernel<DLE_F16>, dim3(grid), dim3(256), 0, stream, (unsigned short*)scores, mask_add, (long long)rows, L, rows_per_batch, scale, (unsigned short*)nullptr, nodrop);
  else hipLaunchKernelGGL(softmax_fwd_kernel<DLE_BF16>, dim3(grid), dim3(256), 0, stream, (unsigned short*)scores, mask_add, (long long)rows, L, rows_per_batch, scale, (unsigned short*)nullptr, nodrop);
  DLE_LAUNCH_CHECK();
  return 0;
}

// probs = softmax(scores * scale + mask) in place, dropped = dropout(probs) (the P V operand), mask: rows * L / 8 bytes
// (modeling.py:366-370: attention_probs = self.dropout(self.softmax(attention_scores)))
extern "C" int dle_softmax_dropout_fwd(void* scores, void* dropped, void* mask, const float* mask_add, int64_t rows, int L,
                                       int rows_per_batch, float scale, float p, uint64_t seed, uint64_t offset,
                                       const uint64_t* offset_base,
                                       int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "softmax_dropout_fwd: 16-bit scores only");
  DLE_CHECK_ARG(softmax_len_ok(L), "softmax_dropout_fwd: row length %d must be a power of two in [8, 512]", L);
  DLE_CHECK_ARG(p >= 0.f && p < 1.f, "softmax_dropout_fwd: p must be in [0, 1)");
  if (rows == 0) return 0;
  DLE_CHECK_ARG(scores && dropped && mask && rows_per_batch > 0, "softmax_dropout_fwd: bad arguments");
  const int grid = tf_grid(rows, 4 * (64 / (L / 8)), 4096);
  const DropArgs d = make_drop(mask, p, seed, offset, offset_base);
  if (dtype == DLE_F16) hipLaunchKernelGGL(softmax_fwd_kernel<DLE_F16>, dim3(grid), dim3(256), 0, stream, (unsigned short*)scores, mask_add, (long long)rows, L, rows_per_batch, scale, (unsigned short*)dropped, d);
  else hipLaunchKernelGGL(softmax_fwd_kernel<DLE_BF16>, dim3(grid), dim3(256), 0, stream, (unsigned short*)scores, mask_add, (long long)rows, L, rows_per_batch, scale, (unsigned short*)dropped, d);
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_softmax_bwd(const void* probs, void* dprobs, int64_t rows, int L, float scale, int dtype,
                               hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "softmax_bwd: 16-bit scores only");
  DLE_CHECK_ARG(softmax_len_ok(L), "softmax_bwd: row length %d must be a power of two in [8, 512]", L);
  if (rows == 0) return 0;
  DLE_CHECK_ARG(probs && dprobs, "softmax_bwd: null pointer");
  const int grid = tf_grid(rows, 4 * (64 / (L / 8)), 4096);
  if (dtype == DLE_F16) hipLaunchKernelGGL(softmax_bwd_kernel<DLE_F16>, dim3(grid), dim3(256), 0, stream, (const unsigned short*)probs, (unsigned short*)dprobs, (long long)rows, L, scale, (const unsigned char*)nullptr, 1.0f);
  else hipLaunchKernelGGL(softmax_bwd_kernel<DLE_BF16>, dim3(grid), dim3(256), 0, stream, (const unsigned short*)probs, (unsigned short*)dprobs, (long long)rows, L, scale, (const unsigned char*)nullptr, 1.0f);
  DLE_LAUNCH_CHECK();
  return 0;
}

// dS = P * (g - sum_j g_j P_j) * scale with g = dropout_backward(dP) (mask bits of dle_softmax_dropout_fwd), over dP
extern "C" int dle_softmax_dropout_bwd(const void* probs, void* dprobs, const void* mask, int64_t rows, int L, float scale,
                                       float p, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "softmax_dropout_bwd: 16-bit scores only");
  DLE_CHECK_ARG(softmax_len_ok(L), "softmax_dropout_bwd: row length %d must be a power of two in [8, 512]", L);
  if (rows == 0) return 0;
  DLE_CHECK_ARG(probs && dprobs && mask, "softmax_dropout_bwd: null pointer");
  const int grid = tf_grid(rows, 4 * (64 / (L / 8)), 4096);
  const DropArgs d = make_drop(nullptr, p, 0, 0);
  if (dtype == DLE_F16) hipLaunchKernelGGL(softmax_bwd_kernel<DLE_F16>, dim3(grid), dim3(256), 0, stream, (const unsigned short*)probs, (unsigned short*)dprobs, (long long)rows, L, scale, (const unsigned char*)mask, d.inv_keep);
  else hipLaunchKernelGGL(softmax_bwd_kernel<DLE_BF16>, dim3(grid), dim3(256), 0, stream, (const unsigned short*)probs, (unsigned short*)dprobs, (long long)rows, L, scale, (const unsigned char*)mask, d.inv_keep);
  DLE_LAUNCH_CHECK();
  return 0;
}


// ------------------------------------------------------------------ standalone dropout (embedding output, modeling.py:296)
template <int DT, bool BWD>
__global__ __launch_bounds__(256) void dropout_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y,
                                                      unsigned char* __restrict__ mask, long long chunks, DropArgs drop) {
  drop_resolve(drop);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += (long long)gridDim.x * blockDim.x) {
    float v[8];
    unpack8<DT>(((const ushort8_t*)x)[i], v);
    unsigned bits;
    if (BWD) bits = mask[i];
    else { bits = drop_bits(drop, i); mask[i] = (unsigned char)bits; }
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = ((bits >> k) & 1u) ? v[k] * drop.inv_keep : 0.f;
    ((ushort8_t*)y)[i] = pack8<DT>(v);
  }
}

// y = dropout(x): n elements (multiple of 8), mask n / 8 bytes.  x == y is allowed.
extern "C" int dle_dropout_fwd(const void* x, void* y, void* mask, int64_t n, float p, uint64_t seed, uint64_t offset,
                               const uint64_t* offset_base,
                               int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "dropout_fwd: 16-bit tensors only");
  DLE_CHECK_ARG(n >= 0 && n % 8 == 0, "dropout_fwd: n must be a multiple of 8");
  DLE_CHECK_ARG(p >= 0.f && p < 1.f, "dropout_fwd: p must be in [0, 1)");
  if (n == 0) return 0;
  DLE_CHECK_ARG(x && y && mask, "dropout_fwd: null pointer");
  const DropArgs d = make_drop(mask, p, seed, offset, offset_base);
  const int grid = tf_grid(n / 8, 256);
  if (dtype == DLE_F16) hipLaunchKernelGGL((dropout_kernel<DLE_F16, false>), dim3(grid), dim3(256), 0, stream, (const unsigned short*)x, (unsigned short*)y, (unsigned char*)mask, (long long)(n / 8), d);
  else hipLaunchKernelGGL((dropout_kernel<DLE_BF16, false>), dim3(grid), dim3(256), 0, stream, (const unsigned short*)x, (unsigned short*)y, (unsigned char*)mask, (long long)(n / 8), d);
  DLE_LAUNCH_CHECK();
  return 0;
}

// dx = dy * mask / (1 - p)   (dy == dx allowed)
extern "C" int dle_dropout_bwd(const void* dy, const void* mask, void* dx, int64_t n, float p, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "dropout_bwd: 16-bit tensors only");
  DLE_CHECK_ARG(n >= 0 && n % 8 == 0, "dropout_bwd: n must be a multiple of 8");
  if (n == 0) return 0;
  DLE_CHECK_ARG(dy && dx && mask, "dropout_bwd: null pointer");
  const DropArgs d = make_drop(nullptr, p, 0, 0);
  const int grid = tf_grid(n / 8, 256);
  if (dtype == DLE_F16) hipLaunchKernelGGL((dropout_kernel<DLE_F16, true>), dim3(grid), dim3(256), 0, stream, (const unsigned short*)dy, (unsigned short*)dx, (unsigned char*)mask, (long long)(n / 8), d);
  else hipLaunchKernelGGL((dropout_kernel<DLE_BF16, true>), dim3(grid), dim3(256), 0, stream, (const unsigned short*)dy, (unsigned short*)dx, (unsigned char*)mask, (long long)(n / 8), d);
  DLE_LAUNCH_CHECK();
  return 0;
}
