// BatchNorm backward (second pass) on the operand load of the 1x1 data gradient that consumes it, for gfx950 -- the backward
// counterpart of conv_bnload.hip, for the conv3 / bn3 unit of a ResNet bottleneck (Classification/ConvNets/image_classification/
// models/resnet.py:148-175, backward of out = bn3(conv3(.)); relu(out + identity)):
//
//   g[m, :]   = dy[m, :] under the keep bits of the block's output ReLU
//   dt[m, :]  = ka * (g - kb - xhat * kg),  xhat = (t - mean) * rstd      (ka = gamma rstd, kb = dbeta / M, kg = dgamma / M)
//   dx[m, :]  = dt[m, :] W                                                (W [K = Ko][N = C]: the data gradient of conv3)
//   side output: dt (16-bit) -- the weight gradient of conv3 reads it.
// As separate launches bn_bwd_apply writes dt (411 MB at 56 x 56 x 256, batch 256) and the data-gradient GEMM reads it back; here
// the gradient goes from the apply arithmetic straight into the MFMA operand registers.  Arithmetic and rounding points are
// those of bn_bwd_apply_pf_kernel followed by the GEMM: dt and dx are bit-identical to the two launches
// (tests/test_gpu_conv_bnbwd.py).  Built like conv_bnload.hip: streaming kernel, weight tile resident in LDS (transposed on the
// way in: conv3's [Ko][C] weight is n-contiguous for this product), A fragments global -> VGPR in MFMA layout, the next row
// tile's loads issued in front of this tile's stores.
#include "gemm_tiles.h"

#define BB_PAD 8

struct BnbArgs {
  const unsigned short* DY;     // [M, K] gradient w.r.t. the unit's output
  const unsigned short* T;      // [M, K] the convolution output the BatchNorm normalised
  const unsigned char* bits;    // [M * K / 8] keep bits of the ReLU behind the unit (or NULL: no ReLU)
  const unsigned short* B;      // weights [K][N], n contiguous
  unsigned short* DT;           // [M, K] side output
  unsigned short* C;            // [M, N] data gradient
  const float* mean; const float* rstd; const float* gamma; const float* dgamma; const float* dbeta;
  float inv_m;
  int M, N, K;
  int row_tiles, groups;
  // optional: dx is itself the gradient entering ANOTHER BatchNorm (bn2 of the same bottleneck): its backward reduction -- sum g,
  // sum g xhat with g = dx under bits2, xhat = (t2 - mean2) rstd2 -- is taken from the values about to be stored (bn_reduce_kernel)
  const unsigned short* t2;     // [M, N] or NULL
  const unsigned char* bits2;   // [M * N / 8]
  const float* mean2; const float* rstd2;
  float* partial;               // [groups][2][N]
};

template <int DT> struct BbMfma;
template <> struct BbMfma<DLE_F16> {
  static __device__ __forceinline__ float4_t run(ushort8_t a, ushort8_t b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
  }
};
template <> struct BbMfma<DLE_BF16> {
  static __device__ __forceinline__ float4_t run(ushort8_t a, ushort8_t b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};

__device__ __forceinline__ int bb_pos(int nl) {                       // (the interleaved weight rows of gemm_expand.hip)
  const int j = nl >> 5, r = nl & 31;
  return 32 * j + 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3);
}

// KS = K / 32 k steps, NB 16-column blocks (N = 16 NB, one column tile), NW wavefronts of 16 rows each
// v[16] per lane, 16 lanes (fr = lane & 15) -> acc += sum over the 16 lanes of v[fr] (reduce-scatter: xor-8 / 4 / 2 / 1 exchanges)
__device__ __forceinline__ void bb_reduce_scatter16(const float* v, int fr, float& acc) {
  float w8[8], w4[4], w2[2];
  const bool h8 = fr & 8, h4 = fr & 4, h2 = fr & 2, h1 = fr & 1;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float keep = h8 ? v[8 + q] : v[q], send = h8 ? v[q] : v[8 + q];
    w8[q] = keep + __shfl_xor(send, 8, 64);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float keep = h4 ? w8[4 + q] : w8[q], send = h4 ? w8[q] : w8[4 + q];
    w4[q] = keep + __shfl_xor(send, 4, 64);
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float keep = h2 ? w4[2 + q] : w4[q], send = h2 ? w4[q] : w4[2 + q];
    w2[q] = keep + __shfl_xor(send, 2, 64);
  }
  const float keep = h1 ? w2[1] : w2[0], send = h1 ? w2[0] : w2[1];
  acc += keep + __shfl_xor(send, 1, 64);
}

template <int DT, int KS, int NB, int NW, bool MASK, bool BRED>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(3))) void conv_bnbwd_kernel(BnbArgs p) {
  constexpr int K = KS * 32, LDW = K + BB_PAD, TM = NW * 16, TN = NB * 16, NP = NB / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* wl = (unsigned short*)smem_raw;                      // [TN][LDW]
  float* cf = (float*)(smem_raw + TN * LDW * 2);                       // [5][K]: ka | mean | rstd | kb | kg
  float* mu2 = cf + 5 * K;                                            // BRED: mean2 | rstd2 [TN] each
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, kg = lane >> 4;
  const int g = blockIdx.x;
  for (int c = threadIdx.x; c < K * (TN / 8); c += NW * 64) {          // [K][N] -> LDS rows of n, k contiguous
    const int k = c / (TN / 8), nc = c - k * (TN / 8);
    const ushort8_t v = *(const ushort8_t*)(p.B + (long long)k * p.N + nc * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) wl[bb_pos(nc * 8 + e) * LDW + k] = v[e];
  }
  for (int c = threadIdx.x; c < K; c += NW * 64) {                     // (the roundings of bn_bwd_apply_pf_kernel)
    const float rs = p.rstd[c];
    cf[c] = p.gamma[c] * rs;
    cf[K + c] = p.mean[c];
    cf[2 * K + c] = rs;
    cf[3 * K + c] = p.dbeta[c] * p.inv_m;
    cf[4 * K + c] = p.dgamma[c] * p.inv_m;
  }
  if (BRED) {
    for (int c = threadIdx.x; c < TN; c += NW * 64) { mu2[c] = p.mean2[c]; mu2[TN + c] = p.rstd2[c]; }
  }
  __syncthreads();
  const unsigned short* wrow = wl + fr * LDW + kg * 8;
  const int mrow = wave * 16 + fr;
  ushort8_t fg[KS], fx[KS];
  uint4_t mb0, mb1;                                                    // the row's K / 8 = 32 mask bytes (K = 256)
  ushort8_t t2c[BRED ? NP : 1];                                        // BRED: the row of t2
  uint2_t m2c;                                                         // ... and its N / 8 = 8 bytes of bits2
  float r1 = 0.f, r2 = 0.f;                                            // BRED: this lane's column (value fr of its 16) over every tile so far
  auto load_rows = [&](int m) __attribute__((always_inline)) {
    const long long mr = m < p.M ? m : p.M - 1;
    const long long o = mr * K + kg * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      fg[ks] = *(const ushort8_t*)(p.DY + o + ks * 32);
      fx[ks] = *(const ushort8_t*)(p.T + o + ks * 32);
    }
    if (MASK) {
      const uint4_t* mp = (const uint4_t*)(p.bits + mr * (K / 8));
      mb0 = mp[0];
      if (KS > 4) mb1 = mp[1];
    }
    if (BRED) {
#pragma unroll
      for (int j = 0; j < NP; ++j) t2c[j] = *(const ushort8_t*)(p.t2 + mr * TN + 32 * j + kg * 8);
      m2c = *(const uint2_t*)(p.bits2 + mr * (TN / 8));
    }
  };
  load_rows(g * TM + mrow);
  for (int tm = g; tm < p.row_tiles; tm += p.groups) {
    const int m_cur = tm * TM + mrow;
    const bool live = m_cur < p.M;
    // ---- BatchNorm backward on the fragments; side output dt
    {
      const long long o = (long long)(live ? m_cur : 0) * K + kg * 8;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int c0 = ks * 32 + kg * 8;
        float gf[8], xf[8], of[8];
        unpack8<DT>(fg[ks], gf);
        unpack8<DT>(fx[ks], xf);
        // byte ks * 4 + kg of the row's mask bytes: dword ks of the 8, byte kg
        const unsigned mw = ks < 4 ? mb0[ks & 3] : mb1[ks & 3];
        const unsigned bits = MASK ? (mw >> (8 * kg)) & 0xffu : 0xffu;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (MASK) { if (!((bits >> k) & 1u)) gf[k] = 0.f; }
          const float xh = (xf[k] - cf[K + c0 + k]) * cf[2 * K + c0 + k];
          of[k] = cf[c0 + k] * (gf[k] - cf[3 * K + c0 + k] - xh * cf[4 * K + c0 + k]);
        }
        fg[ks] = pack8<DT>(of);
        if (live) *(ushort8_t*)(p.DT + o + ks * 32) = fg[ks];
      }
    }
    // ---- product
    float4_t acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const ushort8_t fw = *(const ushort8_t*)(wrow + b * 16 * LDW + ks * 32);
        acc[b] = BbMfma<DT>::run(fw, fg[ks], acc[b]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue: pair j = blocks 2j, 2j + 1 -> columns 32 j + 8 kg + {0..7}
    ushort8_t outv[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = acc[2 * j][r]; v[4 + r] = acc[2 * j + 1][r]; }
      outv[j] = pack8<DT>(v);
    }
    if constexpr (BRED) {
      static_assert(NP == 2, "16 output columns per lane");
      float tg[16], tx[16];
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        float z[8], tf[8];
        unpack8<DT>(outv[j], z);
        unpack8<DT>(t2c[j], tf);
        const unsigned b2 = (m2c[j] >> (8 * kg)) & 0xffu;               // byte 4 j + kg of the row's 8
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int cl = 32 * j + 8 * kg + r;
          const float gz = (live && ((b2 >> r) & 1u)) ? z[r] : 0.f;
          tg[8 * j + r] = gz;
          tx[8 * j + r] = gz * (tf[r] - mu2[cl]) * mu2[TN + cl];
        }
      }
      bb_reduce_scatter16(tg, fr, r1);
      bb_reduce_scatter16(tx, fr, r2);
    }
    __builtin_amdgcn_sched_barrier(0);
    // the next tile's rows, in front of this tile's output stores (vmcnt is in order) and behind the last use of this tile's t2 row
    load_rows((tm + p.groups) * TM + mrow);
    __builtin_amdgcn_sched_barrier(0);
    {
      // full 128-byte lines per store instruction (gemm_expand.hip): neighbouring rows swap half of their pieces
      const bool odd = fr & 1;
      const bool live_other = (odd ? m_cur - 1 : m_cur + 1) < p.M;
      const long long o_own = (long long)m_cur * p.N + kg * 8;
      const long long o_other = odd ? o_own - p.N : o_own + p.N;
#pragma unroll
      for (int h = 0; h < NP / 2; ++h) {
        const uint4_t mine0 = __builtin_bit_cast(uint4_t, outv[2 * h]), mine1 = __builtin_bit_cast(uint4_t, outv[2 * h + 1]);
        const uint4_t give = odd ? mine0 : mine1;
        uint4_t got;
#pragma unroll
        for (int q = 0; q < 4; ++q) got[q] = (unsigned)__shfl_xor((int)give[q], 1, 64);
        const uint4_t first = odd ? got : mine0, second = odd ? mine1 : got;
        const long long o_first = (odd ? o_other : o_own) + 64 * h + (odd ? 32 : 0);
        const long long o_second = (odd ? o_own : o_other) + 64 * h + (odd ? 32 : 0);
        const bool live_first = odd ? live_other : live, live_second = odd ? live : live_other;
        if (live_first) *(uint4_t*)(p.C + o_first) = first;
        if (live_second) *(uint4_t*)(p.C + o_second) = second;
      }
    }
  }
  if (BRED) {
    // lane fr holds value fr of its column group's 16: value i = column 32 (i >> 3) + 8 kg + (i & 7)
    __syncthreads();
    float* red = (float*)smem_raw;                                      // [NW waves][2][TN]
    const int col = 32 * (fr >> 3) + 8 * kg + (fr & 7);
    red[(wave * 2 + 0) * TN + col] = r1;
    red[(wave * 2 + 1) * TN + col] = r2;
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * TN; t += NW * 64) {
      const int which = t / TN, col2 = t - which * TN;
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) tot += red[(w * 2 + which) * TN + col2];
      p.partial[((long long)g * 2 + which) * p.N + col2] = tot;
    }
  }
}

// dt [M, K] = BatchNorm backward of (dy under the keep bits, t), dx [M, N] = dt W with W [K][N] n-contiguous.
// 1: launched; 0: outside the envelope (K = 256, N = 64, M >= 4096, dense 16-byte aligned operands; the caller runs
// dle_bn_bwd_apply + dle_gemm); > 1: launch error.  dgamma / dbeta: the sums dle_bn_bwd_reduce left (fp32 [K]).
// 768 three-per-CU workgroups at most: the rows of `partial` the BRED form writes
extern "C" int dle_conv1x1_bnbwd_groups(int M) {
  const int row_tiles = (M + 63) / 64;
  return row_tiles < 768 ? row_tiles : 768;
}

// t2 / bits2 / mean2 / rstd2 / partial (all or none): dx is the gradient that enters a SECOND BatchNorm (bn2 of the bottleneck);
// partial [dle_conv1x1_bnbwd_groups(M)][2][N] receives its backward reduction (fold with dle_bn_bwd_finish).
extern "C" int dle_conv1x1_bnbwd_dgrad(const void* dy, const void* t, const void* relu_mask, const void* w, void* dt, void* dx,
                                       const float* mean, const float* rstd, const float* gamma, const float* dgamma,
                                       const float* dbeta, const void* t2, const void* bits2, const float* mean2,
                                       const float* rstd2, float* partial, int64_t partial_bytes, int M, int N, int K, int dtype,
                                       hipStream_t stream) {
  static const char* pin = getenv("DLE_CONV_BNBWD");
  if (pin && atoi(pin) == 0) return 0;
  if (dtype != DLE_F16 && dtype != DLE_BF16) return 0;
  if (M < 4096 || K != 256 || N != 64) return 0;
  if (!dy || !t || !w || !dt || !dx || !mean || !rstd || !gamma || !dgamma || !dbeta) return 0;
  if (((((uintptr_t)dy) | ((uintptr_t)t) | ((uintptr_t)w) | ((uintptr_t)dt) | ((uintptr_t)dx) | ((uintptr_t)relu_mask)) & 15) != 0)
    return 0;
  const bool bred = t2 != nullptr;
  if (bred && (!bits2 || !mean2 || !rstd2 || !partial || ((((uintptr_t)t2) | ((uintptr_t)bits2)) & 15) != 0)) return 0;
  constexpr int NWv = 4, TMv = NWv * 16;
  BnbArgs p;
  p.t2 = (const unsigned short*)t2; p.bits2 = (const unsigned char*)bits2; p.mean2 = mean2; p.rstd2 = rstd2; p.partial = partial;
  p.DY = (const unsigned short*)dy; p.T = (const unsigned short*)t; p.bits = (const unsigned char*)relu_mask;
  p.B = (const unsigned short*)w; p.DT = (unsigned short*)dt; p.C = (unsigned short*)dx;
  p.mean = mean; p.rstd = rstd; p.gamma = gamma; p.dgamma = dgamma; p.dbeta = dbeta; p.inv_m = 1.0f / (float)M;
  p.M = M; p.N = N; p.K = K;
  p.row_tiles = (M + TMv - 1) / TMv;
  int groups = 256 * 3;                                                // three 4-wave workgroups per CU (34 KiB weights + coefficients)
  if (groups > p.row_tiles) groups = p.row_tiles;
  p.groups = groups;
  if (bred && partial_bytes < (long long)groups * 2 * N * 4) return 0;
  const size_t lds = (size_t)N * (K + BB_PAD) * 2 + 5 * K * 4 + 2 * N * 4;
  const dim3 grid((unsigned)groups), block(NWv * 64);
#define BB_GO(DT, MK, BR) do { static bool attr_set = false; \
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)conv_bnbwd_kernel<DT, 8, 4, NWv, MK, BR>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); attr_set = true; } \
    hipLaunchKernelGGL((conv_bnbwd_kernel<DT, 8, 4, NWv, MK, BR>), grid, block, lds, stream, p); } while (0)
#define BB_M(DT) do { if (relu_mask) { if (bred) BB_GO(DT, true, true); else BB_GO(DT, true, false); } \
                      else { if (bred) BB_GO(DT, false, true); else BB_GO(DT, false, false); } } while (0)
  if (dtype == DLE_F16) BB_M(DLE_F16); else BB_M(DLE_BF16);
#undef BB_M
#undef BB_GO
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { dle_set_error("conv1x1_bnbwd_dgrad launch failed: %s", hipGetErrorString(e)); return (int)e + 1000; }
  return 1;
}
