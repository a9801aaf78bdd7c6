// MFMA GEMM with fused epilogues for gfx950 -- the dense contraction behind every Linear /
// MLP layer of the hot path (BERT/modeling.py:130-165,316-318 Linear + bias + GELU;
// DLRM apex mlp_cuda F1 = (GEMM + bias + ReLU) x L, DLRM/dlrm/nn/mlps.py:18-43; RN50 FC and the
// 1x1 convolutions in NHWC, which are plain GEMMs).
//
//   C[M,N] = epilogue( alpha * sum_k A(m,k) * B(n,k) )
//
// Operand storage (element (m,k) of A, (n,k) of B):
//   A_KC = true : A stored [M][lda]  (k contiguous)      A_KC = false : A stored [K][lda] (m contiguous)
//   B_KC = true : B stored [N][ldb]  (k contiguous)      B_KC = false : B stored [K][ldb] (n contiguous)
// which covers forward  Y = X W^T          (A_KC, B_KC)
//              dgrad    dX = dY W          (A_KC, !B_KC)
//              wgrad    dW = dY^T X        (!A_KC, !B_KC)   -- no transposed copies in HBM.
//
// Structure (wave64 / CDNA4):
//  * 128x128 output tile, BK = 64, 256 threads = 4 wavefronts in 2x2, 64x64 per wavefront as
//    4x4 v_mfma_f32_16x16x32 tiles (64 fp32 accumulator VGPRs per lane).
//  * LDS tiles are [row][64 k] halves (128 B rows) with the 16-byte chunk index XOR-swizzled by
//    f(row) = (row ^ row>>3) & 7, so ds_read_b128 fragment reads and both loader kinds stay
//    (almost) bank-conflict free; two LDS stages (64 KiB) -> 2 workgroups per CU.
//  * register-staged software pipeline: the global loads of tile t+1 are issued before the MFMA
//    block of tile t and written to the other LDS stage afterwards (one barrier per K-tile).
//  * m/n-contiguous operands (dgrad/wgrad) are transposed on the fly: each thread loads 4 k-rows x
//    8 elements and emits 8 ds_write_b64 (4 consecutive k for one row).
//  * the MFMA is issued with swapped operands so every lane owns 4 consecutive output columns
//    (8 B / 16 B stores); bias, ReLU, tanh-GELU (+ pre-activation side output), ReLU-backward
//    masking, fp32 accumulation and split-K (fp32 atomics) are fused in the epilogue.
//  * workgroup ids are remapped so the 8 XCDs each walk a contiguous band of tiles (private L2s).
#include "common.h"
#include <stdlib.h>

#define BM 128
#define BN 128
#define BK 64
#define SLAB_MODE(p) false

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_RELU_BWD = 3 };

struct GemmArgs {
  const unsigned short* A;
  const unsigned short* B;
  void* C;
  void* aux;                       // optional pre-activation output (same dtype/ld as C)
  const float* bias;               // optional fp32 bias[N]
  const unsigned short* mask_src;  // ACT_RELU_BWD: forward activation output, same shape/ld as C
  int M, N, K;
  long long lda, ldb, ldc;
  int out_dtype;                   // DLE_F32 / DLE_F16 / DLE_BF16
  int act;
  int splitk;
  int accumulate;                  // C += result (fp32 output only)
  float alpha;
};

template <int DT> struct Mfma16;
template <> struct Mfma16<DLE_F16> {
  static __device__ __forceinline__ float4_t run(ushort8_t a, ushort8_t b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a),
                                                  __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
  }
};
template <> struct Mfma16<DLE_BF16> {
  static __device__ __forceinline__ float4_t run(ushort8_t a, ushort8_t b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};

__device__ __forceinline__ int swz(int row) { return (row ^ (row >> 3)) & 7; }

// 8 consecutive elements along the contiguous dimension starting at p[0]; `n_valid` of them
// are inside the matrix (0..8); vec = the 16-byte fast path is legal for this launch.
__device__ __forceinline__ ushort8_t load8(const unsigned short* p, int n_valid, bool vec) {
  ushort8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
  if (n_valid >= 8 && vec) {
    v = *(const ushort8_t*)p;
  } else if (n_valid > 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < n_valid) v[j] = p[j];
  }
  return v;
}

// ---- tile loaders: global -> registers (4 x 16 B per thread per operand) ------------------
// k-contiguous operand: rows = tile rows (m or n), ld = row stride.
__device__ __forceinline__ void gload_kc(ushort8_t (&r)[4], const unsigned short* base, long long ld,
                                         int row0, int nrows, int k0, int kend, bool vec) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int q = threadIdx.x + it * 256;
    const int row = q >> 3, ch = q & 7;
    const int g = row0 + row, gk = k0 + ch * 8;
    int nv = kend - gk;
    if (g >= nrows) nv = 0;
    r[it] = load8(base + (long long)g * ld + gk, nv, vec);
  }
}
__device__ __forceinline__ void swrite_kc(const ushort8_t (&r)[4], unsigned short* tile) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int q = threadIdx.x + it * 256;
    const int row = q >> 3, ch = q & 7;
    *(ushort8_t*)(tile + row * BK + ((ch ^ swz(row)) << 3)) = r[it];
  }
}
// row-contiguous (transposed) operand: stored [K][ld], tile rows along the contiguous dim.
__device__ __forceinline__ void gload_tr(ushort8_t (&r)[4], const unsigned short* base, long long ld,
                                         int row0, int nrows, int k0, int kend, bool vec) {
  const int kg = threadIdx.x & 15, mc = threadIdx.x >> 4;
  const int g0 = row0 + mc * 8;
  const int nv_row = nrows - g0;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int gk = k0 + kg * 4 + rr;
    r[rr] = load8(base + (long long)gk * ld + g0, gk < kend ? nv_row : 0, vec);
  }
}
__device__ __forceinline__ void swrite_tr(const ushort8_t (&r)[4], unsigned short* tile) {
  const int kg = threadIdx.x & 15, mc = threadIdx.x >> 4;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = mc * 8 + j;
    ushort4_t w = {r[0][j], r[1][j], r[2][j], r[3][j]};
    *(ushort4_t*)(tile + row * BK + ((((kg >> 1) ^ swz(row))) << 3) + ((kg & 1) << 2)) = w;
  }
}

__device__ __forceinline__ float gelu_tanh(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float u = k0 * (x + k1 * x * x * x);
  return 0.5f * x * (1.0f + fast_tanh(u));
}

template <int DT, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* lds = (unsigned short*)smem_raw;   // [2 stages][A tile | B tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile walk: blocks b, b+8, b+16.. share an XCD -> give each XCD a contiguous band
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int ntiles = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // split-K range (multiples of BK)
  // balanced K slices: slice z owns k tiles [z*T/S, (z+1)*T/S) -- non-empty for every z when S <= T
  const int ktiles = (p.K + BK - 1) / BK;
  const int kt0 = (int)((long long)blockIdx.y * ktiles / p.splitk);
  const int kt1 = (int)((long long)(blockIdx.y + 1) * ktiles / p.splitk);
  if (kt0 >= kt1 && p.splitk > 1 && !SLAB_MODE(p)) return;
  const int kend = (kt1 * BK < p.K) ? kt1 * BK : p.K;

  const bool vecA = ((p.lda & 7) == 0) && ((((uintptr_t)p.A) & 15) == 0);
  const bool vecB = ((p.ldb & 7) == 0) && ((((uintptr_t)p.B) & 15) == 0);

  float4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

  ushort8_t ra[4], rb[4];
  auto issue = [&](int kt) {
    const int k0 = kt * BK;
    if (A_KC) gload_kc(ra, p.A, p.lda, m0, p.M, k0, kend, vecA);
    else gload_tr(ra, p.A, p.lda, m0, p.M, k0, kend, vecA);
    if (B_KC) gload_kc(rb, p.B, p.ldb, n0, p.N, k0, kend, vecB);
    else gload_tr(rb, p.B, p.ldb, n0, p.N, k0, kend, vecB);
  };
  auto commit = [&](int stage) {
    unsigned short* ta = lds + stage * (BM * BK + BN * BK);
    unsigned short* tb = ta + BM * BK;
    if (A_KC) swrite_kc(ra, ta); else swrite_tr(ra, ta);
    if (B_KC) swrite_kc(rb, tb); else swrite_tr(rb, tb);
  };

  if (kt0 < kt1) {
    issue(kt0);
    commit(0);
  }
  __syncthreads();

  const int fr = lane & 15, fg = lane >> 4;
  for (int kt = kt0; kt < kt1; ++kt) {
    const int stage = (kt - kt0) & 1;
    if (kt + 1 < kt1) issue(kt + 1);
    const unsigned short* ta = lds + stage * (BM * BK + BN * BK);
    const unsigned short* tb = ta + BM * BK;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      ushort8_t fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wm * 64 + i * 16 + fr;
        fa[i] = *(const ushort8_t*)(ta + row * BK + (((kk * 4 + fg) ^ swz(row)) << 3));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wn * 64 + j * 16 + fr;
        fb[j] = *(const ushort8_t*)(tb + row * BK + (((kk * 4 + fg) ^ swz(row)) << 3));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = Mfma16<DT>::run(fb[j], fa[i], acc[i][j]);
    }
    if (kt + 1 < kt1) commit(stage ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane owns C[m][n..n+3], m = fr-th row of the 16x16 tile, n = 4*fg + r
  const bool vec_c = (p.ldc & 3) == 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + fr;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + fg * 4;
      if (n >= p.N) continue;
      float4_t v = acc[i][j] * p.alpha;
      const int nval = (p.N - n) < 4 ? (p.N - n) : 4;
      const long long off = (long long)m * p.ldc + n;
      if (p.splitk > 1) {
        float* c = (float*)p.C + off;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r < nval) unsafeAtomicAdd(c + r, v[r]);
        continue;
      }
      if (p.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r < nval) v[r] += p.bias[n + r];
      }
      float4_t pre = v;
      if (p.act == ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
      } else if (p.act == ACT_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_tanh(v[r]);
      } else if (p.act == ACT_RELU_BWD) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r < nval) {
            const float y = DT == DLE_F16 ? Elem<DLE_F16>::to_f32(p.mask_src[off + r])
                                          : Elem<DLE_BF16>::to_f32(p.mask_src[off + r]);
            v[r] = y > 0.f ? v[r] : 0.f;
          }
      }
      if (p.out_dtype == DLE_F32) {
        float* c = (float*)p.C + off;
        if (p.accumulate) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (r < nval) v[r] += c[r];
        }
        if (nval == 4 && vec_c) *(float4_t*)c = v;
        else
          for (int r = 0; r < nval; ++r) c[r] = v[r];
        if (p.aux) {
          float* a = (float*)p.aux + off;
          for (int r = 0; r < nval; ++r) a[r] = pre[r];
        }
      } else {
        ushort4_t o, po;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (p.out_dtype == DLE_F16) { o[r] = Elem<DLE_F16>::from_f32(v[r]); po[r] = Elem<DLE_F16>::from_f32(pre[r]); }
          else { o[r] = Elem<DLE_BF16>::from_f32(v[r]); po[r] = Elem<DLE_BF16>::from_f32(pre[r]); }
        }
        unsigned short* c = (unsigned short*)p.C + off;
        if (nval == 4 && vec_c) *(ushort4_t*)c = o;
        else
          for (int r = 0; r < nval; ++r) c[r] = o[r];
        if (p.aux) {
          unsigned short* a = (unsigned short*)p.aux + off;
          if (nval == 4 && vec_c) *(ushort4_t*)a = po;
          else
            for (int r = 0; r < nval; ++r) a[r] = po[r];
        }
      }
    }
  }
}

// ---- column sums: out[n] (+)= sum_m X[m][n]   (bias gradients), 16-bit or fp32 input -> fp32
// 16 B per lane along the row (8 x 16-bit / 4 x fp32 columns per lane); LPR lanes cover one row, a wavefront
// sweeps 64/LPR rows per trip; partial sums meet in LDS, one fp32 atomic per column per workgroup.
template <int DT>
__global__ __launch_bounds__(256) void colsum_kernel(const void* __restrict__ x, float* __restrict__ out,
                                                     long long M, int N, long long ld, long long rows_per_block,
                                                     int lpr, float* __restrict__ partial, const long long* __restrict__ table = nullptr) {
  constexpr int V = DT == DLE_F32 ? 4 : 8;
  __shared__ float red[256 * 8];
  if (table) {                           // batched form: matrix blockIdx.z of the table, its own slab of partial rows
    x = (const void*)table[2 * blockIdx.z];
    partial += (long long)blockIdx.z * gridDim.y * N;
  }
  const int cl = threadIdx.x % lpr, rl = threadIdx.x / lpr, rstep = 256 / lpr;
  const int c0 = (blockIdx.x * lpr + cl) * V;
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  long long r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  float s[V];
#pragma unroll
  for (int k = 0; k < V; ++k) s[k] = 0.f;
  const bool vec = (ld % V) == 0 && (((uintptr_t)x) & 15) == 0 && c0 + V <= N;
  if (c0 < N) {
    long long r = r0 + rl;
    if (vec && DT != DLE_F32) {
      // 2 independent 16-byte loads in flight per lane: a read-only sweep peaks with FEW loads per lane once ~1000
      // workgroups are resident (tools/probes/read_bw.hip: 6.4 TB/s with 1-2, 4.8-5.3 with 8)
      constexpr int U = 2;
      for (; r + (long long)(U - 1) * rstep < r1; r += (long long)U * rstep) {
        ushort8_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *(const ushort8_t*)((const unsigned short*)x + (r + (long long)u * rstep) * ld + c0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          float f[8];
          unpack8<DT == DLE_F32 ? DLE_BF16 : DT>(v[u], f);
#pragma unroll
          for (int k = 0; k < V; ++k) s[k] += f[k];
        }
      }
    }
    for (; r < r1; r += rstep) {
      if (vec) {
        if (DT == DLE_F32) {
          const float4_t v = *(const float4_t*)((const float*)x + r * ld + c0);
#pragma unroll
          for (int k = 0; k < 4; ++k) s[k] += v[k];
        } else {
          const ushort8_t v = *(const ushort8_t*)((const unsigned short*)x + r * ld + c0);
#pragma unroll
          for (int k = 0; k < V; ++k) s[k] += DT == DLE_F16 ? Elem<DLE_F16>::to_f32(v[k]) : Elem<DLE_BF16>::to_f32(v[k]);
        }
      } else {
        for (int k = 0; k < V && c0 + k < N; ++k) {
          if (DT == DLE_F32) s[k] += ((const float*)x)[r * ld + c0 + k];
          else if (DT == DLE_F16) s[k] += Elem<DLE_F16>::to_f32(((const unsigned short*)x)[r * ld + c0 + k]);
          else s[k] += Elem<DLE_BF16>::to_f32(((const unsigned short*)x)[r * ld + c0 + k]);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < V; ++k) red[threadIdx.x * 8 + k] = s[k];
  __syncthreads();
  if (rl == 0 && c0 < N) {
    for (int k = 0; k < V && c0 + k < N; ++k) {
      float t = 0.f;
      for (int q = 0; q < rstep; ++q) t += red[(q * lpr + cl) * 8 + k];
      if (partial) partial[(long long)blockIdx.y * N + c0 + k] = t;
      else unsafeAtomicAdd(out + c0 + k, t);
    }
  }
}

// out[n] (+)= sum_g partial[g][n]; 16 columns x 16 group slices per workgroup (latency-bound: spread wide,
// 4 loads in flight per lane)
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                            int N, int groups, int accumulate, const long long* __restrict__ table = nullptr) {
  __shared__ float red[256];
  if (table) {
    out = (float*)table[2 * blockIdx.y + 1];
    partial += (long long)blockIdx.y * groups * N;
  }
  const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int n = blockIdx.x * 16 + cl;
  float s = 0.f;
  if (n < N) {
    int g = sl;
    for (; g + 48 < groups; g += 64) {
      float a[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = partial[(long long)(g + 16 * u) * N + n];
      s += (a[0] + a[1]) + (a[2] + a[3]);
    }
    for (; g < groups; g += 16) s += partial[(long long)g * N + n];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (sl == 0 && n < N) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q * 16 + cl];
    out[n] = accumulate ? out[n] + t : t;
  }
}

extern "C" int dle_colsum(const void* x, float* out, int64_t M, int N, int64_t ld, int dtype,
                          int accumulate, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  DLE_CHECK_ARG(x && out && N > 0 && M >= 0, "colsum: bad args");
  DLE_CHECK_ARG(dtype == DLE_F32 || dtype == DLE_F16 || dtype == DLE_BF16, "colsum: bad dtype %d", dtype);
  if (M == 0) {
    if (!accumulate) {
      hipError_t e = hipMemsetAsync(out, 0, (size_t)N * 4, stream);
      if (e != hipSuccess) { dle_set_error("colsum memset: %s", hipGetErrorString(e)); return (int)e; }
    }
    return 0;
  }
  const int V = dtype == DLE_F32 ? 4 : 8;
  const int cols_v = (N + V - 1) / V;
  int lpr = 1;
  while (lpr < cols_v && lpr < 32) lpr <<= 1;            // lanes per row (power of two, <= 32: wide matrices split their
                                                         // columns over blockIdx.x -- fewer row groups, smaller partials)
  const int gx = (cols_v + lpr - 1) / lpr;
  long long want = 1024 / gx;                              // ~4 workgroups per CU in total (as the BatchNorm reductions)
  if (want < 1) want = 1;
  long long rpb = (M + want - 1) / want;
  const long long min_rows = 8LL * (256 / lpr);
  if (rpb < min_rows) rpb = min_rows;
  long long gy = (M + rpb - 1) / rpb;
  // Row groups meet through plain stores into the caller's workspace + a tiny finishing pass.  Without a
  // workspace they meet through fp32 atomics on the SAME N addresses, which serialise per cache line, so
  // the row split is kept coarse in that case.
  float* partial = nullptr;
  if (workspace && (((uintptr_t)workspace) & 3) == 0 && workspace_bytes >= gy * (long long)N * 4) {
    partial = (float*)workspace;
  } else {
    if (gy > 64) { gy = 64; rpb = (M + gy - 1) / gy; gy = (M + rpb - 1) / rpb; }
    if (!accumulate) {
      hipError_t e = hipMemsetAsync(out, 0, (size_t)N * 4, stream);
      if (e != hipSuccess) { dle_set_error("colsum memset: %s", hipGetErrorString(e)); return (int)e; }
    }
  }
  dim3 grid(gx, (unsigned)gy), block(256);
  if (dtype == DLE_F32) hipLaunchKernelGGL(colsum_kernel<DLE_F32>, grid, block, 0, stream, x, out, (long long)M, N, (long long)ld, rpb, lpr, partial);
  else if (dtype == DLE_F16) hipLaunchKernelGGL(colsum_kernel<DLE_F16>, grid, block, 0, stream, x, out, (long long)M, N, (long long)ld, rpb, lpr, partial);
  else hipLaunchKernelGGL(colsum_kernel<DLE_BF16>, grid, block, 0, stream, x, out, (long long)M, N, (long long)ld, rpb, lpr, partial);
  DLE_LAUNCH_CHECK();
  if (partial) {
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((N + 15) / 16), dim3(256), 0, stream, (const float*)partial, out, N, (int)gy, accumulate);
    DLE_LAUNCH_CHECK();
  }
  return 0;
}

// Column sums of n same-shaped 16-bit matrices in ONE launch pair: table = n x { source address, fp32 destination address }
// (device, int64).  workspace: n * groups * N floats, groups = dle_colsum_batched_groups(n, M, N).
static void colsum_batched_plan(int n, long long M, int N, int& lpr, int& gx, long long& rpb, long long& gy) {
  const int cols_v = (N + 7) / 8;
  lpr = 1;
  while (lpr < cols_v && lpr < 32) lpr <<= 1;
  gx = (cols_v + lpr - 1) / lpr;
  long long want = 2048 / ((long long)gx * n);           // ~8 workgroups per CU over all matrices
  if (want < 1) want = 1;
  rpb = (M + want - 1) / want;
  const long long min_rows = 8LL * (256 / lpr);
  if (rpb < min_rows) rpb = min_rows;
  gy = (M + rpb - 1) / rpb;
}
extern "C" int64_t dle_colsum_batched_workspace_bytes(int n, int64_t M, int N) {
  int lpr, gx; long long rpb, gy;
  if (n <= 0 || M <= 0 || N <= 0) return 0;
  colsum_batched_plan(n, M, N, lpr, gx, rpb, gy);
  return (int64_t)n * gy * N * 4;
}
extern "C" int dle_colsum_batched(const int64_t* table_dev, int n, int64_t M, int N, int64_t ld, int dtype, void* workspace,
                                  int64_t workspace_bytes, hipStream_t stream) {
  DLE_CHECK_ARG(table_dev && n > 0 && n <= 65535 && M > 0 && N > 0 && ld >= N, "colsum_batched: bad args");
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "colsum_batched: 16-bit matrices only (got %d)", dtype);
  int lpr, gx; long long rpb, gy;
  colsum_batched_plan(n, M, N, lpr, gx, rpb, gy);
  DLE_CHECK_ARG(workspace && workspace_bytes >= (long long)n * gy * N * 4, "colsum_batched: workspace too small");
  dim3 grid(gx, (unsigned)gy, (unsigned)n), block(256);
  if (dtype == DLE_F16) hipLaunchKernelGGL(colsum_kernel<DLE_F16>, grid, block, 0, stream, (const void*)nullptr, (float*)nullptr, (long long)M, N, (long long)ld, rpb, lpr, (float*)workspace, (const long long*)table_dev);
  else hipLaunchKernelGGL(colsum_kernel<DLE_BF16>, grid, block, 0, stream, (const void*)nullptr, (float*)nullptr, (long long)M, N, (long long)ld, rpb, lpr, (float*)workspace, (const long long*)table_dev);
  DLE_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((N + 15) / 16, (unsigned)n), dim3(256), 0, stream, (const float*)workspace, (float*)nullptr, N, (int)gy, 0, (const long long*)table_dev);
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_gemm_dma_try(const void* A, const void* B, void* C, void* aux, const float* bias,
                                const void* mask_src, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                                int a_kc, int b_kc, int in_dtype, int out_dtype, int act, int splitk,
                                int accumulate, float alpha, void* workspace, int64_t workspace_bytes,
                                hipStream_t stream);

extern "C" int dle_gemm8_try(const void* A, const void* B, void* C, void* aux, const float* bias, const void* src, int M, int N,
                             int K, int64_t lda, int64_t ldb, int64_t ldc, int a_kc, int b_kc, int in_dtype, int out_dtype,
                             int act, int splitk, int accumulate, float alpha, float* ws, float* stats, hipStream_t stream);   // gemm8.hip

extern "C" int dle_gemm_smallm_try(const void* A, const void* B, void* C, const float* bias, const void* src, int M, int N, int K,
                                   int64_t lda, int64_t ldb, int64_t ldc, int in_dtype, int out_dtype, int act_add, int accumulate,
                                   float alpha, hipStream_t stream);     // gemm_smallm.hip

extern "C" int dle_gemm_expand_try(const void* A, const void* B, void* C, const void* src, const void* bits, float* stats, int M,
                                   int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int b_kc, int in_dtype, int out_dtype,
                                   int act, hipStream_t stream);           // gemm_expand.hip

// C ABI.  a_kc / b_kc: operand stored with the contraction dimension contiguous (see header).
extern "C" int dle_gemm(const void* A, const void* B, void* C, void* aux, const float* bias,
                        const void* mask_src, int M, int N, int K, int64_t lda, int64_t ldb,
                        int64_t ldc, int a_kc, int b_kc, int in_dtype, int out_dtype, int act,
                        int splitk, int accumulate, float alpha, void* workspace, int64_t workspace_bytes,
                        hipStream_t stream) {
  DLE_CHECK_ARG(A && B && C, "gemm: null pointer");
  DLE_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "gemm: negative dimension");
  DLE_CHECK_ARG(in_dtype == DLE_F16 || in_dtype == DLE_BF16, "gemm: inputs must be f16/bf16 (got %d)", in_dtype);
  DLE_CHECK_ARG(out_dtype == DLE_F32 || out_dtype == DLE_F16 || out_dtype == DLE_BF16, "gemm: bad out dtype");
  DLE_CHECK_ARG(!(a_kc == 0 && b_kc != 0), "gemm: (A m-contiguous, B k-contiguous) is not a hot-path layout");
  const bool needs_src = act == ACT_RELU_BWD || act == 4 || act == 5 || act == 7 || act == 8 || act == 9;   // RELU_BWD, ADD, GELU_BWD, TANH_BWD, ADD_MASKED, MUL
  DLE_CHECK_ARG(act >= 0 && act <= 10, "gemm: unknown epilogue %d", act);
  DLE_CHECK_ARG(!needs_src || mask_src, "gemm: this epilogue needs mask_src");
  DLE_CHECK_ARG(act != 8 || (aux && (ldc & 7) == 0), "gemm: the masked add reads its keep bits through aux (ldc a multiple of 8)");
  DLE_CHECK_ARG(act != 10 || aux, "gemm: DLE_ACT_GELU_DAUX writes the derivative to aux");
  DLE_CHECK_ARG(!needs_src || out_dtype == in_dtype, "gemm: mask_src dtype = in dtype = out dtype");
  if (splitk < 1) splitk = 1;
  {
    const int kt = K > 0 ? (K + BK - 1) / BK : 1;
    if (splitk > kt) splitk = kt;          // every K slice owns at least one K tile
  }
  if (splitk > 1)
    DLE_CHECK_ARG(out_dtype == DLE_F32 && !bias && act == ACT_NONE && !aux, "gemm: split-K needs a plain fp32 output");
  else
    DLE_CHECK_ARG(!accumulate || out_dtype == DLE_F32, "gemm: accumulate needs fp32 output");
  if (M == 0 || N == 0) return 0;
  {
    // fast path: LDS-DMA fed kernel (gemm_dma.hip); DLE_GEMM_LEGACY=1 pins the register-staged kernel
    static const bool legacy = getenv("DLE_GEMM_LEGACY") != nullptr && getenv("DLE_GEMM_LEGACY")[0] == '1';
    // few rows (recurrent steps, heads): the weight-streaming kernel of gemm_smallm.hip -- N / 16..32 workgroups instead of a
    // dozen 128x128 tiles
    if (!legacy && K > 0 && M <= 256 && a_kc && b_kc && splitk == 1 && !aux && (act == ACT_NONE || act == 4) &&
        (!accumulate || out_dtype == DLE_F32)) {
      const int r = dle_gemm_smallm_try(A, B, C, bias, mask_src, M, N, K, lda, ldb, ldc, in_dtype, out_dtype, act == 4, accumulate,
                                        alpha, stream);
      if (r == 1) return 0;
      if (r != 0) return r;
    }
    // the masked-addend data gradient of the deepest stage's conv1 (K >= 512, M <= 65536 rows: 7 x 7 at batch 256): the
    // ping-pong kernel's source-tensor epilogue with the keep bits (gemm8_kernel.h, ACT_ADD_MASKED); DLE_GEMM8_MASKED=0 keeps the
    // streaming kernel below
    // (K >= 512 only: at K = 256 -- 50176 x 1024 x 256, four K tiles per item -- the item is all epilogue and the streaming kernel
    //  below is faster, 63 against 71 us; at K = 512 the ping-pong kernel wins, 57 against 63 us: profiles/r06_rn50_shapes_*.txt)
    if (!legacy && act == 8 && a_kc && !b_kc && splitk == 1 && !accumulate && !bias && alpha == 1.0f && K >= 512) {
      static const int on = getenv("DLE_GEMM8_MASKED") ? atoi(getenv("DLE_GEMM8_MASKED")) : 1;
      static const long long maxm = getenv("DLE_GEMM8_MASKED_MAXM") ? atoll(getenv("DLE_GEMM8_MASKED_MAXM")) : 65536;
      if (on && M <= maxm) {
        const int r = dle_gemm8_try(A, B, C, aux, bias, mask_src, M, N, K, lda, ldb, ldc, a_kc, b_kc, in_dtype, out_dtype, act, 1, 0,
                                    alpha, nullptr, nullptr, stream);
        if (r == 1) return 0;
        if (r > 1) return r;
      }
    }
    // many rows, K <= 256, N >= 2 K (the channel-widening 1x1 convolutions): the streaming kernel of gemm_expand.hip
    // (store-only products with K = 64 stay on the tile kernel's PLAIN epilogue: 135 vs 152 us at 802816 x 256 x 64)
    if (!legacy && a_kc && splitk == 1 && !accumulate && !bias && alpha == 1.0f &&
        (act == ACT_NONE ? (!aux && K >= 128) : act == 4 ? !aux : act == 8)) {
      const char* pin = getenv("DLE_GEMM_EXPAND");                          // probes / tests: "0" pins the tile kernels
      if (!pin || atoi(pin) != 0) {
        const int r = dle_gemm_expand_try(A, B, C, mask_src, act == 8 ? aux : nullptr, nullptr, M, N, K, lda, ldb, ldc, b_kc,
                                          in_dtype, out_dtype, act == 0 ? 0 : act == 4 ? 1 : 2, stream);
        if (r == 1) return 0;
        if (r != 0) return r;
      }
    }
    if (!legacy && K > 0) {
      const int r = dle_gemm_dma_try(A, B, C, aux, bias, mask_src, M, N, K, lda, ldb, ldc, a_kc, b_kc, in_dtype,
                                     out_dtype, act, splitk, accumulate, alpha, workspace, workspace_bytes, stream);
      if (r == 1) return 0;
      if (r != 0) return r;
    }
  }
  DLE_CHECK_ARG(act <= ACT_RELU_BWD, "gemm: epilogue %d needs the aligned (LDS-DMA) path: K, lda, ldb multiples of 8, 16-byte "
                "aligned operands", act);
  if (splitk > 1 && !accumulate) {   // register-staged kernel: fp32 atomics into a cleared C
    hipError_t e = ldc == N ? hipMemsetAsync(C, 0, (size_t)M * N * 4, stream)
                            : hipMemset2DAsync(C, (size_t)ldc * 4, 0, (size_t)N * 4, (size_t)M, stream);
    if (e != hipSuccess) { dle_set_error("gemm memset: %s", hipGetErrorString(e)); return (int)e; }
  }
  GemmArgs p;
  p.A = (const unsigned short*)A; p.B = (const unsigned short*)B; p.C = C; p.aux = aux; p.bias = bias;
  p.mask_src = (const unsigned short*)mask_src;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.out_dtype = out_dtype; p.act = act; p.splitk = splitk; p.accumulate = accumulate; p.alpha = alpha;
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  dim3 grid(tiles, splitk), block(256);
  const size_t lds = 2 * (BM * BK + BN * BK) * 2;
#define GO(DT, AK, BKC) hipLaunchKernelGGL((gemm_kernel<DT, AK, BKC>), grid, block, lds, stream, p)
  if (in_dtype == DLE_F16) {
    if (a_kc && b_kc) GO(DLE_F16, true, true);
    else if (a_kc) GO(DLE_F16, true, false);
    else GO(DLE_F16, false, false);
  } else {
    if (a_kc && b_kc) GO(DLE_BF16, true, true);
    else if (a_kc) GO(DLE_BF16, true, false);
    else GO(DLE_BF16, false, false);
  }
#undef GO
  DLE_LAUNCH_CHECK();
  return 0;
}
