// "Expand" GEMM for gfx950: C[M, N] = A[M, K] B^T (+ addend) with M in the 10^5..10^6, K <= 256 and N >= 2 K -- the 1x1
// convolutions of a ResNet bottleneck that WIDEN the channel count (Classification/ConvNets/image_classification/models/
// resnet.py:148-175: conv3 forward, conv1 data gradient + the masked residual gradient of the block input).  At 28-110 flop/B
// these products sit far on the HBM side of the ridge (312 flop/B): per output row they read 2 K bytes of A and write (and, with
// an addend, read) 2 N >= 4 K bytes.  The MFMA tile kernels of gemm_dma.hip run them at 2.4-3.0 TB/s: one or two workgroups per
// CU walk load -> MFMA -> addend -> store phases one after the other.  This kernel is a STREAMING kernel with a matrix product
// in the middle, built like the BatchNorm apply passes (5.5 TB/s) rather than like a GEMM:
//  * the weight tile (128 output columns x K, <= 66 KiB) is loaded ONCE per workgroup into LDS and the workgroup then walks
//    row tiles (persistent along M): after the prologue every byte a wave touches is stream data;
//  * 4 wavefronts per workgroup, each owning 16 rows x 128 columns; 3-4 workgroups per CU (LDS 18-66 KiB, <= 128 VGPRs), so
//    12-16 waves per CU are in different phases and the loads of some cover the stores of others;
//  * no LDS and no DMA on the stream side: A fragments, addend rows and mask bytes go global -> VGPR in the layout the MFMA /
//    the epilogue need.  v_mfma_f32_16x16x32 with the WEIGHT rows as the first operand leaves a lane with 4 consecutive output
//    columns of one row; the weight rows of two MFMA blocks are interleaved (LDS position 32 j + 16 q + i holds column
//    32 j + 8 (i >> 2) + 4 q + (i & 3)), so that the pair leaves the lane with 8 CONSECUTIVE columns = one 16-byte load of the
//    addend and one 16-byte store of the output, 64-byte segments per row across the 4 lane groups;
//  * weights in either layout ([N, K] k-contiguous or [K, N]): the transposition happens on the way into LDS.
// Epilogues: none / + addend (DLE_ACT_ADD) / + addend under bit-packed keep bits (DLE_ACT_ADD_MASKED); optional column sums and
// sums of squares of the ROUNDED output for the BatchNorm that follows (one partial row per workgroup, no atomics).
#include "gemm_tiles.h"

#define EX_TN 128
#define EX_PAD 8                        // halves of padding per LDS weight row: (K + 8) / 2 dwords = 4 (mod 8) -> conflict-free b128 reads

struct ExpandArgs {
  const unsigned short* A;
  const unsigned short* B;
  unsigned short* C;
  const unsigned short* src;
  const unsigned char* bits;
  float* stats;                         // [groups][2][N] or NULL
  int M, N, K;
  long long lda, ldb, ldc;
  int b_kc, row_tiles, groups, col_tiles;
  // ACT == 3: the addend is the ZERO-STUFFED form of a compact [n, H/2, W/2, N] tensor (the data gradient of a 1x1 stride-2
  // convolution on its own P x Q grid): row m = (n, h, w) reads compact row (n, h/2, w/2) when h and w are even, nothing otherwise
  FastDiv dHW, dW;
  int up_HW, up_W, up_P, up_Q;
  long long ld_src;
  // BRED: the output is also the gradient that enters ANOTHER BatchNorm (the previous block's bn3): its backward reduction --
  // sum g, sum g xhat with g = C under bits2, xhat = (t2 - mean2) rstd2 -- is taken here, from the values about to be stored
  const unsigned short* t2;             // [M, N], pitch ldc
  const unsigned char* bits2;           // keep bits, same indexing as `bits`
  const float* mean2;
  const float* rstd2;
};

template <int DT> struct ExMfma;
template <> struct ExMfma<DLE_F16> {
  static __device__ __forceinline__ float4_t run(ushort8_t a, ushort8_t b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
  }
};
template <> struct ExMfma<DLE_BF16> {
  static __device__ __forceinline__ float4_t run(ushort8_t a, ushort8_t b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};

// LDS row position of local output column nl (0..127): see the header comment
__device__ __forceinline__ int ex_pos(int nl) {
  const int j = nl >> 5, r = nl & 31;
  return 32 * j + 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3);
}

// The stream registers of one row tile of a wave: A fragments, addend, keep bits (all loads unconditional: clamped row).
template <int KS, int ACT, bool BRED = false>
struct ExStream {
  ushort8_t fa[KS];
  ushort8_t sv[ACT ? 4 : 1];
  ushort8_t tv[BRED ? 4 : 1];                                          // BRED: the row of t2
  uint4_t mb;                                                          // ACT == 2: 16 mask bytes = columns n0 .. n0 + 127 of the row
  uint4_t mb2;                                                         // BRED: the row's 16 bytes of bits2
  long long o0;
  bool live;
  bool has;                                                            // ACT == 3: this row has an addend row
  __device__ __forceinline__ void load_a(const ExpandArgs& p, int m, int kg) {
    const long long mr = m < p.M ? m : p.M - 1;
    const unsigned short* arow = p.A + mr * p.lda + kg * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) fa[ks] = *(const ushort8_t*)(arow + ks * 32);
  }
  __device__ __forceinline__ void load_src(const ExpandArgs& p, int m, int n0, int kg) {
    live = m < p.M;
    const long long mr = live ? m : p.M - 1;
    o0 = mr * p.ldc + n0 + kg * 8;                                     // + 32 j
    if (ACT == 3) {
      const int mi = (int)mr;
      const int n = fd_div(mi, p.dHW), rem = mi - n * p.up_HW;
      const int h = fd_div(rem, p.dW), w = rem - h * p.up_W;
      has = !((h | w) & 1);
      const long long cr = has ? ((long long)n * p.up_P + (h >> 1)) * p.up_Q + (w >> 1) : 0;      // (unconditional, clamped load)
      const unsigned short* srow = p.src + cr * p.ld_src + n0 + kg * 8;
#pragma unroll
      for (int j = 0; j < 4; ++j) sv[j] = *(const ushort8_t*)(srow + 32 * j);
    } else if (ACT) {
#pragma unroll
      for (int j = 0; j < 4; ++j) sv[j] = *(const ushort8_t*)(p.src + o0 + 32 * j);
    }
    if (ACT == 2) mb = *(const uint4_t*)(p.bits + ((mr * p.ldc + n0) >> 3));   // (ldc, n0 multiples of 128: 16-byte aligned)
    if (BRED) {
#pragma unroll
      for (int j = 0; j < 4; ++j) tv[j] = *(const ushort8_t*)(p.t2 + o0 + 32 * j);
      mb2 = *(const uint4_t*)(p.bits2 + ((mr * p.ldc + n0) >> 3));
    }
  }
};

// v[32] per lane, 16 lanes (fr = lane & 15) -> acc[b] += sum over the 16 lanes of v[2 fr + b]
__device__ __forceinline__ void ex_reduce_scatter16(const float* v, int fr, float* acc) {
  float w16[16], w8[8], w4[4];
  const bool h8 = fr & 8, h4 = fr & 4, h2 = fr & 2, h1 = fr & 1;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const float keep = h8 ? v[16 + q] : v[q], send = h8 ? v[q] : v[16 + q];
    w16[q] = keep + __shfl_xor(send, 8, 64);
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float keep = h4 ? w16[8 + q] : w16[q], send = h4 ? w16[q] : w16[8 + q];
    w8[q] = keep + __shfl_xor(send, 4, 64);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float keep = h2 ? w8[4 + q] : w8[q], send = h2 ? w8[q] : w8[4 + q];
    w4[q] = keep + __shfl_xor(send, 2, 64);
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float keep = h1 ? w4[2 + q] : w4[q], send = h1 ? w4[q] : w4[2 + q];
    acc[q] += keep + __shfl_xor(send, 1, 64);
  }
}

// ACT: 0 none, 1 + addend, 2 + addend under keep bits.  KS = K / 32 (2, 4 or 8).  STATS: column sums of the rounded output.
// NW wavefronts of 16 rows each: 4, or 8 for K = 256 (its 66 KiB weight tile allows two workgroups per CU; 16 waves per CU either way).
template <int DT, int KS, int ACT, bool STATS, int NW, bool BRED = false>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu((STATS || BRED) ? 3 : 4)))
void gemm_expand_kernel(ExpandArgs p) {
  constexpr int K = KS * 32, LDW = K + EX_PAD, TM = NW * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* wl = (unsigned short*)smem_raw;                     // [128][LDW]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, kg = lane >> 4;
  // workgroups are dealt to the 8 XCDs round-robin (blockIdx & 7): the column tiles of one row group share their A rows, so they
  // are given consecutive slots of ONE XCD and meet in its L2 (groups is a multiple of 8)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tn = slot % p.col_tiles, g = (slot / p.col_tiles) * 8 + xcd;
  const int n0 = tn * EX_TN;
  // ---- weight tile -> LDS (once)
  if (p.b_kc) {
    constexpr int CPR = K / 8;                                         // 16-byte chunks per row
    for (int c = threadIdx.x; c < EX_TN * CPR; c += NW * 64) {
      const int nl = c / CPR, kc = c - nl * CPR;
      const ushort8_t v = *(const ushort8_t*)(p.B + (long long)(n0 + nl) * p.ldb + kc * 8);
      *(ushort8_t*)(wl + ex_pos(nl) * LDW + kc * 8) = v;
    }
  } else {
    for (int c = threadIdx.x; c < K * (EX_TN / 8); c += NW * 64) {
      const int k = c >> 4, nc = c & 15;
      const ushort8_t v = *(const ushort8_t*)(p.B + (long long)k * p.ldb + n0 + nc * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) wl[ex_pos(nc * 8 + e) * LDW + k] = v[e];
    }
  }
  float* mu2 = (float*)(smem_raw + EX_TN * LDW * 2);                  // BRED: mean2 | rstd2 of this column tile
  if (BRED) {
    for (int c = threadIdx.x; c < EX_TN; c += NW * 64) { mu2[c] = p.mean2[n0 + c]; mu2[EX_TN + c] = p.rstd2[n0 + c]; }
  }
  __syncthreads();
  float s1[STATS ? 32 : 1], s2[STATS ? 32 : 1];
  if (STATS) {
#pragma unroll
    for (int i = 0; i < 32; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
  }
  float b1[2] = {0.f, 0.f}, b2s[2] = {0.f, 0.f};     // BRED: this lane's two columns (2 fr, 2 fr + 1 of its 32) over every tile so far
  const unsigned short* wrow = wl + fr * LDW + kg * 8;                  // + (32 j + 16 q) * LDW + 32 ks
  const int mrow = wave * 16 + fr;
  // Software pipeline over the row tiles: the NEXT tile's loads are issued before this tile's stores, into the registers the
  // product / the epilogue arithmetic have just finished with.  vmcnt retires in order and counts stores: a wave that stores and
  // then loads must see its stores acknowledged before the loaded data can be waited for; with the loads in front, the stores of
  // tile i are only waited for together with the loads of tile i + 2.
  ExStream<KS, ACT, BRED> cur;
  cur.load_a(p, g * TM + mrow, kg);
  cur.load_src(p, g * TM + mrow, n0, kg);
  for (int tm = g; tm < p.row_tiles; tm += p.groups) {
    const int m_next = (tm + p.groups) * TM + mrow;                    // past the end: the clamped last row, never stored
    // ---- product: 8 blocks of 16 columns
    float4_t acc[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[b] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const ushort8_t fw = *(const ushort8_t*)(wrow + b * 16 * LDW + ks * 32);
        acc[b] = ExMfma<DT>::run(fw, cur.fa[ks], acc[b]);
      }
      // one k step's weight fragments at a time: without the fence hipcc hoists every LDS read of the unrolled product above the
      // first MFMA (8 KS fragments = up to 256 VGPRs, occupancy 1)
      __builtin_amdgcn_sched_barrier(0);
    }
    cur.load_a(p, m_next, kg);
    // ---- epilogue: pair j = blocks 2j, 2j + 1 -> columns n0 + 32 j + 8 kg + {0..7} of row m
    ushort8_t outv[4];
    float tg[BRED ? 32 : 1], tx[BRED ? 32 : 1];         // BRED: this row's 32 values of g and g xhat
    const bool live_cur = cur.live;
    const bool has_cur = ACT == 3 ? cur.has : true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = acc[2 * j][r]; v[4 + r] = acc[2 * j + 1][r]; }
      if (ACT) {
        float y[8];
        unpack8<DT>(cur.sv[j], y);
        const unsigned int bits8 = ACT == 2 ? (cur.mb[j] >> (8 * kg)) & 0xffu : 0xffu;   // byte 4 j + kg of the row's 16
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          if (ACT == 2) { if ((bits8 >> r) & 1u) v[r] += y[r]; }
          else if (ACT == 3) { if (has_cur) v[r] += y[r]; }
          else v[r] += y[r];
        }
      }
      const ushort8_t ov = pack8<DT>(v);
      outv[j] = ov;
      if (STATS) {
        float z[8];
        unpack8<DT>(ov, z);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float zz = live_cur ? z[r] : 0.f;
          s1[8 * j + r] += zz;
          s2[8 * j + r] += zz * zz;
        }
      }
      if (BRED) {                                      // (the arithmetic of bn_reduce_kernel on the rounded output)
        float z[8], tf[8];
        unpack8<DT>(ov, z);
        unpack8<DT>(cur.tv[j], tf);
        const unsigned b2 = (cur.mb2[j] >> (8 * kg)) & 0xffu;
        const float4_t m0 = *(const float4_t*)(mu2 + 32 * j + 8 * kg), m1 = *(const float4_t*)(mu2 + 32 * j + 8 * kg + 4);
        const float4_t r0 = *(const float4_t*)(mu2 + EX_TN + 32 * j + 8 * kg), r1 = *(const float4_t*)(mu2 + EX_TN + 32 * j + 8 * kg + 4);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float gz = (live_cur && ((b2 >> r) & 1u)) ? z[r] : 0.f;
          const float mu = r < 4 ? m0[r & 3] : m1[r & 3], rs = r < 4 ? r0[r & 3] : r1[r & 3];
          tg[8 * j + r] = gz;
          tx[8 * j + r] = gz * (tf[r] - mu) * rs;
        }
      }
    }
    if (BRED) {
      // 64 accumulators per lane do not fit beside the stream registers (the kernel spilled 150-190 VGPRs): the 16 row lanes of
      // a column group reduce-scatter the tile's values instead -- after xor-8 / 4 / 2 / 1 exchanges lane fr holds the sums of
      // the 16 rows for values 2 fr and 2 fr + 1 of the 32
      ex_reduce_scatter16(tg, fr, b1);
      ex_reduce_scatter16(tx, fr, b2s);
    }
    __builtin_amdgcn_sched_barrier(0);
    cur.load_src(p, m_next, n0, kg);
    __builtin_amdgcn_sched_barrier(0);
    // Full 128-byte lines per store instruction: rows come in (even, odd) lane pairs (fr, fr ^ 1); of a pair's two 64-byte
    // pieces (column blocks 2 h, 2 h + 1) the even lane stores row 2a's left piece then row 2a+1's left piece, the odd lane the
    // right pieces -- one dword-for-dword swap with the neighbour lane per piece pair (DPP), 8 lanes x 16 B = one full line of one
    // row per instruction instead of two half lines of two rows.
    {
      const bool odd = fr & 1;
      const int m_cur = tm * TM + mrow;                                 // (the load offsets are of the CLAMPED row: recompute)
      const bool live_other = (odd ? m_cur - 1 : m_cur + 1) < p.M;
      const long long o_own = (long long)m_cur * p.ldc + n0 + kg * 8;
      const long long o_other = odd ? o_own - p.ldc : o_own + p.ldc;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint4_t mine0 = __builtin_bit_cast(uint4_t, outv[2 * h]), mine1 = __builtin_bit_cast(uint4_t, outv[2 * h + 1]);
        const uint4_t give = odd ? mine0 : mine1;
        uint4_t got;
#pragma unroll
        for (int q = 0; q < 4; ++q) got[q] = (unsigned)__shfl_xor((int)give[q], 1, 64);
        // even lane: (row 2a, block 2h) own, then (row 2a+1, block 2h) received; odd lane: (row 2a, block 2h+1) received, then own
        const uint4_t first = odd ? got : mine0, second = odd ? mine1 : got;
        const long long o_first = (odd ? o_other : o_own) + 64 * h + (odd ? 32 : 0);
        const long long o_second = (odd ? o_own : o_other) + 64 * h + (odd ? 32 : 0);
        const bool live_first = odd ? live_other : live_cur, live_second = odd ? live_cur : live_other;
        if (live_first) *(uint4_t*)(p.C + o_first) = first;
        if (live_second) *(uint4_t*)(p.C + o_second) = second;
      }
    }
  }
  if (BRED) {
    // lane fr holds values 2 fr, 2 fr + 1 of its column group's 32: value i = column 32 (i >> 3) + 8 kg + (i & 7)
    __syncthreads();
    float* red = (float*)smem_raw;                                      // [NW waves][2][128]
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int i = 2 * fr + b, col = 32 * (i >> 3) + 8 * kg + (i & 7);
      red[(wave * 2 + 0) * EX_TN + col] = b1[b];
      red[(wave * 2 + 1) * EX_TN + col] = b2s[b];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * EX_TN; t += NW * 64) {
      const int which = t / EX_TN, col = t - which * EX_TN;
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) tot += red[(w * 2 + which) * EX_TN + col];
      p.stats[((long long)g * 2 + which) * p.N + n0 + col] = tot;
    }
  }
  if (STATS) {
    // rows: the 16 fr lanes of a wave, then the waves (LDS, reusing the weight tile's space after a barrier)
#pragma unroll
    for (int i = 0; i < 32; ++i) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        s1[i] += __shfl_xor(s1[i], o, 64);
        s2[i] += __shfl_xor(s2[i], o, 64);
      }
    }
    __syncthreads();
    float* red = (float*)smem_raw;                                      // [NW waves][2][128]
    if (fr == 0) {
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const int col = 32 * (i >> 3) + 8 * kg + (i & 7);
        red[(wave * 2 + 0) * EX_TN + col] = s1[i];
        red[(wave * 2 + 1) * EX_TN + col] = s2[i];
      }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * EX_TN; t += NW * 64) {
      const int which = t / EX_TN, col = t - which * EX_TN;
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) tot += red[(w * 2 + which) * EX_TN + col];
      p.stats[((long long)g * 2 + which) * p.N + n0 + col] = tot;
    }
  }
}

// Number of partial rows the stats variant writes for (M, N): the caller sizes `stats` for it and folds that many rows.
static int ex_rows_per_tile(int K) { return K > 128 ? 128 : 64; }
extern "C" int dle_gemm_expand_groups(int M, int N, int K) {
  const int tm = ex_rows_per_tile(K), row_tiles = (M + tm - 1) / tm, col_tiles = N / EX_TN;
  static const int per_cu_env = getenv("DLE_EXPAND_WG_PER_CU") ? atoi(getenv("DLE_EXPAND_WG_PER_CU")) : 0;   // probe knob
  const int per_cu = per_cu_env > 0 ? per_cu_env : (K > 128 ? 2 : 4);  // workgroups a CU holds at once (16 waves per CU)
  const int resident = 256 * per_cu;
  int groups = (resident + col_tiles - 1) / (col_tiles > 0 ? col_tiles : 1);
  if (groups > row_tiles) groups = row_tiles;
  return (groups + 7) / 8 * 8;                                         // XCD-aware dealing (see the kernel); idle groups run no tile
}

// 1: launched; 0: outside the envelope (the caller goes on to the tile kernels); > 1: error.
// act: 0 none, 1 DLE_ACT_ADD, 2 DLE_ACT_ADD_MASKED (bits = keep bits of the addend, bit (m ldc + n) & 7 of byte (m ldc + n) >> 3).
extern "C" int dle_gemm_expand_try(const void* A, const void* B, void* C, const void* src, const void* bits, float* stats, int M,
                                   int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int b_kc, int in_dtype, int out_dtype,
                                   int act, hipStream_t stream) {
  if (M < 4096 || (K != 64 && K != 128 && K != 256) || (N % EX_TN) != 0 || N < 2 * K || out_dtype != in_dtype) return 0;
  if (((((uintptr_t)A) | ((uintptr_t)B) | ((uintptr_t)C) | ((uintptr_t)src)) & 15) != 0 || (lda & 7) || (ldb & 7) || (ldc & 7)) return 0;
  if (act < 0 || act > 2) return 0;
  if (act == 2 && !bits) return 0;
  // the masked epilogue reads a row's 16 keep bytes with ONE 16-byte load at bits + ((m ldc + n0) >> 3): needs ldc % 128 == 0 and
  // a 16-byte aligned bit plane (other pitches go to the tile kernel's per-byte form)
  if (act == 2 && ((ldc % EX_TN) != 0 || (((uintptr_t)bits) & 15) != 0)) return 0;
  if (act && !src) return 0;
  ExpandArgs p = {(const unsigned short*)A, (const unsigned short*)B, (unsigned short*)C, (const unsigned short*)src,
                  (const unsigned char*)bits, stats, M, N, K, (long long)lda, (long long)ldb, (long long)ldc, b_kc, 0, 0, 0};
  const int tmr = ex_rows_per_tile(K);
  p.row_tiles = (M + tmr - 1) / tmr;
  p.col_tiles = N / EX_TN;
  p.groups = dle_gemm_expand_groups(M, N, K);
  const int lds_bytes = EX_TN * (K + EX_PAD) * 2 > 8 * 2 * EX_TN * 4 ? EX_TN * (K + EX_PAD) * 2 : 8 * 2 * EX_TN * 4;
  const dim3 grid((unsigned)(p.groups * p.col_tiles)), block(tmr * 4);
#define GO(DT, KS, ACT, ST)                                                                                              \
  do {                                                                                                                   \
    constexpr int NW_ = KS > 4 ? 8 : 4;                                                                                  \
    static bool attr_set = false;                                                                                        \
    if (!attr_set) {                                                                                                     \
      (void)hipFuncSetAttribute((const void*)gemm_expand_kernel<DT, KS, ACT, ST, NW_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                EX_TN * (KS * 32 + EX_PAD) * 2);                                                         \
      attr_set = true;                                                                                                   \
    }                                                                                                                    \
    hipLaunchKernelGGL((gemm_expand_kernel<DT, KS, ACT, ST, NW_>), grid, block, lds_bytes, stream, p);                   \
  } while (0)
#define PICK_ACT(DT, KS)                                              \
  do {                                                                \
    if (stats) { if (act != 0) return 0; GO(DT, KS, 0, true); }       \
    else if (act == 0) GO(DT, KS, 0, false);                          \
    else if (act == 1) GO(DT, KS, 1, false);                          \
    else GO(DT, KS, 2, false);                                        \
  } while (0)
#define PICK_K(DT)                                   \
  do {                                               \
    if (K == 64) PICK_ACT(DT, 2);                    \
    else if (K == 128) PICK_ACT(DT, 4);              \
    else PICK_ACT(DT, 8);                            \
  } while (0)
  if (in_dtype == DLE_F16) PICK_K(DLE_F16); else PICK_K(DLE_BF16);
#undef PICK_K
#undef PICK_ACT
#undef GO
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { dle_set_error("gemm_expand launch failed: %s", hipGetErrorString(e)); return (int)e + 1000; }
  return 1;
}

// The masked-addend form (act 2) that ALSO takes the backward reduction of the BatchNorm its output flows into (ExpandArgs BRED):
// partial [dle_gemm_expand_groups(M, N, K)][2][N] receives per-workgroup-group rows of (sum g, sum g xhat); the caller folds them
// with dle_bn_bwd_finish.  In a ResNet bottleneck chain the output of conv1's data gradient IS the gradient of the previous
// block's output: its bn3 reduction re-read dy + t + mask (848 MB at 56 x 56 x 256, batch 256) -- here t and the mask are read
// once more beside the store, dy not at all.  1: launched; 0: outside the envelope.
extern "C" int dle_gemm_expand_masked_bnred(const void* A, const void* B, void* C, const void* src, const void* bits, const void* t2,
                                            const void* bits2, const float* mean2, const float* rstd2, float* partial,
                                            int64_t partial_bytes, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                                            int b_kc, int dtype, hipStream_t stream) {
  static const char* pin = getenv("DLE_GEMM_EXPAND");
  if (pin && atoi(pin) == 0) return 0;
  // (K = 256 is built and correct, but its 8-wave workgroup spills and runs slower than the two launches: 104 us against 60 + 39 at
  //  50176 x 1024 x 256; DLE_GEMM_BNRED_K256=1 lets it through for measurement)
  const char* k256e = getenv("DLE_GEMM_BNRED_K256");                     // (read per call: tests switch it inside one process)
  const int k256 = k256e ? atoi(k256e) : 0;
  if (M < 4096 || (K != 64 && K != 128 && !(K == 256 && k256)) || (N % EX_TN) != 0 || N < 2 * K) return 0;
  if (dtype != DLE_F16 && dtype != DLE_BF16) return 0;
  if (!A || !B || !C || !src || !bits || !t2 || !bits2 || !mean2 || !rstd2 || !partial) return 0;
  if (((((uintptr_t)A) | ((uintptr_t)B) | ((uintptr_t)C) | ((uintptr_t)src) | ((uintptr_t)t2) | ((uintptr_t)bits) | ((uintptr_t)bits2)) & 15) != 0 ||
      (lda & 7) || (ldb & 7) || (ldc % EX_TN) != 0)
    return 0;
  ExpandArgs p = {(const unsigned short*)A, (const unsigned short*)B, (unsigned short*)C, (const unsigned short*)src,
                  (const unsigned char*)bits, partial, M, N, K, (long long)lda, (long long)ldb, (long long)ldc, b_kc, 0, 0, 0};
  p.t2 = (const unsigned short*)t2; p.bits2 = (const unsigned char*)bits2; p.mean2 = mean2; p.rstd2 = rstd2;
  const int tmr = ex_rows_per_tile(K);
  p.row_tiles = (M + tmr - 1) / tmr;
  p.col_tiles = N / EX_TN;
  p.groups = dle_gemm_expand_groups(M, N, K);
  if (partial_bytes < (long long)p.groups * 2 * N * 4) return 0;
  size_t lds_bytes = (size_t)EX_TN * (K + EX_PAD) * 2 + 2 * EX_TN * 4;
  if (lds_bytes < (size_t)8 * 2 * EX_TN * 4) lds_bytes = (size_t)8 * 2 * EX_TN * 4;
  const dim3 grid((unsigned)(p.groups * p.col_tiles)), block(tmr * 4);
#define GO(DT, KS)                                                                                                       \
  do {                                                                                                                   \
    constexpr int NW_ = KS > 4 ? 8 : 4;                                                                                  \
    static bool attr_set = false;                                                                                        \
    if (!attr_set) {                                                                                                     \
      (void)hipFuncSetAttribute((const void*)gemm_expand_kernel<DT, KS, 2, false, NW_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                EX_TN * (KS * 32 + EX_PAD) * 2 + 2 * EX_TN * 4);                                         \
      attr_set = true;                                                                                                   \
    }                                                                                                                    \
    hipLaunchKernelGGL((gemm_expand_kernel<DT, KS, 2, false, NW_, true>), grid, block, lds_bytes, stream, p);            \
  } while (0)
#define PICK_K(DT) do { if (K == 64) GO(DT, 2); else if (K == 128) GO(DT, 4); else GO(DT, 8); } while (0)
  if (dtype == DLE_F16) PICK_K(DLE_F16); else PICK_K(DLE_BF16);
#undef PICK_K
#undef GO
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { dle_set_error("gemm_expand_masked_bnred launch failed: %s", hipGetErrorString(e)); return (int)e + 1000; }
  return 1;
}

// C [M = N_img * H * W, N] = A [M, K] B^T + zero-stuffed(compact [N_img * (H/2) * (W/2), N]): the 1x1 data gradient of a
// bottleneck's first convolution + the gradient of the stride-2 1x1 downsample branch, whose zero-stuffed full-resolution form
// (dle_upsample_zero: 4x the bytes, written and read once) is never materialised.  1: launched; 0: outside the envelope.
extern "C" int dle_gemm_expand_add_up2(const void* A, const void* B, void* C, const void* compact, int M, int N, int K, int64_t lda,
                                       int64_t ldb, int64_t ldc, int64_t ld_compact, int b_kc, int H, int W, int dtype,
                                       hipStream_t stream) {
  if (M < 4096 || (K != 64 && K != 128 && K != 256) || (N % EX_TN) != 0 || N < 2 * K) return 0;
  if (dtype != DLE_F16 && dtype != DLE_BF16) return 0;
  if ((H & 1) || (W & 1) || H <= 0 || W <= 0 || (M % (H * W)) != 0 || !compact) return 0;
  if (((((uintptr_t)A) | ((uintptr_t)B) | ((uintptr_t)C) | ((uintptr_t)compact)) & 15) != 0 || (lda & 7) || (ldb & 7) || (ldc & 7) ||
      (ld_compact & 7)) return 0;
  static const char* pin = getenv("DLE_GEMM_EXPAND");
  if (pin && atoi(pin) == 0) return 0;
  ExpandArgs p = {(const unsigned short*)A, (const unsigned short*)B, (unsigned short*)C, (const unsigned short*)compact, nullptr,
                  nullptr, M, N, K, (long long)lda, (long long)ldb, (long long)ldc, b_kc, 0, 0, 0};
  p.dHW = make_fastdiv(H * W); p.dW = make_fastdiv(W);
  p.up_HW = H * W; p.up_W = W; p.up_P = H / 2; p.up_Q = W / 2; p.ld_src = ld_compact;
  const int tmr = ex_rows_per_tile(K);
  p.row_tiles = (M + tmr - 1) / tmr;
  p.col_tiles = N / EX_TN;
  p.groups = dle_gemm_expand_groups(M, N, K);
  const int lds_bytes = EX_TN * (K + EX_PAD) * 2;
  const dim3 grid((unsigned)(p.groups * p.col_tiles)), block(tmr * 4);
#define GO3(DT, KS)                                                                                                      \
  do {                                                                                                                   \
    constexpr int NW_ = KS > 4 ? 8 : 4;                                                                                  \
    static bool attr_set = false;                                                                                        \
    if (!attr_set) {                                                                                                     \
      (void)hipFuncSetAttribute((const void*)gemm_expand_kernel<DT, KS, 3, false, NW_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                EX_TN * (KS * 32 + EX_PAD) * 2);                                                         \
      attr_set = true;                                                                                                   \
    }                                                                                                                    \
    hipLaunchKernelGGL((gemm_expand_kernel<DT, KS, 3, false, NW_>), grid, block, lds_bytes, stream, p);                  \
  } while (0)
#define PICK3(DT) do { if (K == 64) GO3(DT, 2); else if (K == 128) GO3(DT, 4); else GO3(DT, 8); } while (0)
  if (dtype == DLE_F16) PICK3(DLE_F16); else PICK3(DLE_BF16);
#undef PICK3
#undef GO3
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { dle_set_error("gemm_expand_add_up2 launch failed: %s", hipGetErrorString(e)); return (int)e + 1000; }
  return 1;
}
