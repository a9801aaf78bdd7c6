// Few-row GEMM for gfx950: C[M <= 256, N] = epilogue(alpha * A[M, K] B[N, K]^T), both operands k-contiguous.
//
// The recurrent loops of the path (Tacotron2's attention / decoder LSTMCells and their data gradients,
// SpeechSynthesis/Tacotron2/tacotron2/model.py:405-455, one row per sample: M = batch = 128; the encoder LSTM steps; the
// pooler / NSP heads of BERT) multiply a handful of rows by a wide weight matrix once per time step.  On the 128x128 tile of
// gemm_dma.hip such a product is 12-32 workgroups on 256 CUs, each walking the whole K range alone: 28-54 us per launch,
// 64 TFLOP/s, and ~9,000 of them per Tacotron2 iteration.  These products are bound by how fast the WEIGHTS (12-21 MB per
// launch, L2 / Infinity-Cache resident across steps) reach the CUs, so the tile is chosen for bytes per workgroup, not for MFMA
// reuse: 64 rows x TN (32 | 16) columns x the full K range per workgroup -> ceil(M / 64) x N / TN workgroups (256 for a
// 128 x 4096 gate matrix), K * 2 * (64 + TN) bytes each.
//  * operands stream HBM/L2 -> LDS by LDS-DMA (buffer_load ... lds, 16 B / lane, 1 KiB per wave instruction) in K chunks of
//    128 elements (256-byte rows, full cache lines), 3-7 stages (all but one in flight while one chunk is multiplied);
//    counted vmcnt (never 0 inside the loop), LDS-only barriers; every DMA goes through inline asm (gemm_tiles.h) so hipcc's
//    wait-count pass inserts nothing;
//  * the LDS image of a DMA is lane-linear, so bank conflicts are removed on the SOURCE address: slot s of row r holds the
//    16-byte chunk s ^ (r & 15); a fragment read (16 rows x one k chunk per 16-lane group) then touches 16 distinct slots of
//    the 256-byte bank window;
//  * v_mfma_f32_16x16x32 with the WEIGHT rows as the first operand: a lane ends up with 4 consecutive output columns of one row
//    (8 / 16-byte stores); one 16 x 16 block per wavefront (8 wavefronts for TN = 32).
// Epilogue: alpha, bias[n], 16-bit addend (DLE_ACT_ADD), fp32 accumulate, fp32 or 16-bit output.
#include "gemm_tiles.h"

#define SM_BKE 128                     // K elements per chunk
#define SM_TS 64                       // rows of A per workgroup

struct SmallMArgs {
  const unsigned short* A;
  const unsigned short* B;
  void* C;
  const float* bias;
  const unsigned short* src;
  int M, N, K;
  long long lda, ldb, ldc;
  int out_dtype, act_add, accumulate;
  float alpha;
};

template <int DT> struct Mfma16x32;
template <> struct Mfma16x32<DLE_F16> {
  static __device__ __forceinline__ float4_t run(ushort8_t a, ushort8_t b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
  }
};
template <> struct Mfma16x32<DLE_BF16> {
  static __device__ __forceinline__ float4_t run(ushort8_t a, ushort8_t b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};

// One K chunk of the workgroup's (SM_TS + TN)-row operand panel into `stage`.  NPW pieces per wavefront; piece q covers rows
// 4q .. 4q + 3 (A rows first); lane l -> row 4q + (l >> 4), LDS slot l & 15, source chunk slot ^ (row & 15).
template <int TN, int NW>
struct SmallMLoader {
  static constexpr int R = SM_TS + TN, NPW = R / 4 / NW, NPA = SM_TS / 4 / NW;     // pieces per wave: all / of the A rows
  unsigned off[NPW];        // byte offset of (row, chunk) inside its operand, k0 = 0
  int kin[NPW];             // first k element of the lane's chunk
  bool ok[NPW];
  __device__ __forceinline__ void init(const SmallMArgs& p, int wave, int lane, int m0, int n0) {
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      const int q = wave + NW * j, row = 4 * q + (lane >> 4), chunk = (lane & 15) ^ (row & 15);
      const bool is_a = j < NPA;                 // (wave + NW j < 16 <=> j < 16 / NW: a piece lies entirely in one operand)
      const int g = is_a ? m0 + row : n0 + row - SM_TS;
      ok[j] = is_a ? g < p.M : g < p.N;
      kin[j] = chunk * 8;
      off[j] = (unsigned)(((long long)g * (is_a ? p.lda : p.ldb) + chunk * 8) * 2);
    }
  }
  __device__ __forceinline__ void issue(const SmallMArgs& p, unsigned short* stage, int wave, int k0) {
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      const int q = wave + NW * j;
      const bool valid = ok[j] && k0 + kin[j] < p.K;
      const unsigned voff = valid ? off[j] + (unsigned)k0 * 2u : OOB_OFF;
      // (the descriptor is rebuilt at the use: its words reach the asm statement straight from readfirstlane and stay in SGPRs;
      //  `wave` itself must come through readfirstlane too, or hipcc treats everything derived from it as divergent)
      if (j < NPA) dma16_raw(rsrc_words(p.A), stage + q * 512, voff);
      else dma16_raw(rsrc_words(p.B), stage + q * 512, voff);
    }
  }
};

// s_waitcnt vmcnt(C * NPW): this wave's pieces of everything but its C youngest chunks have landed (vmcnt retires in order)
template <int N> __device__ __forceinline__ void sm_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int NPW, int MAXC>
__device__ __forceinline__ void sm_wait_chunks(int c) {
  static_for<0, MAXC + 1>([&](auto I) __attribute__((always_inline)) {
    if (c == decltype(I)::value) sm_wait_vm<decltype(I)::value * NPW>();
  });
}

// NST stages of (64 + TN) x 256 B: NST - 1 chunks in flight while one is multiplied.  (Deeper rings -- 6-7 stages, ~140 KiB per
// workgroup -- measured 5-20 % SLOWER: the limit was the issue cost of the pieces, next comment, not the bytes in flight.)
// NW wavefronts: 8 for the 64 x 32 tile, 4 for 64 x 16 -- ONE 16 x 16 output block per wavefront.  An LDS-DMA piece costs the
// issuing wavefront ~100-180 cycles (MI355X_MICROARCH.md): with 4 wavefronts issuing 6 pieces per chunk each, the first version
// of this kernel was bound by that issue cost (42 GB/s per CU whatever the pipeline depth); 8 wavefronts issue 3 each.
template <int DT, int TN, int NW, int NST>
__global__ __launch_bounds__(NW * 64) void gemm_smallm_kernel(SmallMArgs p) {
  typedef SmallMLoader<TN, NW> L;
  static_assert(4 * (TN / 16) == NW, "one 16x16 block per wavefront");
  constexpr int R = L::R, STAGE = R * SM_BKE;      // halves per stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* lds = (unsigned short*)smem_raw;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tiles_n = (p.N + TN - 1) / TN;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int m0 = tm * SM_TS, n0 = tn * TN;
  L ld;
  ld.init(p, wave, lane, m0, n0);
  const int nk = (p.K + SM_BKE - 1) / SM_BKE;
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) ld.issue(p, lds + s * STAGE, wave, s * SM_BKE);
  float4_t acc = {0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, kg = lane >> 4;
  const int mb = wave & 3, nb = wave >> 2;         // this wavefront's block: rows 16 mb.., columns 16 nb..
  for (int kt = 0; kt < nk; ++kt) {
    // this wave's pieces of chunk kt have landed; the (at most NST - 2) younger chunks stay in flight
    const int younger = nk - 1 - kt;
    sm_wait_chunks<L::NPW, NST - 2>(younger < NST - 2 ? younger : NST - 2);
    lds_barrier();                           // ... and everybody's; the stage the next chunk will overwrite is no longer read
    if (kt + NST - 1 < nk) ld.issue(p, lds + ((kt + NST - 1) % NST) * STAGE, wave, (kt + NST - 1) * SM_BKE);
    const unsigned short* st = lds + (kt % NST) * STAGE;
    const unsigned short* xa = st + (mb * 16 + fr) * SM_BKE;
    const unsigned short* wa = st + (SM_TS + nb * 16 + fr) * SM_BKE;
#pragma unroll
    for (int ks = 0; ks < SM_BKE / 32; ++ks) {
      const int slot = (ks * 4 + kg) ^ fr;
      const ushort8_t fx = *(const ushort8_t*)(xa + slot * 8);
      const ushort8_t fw = *(const ushort8_t*)(wa + slot * 8);
      acc = Mfma16x32<DT>::run(fw, fx, acc);
    }
  }
  // lane: row m = m0 + 16 mb + (lane & 15), columns n0 + 16 nb + 4 (lane >> 4) + {0..3}
  const int m = m0 + mb * 16 + fr;
  if (m >= p.M) return;
  {
    const int n = n0 + nb * 16 + 4 * kg;
    if (n >= p.N) return;
    const int nval = p.N - n < 4 ? p.N - n : 4;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = acc[r] * p.alpha;
    if (p.bias) {
#pragma unroll
      for (int r = 0; r < 4; ++r) if (r < nval) v[r] += p.bias[n + r];
    }
    const long long o = (long long)m * p.ldc + n;
    const bool vec = nval == 4 && (p.ldc & 3) == 0;
    if (p.act_add) {
      const unsigned short* s = p.src + o;
      if (vec) {
        const ushort4_t sv = *(const ushort4_t*)s;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += Elem<DT>::to_f32(sv[r]);
      } else {
        for (int r = 0; r < nval; ++r) v[r] += Elem<DT>::to_f32(s[r]);
      }
    }
    if (p.out_dtype == DLE_F32) {
      float* c = (float*)p.C + o;
      if (p.accumulate) for (int r = 0; r < nval; ++r) v[r] += c[r];
      if (vec) *(float4_t*)c = (float4_t){v[0], v[1], v[2], v[3]};
      else for (int r = 0; r < nval; ++r) c[r] = v[r];
    } else {
      unsigned short* c = (unsigned short*)p.C + o;
      ushort4_t ov;
#pragma unroll
      for (int r = 0; r < 4; ++r) ov[r] = p.out_dtype == DLE_F16 ? Elem<DLE_F16>::from_f32(v[r]) : Elem<DLE_BF16>::from_f32(v[r]);
      if (vec) *(ushort4_t*)c = ov;
      else for (int r = 0; r < nval; ++r) c[r] = ov[r];
    }
  }
}

// 1: launched; 0: outside the envelope (the caller goes on to the other kernels); > 1: error.
extern "C" int dle_gemm_smallm_try(const void* A, const void* B, void* C, const float* bias, const void* src, int M, int N, int K,
                                   int64_t lda, int64_t ldb, int64_t ldc, int in_dtype, int out_dtype, int act_add, int accumulate,
                                   float alpha, hipStream_t stream) {
  static const int mode = getenv("DLE_GEMM_SMALLM") ? atoi(getenv("DLE_GEMM_SMALLM")) : 1;
  if (!mode || M < 1 || M > 256 || N < 8 || K < 8) return 0;
  if (((((uintptr_t)A) | ((uintptr_t)B)) & 15) != 0 || (lda & 7) != 0 || (ldb & 7) != 0 || (K & 7) != 0) return 0;
  if ((long long)M * lda * 2 >= 0xFFFFFFE0LL || (long long)N * ldb * 2 >= 0xFFFFFFE0LL) return 0;
  if (out_dtype != DLE_F32 && out_dtype != in_dtype) return 0;
  if (out_dtype != DLE_F32 && ((((uintptr_t)C) | ((uintptr_t)src)) & 7) != 0) return 0;
  if (out_dtype == DLE_F32 && (((uintptr_t)C) & 15) != 0) return 0;
  if (act_add && (out_dtype != in_dtype || (((uintptr_t)src) & 7) != 0)) return 0;
  // worth it only when the K range is long enough to stream (a 128-element chunk per stage) or the other kernels would leave
  // most of the chip idle anyway
  SmallMArgs p = {(const unsigned short*)A, (const unsigned short*)B, C, bias, (const unsigned short*)src, M, N, K,
                  (long long)lda, (long long)ldb, (long long)ldc, out_dtype, act_add, accumulate, alpha};
  const int tm = (M + SM_TS - 1) / SM_TS;
  // 64 x 32 tiles (8 wavefronts) unless that leaves more than half of the CUs without a workgroup while 64 x 16 tiles would not
  const int t32 = tm * ((N + 31) / 32), t16 = tm * ((N + 15) / 16);
  bool wide = N > 16 && (t32 >= 128 || t16 > 256);
  int nst_pin = 0;
  {
    // tools/probes/smallm_policy.py: DLE_GEMM_SMALLM_TN = 16 | 32 pins the tile, DLE_GEMM_SMALLM_NST the ring depth (read per
    // call)
    const char* pin = getenv("DLE_GEMM_SMALLM_TN");
    if (pin) {
      const int tn = atoi(pin);
      if (tn == 16) wide = false;
      else if (tn == 32 && N > 16) wide = true;
      if (getenv("DLE_GEMM_SMALLM_NST")) nst_pin = atoi(getenv("DLE_GEMM_SMALLM_NST"));
    }
  }
  const int tiles = wide ? t32 : t16;
  const bool deep = tiles <= 256;                 // at most one workgroup per CU: four stages; else three (two workgroups per CU)
#define GO(DT, TN, NW, NST)                                                                                                    \
  do {                                                                                                                         \
    constexpr int lds_bytes = NST * (SM_TS + TN) * SM_BKE * 2;                                                                 \
    static bool attr_set = false;                                                                                              \
    if (!attr_set) {                                                                                                           \
      (void)hipFuncSetAttribute((const void*)gemm_smallm_kernel<DT, TN, NW, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); \
      attr_set = true;                                                                                                         \
    }                                                                                                                          \
    hipLaunchKernelGGL((gemm_smallm_kernel<DT, TN, NW, NST>), dim3(tiles), dim3(NW * 64), lds_bytes, stream, p);               \
  } while (0)
#define PICK(DT)                                                            \
  do {                                                                      \
    if (wide) {                                                             \
      if (nst_pin == 6) GO(DT, 32, 8, 6);                                   \
      else if (nst_pin == 3 || (!deep && nst_pin != 4)) GO(DT, 32, 8, 3);   \
      else GO(DT, 32, 8, 4);                                                \
    } else {                                                                \
      if (nst_pin == 7) GO(DT, 16, 4, 7); else GO(DT, 16, 4, 4);            \
    }                                                                       \
  } while (0)
  if (in_dtype == DLE_F16) PICK(DLE_F16); else PICK(DLE_BF16);
#undef PICK
#undef GO
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { dle_set_error("gemm_smallm launch failed: %s", hipGetErrorString(e)); return (int)e + 1000; }
  return 1;
}
