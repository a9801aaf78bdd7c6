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
  // fused LSTM cell (LSTM instantiation only): N = 4 H gate columns (i | f | g | o blocks of H); tile tn owns units 8 tn .. 8 tn + 7
  int H;
  const float* c_prev;
  float* c_out;
  unsigned short* hd[3];
  long long ldh[3];
  const unsigned char* keep;
  long long keep_index;
  float inv_keep;
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
template <int TN, int NW, bool LSTM = false>
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
      // LSTM: tile-local weight row r <-> gate r >> 3, unit n0 / 4 + (r & 7): the four gate rows of 8 units (n0 = 32 tn)
      const int rl = row - SM_TS;
      const int g = is_a ? m0 + row : LSTM ? (rl >> 3) * p.H + (n0 >> 2) + (rl & 7) : n0 + rl;
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
__device__ __forceinline__ float sm_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }

template <int DT, int TN, int NW, int NST, bool LSTM = false>
__global__ __launch_bounds__(NW * 64) void gemm_smallm_kernel(SmallMArgs p) {
  typedef SmallMLoader<TN, NW, LSTM> L;
  static_assert(!LSTM || (TN == 32 && NW == 8), "the fused cell runs on the 64 x 32 tile");
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
  // LSTM: everything the cell reads besides the product (bias, addend, previous cell state, keep bits) is requested NOW: behind the
  // K loop these would be two dependent global round trips (~2.5 us of a ~12 us launch) with nothing to hide them
  float4_t pf_bias = {0.f, 0.f, 0.f, 0.f};
  ushort4_t pf_src = {0, 0, 0, 0};
  float pf_c = 0.f;
  unsigned pf_keep = 0xffu;
  if constexpr (LSTM) {
    const int fr_ = lane & 15, kg_ = lane >> 4, mb_ = wave & 3, nb_ = wave >> 2;
    const int m_ = m0 + mb_ * 16 + fr_, c_ = nb_ * 16 + 4 * kg_;
    const long long ng_ = (long long)(c_ >> 3) * p.H + (n0 >> 2) + (c_ & 7);
    const int mc_ = m_ < p.M ? m_ : p.M - 1;
    if (p.bias) pf_bias = *(const float4_t*)(p.bias + ng_);
    if (p.act_add) pf_src = *(const ushort4_t*)(p.src + (long long)mc_ * p.ldc + ng_);
    const int row2 = threadIdx.x >> 3, m2 = m0 + row2 < p.M ? m0 + row2 : p.M - 1;
    const long long idx2 = (long long)m2 * p.H + (n0 >> 2) + (threadIdx.x & 7);
    pf_c = p.c_prev[idx2];
    if (p.keep) pf_keep = p.keep[(p.keep_index + idx2) >> 3];
  }
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
  if constexpr (LSTM) {
    // ---- fused LSTM cell (tacotron2/model.py:425-444: nn.LSTMCell + F.dropout on the hidden state).  The tile holds the four
    // gate pre-activations of 8 units for 64 samples.  They meet through LDS (the stages are idle now): [64][32] fp32, column
    // 8 gate + unit; then one thread per (sample, unit) runs the cell, and the activations / hidden states leave as 16-byte rows.
    const int H = p.H, u0 = n0 >> 2;
    __syncthreads();                                   // every wave is done reading the last stage
    float* pre = (float*)smem_raw;                     // [64][33] (padded)
    unsigned short* hb = (unsigned short*)(pre + 64 * 33);   // [64][8] hidden states (16-bit)
    {
      const int row = mb * 16 + fr, m = m0 + row;
      const int c = nb * 16 + 4 * kg;                  // tile column of acc[0]: gate c >> 3, units (c & 7) .. + 3
      const int gate = c >> 3, un = c & 7;
      const long long ng = (long long)gate * H + u0 + un;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[r] * p.alpha;
      (void)ng; (void)m;
      if (p.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += pf_bias[r];
      }
      if (p.act_add) {                                 // (same order of additions as the unfused epilogue: bit-identical gates)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += Elem<DT>::to_f32(pf_src[r]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)                      // the pre-activation is a 16-bit tensor in the unfused sequence: same rounding
        pre[row * 33 + c + r] = Elem<DT>::to_f32(Elem<DT>::from_f32(v[r]));
    }
    __syncthreads();
    {
      const int row = threadIdx.x >> 3, un = threadIdx.x & 7, m = m0 + row;
      float* pr = pre + row * 33 + un;
      const float gi = sm_sigmoid(pr[0]), gf = sm_sigmoid(pr[8]), gg = fast_tanh(pr[16]), go = sm_sigmoid(pr[24]);
      const long long idx = (long long)(m < p.M ? m : p.M - 1) * H + u0 + un;
      const float c = gf * pf_c + gi * gg;
      float h = go * fast_tanh(c);
      if (p.keep) h = ((pf_keep >> ((p.keep_index + idx) & 7)) & 1u) ? h * p.inv_keep : 0.f;
      if (m < p.M) p.c_out[idx] = c;
      pr[0] = gi; pr[8] = gf; pr[16] = gg; pr[24] = go;
      hb[row * 8 + un] = Elem<DT>::from_f32(h);
    }
    __syncthreads();
    if (threadIdx.x < 256) {                           // activations: 64 rows x 4 gates, 16 bytes each
      const int row = threadIdx.x >> 2, gate = threadIdx.x & 3, m = m0 + row;
      if (m < p.M) {
        float a[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) a[r] = pre[row * 33 + gate * 8 + r];
        *(ushort8_t*)((unsigned short*)p.C + (long long)m * p.ldc + (long long)gate * H + u0) = pack8<DT>(a);
      }
    } else if (threadIdx.x < 256 + 192) {              // hidden state: 64 rows x up to 3 destinations
      const int t = threadIdx.x - 256, row = t & 63, d = t >> 6, m = m0 + row;
      if (m < p.M && p.hd[d]) *(ushort8_t*)(p.hd[d] + (long long)m * p.ldh[d] + u0) = *(const ushort8_t*)(hb + row * 8);
    }
    return;
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


// One decoder-step LSTMCell in ONE launch: gates = x W^T (+ bias) (+ addend), cell, dropout, hidden state to its consumers'
// operand buffers.  See include/dle_mi355x.h (dle_t2_lstm_gemm_fwd).  0 = launched, -1 = argument error.
extern "C" int dle_t2_lstm_gemm_fwd(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* bias, const void* addend,
                                    const float* c_prev, float* c_out, void* gates, int64_t ld_g, void* d0, int64_t ld0, void* d1,
                                    int64_t ld1, void* d2, int64_t ld2, const void* keep, int64_t keep_index, float inv_keep, int B,
                                    int H, int K, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(x && w && c_prev && c_out && gates && B > 0 && H > 0 && K > 0, "t2_lstm_gemm_fwd: bad args");
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "t2_lstm_gemm_fwd: 16-bit dtypes only (got %d)", dtype);
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  DLE_CHECK_ARG((H & 7) == 0 && (K & 7) == 0 && (ldx & 7) == 0 && (ldw & 7) == 0 && (ld_g & 7) == 0 && al16(x) && al16(w) &&
                al16(gates) && al16(c_prev) && al16(c_out) && (!bias || al16(bias)) && (!addend || (((uintptr_t)addend) & 7) == 0) &&
                (!d0 || (al16(d0) && (ld0 & 7) == 0)) && (!d1 || (al16(d1) && (ld1 & 7) == 0)) && (!d2 || (al16(d2) && (ld2 & 7) == 0)),
                "t2_lstm_gemm_fwd: H, K and every row pitch must be multiples of 8 with 16-byte aligned bases");
  DLE_CHECK_ARG((long long)B * ldx * 2 < 0xFFFFFFE0LL && (long long)4 * H * ldw * 2 < 0xFFFFFFE0LL, "t2_lstm_gemm_fwd: operand above 4 GiB");
  SmallMArgs p = {(const unsigned short*)x, (const unsigned short*)w, gates, bias, (const unsigned short*)addend, B, 4 * H, K,
                  (long long)ldx, (long long)ldw, (long long)ld_g, dtype, addend ? 1 : 0, 0, 1.0f};
  p.H = H; p.c_prev = c_prev; p.c_out = c_out;
  p.hd[0] = (unsigned short*)d0; p.hd[1] = (unsigned short*)d1; p.hd[2] = (unsigned short*)d2;
  p.ldh[0] = ld0; p.ldh[1] = ld1; p.ldh[2] = ld2;
  p.keep = (const unsigned char*)keep; p.keep_index = keep_index; p.inv_keep = inv_keep;
  const int tiles = ((B + SM_TS - 1) / SM_TS) * (H / 8);
  const bool deep = tiles <= 256;
#define GOL(DT, NST)                                                                                                          \
  do {                                                                                                                         \
    constexpr int lds_bytes = NST * (SM_TS + 32) * SM_BKE * 2;                                                                 \
    static bool attr_set = false;                                                                                              \
    if (!attr_set) {                                                                                                           \
      (void)hipFuncSetAttribute((const void*)gemm_smallm_kernel<DT, 32, 8, NST, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); \
      attr_set = true;                                                                                                         \
    }                                                                                                                          \
    hipLaunchKernelGGL((gemm_smallm_kernel<DT, 32, 8, NST, true>), dim3(tiles), dim3(512), lds_bytes, stream, p);              \
  } while (0)
  if (dtype == DLE_F16) { if (deep) GOL(DLE_F16, 4); else GOL(DLE_F16, 3); }
  else { if (deep) GOL(DLE_BF16, 4); else GOL(DLE_BF16, 3); }
#undef GOL
  DLE_LAUNCH_CHECK();
  return 0;
}
