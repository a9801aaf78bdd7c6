// MFMA GEMM, LDS-DMA fed (buffer_load ... lds), for gfx950 -- the fast path behind dle_gemm.
//
//   C[M,N] = epilogue( alpha * sum_k A(m,k) * B(n,k) )      (same contract as gemm.hip, which stays as the
//                                                            fallback for unaligned / tiny shapes)
// Why a second kernel: gemm.hip stages tiles through VGPRs (global_load -> ds_write), which costs staging
// registers, a ds_write pass and VALU address work per tile; it tops out at 0.5-0.75 PFLOP/s.  Here
//  * both operand tiles go HBM -> LDS by LDS-DMA (16 B/lane, 1 KiB per wave instruction, lane-linear LDS
//    image, per-lane global address), double buffered: the DMA of tile t+1 flies under the MFMAs of tile t,
//    one barrier per K tile; out-of-range rows / K tail / conv padding read as ZERO through the buffer
//    descriptor's range check (forced out-of-range offset) -- no predicated code in the loop.
//  * bank conflicts are removed on the SOURCE side (the DMA image is linear): LDS slot (row, c) holds the
//    global 16-byte chunk c ^ f(row); fragment reads apply the same involution.
//      k-contiguous operand  [128 rows][64 k]  : f(row) = (row >> 1) & 7, read with ds_read_b128
//      row-contiguous operand [64 k][128|256 rows] : 32-byte pair index ^ ((k & 3) << 1) -- the 8 (k line, 16-row
//                                                half) pairs a 32-lane read group touches land in 8 distinct 32-byte bank slots,
//                                                read with ds_read_b64_tr_b16 (LDS transpose read), so
//                                                dgrad/wgrad need no transposed copies and no register shuffles.
//  * 128x128x64 tile, 4 wavefronts (2x2), 64x64 per wavefront as 2x2 v_mfma_f32_32x32x16 (64 acc VGPRs),
//    2 workgroups per CU (64 KiB LDS each).  MFMA operands are swapped so a lane owns 4 consecutive output
//    columns; the epilogue (bias, ReLU, tanh-GELU + pre-activation, ReLU-backward mask, fp32 accumulate,
//    split-K atomics) is the one of gemm.hip.
//  * XCD-aware tile walk (private L2 per XCD).
// A_MODE: 0 = k-contiguous matrix, 1 = row-contiguous matrix, 2 = implicit im2col of an NHWC tensor
// (convolution forward / data-gradient: A(m,k) = X[n, p*stride - pad + r, q*stride - pad + s, c]).
#include "gemm_tiles.h"

#define BM 128
#define BN 128
#define SLAB_MODE(p) ((p).ws != nullptr)
// operand stages (2 x 32 KiB); the fp32 epilogue staging tile 64 x (128+4) reuses them
#define GEMM2_LDS_BYTES (2 * (BM * BK + BN * BK) * 2)

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_RELU_BWD = 3, ACT_ADD = 4, ACT_GELU_BWD = 5, ACT_TANH = 6,
       ACT_TANH_BWD = 7,
       ACT_ADD_MASKED = 8,   // C = acc + (bit ? mask_src : 0); aux = bit-packed keep bits of the addend (INPUT, 1 bit / element)
       ACT_MUL = 9,          // C = acc * mask_src (e.g. the GELU derivative the forward GEMM left behind)
       ACT_GELU_DAUX = 10 }; // C = gelu(v), aux = gelu'(v) (instead of the pre-activation): the backward is a plain multiply

struct Gemm2Args {
  const unsigned short* A;
  const unsigned short* B;
  void* C;
  void* aux;
  const float* bias;
  const unsigned short* mask_src;
  int M, N, K;
  long long lda, ldb, ldc;
  int out_dtype, act, splitk, accumulate;
  float alpha;
  float* ws;             // split-K partial slabs [splitk][M][N] fp32 (NULL: fp32 atomics straight into C)
  ConvGeom cg;
  // batched mode (grid.z = batch): operand z = (zo, zi) = (z / batch_inner, z % batch_inner) starts at
  // base + zo * stride_outer + zi * stride_inner (elements) -- e.g. (sequence, head) slices of a [T, 3H] buffer
  int batch_inner, batch_count;
  int debug_skip;        // reserved (profiling ablations)
  float* stats;          // per-tile-row column sums of the ROUNDED output: [tiles_m][2][N] (sum, sum of squares); NULL: off
  int force_small;       // keep the 128x128 tile (stats layout is indexed by 128-row tiles)
  int stats_sums;        // stats holds column SUMS only, one partial row per tile row: [tiles_m][N] (bias gradients); 0: [tiles_m][2][N]
  long long row_extra;   // output row m lives at m * ldc + (m / row_div.d) * row_extra (+ column): rows of a strided sub-grid
  FastDiv row_div;
  int lds_src;           // 256x256 tile: 32 KiB of LDS beyond the two stages are available for epilogue source rows
  int gm;                // tile rows per walk group (tile_coords); DLE_GEMM_GM, default 8
  int persist;           // 128x128 tile only: the grid is smaller than the tile list, a workgroup walks tiles bid, bid + grid, ...
  long long sa_o, sa_i, sb_o, sb_i, sc_o, sc_i;
};

__device__ __forceinline__ float gelu_tanh2(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float u = k0 * (x + k1 * x * x * x);
  return 0.5f * x * (1.0f + fast_tanh(u));
}
// gelu(x) and d gelu / dx from one tanh
__device__ __forceinline__ float gelu_tanh2_d(float x, float& d) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float x2 = x * x;
  const float th = fast_tanh(k0 * (x + k1 * x2 * x));
  const float hp = 0.5f * (1.0f + th);
  d = hp + 0.5f * x * (1.f - th * th) * k0 * (1.f + 3.f * k1 * x2);
  return x * hp;
}

// Epilogue of 8 consecutive output columns of row m (v = alpha * accumulators): split-K partials, or bias /
// activation / mask / addend math and the 16-byte stores of C (+ the pre-activation side output).
template <int DT>
__device__ __forceinline__ void epi_store8(const Gemm2Args& p, float* v, int m, int n, int nval, bool vec16, int ky) {
  const long long off = (long long)m * p.ldc + n + (p.row_extra ? (long long)fd_div(m, p.row_div) * p.row_extra : 0LL);
  if (p.splitk > 1) {
    if (p.ws) {
      float* c = p.ws + ((long long)ky * p.M + m) * p.N + n;
      if (nval == 8 && (p.N & 3) == 0) {
        *(float4_t*)c = (float4_t){v[0], v[1], v[2], v[3]};
        *(float4_t*)(c + 4) = (float4_t){v[4], v[5], v[6], v[7]};
      } else {
        for (int r = 0; r < nval; ++r) c[r] = v[r];
      }
    } else {
      float* c = (float*)p.C + off;
      for (int r = 0; r < nval; ++r) unsafeAtomicAdd(c + r, v[r]);
    }
    return;
  }
  const bool full = nval == 8 && vec16;
  if (p.bias) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
      if (r < nval) v[r] += p.bias[n + r];
  }
  float pre[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) pre[r] = v[r];
  if (p.act == ACT_RELU) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
  } else if (p.act == ACT_GELU) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = gelu_tanh2(v[r]);
  } else if (p.act == ACT_TANH) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = fast_tanh(v[r]);
  } else if (p.act == ACT_GELU_DAUX) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = gelu_tanh2_d(v[r], pre[r]);
  } else if (p.act == ACT_RELU_BWD || p.act == ACT_ADD || p.act == ACT_GELU_BWD || p.act == ACT_TANH_BWD ||
             p.act == ACT_ADD_MASKED || p.act == ACT_MUL) {
    ushort8_t sv = {0, 0, 0, 0, 0, 0, 0, 0};
    if (full) sv = *(const ushort8_t*)(p.mask_src + off);
    else
      for (int r = 0; r < nval; ++r) sv[r] = p.mask_src[off + r];
    float yv[8];
    unpack8<DT>(sv, yv);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float y = yv[r];
      if (p.act == ACT_RELU_BWD) v[r] = y > 0.f ? v[r] : 0.f;
      else if (p.act == ACT_ADD) v[r] += y;
      else if (p.act == ACT_MUL) v[r] *= y;
      else if (p.act == ACT_ADD_MASKED) {
        const long long e = off + r;                                     // bit index = element index of the addend
        if (r < nval && ((((const unsigned char*)p.aux)[e >> 3] >> (e & 7)) & 1u)) v[r] += y;
      }
      else if (p.act == ACT_TANH_BWD) v[r] *= (1.f - y * y);           // y = tanh output of the forward
      else {                                                            // y = GELU pre-activation
        const float k0 = 0.7978845608028654f, k1 = 0.044715f;
        const float th = fast_tanh(k0 * (y + k1 * y * y * y));
        v[r] *= 0.5f * (1.f + th) + 0.5f * y * (1.f - th * th) * k0 * (1.f + 3.f * k1 * y * y);
      }
    }
  }
  if (p.out_dtype == DLE_F32) {
    float* c = (float*)p.C + off;
    if (p.accumulate)
      for (int r = 0; r < nval; ++r) v[r] += c[r];
    if (nval == 8 && (p.ldc & 3) == 0) {
      *(float4_t*)c = (float4_t){v[0], v[1], v[2], v[3]};
      *(float4_t*)(c + 4) = (float4_t){v[4], v[5], v[6], v[7]};
    } else {
      for (int r = 0; r < nval; ++r) c[r] = v[r];
    }
    if (p.aux && p.act != ACT_ADD_MASKED) {
      float* a = (float*)p.aux + off;
      for (int r = 0; r < nval; ++r) a[r] = pre[r];
    }
  } else {
    const ushort8_t o = p.out_dtype == DLE_F16 ? pack8<DLE_F16>(v) : pack8<DLE_BF16>(v);
    ushort8_t po = o;
    const bool side = p.aux && p.act != ACT_ADD_MASKED;
    if (side) po = p.out_dtype == DLE_F16 ? pack8<DLE_F16>(pre) : pack8<DLE_BF16>(pre);
    unsigned short* c = (unsigned short*)p.C + off;
    if (full) *(ushort8_t*)c = o;
    else
      for (int r = 0; r < nval; ++r) c[r] = o[r];
    if (side) {
      unsigned short* a = (unsigned short*)p.aux + off;
      if (full) *(ushort8_t*)a = po;
      else
        for (int r = 0; r < nval; ++r) a[r] = po[r];
    }
  }
}

// BIG = 0: 128x128 tile, 4 wavefronts (2x2), 64x64 per wavefront (64 accumulator VGPRs), 2 stages of 32 KiB, two
//          workgroups per CU.
// BIG = 1: 256x256 tile, 8 wavefronts (2x4), 128x64 per wavefront (128 accumulator VGPRs), 2 stages of 64 KiB, ONE
//          workgroup per CU = two wavefronts per SIMD.  Four times the flops per DMA byte and half the LDS fragment
//          bytes per flop of the 128x128 tile; the second wavefront of each SIMD keeps the matrix pipe busy while the
//          first one sits in an LDS-DMA issue, an LDS wait or the barrier.  The variant for the big compute-bound GEMMs
//          (BERT / DLRM linear layers).
// NSTAGE = 2: double-buffered operand stages.
// NSTAGE = 1 (BIG = 0 only): one stage (32 KiB) and a 128-VGPR budget -> 4 workgroups per CU.  For K <= 128 a tile is
// one or two K tiles: its lifetime is dominated by the DMA latency at the start and the store drain at the end, so the
// bytes in flight per CU (= resident workgroups) set the throughput, not the overlap inside one workgroup.
// PLAIN = 1: the epilogue is the fast path with no bias, activation, addend or side output (stores and BatchNorm partials
// only), decided by the launcher.  Such an instantiation issues NO load the compiler knows about inside the persistent tile
// loop (its LDS-DMA goes through inline asm), so hipcc's wait-count pass has nothing to drain: with the general epilogue in
// the same code it inserts s_waitcnt vmcnt(0) at the head of every tile (a bias / addend load of some path might still be
// in flight when a register is reused), which waits for the previous tile's STORES before the next DMA can be issued.
// BST = 1: the ReLU-mask / activation-derivative epilogues (ACT_RELU_BWD, ACT_MUL) also leave the column sums of their rounded
// output (p.stats, p.stats_sums) -- the bias gradient of the layer whose activation derivative they apply.  A separate instantiation: the 8 extra live registers spill in the 256 x 256
// kernel (8 -> 20 VGPRs), which the BERT / WaveGlow GEMMs must not pay for.
template <int DT, int A_MODE, int B_MODE, int NSTAGE, int BIG, int PLAIN = 0, int BST = 0>
__global__ __launch_bounds__(BIG ? 512 : 256, BIG ? 1 : (NSTAGE == 1 ? 4 : 2)) void gemm2_kernel(Gemm2Args p) {
  constexpr int WGN = BIG ? 4 : 2;                     // wave grid 2 x WGN
  constexpr int NW = 2 * WGN, NT = 64 * NW;
  constexpr int WTM = BIG ? 4 : 2, WTN = 2;            // 32x32 MFMA blocks per wave (rows, columns)
  constexpr int TM = 2 * WTM * 32, TN = WGN * WTN * 32;   // workgroup tile: 128x128 / 256x256
  constexpr int STAGE = (TM + TN) * BK;                // halves per stage (A tile | B tile)
  constexpr bool RAW_DMA = !BIG && A_MODE <= 1 && B_MODE <= 1;     // the persistent-walk kernels (counted waits, below)
  typedef Loader<A_MODE, TM, NW, RAW_DMA> LA;
  typedef Loader<B_MODE, TN, NW, RAW_DMA> LB;
  static_assert(LA::NP == 4 && LB::NP == 4, "4 DMA pieces per wave per operand per K tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* lds = (unsigned short*)smem_raw;   // [stages][A tile | B tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  if (p.batch_inner > 0) {
    const int zo = blockIdx.z / p.batch_inner, zi = blockIdx.z - zo * p.batch_inner;
    p.A += zo * p.sa_o + zi * p.sa_i;
    p.B += zo * p.sb_o + zi * p.sb_i;
    const long long co = zo * p.sc_o + zi * p.sc_i;
    p.C = p.out_dtype == DLE_F32 ? (void*)((float*)p.C + co) : (void*)((unsigned short*)p.C + co);
  }

  const int tiles_m = (p.M + TM - 1) / TM, tiles_n = (p.N + TN - 1) / TN;
  const int ntiles = tiles_m * tiles_n;
  // Each XCD (private L2) works through a contiguous chunk of the tile list; the list is ordered in groups of GM tile
  // rows walked column by column, so the ~32-64 tiles an XCD runs at once form a GM x (32..64/GM) block: per K step
  // they pull GM A tiles + a few B tiles through L2 instead of 1 + 32 (a 1 x 32 strip re-reads the whole B matrix
  // once per tile row: 4 GB of L2 fills for an 8192^3 GEMM).
  const int GM = p.gm;
  auto tile_coords = [&](int bid, int& tm_, int& tn_) __attribute__((always_inline)) {
    const int q = ntiles >> 3, r8 = ntiles & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + loc;
    const int per_group = GM * tiles_n;
    const int g = bid / per_group, r = bid - g * per_group;
    const int rows = (tiles_m - g * GM) < GM ? (tiles_m - g * GM) : GM;
    tm_ = g * GM + r % rows;
    tn_ = r / rows;
  };
  // Persistent walk (128x128 tile, no split-K): the workgroup's NEXT tile is prefetched (its first K tile, into the
  // stage the last K tile of the current one does not occupy) before the current tile's epilogue, so the DMA latency
  // at the head of a tile and the store drain at its tail overlap -- the K <= 256 layers of ResNet-50 (1-4 K tiles per
  // tile) ran at half the HBM rate with two one-shot workgroups per CU.
  const int vstep = (int)gridDim.x;
  int vbid = blockIdx.x;
  int tm, tn;
  tile_coords(vbid, tm, tn);
  int m0 = tm * TM, n0 = tn * TN;          // current tile (epilogue)
  int lm0 = m0, ln0 = n0;                  // the tile the loaders address (runs one tile ahead at tile boundaries)

  // balanced K slices: slice z owns k tiles [z*T/S, (z+1)*T/S) -- non-empty for every z when S <= T
  const int ktiles = (p.K + BK - 1) / BK;
  const int kt0 = (int)((long long)blockIdx.y * ktiles / p.splitk);
  const int kt1 = (int)((long long)(blockIdx.y + 1) * ktiles / p.splitk);
  if (kt0 >= kt1 && p.splitk > 1 && !SLAB_MODE(p)) return;
  const int kend = (kt1 * BK < p.K) ? kt1 * BK : p.K;

  LA la;
  LB lb;
  la.init(wave, lane, lm0, p.M, p.lda, p.cg);
  lb.init(wave, lane, ln0, p.N, p.ldb, p.cg);

  // DMA piece J of operand A / B of K tile kt into `stage` (one wave instruction each)
  auto issue_a = [&](int kt, int stage, auto J) __attribute__((always_inline)) {
    const int k0 = kt * BK;
    const unsigned short* ba = A_MODE == 0 ? p.A + (long long)lm0 * p.lda + k0
                             : A_MODE == 1 ? p.A + (long long)k0 * p.lda + lm0 : p.A;
    la.template issue<decltype(J)::value, decltype(J)::value + 1>(ba, lds + stage * STAGE, wave, kend - k0, k0, p.cg);
  };
  auto issue_b = [&](int kt, int stage, auto J) __attribute__((always_inline)) {
    const int k0 = kt * BK;
    const unsigned short* bb = B_MODE == 0 ? p.B + (long long)ln0 * p.ldb + k0
                             : B_MODE == 1 ? p.B + (long long)k0 * p.ldb + ln0 : p.B;
    lb.template issue<decltype(J)::value, decltype(J)::value + 1>(bb, lds + stage * STAGE + TM * BK, wave, kend - k0, k0, p.cg);
  };
  auto issue_all = [&](int kt, int stage) __attribute__((always_inline)) {
    static_for<0, 4>([&](auto J) __attribute__((always_inline)) { issue_a(kt, stage, J); });
    static_for<0, 4>([&](auto J) __attribute__((always_inline)) { issue_b(kt, stage, J); });
  };

  float16_t acc[WTM][WTN];
  if (kt0 < kt1) issue_all(kt0, 0);
  int s0 = 0;                              // stage of this tile's first K tile (128x128 tile)
  int stage_last = 0;
  // vmcnt retires IN ORDER and counts stores: waiting with vmcnt(0) for the next tile's first K tile (whose DMA was issued
  // BEFORE the epilogue) would also wait for every store of that epilogue to be acknowledged -- several microseconds per
  // tile under a write-heavy load, with nothing of this workgroup in flight meanwhile (K <= 256 layers ran at ~3 TB/s).
  // After an interior tile on the fast epilogue every wave has issued at least 2 * ITERS = 8 memory instructions behind
  // that DMA, so vmcnt(8) proves the DMA has landed and leaves the youngest 8 (stores) in flight.
  bool counted_wait = false;
  for (;;) {                               // tiles of this workgroup (one trip unless p.persist)
  constexpr bool CAN_PERSIST = !BIG && A_MODE <= 1 && B_MODE <= 1;      // (the convolution loaders are too register-heavy)
  const bool has_next = CAN_PERSIST && p.persist && vbid + vstep < ntiles;
  if (CAN_PERSIST && vbid != (int)blockIdx.x) {   // (re-derived here, not kept alive across the epilogue: registers)
    la.init(wave, lane, lm0, p.M, p.lda, p.cg);
    lb.init(wave, lane, ln0, p.N, p.ldb, p.cg);
  }
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;          // 32x32x16 fragment: row fr, k group fh (8 elements)
  constexpr bool RCA = LA::RC, RCB = LB::RC;
  constexpr int NM = WTM * WTN, NR = WTM + WTN;      // MFMAs / fragment reads per k-step
  if constexpr (!BIG) {
    // 128x128 tile: DMA of the next K tile at the top, then 4 k-steps; hipcc schedules the body (2-4 workgroups
    // per CU overlap each other's bubbles)
    // this wave's DMA pieces of the first K tile have landed (issued by the prologue, or under the previous tile)
    if (CAN_PERSIST && counted_wait) __builtin_amdgcn_s_waitcnt(0x0F78);   // vmcnt(8), see counted_wait
    else __builtin_amdgcn_s_waitcnt(0x0F70);                                // vmcnt(0)
    for (int kt = kt0; kt < kt1; ++kt) {
      const int stage = NSTAGE == 1 ? 0 : s0 ^ ((kt - kt0) & 1);
      stage_last = stage;
      if (NSTAGE == 1 && kt > kt0) {
        __syncthreads();                    // everyone is done reading the single stage
        issue_all(kt, 0);
      }
      if (kt > kt0) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's DMA pieces of tile kt have landed
      lds_barrier();                        // ... and everybody's; everyone is done reading the other stage (no global
                                            // fence: __syncthreads() would wait for the previous tile's stores again)
      if (NSTAGE == 2 && kt + 1 < kt1) issue_all(kt + 1, stage ^ 1);
      else if (CAN_PERSIST && NSTAGE == 2 && has_next) {   // last K tile: the loaders move on to the next tile of this workgroup
        int ntm, ntn;
        tile_coords(vbid + vstep, ntm, ntn);
        lm0 = ntm * TM; ln0 = ntn * TN;
        la.init(wave, lane, lm0, p.M, p.lda, p.cg);
        lb.init(wave, lane, ln0, p.N, p.ldb, p.cg);
        issue_all(kt0, stage ^ 1);
      }
      const unsigned short* ta = lds + stage * STAGE;
      const unsigned short* tb = ta + TM * BK;
      if constexpr (!(RCA || RCB)) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {      // 4 k-steps of 16
          ushort8_t fa[WTM], fb[WTN];
#pragma unroll
          for (int i = 0; i < WTM; ++i) fa[i] = frag_issue<false, TM>(ta, wm * (WTM * 32) + i * 32, ks, lane);
#pragma unroll
          for (int j = 0; j < WTN; ++j) fb[j] = frag_issue<false, TN>(tb, wn * (WTN * 32) + j * 32, ks, lane);
#pragma unroll
          for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j) acc[i][j] = Mfma32x16<DT>::run(fb[j], fa[i], acc[i][j]);
        }
      } else {
        // transpose-read operands are issued by inline asm (gemm_tiles.h): software pipeline by hand -- the fragments of
        // k-step ks + 1 are requested before the MFMAs of k-step ks, one explicit wait per k-step
        typename FragT<RCA>::type fa[2][WTM];
        typename FragT<RCB>::type fb[2][WTN];
#pragma unroll
        for (int i = 0; i < WTM; ++i) fa[0][i] = frag_issue<RCA, TM>(ta, wm * (WTM * 32) + i * 32, 0, lane);
#pragma unroll
        for (int j = 0; j < WTN; ++j) fb[0][j] = frag_issue<RCB, TN>(tb, wn * (WTN * 32) + j * 32, 0, lane);
        static_for<0, 4>([&](auto KS) __attribute__((always_inline)) {
          constexpr int ks = decltype(KS)::value, cur = ks & 1, nxt = cur ^ 1;
          frag_wait<true>();
          if constexpr (ks < 3) {
#pragma unroll
            for (int i = 0; i < WTM; ++i) fa[nxt][i] = frag_issue<RCA, TM>(ta, wm * (WTM * 32) + i * 32, ks + 1, lane);
#pragma unroll
            for (int j = 0; j < WTN; ++j) fb[nxt][j] = frag_issue<RCB, TN>(tb, wn * (WTN * 32) + j * 32, ks + 1, lane);
          }
          ushort8_t va[WTM], vb[WTN];
#pragma unroll
          for (int i = 0; i < WTM; ++i) va[i] = frag_value(fa[cur][i]);
#pragma unroll
          for (int j = 0; j < WTN; ++j) vb[j] = frag_value(fb[cur][j]);
#pragma unroll
          for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j) acc[i][j] = Mfma32x16<DT>::run(vb[j], va[i], acc[i][j]);
        });
      }
    }
  } else {
    // 256x256 tile, two waves per SIMD.  The K loop is laid out by hand, one MFMA per slot, and pinned with
    // sched_barrier (hipcc otherwise bunches the DMA issues and the LDS reads):
    //  * fragments are double-buffered in registers; the LDS reads of the next k-step sit behind the MFMAs of the
    //    FIRST half of the current one, so the last read has half a k-step of matrix time to come back;
    //  * ONE barrier per K tile, after k-step 2: by then every wave has finished reading the current stage (its
    //    k-step-3 fragments are in registers: lgkmcnt(0)) and its DMA pieces of the next tile have landed
    //    (vmcnt(0)).  k-step 3 then overlaps its MFMAs with the first fragment reads of the NEXT tile, so no LDS
    //    round trip is exposed at the tile boundary;
    //  * the 8 LDS-DMA pieces per wave of a tile are spread one per MFMA or two over k-step 3 of the previous tile
    //    and k-step 0 of the current one (a piece costs ~60-180 issue cycles during which this wave issues nothing
    //    else -- never in a burst), which leaves them two k-steps to land before the barrier.  Past the last K
    //    tile every piece is out of range and zero-fills the idle stage (no branch in the loop).
    typename FragT<RCA>::type fa[2][WTM];
    typename FragT<RCB>::type fb[2][WTN];
    auto kstep = [&](auto CUR, const unsigned short* ra, const unsigned short* rb, auto RKS, int dkt, int dstage,
                     auto HALF) {
      constexpr int cur = decltype(CUR)::value, nxt = cur ^ 1, rks = decltype(RKS)::value, half = decltype(HALF)::value;
      frag_wait<RCA || RCB>();             // asm-issued transpose reads of the previous k-step (no-op for k-contiguous operands)
      ushort8_t va[WTM], vb[WTN];
#pragma unroll
      for (int i = 0; i < WTM; ++i) va[i] = frag_value(fa[cur][i]);
#pragma unroll
      for (int j = 0; j < WTN; ++j) vb[j] = frag_value(fb[cur][j]);
      static_for<0, NM>([&](auto MI) {
        constexpr int m = decltype(MI)::value;
        constexpr int i = m / WTN, j = m % WTN;
        acc[i][j] = Mfma32x16<DT>::run(vb[j], va[i], acc[i][j]);
        static_for<0, NR>([&](auto U) {
          constexpr int u = decltype(U)::value;
          if constexpr ((u * (NM / 2)) / NR == m) {
            if constexpr (u < WTM) fa[nxt][u] = frag_issue<RCA, TM>(ra, wm * (WTM * 32) + u * 32, rks, lane);
            else fb[nxt][u - WTM] = frag_issue<RCB, TN>(rb, wn * (WTN * 32) + (u - WTM) * 32, rks, lane);
          }
        });
        if constexpr (half >= 0) {
          static_for<0, 4>([&](auto D) {
            constexpr int d = decltype(D)::value;
            if constexpr (((2 * d + 1) * NM) / 8 == m) {
              if constexpr ((d & 1) == 0) issue_a(dkt, dstage, std::integral_constant<int, 2 * half + (d >> 1)>());
              else issue_b(dkt, dstage, std::integral_constant<int, 2 * half + (d >> 1)>());
            }
          });
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    typedef std::integral_constant<int, 0> C0;
    typedef std::integral_constant<int, 1> C1;
    typedef std::integral_constant<int, 2> C2;
    typedef std::integral_constant<int, 3> C3;
    typedef std::integral_constant<int, -1> CN;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < WTM; ++i) fa[0][i] = frag_issue<RCA, TM>(lds, wm * (WTM * 32) + i * 32, 0, lane);
#pragma unroll
    for (int j = 0; j < WTN; ++j) fb[0][j] = frag_issue<RCB, TN>(lds + TM * BK, wn * (WTN * 32) + j * 32, 0, lane);
    static_for<0, 2>([&](auto J) { issue_a(kt0 + 1, 1, J); issue_b(kt0 + 1, 1, J); });
    __builtin_amdgcn_sched_barrier(0);
    for (int kt = kt0; kt < kt1; ++kt) {
      const int stage = (kt - kt0) & 1;
      const unsigned short* ta = lds + stage * STAGE;
      const unsigned short* tb = ta + TM * BK;
      const unsigned short* na = lds + (stage ^ 1) * STAGE;
      const unsigned short* nb = na + TM * BK;
      kstep(C0(), ta, tb, C1(), kt + 1, stage ^ 1, C1());     // k-step 0 | read k-step 1 | DMA 2nd half of tile kt+1
      kstep(C1(), ta, tb, C2(), 0, 0, CN());                  // k-step 1 | read k-step 2
      kstep(C0(), ta, tb, C3(), 0, 0, CN());                  // k-step 2 | read k-step 3
      __builtin_amdgcn_s_waitcnt(0x0070);                     // vmcnt(0) & lgkmcnt(0)
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      kstep(C1(), na, nb, C0(), kt + 2, stage, C0());         // k-step 3 | read k-step 0 of tile kt+1 | DMA 1st half of kt+2
    }
  }

  // ---- epilogue.  D = (B A^T) tile: lane owns C[m = fr-th row][n = 8*(r>>2) + 4*fh + (r&3)].
  // The fp32 tile is transposed through LDS so that the bias / activation / mask / addend math and the global
  // stores run with 8 consecutive columns per lane: 16-byte loads of mask/addend, 16-byte stores of C, 256-byte
  // contiguous row segments per 16 lanes.  Two passes of TM/2 rows (the wm = 0 waves, then the wm = 1 waves);
  // 16-byte slots XOR-swizzled by the row keep the staging tile at exactly the size of the operand stages
  // (32 KiB / 128 KiB) with conflict-free ds_write_b128 / ds_read_b128.
  float* epi = (float*)smem_raw + (BIG ? 0 : stage_last * (STAGE / 2));   // (the other stage may hold the prefetched next tile)
  const bool vec16 = (p.ldc & 7) == 0 && ((((uintptr_t)p.C) | ((uintptr_t)p.aux) | ((uintptr_t)p.mask_src)) & 15) == 0;
  // Fast path (interior tile, 16-bit output of the input type, 16-byte aligned rows, no split-K): everything that does
  // not depend on the row -- the lane's 8 columns, its bias values, the swizzled LDS slots, the activation kind -- is
  // hoisted out of the store loop, which is then ~30 VALU instructions per 16-byte store instead of ~100 VALU + 60
  // SALU (the generic loop below re-decides every runtime flag per store; a K <= 256 GEMM was instruction-bound in it).
  // (edge tiles included: N % 8 == 0 makes every 8-column group entirely inside or outside, rows are checked per trip)
  const bool fast = PLAIN || (p.splitk == 1 && p.out_dtype == DT && vec16 && (p.N & 7) == 0);
  const bool fast_slab = !PLAIN && p.splitk > 1 && p.ws != nullptr && (p.N & 7) == 0;
  constexpr int RPI = NT / (TN / 8), ITERS = (TM / 2) / RPI;      // rows per store-loop trip (16), trips per half
  const int f_ml0 = tid / (TN / 8), f_nl = (tid % (TN / 8)) << 3;
  float fbias[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) fbias[r] = 0.f;
  const bool f_col_ok = n0 + f_nl < p.N;
  if (!PLAIN && fast && p.bias && f_col_ok) {
    const float4_t b0 = *(const float4_t*)(p.bias + n0 + f_nl), b1 = *(const float4_t*)(p.bias + n0 + f_nl + 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) { fbias[r] = b0[r]; fbias[4 + r] = b1[r]; }
  }
  float st0[8], st1[8];                  // column sums / sums of squares over this thread's rows (p.stats only)
#pragma unroll
  for (int r = 0; r < 8; ++r) { st0[r] = 0.f; st1[r] = 0.f; }
  auto fast_pass = [&](auto ACTC, int half) __attribute__((always_inline)) {
    constexpr int act = decltype(ACTC)::value;
    constexpr bool needs_src = act == ACT_RELU_BWD || act == ACT_ADD || act == ACT_GELU_BWD || act == ACT_TANH_BWD ||
                               act == ACT_ADD_MASKED || act == ACT_MUL;
    const float* e0 = epi + f_ml0 * TN;
    const int c4 = f_nl >> 2;
    const int olo = (c4 ^ f_ml0) << 2, ohi = ((c4 + 1) ^ f_ml0) << 2;
    const int mrow0 = m0 + half * (TM / 2) + f_ml0;
    const long long off0 = (long long)mrow0 * p.ldc + n0 + f_nl;
    const long long step = (long long)RPI * p.ldc;
    const bool remap = !PLAIN && p.row_extra != 0;           // strided sub-grid output (stride-2 data gradient classes)
    unsigned short* c = (unsigned short*)p.C + off0;
    unsigned short* ax = (!PLAIN && p.aux && act != ACT_ADD_MASKED) ? (unsigned short*)p.aux + off0 : nullptr;
    const unsigned short* ms = needs_src ? p.mask_src + off0 : nullptr;
    const unsigned char* bits = act == ACT_ADD_MASKED ? (const unsigned char*)p.aux + (off0 >> 3) : nullptr;   // off0 % 8 == 0 (vec16)
    const int rows_left = p.M - (m0 + half * (TM / 2) + f_ml0);      // trips with it * RPI < rows_left are inside
    if (!f_col_ok) return;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      if (it * RPI >= rows_left) break;
      const float* e = e0 + it * RPI * TN;
      const int x = (it & 1) << 6;                       // row & 31 alternates between ml0 and ml0 + 16
      const float4_t lo = *(const float4_t*)(e + (olo ^ x)), hi = *(const float4_t*)(e + (ohi ^ x));
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = lo[r] + fbias[r]; v[4 + r] = hi[r] + fbias[4 + r]; }
      if constexpr (act == ACT_GELU_DAUX) {
        float dv[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = gelu_tanh2_d(v[r], dv[r]);
        if (ax) *(ushort8_t*)(ax + it * step) = pack8<DT>(dv);   // derivative side output
      } else if (ax) {
        *(ushort8_t*)(ax + it * step) = pack8<DT>(v);            // pre-activation side output
      }
      if constexpr (act == ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
      } else if constexpr (act == ACT_GELU) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = gelu_tanh2(v[r]);
      } else if constexpr (act == ACT_TANH) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = fast_tanh(v[r]);
      } else if constexpr (needs_src) {
        float yv[8];
        unpack8<DT>(*(const ushort8_t*)(ms + it * step), yv);
        unsigned mbits = 0;
        if constexpr (act == ACT_ADD_MASKED) mbits = bits[(it * step) >> 3];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float y = yv[r];
          if constexpr (act == ACT_RELU_BWD) v[r] = y > 0.f ? v[r] : 0.f;
          else if constexpr (act == ACT_ADD) v[r] += y;
          else if constexpr (act == ACT_MUL) v[r] *= y;
          else if constexpr (act == ACT_ADD_MASKED) { if ((mbits >> r) & 1u) v[r] += y; }
          else if constexpr (act == ACT_TANH_BWD) v[r] *= (1.f - y * y);
          else {
            const float k0 = 0.7978845608028654f, k1 = 0.044715f;
            const float th = fast_tanh(k0 * (y + k1 * y * y * y));
            v[r] *= 0.5f * (1.f + th) + 0.5f * y * (1.f - th * th) * k0 * (1.f + 3.f * k1 * y * y);
          }
        }
      }
      const ushort8_t ov = pack8<DT>(v);
      if (remap) *(ushort8_t*)(c + it * step + (long long)fd_div(mrow0 + it * RPI, p.row_div) * p.row_extra) = ov;
      else *(ushort8_t*)(c + it * step) = ov;
      if ((BST ? (act == ACT_RELU_BWD || act == ACT_MUL || (!BIG && act == ACT_NONE)) : (!BIG && act == ACT_NONE)) && p.stats) {
        // column sums of the ROUNDED output: BatchNorm statistics of what the next pass will read (sums + squares), or the
        // bias gradient of the layer whose ReLU mask this epilogue applies (sums only)
        float vr[8];
        unpack8<DT>(ov, vr);
#pragma unroll
        for (int r = 0; r < 8; ++r) { st0[r] += vr[r]; if (!BIG) st1[r] += vr[r] * vr[r]; }
      }
    }
  };
  // ---- 256x256 tile, epilogues that read a 16-bit source tensor (addend, activation derivative, ReLU source), interior tile.
  // Loaded per trip from global memory, each source row sat behind the previous trip's store in the in-order vmcnt queue and
  // one 16-byte load per lane was in flight: 32768x4096x1024 ran 443 us against 276 us without a source.  A register window
  // spills in this kernel (128 accumulators live until staged).  Here the source arrives through the 32 KiB of LDS the two
  // operand stages leave free: 32-row chunks (16 KiB, two LDS-DMA pieces per wavefront) double-buffered, chunk g + 1 requested
  // BEFORE the stores of chunk g and waited for with vmcnt(2) = the two store instructions issued behind it; the trips read
  // their source rows with ds_read_b128.  No global load the compiler knows about remains in the epilogue.
  bool epilogue_done = false;
  if constexpr (BIG && !PLAIN) {
    const bool src_act = p.act == ACT_RELU_BWD || p.act == ACT_ADD || p.act == ACT_GELU_BWD || p.act == ACT_TANH_BWD ||
                         p.act == ACT_MUL;
    if (p.lds_src && fast && src_act && p.mask_src != nullptr && p.row_extra == 0 && m0 + TM <= p.M && n0 + TN <= p.N) {
      __builtin_amdgcn_s_waitcnt(0x0F70);        // the zero-fill DMA issued under the last K tile has landed
      unsigned short* srcbuf = (unsigned short*)smem_raw + 65536;       // halves: behind the 128 KiB of the two stages
      const int4v_t rs_src = rsrc_words(p.mask_src);
      auto issue_src = [&](int g) __attribute__((always_inline)) {      // chunk g = rows m0 + 32 g .. + 31
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int piece = wave * 2 + j;                                // 16 pieces of 2 rows x 512 B
          const int row = m0 + g * 32 + piece * 2 + (lane >> 5);
          const unsigned voff = (unsigned)(((long long)row * p.ldc + n0 + (lane & 31) * 8) * 2);
          dma16_raw(rs_src, srcbuf + (g & 1) * 8192 + piece * 512, voff);
        }
      };
      auto src_pass = [&](auto ACTC) __attribute__((always_inline)) {
        constexpr int act = decltype(ACTC)::value;
        const float* e0 = epi + f_ml0 * TN;
        const int c4 = f_nl >> 2;
        const int olo = (c4 ^ f_ml0) << 2, ohi = ((c4 + 1) ^ f_ml0) << 2;
        const float alpha_s = p.alpha;
        issue_src(0);
        // (alpha is applied when a trip reads the staged tile: `acc * alpha` in the staging loop is hoisted by hipcc out of the
        //  half loop as a second 128-register copy of the accumulators and spilled)
        for (int half = 0; half < 2; ++half) {
          lds_barrier();                         // operand stages (half 0) / previous half's tile are no longer read
          if (wm == half) {
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
              for (int j = 0; j < WTN; ++j)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                  float4_t v = {acc[i][j][qd * 4 + 0], acc[i][j][qd * 4 + 1], acc[i][j][qd * 4 + 2], acc[i][j][qd * 4 + 3]};
                  const int row = i * 32 + fr, cq = (wn * (WTN * 32) + j * 32 + qd * 8 + fh * 4) >> 2;
                  *(float4_t*)(epi + row * TN + ((cq ^ (row & 31)) << 2)) = v;
                }
          }
          const long long off0 = (long long)(m0 + half * (TM / 2) + f_ml0) * p.ldc + n0 + f_nl;
          const long long step = (long long)RPI * p.ldc;
          unsigned short* c = (unsigned short*)p.C + off0;
          unsigned short* ax = p.aux ? (unsigned short*)p.aux + off0 : nullptr;
#pragma unroll
          for (int ch = 0; ch < ITERS / 2; ++ch) {
            const int g = half * (ITERS / 2) + ch;
            if (g == 0) __builtin_amdgcn_s_waitcnt(0x0F70);   // chunk 0: nothing of this workgroup is behind it
            else __builtin_amdgcn_s_waitcnt(0x0F72);          // vmcnt(2): the >= 2 stores of chunk g - 1 stay in flight
            lds_barrier();                        // everybody's pieces of chunk g landed; chunk g - 1 (and, at ch == 0, the
                                                  // staging writes of this half) are visible / no longer read
            if (g + 1 < ITERS) issue_src(g + 1);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const int it = ch * 2 + t;
              const float* e = e0 + it * RPI * TN;
              const int x = (it & 1) << 6;
              const float4_t lo = *(const float4_t*)(e + (olo ^ x)), hi = *(const float4_t*)(e + (ohi ^ x));
              float v[8], yv[8];
#pragma unroll
              for (int r = 0; r < 4; ++r) { v[r] = lo[r] * alpha_s + fbias[r]; v[4 + r] = hi[r] * alpha_s + fbias[4 + r]; }
              if (ax) *(ushort8_t*)(ax + it * step) = pack8<DT>(v);            // pre-activation side output
              unpack8<DT>(*(const ushort8_t*)(srcbuf + (g & 1) * 8192 + (f_ml0 + RPI * t) * TN + f_nl), yv);
#pragma unroll
              for (int r = 0; r < 8; ++r) {
                const float y = yv[r];
                if constexpr (act == ACT_RELU_BWD) v[r] = y > 0.f ? v[r] : 0.f;
                else if constexpr (act == ACT_ADD) v[r] += y;
                else if constexpr (act == ACT_MUL) v[r] *= y;
                else if constexpr (act == ACT_TANH_BWD) v[r] *= (1.f - y * y);
                else {
                  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
                  const float th = fast_tanh(k0 * (y + k1 * y * y * y));
                  v[r] *= 0.5f * (1.f + th) + 0.5f * y * (1.f - th * th) * k0 * (1.f + 3.f * k1 * y * y);
                }
              }
              const ushort8_t ov = pack8<DT>(v);
              *(ushort8_t*)(c + it * step) = ov;
              if constexpr (BST && (act == ACT_RELU_BWD || act == ACT_MUL)) {
                if (p.stats) {                    // bias gradient of the layer below: column sums of the rounded, masked output
                  float vr[8];
                  unpack8<DT>(ov, vr);
#pragma unroll
                  for (int r = 0; r < 8; ++r) st0[r] += vr[r];
                }
              }
            }
          }
        }
      };
      switch (p.act) {
        case ACT_RELU_BWD: src_pass(std::integral_constant<int, ACT_RELU_BWD>()); break;
        case ACT_ADD: src_pass(std::integral_constant<int, ACT_ADD>()); break;
        case ACT_MUL: src_pass(std::integral_constant<int, ACT_MUL>()); break;
        case ACT_TANH_BWD: src_pass(std::integral_constant<int, ACT_TANH_BWD>()); break;
        default: src_pass(std::integral_constant<int, ACT_GELU_BWD>()); break;
      }
      epilogue_done = true;
    }
  }
  if (!epilogue_done) {
  if (BIG) __builtin_amdgcn_s_waitcnt(0x0F70);    // the zero-fill DMA issued under the last K tile has landed
  for (int half = 0; half < 2; ++half) {
  lds_barrier();                         // operand stages (half 0) / previous half's tile are no longer read
  if (wm == half) {
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int j = 0; j < WTN; ++j)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          float4_t v = {acc[i][j][qd * 4 + 0], acc[i][j][qd * 4 + 1], acc[i][j][qd * 4 + 2], acc[i][j][qd * 4 + 3]};
          const int row = i * 32 + fr, c4 = (wn * (WTN * 32) + j * 32 + qd * 8 + fh * 4) >> 2;
          *(float4_t*)(epi + row * TN + ((c4 ^ (row & 31)) << 2)) = v * p.alpha;
        }
  }
  lds_barrier();
  if (fast_slab) {                                   // split-K partial tile -> fp32 slab, two 16-byte stores per trip
    const float* e0 = epi + f_ml0 * TN;
    const int c4 = f_nl >> 2;
    const int olo = (c4 ^ f_ml0) << 2, ohi = ((c4 + 1) ^ f_ml0) << 2;
    float* c = p.ws + ((long long)blockIdx.y * p.M + m0 + half * (TM / 2) + f_ml0) * p.N + n0 + f_nl;
    const long long step = (long long)RPI * p.N;
    const int rows_left = p.M - (m0 + half * (TM / 2) + f_ml0);
    if (n0 + f_nl < p.N) {
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        if (it * RPI >= rows_left) break;
        const float* e = e0 + it * RPI * TN;
        const int x = (it & 1) << 6;
        const float4_t lo = *(const float4_t*)(e + (olo ^ x)), hi = *(const float4_t*)(e + (ohi ^ x));
        *(float4_t*)(c + it * step) = lo;
        *(float4_t*)(c + it * step + 4) = hi;
      }
    }
    continue;
  }
  if constexpr (PLAIN) {
    fast_pass(std::integral_constant<int, ACT_NONE>(), half);
    continue;
  }
  if (fast) {
    switch (p.act) {
      case ACT_NONE: fast_pass(std::integral_constant<int, ACT_NONE>(), half); break;
      case ACT_RELU: fast_pass(std::integral_constant<int, ACT_RELU>(), half); break;
      case ACT_GELU: fast_pass(std::integral_constant<int, ACT_GELU>(), half); break;
      case ACT_RELU_BWD: fast_pass(std::integral_constant<int, ACT_RELU_BWD>(), half); break;
      case ACT_ADD: fast_pass(std::integral_constant<int, ACT_ADD>(), half); break;
      case ACT_GELU_BWD: fast_pass(std::integral_constant<int, ACT_GELU_BWD>(), half); break;
      case ACT_TANH: fast_pass(std::integral_constant<int, ACT_TANH>(), half); break;
      case ACT_ADD_MASKED: fast_pass(std::integral_constant<int, ACT_ADD_MASKED>(), half); break;
      case ACT_MUL: fast_pass(std::integral_constant<int, ACT_MUL>(), half); break;
      case ACT_GELU_DAUX: fast_pass(std::integral_constant<int, ACT_GELU_DAUX>(), half); break;
      default: fast_pass(std::integral_constant<int, ACT_TANH_BWD>(), half); break;
    }
    continue;
  }
#pragma unroll 2
  for (int it = 0; it < (TM / 2) * (TN / 8) / NT; ++it) {
    const int idx = it * NT + tid;
    const int ml = idx / (TN / 8), nl = (idx % (TN / 8)) << 3;
    const int m = m0 + half * (TM / 2) + ml, n = n0 + nl;
    if (m >= p.M || n >= p.N) continue;
    const int nval = (p.N - n) < 8 ? (p.N - n) : 8;
    float v[8];
    {
      const int c4 = nl >> 2;
      const float4_t lo = *(const float4_t*)(epi + ml * TN + ((c4 ^ (ml & 31)) << 2));
      const float4_t hi = *(const float4_t*)(epi + ml * TN + (((c4 + 1) ^ (ml & 31)) << 2));
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = lo[r]; v[4 + r] = hi[r]; }
    }
    epi_store8<DT>(p, v, m, n, nval, vec16, blockIdx.y);
  }
  }
  }   // !epilogue_done
  if ((BST || !BIG) && p.stats && fast) {
    // the 16 threads that share a column group meet in LDS; one plain store per (tile row, statistic, column):
    // deterministic, no atomics -- dle_bn_stats_from_partials / the column-sum fold add the tile rows in a fixed order
    const int nst = (BST && p.stats_sums) ? 1 : 2;
    lds_barrier();
    float* red = epi;                                    // [nst][RPI][TN]
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      red[(0 * RPI + f_ml0) * TN + f_nl + r] = st0[r];
      if (!BIG) { if (nst == 2) red[(1 * RPI + f_ml0) * TN + f_nl + r] = st1[r]; }
    }
    lds_barrier();
    for (int cidx = tid; cidx < nst * TN; cidx += NT) {
      const int which = cidx / TN, col = cidx - which * TN;
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < RPI; ++q) t += red[(which * RPI + q) * TN + col];
      if (n0 + col < p.N) p.stats[((long long)tm * nst + which) * p.N + n0 + col] = t;
    }
  }
  if (!CAN_PERSIST || !has_next) break;
  counted_wait = fast && m0 + TM <= p.M && n0 + TN <= p.N && 2 * ITERS >= 8;
  vbid += vstep;
  tile_coords(vbid, tm, tn);
  m0 = tm * TM; n0 = tn * TN;
  s0 = stage_last ^ 1;
  }
}

// One launcher for every entry point.  modes: (A_MODE, B_MODE) in {(0,0),(0,1),(1,1),(2,0),(4,5),(1,3)}.
//  * plain matrix operands, M, N >= 256, >= 4 K tiles per slice and enough 256x256 tiles to cover most of the chip:
//    the 256x256 tile (DLE_GEMM_BIG=0 disables it, =1 forces it whenever the shape allows);
//  * everything else (convolutions, batched attention slices, small / skinny shapes): the 128x128 tile.
// (A persistent variant of the 256x256 kernel -- one workgroup per CU walking the tile list, the K pipeline running
//  across tile boundaries, the epilogue through the 32 KiB of LDS the stages leave free -- measured 8% SLOWER at
//  16384x4096x1024: its output stores still meet a vmcnt(0) three k-steps later and the 8-pass epilogue costs more
//  than the tile turnover it saves; it was removed.)
static int launch_gemm(const Gemm2Args& p_in, int in_dtype, int amode, int bmode, int batch, hipStream_t stream,
                       int* tile_rows_out = nullptr) {
  Gemm2Args p = p_in;
  static const int gm_env = getenv("DLE_GEMM_GM") ? atoi(getenv("DLE_GEMM_GM")) : 8;
  p.gm = gm_env > 0 ? gm_env : 8;
  p.debug_skip = 0;
  p.batch_count = batch;
  const int ktiles = (p.K + BK - 1) / BK;
  const int kt_per_item = (ktiles + p.splitk - 1) / p.splitk;
  if (amode == 1 && bmode == 0) { dle_set_error("gemm: unsupported operand layout"); return 1; }
  static const int big_mode = getenv("DLE_GEMM_BIG") ? atoi(getenv("DLE_GEMM_BIG")) : -1;
  const long long tiles_big = (long long)((p.M + 255) / 256) * ((p.N + 255) / 256);
  const bool fits = amode <= 1 && bmode <= 1 && p.M >= 256 && p.N >= 256 &&
                    p.lda * 512 <= 0x7FFFFFFFLL && p.ldb * 512 <= 0x7FFFFFFFLL;
  // Split-K weight gradients (fp32 slabs) take the big tile too once their slices cover half the chip: the 128x128 tile
  // issues 8 LDS-DMA pieces per wave per 16 MFMAs and is bound by that issue cost (~590 TFLOP/s); the 256x256 tile
  // halves it.  (It measured SLOWER before the transpose reads went to inline asm: every one of them drained the DMA queue.)
  const long long work_big = tiles_big * (batch > 0 ? batch : 1) * (SLAB_MODE(p) ? p.splitk : 1);
  const bool want = !p.force_small && (big_mode >= 1 || (big_mode != 0 && (p.splitk == 1 || SLAB_MODE(p)) && kt_per_item >= 4 &&
                                                         work_big >= (p.splitk == 1 ? 160 : 128)));
  if (tile_rows_out) *tile_rows_out = (fits && want) ? 256 : BM;
  if (fits && want) {
    dim3 grid((unsigned)tiles_big, p.splitk, batch > 0 ? batch : 1), block(512);
    // 128 KiB of operand stages + (when the device grants a workgroup the whole 160 KiB) 32 KiB for epilogue source rows
    static const int lds_max = [] { int dev = 0, v = 0; hipGetDevice(&dev);
                                    hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev); return v; }();
    static const int lds_src_mode = getenv("DLE_GEMM_LDS_SRC") ? atoi(getenv("DLE_GEMM_LDS_SRC")) : 1;
    const size_t lds_stages = 2 * (256 * BK + 256 * BK) * 2;
    p.lds_src = lds_src_mode && lds_max >= (int)(lds_stages + 32768) && (long long)p.M * p.ldc * 2 < 0xFFFFFFE0LL;
    const size_t lds_big = lds_stages + (lds_max >= (int)(lds_stages + 32768) ? 32768 : 0);
#define GOBIG(DT, AM, BMODE) do { static bool attr_set = false; \
      if (!attr_set) { hipFuncSetAttribute((const void*)gemm2_kernel<DT, AM, BMODE, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big); attr_set = true; } \
      hipLaunchKernelGGL((gemm2_kernel<DT, AM, BMODE, 2, 1>), grid, block, lds_big, stream, p); } while (0)
#define PICKBIG(DT) do { if (amode == 0 && bmode == 0) GOBIG(DT, 0, 0); else if (amode == 0) GOBIG(DT, 0, 1); else GOBIG(DT, 1, 1); } while (0)
#define GOBIGST(DT) do { static bool attr_set = false; \
      if (!attr_set) { hipFuncSetAttribute((const void*)gemm2_kernel<DT, 0, 1, 2, 1, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big); attr_set = true; } \
      hipLaunchKernelGGL((gemm2_kernel<DT, 0, 1, 2, 1, 0, 1>), grid, block, lds_big, stream, p); } while (0)
    if (p.stats) {                                   // (launch_gemm's caller checked: sums only, ReLU-mask epilogue, operands (0, 1))
      if (in_dtype == DLE_F16) GOBIGST(DLE_F16); else GOBIGST(DLE_BF16);
    } else if (in_dtype == DLE_F16) PICKBIG(DLE_F16); else PICKBIG(DLE_BF16);
#undef GOBIGST
#undef GOBIG
#undef PICKBIG
  } else {
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const size_t lds = GEMM2_LDS_BYTES;
    // persistent walk: two resident workgroups per CU, each prefetching its next tile under the current epilogue
    static const int persist_mode = getenv("DLE_GEMM_PERSIST") ? atoi(getenv("DLE_GEMM_PERSIST")) : 1;
    const int resident = 2 * 256;
    p.persist = persist_mode && amode <= 1 && bmode <= 1 && p.splitk == 1 && batch <= 0 && tiles > resident;
    dim3 grid(p.persist ? resident : tiles, p.splitk, batch > 0 ? batch : 1), block(256);
    // (a single-stage, 4-workgroups-per-CU variant for K <= 128 existed; with the hoisted epilogue it measured 2x
    //  SLOWER than this one -- 802816x256x64: 295 vs 144 us -- its 128-VGPR budget spilled; removed)
    // store-only epilogue (see PLAIN at the kernel): no bias / activation / addend / side output, 16-bit output of the input
    // type on the 16-byte fast path
    const bool plain = amode <= 1 && bmode <= 1 && p.splitk == 1 && batch <= 0 && p.act == ACT_NONE && !p.bias && !p.aux &&
                       !p.mask_src && p.out_dtype == in_dtype && p.row_extra == 0 && (p.N & 7) == 0 && (p.ldc & 7) == 0 &&
                       (((uintptr_t)p.C) & 15) == 0;
#define GO(DT, AM, BMODE) hipLaunchKernelGGL((gemm2_kernel<DT, AM, BMODE, 2, 0>), grid, block, lds, stream, p)
#define GOP(DT, AM, BMODE) hipLaunchKernelGGL((gemm2_kernel<DT, AM, BMODE, 2, 0, 1>), grid, block, lds, stream, p)
#define GOST(DT) hipLaunchKernelGGL((gemm2_kernel<DT, 0, 1, 2, 0, 0, 1>), grid, block, lds, stream, p)
#define PICK(DT) do { if (p.stats && p.stats_sums) GOST(DT); else if (plain) { if (amode == 0 && bmode == 0) GOP(DT, 0, 0); else if (amode == 0) GOP(DT, 0, 1); else GOP(DT, 1, 1); } \
      else if (amode == 0 && bmode == 0) GO(DT, 0, 0); else if (amode == 0) GO(DT, 0, 1); else if (amode == 1 && bmode == 1) GO(DT, 1, 1); \
      else if (amode == 2) GO(DT, 2, 0); else if (amode == 4) GO(DT, 4, 5); else GO(DT, 1, 3); } while (0)
    if (in_dtype == DLE_F16) PICK(DLE_F16); else PICK(DLE_BF16);
#undef GO
#undef GOP
#undef GOST
#undef PICK
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { dle_set_error("gemm launch failed: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

// C[m][n] (+)= sum_z ws[z][m][n].  A weight gradient is small (M x N <= a few 10^5) and deep (up to 512 slabs), so
// the pass is spread over BOTH axes: 16 lanes x 16 bytes cover 64 consecutive elements, the 16 lane rows of a workgroup
// take every 16th slab (4 loads in flight each) and meet in LDS.  (The first version gave one thread all slabs of its
// element: 16 workgroups doing 512 dependent loads each, 50 us for 33 MB.)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C,
                                                            int M, int N, long long ldc, int splitk, int accumulate) {
  const long long slab = (long long)M * N;
  if ((N & 3) == 0 && (ldc & 3) == 0) {
    __shared__ float4_t red[256];
    const long long total4 = slab >> 2;
    const int n4 = N >> 2;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    for (long long i0 = (long long)blockIdx.x * 16; i0 < total4; i0 += (long long)gridDim.x * 16) {
      const long long i = i0 + tx;
      float4_t s = {0.f, 0.f, 0.f, 0.f};
      if (i < total4) {
        int z = ty;
        for (; z + 48 < splitk; z += 64) {
          const float4_t a = ((const float4_t*)(ws + (long long)z * slab))[i];
          const float4_t b = ((const float4_t*)(ws + (long long)(z + 16) * slab))[i];
          const float4_t c = ((const float4_t*)(ws + (long long)(z + 32) * slab))[i];
          const float4_t d = ((const float4_t*)(ws + (long long)(z + 48) * slab))[i];
          s += (a + b) + (c + d);
        }
        for (; z < splitk; z += 16) s += ((const float4_t*)(ws + (long long)z * slab))[i];
      }
      red[threadIdx.x] = s;
      __syncthreads();
      if (ty == 0 && i < total4) {
#pragma unroll
        for (int q = 1; q < 16; ++q) s += red[q * 16 + tx];
        const long long m = i / n4;
        float4_t* c = (float4_t*)(C + m * ldc) + (i - m * n4);
        if (accumulate) s += *c;
        *c = s;
      }
      __syncthreads();
    }
  } else {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < slab; i += (long long)gridDim.x * blockDim.x) {
      float s = ws[i];
      for (int z = 1; z < splitk; ++z) s += ws[z * slab + i];
      const long long m = i / N;
      float* c = C + m * ldc + (i - m * N);
      if (accumulate) s += *c;
      *c = s;
    }
  }
}

// Shallow form (S <= 8 slabs: the one-slice-per-CU weight gradients of BERT / DLRM): one 16-byte element per thread, its S
// slab loads issued back to back (S is a template parameter: unconditional loads), no LDS.  In the deep form above only S of
// the 16 lane rows of a workgroup have a slab to read: 4 slabs -> 25 % of the lanes, 36 us for 80 MB (3.5 ms of the BERT step).
template <int S>
__global__ __launch_bounds__(256) void splitk_reduce_shallow_kernel(const float* __restrict__ ws, float* __restrict__ C,
                                                                    int M, int N, long long ldc, int accumulate) {
  const long long slab4 = ((long long)M * N) >> 2;
  const int n4 = N >> 2;
  const bool dense = ldc == N;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < slab4; i += (long long)gridDim.x * blockDim.x) {
    float4_t v[S];
#pragma unroll
    for (int z = 0; z < S; ++z) v[z] = ((const float4_t*)ws)[(long long)z * slab4 + i];
    float4_t* c;
    if (dense) c = (float4_t*)C + i;
    else { const long long m = i / n4; c = (float4_t*)(C + m * ldc) + (i - m * n4); }
    float4_t s = v[0];
#pragma unroll
    for (int z = 1; z < S; ++z) s += v[z];
    if (accumulate) s += *c;
    *c = s;
  }
}

static bool launch_splitk_reduce_shallow(const float* ws, float* C, int M, int N, long long ldc, int splitk, int accumulate,
                                         hipStream_t stream) {
  if (splitk < 2 || splitk > 8 || (N & 3) != 0 || (ldc & 3) != 0) return false;
  const long long slab4 = ((long long)M * N) >> 2;
  long long g = (slab4 + 255) / 256;
  if (g > 4096) g = 4096;
#define GO(S) hipLaunchKernelGGL(splitk_reduce_shallow_kernel<S>, dim3((unsigned)g), dim3(256), 0, stream, ws, C, M, N, ldc, accumulate)
  switch (splitk) {
    case 2: GO(2); break; case 3: GO(3); break; case 4: GO(4); break; case 5: GO(5); break;
    case 6: GO(6); break; case 7: GO(7); break; default: GO(8); break;
  }
#undef GO
  return true;
}

extern "C" int dle_gemm8_try(const void* A, const void* B, void* C, void* aux, const float* bias, const void* src, int M, int N,
                             int K, int64_t lda, int64_t ldb, int64_t ldc, int a_kc, int b_kc, int in_dtype, int out_dtype,
                             int act, int splitk, int accumulate, float alpha, float* ws, float* stats, hipStream_t stream);   // gemm8.hip

// Returns 1 when the DMA kernel took the launch, 0 when the shape/alignment is outside its envelope
// (caller falls back to gemm.hip), <0 / >1 on error.  Called from dle_gemm.
extern "C" int dle_gemm_dma_try(const void* A, const void* B, void* C, void* aux, const float* bias,
                                const void* mask_src, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                                int a_kc, int b_kc, int in_dtype, int out_dtype, int act, int splitk,
                                int accumulate, float alpha, void* workspace, int64_t workspace_bytes,
                                hipStream_t stream) {
  const bool al = ((((uintptr_t)A) | ((uintptr_t)B)) & 15) == 0 && (lda & 7) == 0 && (ldb & 7) == 0;
  if (!al || (K & 7) != 0) return 0;
  if (!a_kc && (M & 7) != 0) return 0;         // row-contiguous operands are fetched in 8-row chunks
  if (!b_kc && (N & 7) != 0) return 0;
  if (M < 1 || N < 8 || K < 8) return 0;
  // 32-bit byte offsets inside one operand tile panel
  if ((long long)lda * (a_kc ? BM : BK) * 2 > 0x7FFFFFFFLL || (long long)ldb * (b_kc ? BN : BK) * 2 > 0x7FFFFFFFLL) return 0;
  Gemm2Args p = {};
  p.A = (const unsigned short*)A; p.B = (const unsigned short*)B; p.C = C; p.aux = aux; p.bias = bias;
  p.mask_src = (const unsigned short*)mask_src;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.out_dtype = out_dtype; p.act = act; p.splitk = splitk; p.accumulate = accumulate; p.alpha = alpha;
  p.cg = make_geom(1, 1, 1, 1, 1, 1, 1, 1, 0, 1);
  p.ws = nullptr;
  p.batch_inner = 0;
  if (splitk > 1) {
    const long long need = (long long)splitk * M * N * 4;
    // (an empty K slice still writes its all-zero slab: the kernel does not early-out in slab mode)
    if (workspace && workspace_bytes >= need && (((uintptr_t)workspace) & 15) == 0) p.ws = (float*)workspace;
    else if (!accumulate) {
      hipError_t e = ldc == N ? hipMemsetAsync(C, 0, (size_t)M * N * 4, stream)
                              : hipMemset2DAsync(C, (size_t)ldc * 4, 0, (size_t)N * 4, (size_t)M, stream);
      if (e != hipSuccess) { dle_set_error("gemm memset: %s", hipGetErrorString(e)); return (int)e + 1000; }
    }
  }
  const int amode = a_kc ? 0 : 1, bmode = b_kc ? 0 : 1;
  if (!a_kc && b_kc) return 0;
  {
    // the big linear layers: the persistent ping-pong kernel of gemm8.hip (split-K only with slabs)
    int r8 = 0;
    if (splitk == 1 || p.ws)
      r8 = dle_gemm8_try(A, B, C, aux, bias, mask_src, M, N, K, lda, ldb, ldc, a_kc, b_kc, in_dtype, out_dtype, act, splitk,
                         accumulate, alpha, p.ws, nullptr, stream);
    if (r8 > 1) return r8;
    if (r8 == 0) { const int rc = launch_gemm(p, in_dtype, amode, bmode, 0, stream); if (rc) return rc + 1000; }
  }
  if (p.ws) {
    long long items = ((long long)M * N + 3) / 4;
    long long g = ((N & 3) == 0 && (ldc & 3) == 0) ? (items + 15) / 16 : (items + 63) / 64;
    if (g > 4096) g = 4096;
    if (!launch_splitk_reduce_shallow((const float*)p.ws, (float*)C, M, N, (long long)ldc, splitk, accumulate, stream))
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)g), dim3(256), 0, stream, (const float*)p.ws, (float*)C, M, N,
                         (long long)ldc, splitk, accumulate);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { dle_set_error("splitk_reduce launch failed: %s", hipGetErrorString(e)); return (int)e + 1000; }
  }
  return 1;
}

// out[n] = sum over the tile rows g of partial[g][n] (fixed order): 16 columns x 16 row slices per workgroup
__global__ __launch_bounds__(256) void colsum_fold_kernel(const float* __restrict__ partial, float* __restrict__ out, int N,
                                                          int groups, int accumulate) {
  __shared__ float red[256];
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

// C[M, N] = f(A[M, K] B[K, N], src[M, N]) AND colsum_out[n] (+)= sum_m C[m, n] of the ROUNDED output: the data gradient of a linear
// layer through the activation derivative of the layer below, together with that layer's bias gradient.  act = DLE_ACT_RELU_BWD
// (C = product where src > 0: Recommendation/DLRM/dlrm/nn/mlps.py:38-43 backward) or DLE_ACT_MUL (C = product * src, src = the
// stored GELU derivative: LanguageModeling/BERT/modeling.py:130-160 backward of bias_gelu).  The separate column-sum pass re-read
// every such gradient the step had just written (114 us per DLRM step at batch 65536, 1.06 ms per BERT-Large step).  One
// partial row per tile row, folded in a fixed order.  1: launched; 0: outside the envelope (the caller runs dle_gemm +
// dle_colsum); > 1: error.
extern "C" int dle_gemm_colsum(const void* A, const void* B, void* C, const void* src, float* colsum_out, int M, int N, int K,
                               int64_t lda, int64_t ldb, int64_t ldc, int dtype, int act, int accumulate_colsum, void* workspace,
                               int64_t workspace_bytes, hipStream_t stream) {
  const void* mask_src = src;
  if (act != ACT_RELU_BWD && act != ACT_MUL) return 0;
  static const int mode = getenv("DLE_GEMM_COLSUM") ? atoi(getenv("DLE_GEMM_COLSUM")) : 1;
  if (!mode || !A || !B || !C || !mask_src || !colsum_out || !workspace) return 0;
  if (dtype != DLE_F16 && dtype != DLE_BF16) return 0;
  const bool al = ((((uintptr_t)A) | ((uintptr_t)B) | ((uintptr_t)C) | ((uintptr_t)mask_src) | ((uintptr_t)workspace)) & 15) == 0 &&
                  (lda & 7) == 0 && (ldb & 7) == 0 && (ldc & 7) == 0;
  if (!al || (K & 7) != 0 || (N & 7) != 0 || M < 1 || N < 8 || K < 8) return 0;
  if ((long long)lda * BM * 2 > 0x7FFFFFFFLL || (long long)ldb * BK * 2 > 0x7FFFFFFFLL) return 0;
  const long long groups_max = (M + BM - 1) / BM;
  if (workspace_bytes < groups_max * N * 4) return 0;
  Gemm2Args p = {};
  p.A = (const unsigned short*)A; p.B = (const unsigned short*)B; p.C = C; p.mask_src = (const unsigned short*)mask_src;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.out_dtype = dtype; p.act = act; p.splitk = 1; p.alpha = 1.0f;
  p.cg = make_geom(1, 1, 1, 1, 1, 1, 1, 1, 0, 1);
  p.stats = (float*)workspace; p.stats_sums = 1;
  int tile_rows = 0;
  {
    // ping-pong kernel: two partial rows per 256-row tile row (one per wavefront row group) = ceil(M / 128) rows when 256 | M
    const int r8 = dle_gemm8_try(A, B, C, nullptr, nullptr, mask_src, M, N, K, lda, ldb, ldc, 1, 0, dtype, dtype, act, 1, 0, 1.0f,
                                 nullptr, (float*)workspace, stream);
    if (r8 > 1) return r8;
    if (r8 == 1) tile_rows = 128;
    else { const int rc = launch_gemm(p, dtype, 0, 1, 0, stream, &tile_rows); if (rc) return rc + 1000; }
  }
  hipLaunchKernelGGL(colsum_fold_kernel, dim3((N + 15) / 16), dim3(256), 0, stream, (const float*)workspace, colsum_out, N,
                     (M + tile_rows - 1) / tile_rows, accumulate_colsum);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { dle_set_error("colsum_fold launch failed: %s", hipGetErrorString(e)); return (int)e + 1000; }
  return 1;
}

extern "C" int dle_gemm8_relu_bwd_bits_try(const void* dY, const void* W, void* dX, const void* bits, float* colsum_partial, int M,
                                           int N, int K, int64_t lddy, int64_t ldw, int dtype, hipStream_t stream);   // gemm8.hip

// dle_gemm_colsum with the ReLU mask of the layer below as ONE BIT per element (the keep bits dle_gemm8_relu_bits_try left
// beside that layer's forward output) instead of the 16-bit activation: C [M, N] = (A [M, K] B [K, N]) where the bit is set,
// colsum_out[n] (+)= sum_m C[m, n] of the rounded output.  The masked data gradient of an MLP layer read 2 bytes per element to
// learn 1 bit (134 MB per 1024-wide layer at batch 65536: Recommendation/DLRM/dlrm/nn/mlps.py:38-43 backward).
// 1: launched; 0: outside the ping-pong kernel's envelope (the caller uses dle_gemm_colsum with the activation); > 1: error.
extern "C" int dle_gemm_colsum_bits(const void* A, const void* B, void* C, const void* bits, float* colsum_out, int M, int N, int K,
                                    int64_t lda, int64_t ldb, int dtype, int accumulate_colsum, void* workspace,
                                    int64_t workspace_bytes, hipStream_t stream) {
  if (!A || !B || !C || !bits || !colsum_out || !workspace) return 0;
  if ((M & 255) != 0 || workspace_bytes < (long long)((M + 127) / 128) * N * 4) return 0;
  const int r8 = dle_gemm8_relu_bwd_bits_try(A, B, C, bits, (float*)workspace, M, N, K, lda, ldb, dtype, stream);
  if (r8 != 1) return r8;
  hipLaunchKernelGGL(colsum_fold_kernel, dim3((N + 15) / 16), dim3(256), 0, stream, (const float*)workspace, colsum_out, N,
                     (M + 127) / 128, accumulate_colsum);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { dle_set_error("colsum_fold launch failed: %s", hipGetErrorString(e)); return (int)e + 1000; }
  return 1;
}


// ---- convolutions as implicit GEMM (NHWC activations, KRSC weights, 16-bit in, fp32 accumulate) -------------
// Replace cuDNN's conv fwd / bwd-data / bwd-filter behind nn.Conv2d(bias=False)
// (Classification/ConvNets/image_classification/models/common.py:31-60, resnet.py:126-175).
extern "C" int dle_conv3x3_try(const void* x, const void* w, void* y, float* stats, long long stats_bytes, int N, int H,
                               int W, int C, int Ko, int dgrad, int dtype, hipStream_t stream);   // conv3x3.hip

static int conv_launch(Gemm2Args& p, int in_dtype, int amode, int bmode, hipStream_t stream) {
  return launch_gemm(p, in_dtype, amode, bmode, 0, stream);
}

static int conv_check(const char* what, int N, int H, int W, int C, int Ko, int R, int S, int stride, int pad,
                      int P, int Q, int dtype) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "%s: 16-bit dtypes only (got %d)", what, dtype);
  DLE_CHECK_ARG(N > 0 && H > 0 && W > 0 && R > 0 && S > 0 && stride > 0 && pad >= 0, "%s: bad geometry", what);
  DLE_CHECK_ARG(C % 8 == 0 && Ko % 8 == 0, "%s: channel counts must be multiples of 8 (C=%d, K=%d)", what, C, Ko);
  DLE_CHECK_ARG(P == (H + 2 * pad - R) / stride + 1 && Q == (W + 2 * pad - S) / stride + 1, "%s: output size mismatch", what);
  DLE_CHECK_ARG((long long)N * H * W * C * 2 < 0xFFFFFFE0LL && (long long)N * P * Q * Ko * 2 < 0xFFFFFFE0LL,
                "%s: tensors above 4 GiB are not addressable by this kernel", what);
  return 0;
}

// y[N,P,Q,Ko] = conv(x[N,H,W,C], w[Ko,R,S,C]); act/aux/bias as in dle_gemm (bias per output channel).
extern "C" int dle_conv2d_fwd(const void* x, const void* w, void* y, const float* bias, int N, int H, int W, int C,
                              int Ko, int R, int S, int stride, int pad, int dtype, int out_dtype, int act,
                              hipStream_t stream) {
  const int P = (H + 2 * pad - R) / stride + 1, Q = (W + 2 * pad - S) / stride + 1;
  if (int rc = conv_check("conv2d_fwd", N, H, W, C, Ko, R, S, stride, pad, P, Q, dtype)) return rc;
  DLE_CHECK_ARG(x && w && y, "conv2d_fwd: null pointer");
  DLE_CHECK_ARG(act == ACT_NONE || act == ACT_RELU, "conv2d_fwd: unsupported epilogue %d", act);
  if (R == 3 && S == 3 && stride == 1 && pad == 1 && !bias && act == ACT_NONE && out_dtype == dtype) {
    const int rc = dle_conv3x3_try(x, w, y, nullptr, 0, N, H, W, C, Ko, 0, dtype, stream);   // halo-tile kernel
    if (rc == 1) return 0;
    if (rc > 1) return rc;
  }
  Gemm2Args p = {};
  p.A = (const unsigned short*)x; p.B = (const unsigned short*)w; p.C = y; p.bias = bias;
  p.M = N * P * Q; p.N = Ko; p.K = R * S * C; p.lda = 0; p.ldb = (long long)R * S * C; p.ldc = Ko;
  p.out_dtype = out_dtype; p.act = act; p.splitk = 1; p.accumulate = 0; p.alpha = 1.f;
  p.cg = make_geom(H, W, C, P, Q, R, S, stride, pad, Ko);
  return conv_launch(p, dtype, 2, 0, stream);
}

extern "C" int dle_gemm8_colstats_try(const void* A, const void* B, void* C, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                                      int dtype, float* stats, hipStream_t stream);   // gemm8.hip
extern "C" int dle_gemm_expand_groups(int M, int N, int K);                // gemm_expand.hip
extern "C" int dle_gemm_expand_try(const void* A, const void* B, void* C, const void* src, const void* bits, float* stats, int M,
                                   int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int b_kc, int in_dtype, int out_dtype,
                                   int act, hipStream_t stream);

// dle_conv2d_fwd (no bias / activation) that ALSO leaves, per 128-row tile of the [N*P*Q, Ko] output, the column sums
// and sums of squares of the ROUNDED output in col_partial[tile_row][2][Ko] -- the BatchNorm that follows gets its
// batch statistics without re-reading the activation (dle_bn_stats_from_partials).  1x1 stride-1 convolutions run as
// plain matrices.  Returns the number of tile rows through *groups.
extern "C" int dle_conv2d_fwd_colstats(const void* x, const void* w, void* y, int N, int H, int W, int C, int Ko, int R,
                                       int S, int stride, int pad, int dtype, float* col_partial,
                                       int64_t col_partial_bytes, int* groups, hipStream_t stream) {
  const int P = (H + 2 * pad - R) / stride + 1, Q = (W + 2 * pad - S) / stride + 1;
  if (int rc = conv_check("conv2d_fwd_colstats", N, H, W, C, Ko, R, S, stride, pad, P, Q, dtype)) return rc;
  DLE_CHECK_ARG(x && w && y && col_partial && groups, "conv2d_fwd_colstats: null pointer");
  DLE_CHECK_ARG((((uintptr_t)y) & 15) == 0, "conv2d_fwd_colstats: output must be 16-byte aligned");
  const long long M = (long long)N * P * Q;
  const int g = (int)((M + BM - 1) / BM);
  DLE_CHECK_ARG(col_partial_bytes >= (long long)g * 2 * Ko * 4, "conv2d_fwd_colstats: partial buffer too small (%lld tile rows)", (long long)g);
  *groups = g;
  if (R == 3 && S == 3 && stride == 1 && pad == 1) {
    const int rc = dle_conv3x3_try(x, w, y, col_partial, col_partial_bytes, N, H, W, C, Ko, 0, dtype, stream);
    if (rc == 1) { *groups = (int)(((long long)N * (H + 1) * (W + 2) + 255) / 256); return 0; }
    if (rc > 1) return rc;
  }
  const bool plain = R == 1 && S == 1 && stride == 1 && pad == 0;
  if (plain && M < 0x7FFFFFFF && (M & 255) == 0) {
    // the deep stages (14 x 14, 7 x 7: M <= 65536 rows, K >= 128): the persistent ping-pong kernel with the statistics in its
    // register epilogue (gemm8_kernel.h EPI 3) -- the streaming / tile kernels run these shapes at 0.4 of their HBM floor.
    // DLE_CONV_STATS_GEMM8=0 pins the older kernels; DLE_CONV_STATS_GEMM8_MAXM moves the row limit (A/B measurements).
    static const int g8on = getenv("DLE_CONV_STATS_GEMM8") ? atoi(getenv("DLE_CONV_STATS_GEMM8")) : 1;
    static const long long g8maxm = getenv("DLE_CONV_STATS_GEMM8_MAXM") ? atoll(getenv("DLE_CONV_STATS_GEMM8_MAXM")) : 65536;
    if (g8on && M <= g8maxm && C >= 128 && Ko >= 256) {
      const int rc = dle_gemm8_colstats_try(x, w, y, (int)M, Ko, C, C, C, Ko, dtype, col_partial, stream);
      if (rc == 1) { *groups = g; return 0; }
      if (rc > 1) return rc;
    }
  }
  if (plain && M < 0x7FFFFFFF) {
    // channel-widening 1x1 convolutions: the streaming kernel of gemm_expand.hip (one partial row per workgroup group)
    const char* pin = getenv("DLE_GEMM_EXPAND");                          // probes / tests: "0" pins the tile kernels
    static const int kmin = getenv("DLE_EXPAND_STATS_KMIN") ? atoi(getenv("DLE_EXPAND_STATS_KMIN")) : 64;
    const int eg = dle_gemm_expand_groups((int)M, Ko, C);
    if ((!pin || atoi(pin) != 0) && C >= kmin && col_partial_bytes >= (long long)eg * 2 * Ko * 4) {
      const int rc = dle_gemm_expand_try(x, w, y, nullptr, nullptr, col_partial, (int)M, Ko, C, C, C, Ko, 1, dtype, dtype, 0, stream);
      if (rc == 1) { *groups = eg; return 0; }
      if (rc > 1) return rc;
    }
  }
  Gemm2Args p = {};
  p.A = (const unsigned short*)x; p.B = (const unsigned short*)w; p.C = y;
  p.M = (int)M; p.N = Ko; p.K = R * S * C; p.ldb = (long long)R * S * C; p.ldc = Ko;
  p.out_dtype = dtype; p.act = ACT_NONE; p.splitk = 1; p.accumulate = 0; p.alpha = 1.f;
  p.stats = col_partial; p.force_small = 1;
  p.lda = plain ? C : 0;
  p.cg = make_geom(H, W, C, P, Q, R, S, stride, pad, Ko);
  return conv_launch(p, dtype, plain ? 0 : 2, 0, stream);
}

// dx[N,H,W,C] = conv_transpose(dy[N,P,Q,Ko], w[Ko,R,S,C]) (+ addend[N,H,W,C] when non-NULL)
extern "C" int dle_conv2d_dgrad(const void* dy, const void* w, void* dx, const void* addend, int N, int H, int W,
                                int C, int Ko, int R, int S, int stride, int pad, int dtype, hipStream_t stream) {
  const int P = (H + 2 * pad - R) / stride + 1, Q = (W + 2 * pad - S) / stride + 1;
  if (int rc = conv_check("conv2d_dgrad", N, H, W, C, Ko, R, S, stride, pad, P, Q, dtype)) return rc;
  DLE_CHECK_ARG(dy && w && dx, "conv2d_dgrad: null pointer");
  if (R == 3 && S == 3 && stride == 1 && pad == 1 && !addend) {
    const int rc = dle_conv3x3_try(dy, w, dx, nullptr, 0, N, H, W, C, Ko, 1, dtype, stream);
    if (rc == 1) return 0;
    if (rc > 1) return rc;
  }
  Gemm2Args p = {};
  p.A = (const unsigned short*)dy; p.B = (const unsigned short*)w; p.C = dx;
  p.mask_src = (const unsigned short*)addend;
  p.M = N * H * W; p.N = C; p.K = R * S * Ko; p.lda = 0; p.ldb = 0; p.ldc = C;
  p.out_dtype = dtype; p.act = addend ? ACT_ADD : ACT_NONE; p.splitk = 1; p.accumulate = 0; p.alpha = 1.f;
  p.cg = make_geom(H, W, C, P, Q, R, S, stride, pad, Ko);
  return conv_launch(p, dtype, 4, 5, stream);
}

extern "C" int dle_conv3x3_wgrad_try(const void* dy, const void* x, float* dw, int N, int H, int W, int C, int Ko, int dtype,
                                     int accumulate, void* workspace, int64_t workspace_bytes, hipStream_t stream);   // conv3x3_wgrad.hip

// dw[Ko,R,S,C] (fp32) (+)= sum over pixels dy[N,P,Q,Ko]^T im2col(x[N,H,W,C]); workspace: split-K slabs.
extern "C" int dle_conv2d_wgrad(const void* dy, const void* x, float* dw, int N, int H, int W, int C, int Ko, int R,
                                int S, int stride, int pad, int dtype, int splitk, int accumulate, void* workspace,
                                int64_t workspace_bytes, hipStream_t stream) {
  const int P = (H + 2 * pad - R) / stride + 1, Q = (W + 2 * pad - S) / stride + 1;
  if (int rc = conv_check("conv2d_wgrad", N, H, W, C, Ko, R, S, stride, pad, P, Q, dtype)) return rc;
  DLE_CHECK_ARG(dy && x && dw, "conv2d_wgrad: null pointer");
  if (R == 3 && S == 3 && stride == 1 && pad == 1) {
    // halo-tile kernel (conv3x3_wgrad.hip): the nine taps share one activation patch and one dy tile per pixel tile
    const int rc = dle_conv3x3_wgrad_try(dy, x, dw, N, H, W, C, Ko, dtype, accumulate, workspace, workspace_bytes, stream);
    if (rc == 1) return 0;
    if (rc > 1) return rc;
  }
  Gemm2Args p = {};
  p.A = (const unsigned short*)dy; p.B = (const unsigned short*)x; p.C = dw;
  p.M = Ko; p.N = R * S * C; p.K = N * P * Q; p.lda = Ko; p.ldb = 0; p.ldc = (long long)R * S * C;
  p.out_dtype = DLE_F32; p.act = ACT_NONE; p.accumulate = accumulate; p.alpha = 1.f;
  p.cg = make_geom(H, W, C, P, Q, R, S, stride, pad, Ko);
  const int ktiles = (p.K + BK - 1) / BK;
  if (splitk < 1) splitk = 1;
  if (splitk > ktiles) splitk = ktiles;
  p.splitk = splitk;
  if (splitk > 1) {
    const long long need = (long long)splitk * p.M * p.N * 4;
    if (workspace && workspace_bytes >= need && (((uintptr_t)workspace) & 15) == 0) p.ws = (float*)workspace;
    else if (!accumulate) {
      hipError_t e = hipMemsetAsync(dw, 0, (size_t)p.M * p.N * 4, stream);
      if (e != hipSuccess) { dle_set_error("conv2d_wgrad memset: %s", hipGetErrorString(e)); return (int)e; }
    }
  }
  if (int rc = conv_launch(p, dtype, 1, 3, stream)) return rc;
  if (p.ws) {
    long long g = (((long long)p.M * p.N + 3) / 4 + 15) / 16;         // 16 float4 elements per workgroup trip
    if (g > 4096) g = 4096;
    if (!launch_splitk_reduce_shallow((const float*)p.ws, dw, p.M, p.N, p.ldc, splitk, accumulate, stream))
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)g), dim3(256), 0, stream, (const float*)p.ws, dw, p.M, p.N,
                         p.ldc, splitk, accumulate);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { dle_set_error("splitk_reduce launch failed: %s", hipGetErrorString(e)); return (int)e; }
  }
  return 0;
}


// ---- 3x3 / stride 2 / pad 1 data gradient without zero work -------------------------------------------------------
// dx[n, h, w, :] receives dy[n, p, q, :] w[:, r, s, :] for the taps with 2p - 1 + r = h, 2q - 1 + s = w: an even h sees only
// r = 1, an odd h sees r = 0 and r = 2 (likewise in w).  The gather form above walks all 9 taps for every pixel and
// multiplies 3/4 zeros; here the four parity classes (h & 1, w & 1) are four stride-1 correlations of dy on the P x Q
// grid with 1, 2, 2 and 4 taps, each written straight into its strided sub-grid of dx (row_extra / row_div in the
// epilogue).  The tap-restricted weights of a class are a [C][taps * Ko] k-contiguous matrix packed into the workspace.
template <int DT>
__global__ __launch_bounds__(256) void dgrad_s2_pack_kernel(const unsigned short* __restrict__ w, unsigned short* __restrict__ out,
                                                            int Ko, int C) {
  // out: class (a, b) at element offset C * Ko * {0, 1, 3, 5}[2a + b]; layout [c][r'][s'][ko]; r' = 0 <-> dy row i, 1 <-> i + 1
  const long long total = 9LL * Ko * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long t = i / C;
    const int rs = (int)(t % 9), ko = (int)(t / 9);
    const int r = rs / 3, sx = rs - r * 3;
    const int a = r == 1 ? 0 : 1, b = sx == 1 ? 0 : 1;            // parity class of the dx pixels this tap feeds
    const int rp = r == 0 ? 1 : 0, sp = sx == 0 ? 1 : 0;          // tap position inside the class kernel
    const int Sp = b ? 2 : 1, taps = (a ? 2 : 1) * Sp;
    const long long base = (long long)C * Ko * (a == 0 ? (b == 0 ? 0 : 1) : (b == 0 ? 3 : 5));
    out[base + ((long long)c * taps + rp * Sp + sp) * Ko + ko] = w[i];
  }
}

// dx [N,H,W,C] = conv_transpose(dy [N,H/2,W/2,Ko], w [Ko,3,3,C]) for stride 2, pad 1 (H, W even, Ko a multiple of 64).
// workspace: >= 9 * Ko * C * 2 bytes (the packed class weights).
extern "C" int dle_conv2d_dgrad_s2(const void* dy, const void* w, void* dx, int N, int H, int W, int C, int Ko,
                                   void* workspace, int64_t workspace_bytes, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "conv2d_dgrad_s2: 16-bit dtypes only");
  DLE_CHECK_ARG(dy && w && dx && workspace, "conv2d_dgrad_s2: null pointer");
  DLE_CHECK_ARG(N > 0 && H > 0 && W > 0 && (H & 1) == 0 && (W & 1) == 0, "conv2d_dgrad_s2: H, W must be even");
  DLE_CHECK_ARG(C % 8 == 0 && Ko % 64 == 0, "conv2d_dgrad_s2: C a multiple of 8, Ko a multiple of 64 (got %d, %d)", C, Ko);
  DLE_CHECK_ARG(workspace_bytes >= 9LL * Ko * C * 2 && (((uintptr_t)workspace) & 15) == 0, "conv2d_dgrad_s2: workspace too small");
  DLE_CHECK_ARG((long long)N * H * W * C * 2 < 0xFFFFFFE0LL, "conv2d_dgrad_s2: tensors above 4 GiB are not addressable");
  const int P = H / 2, Q = W / 2;
  {
    long long g = (9LL * Ko * C + 255) / 256;
    if (g > 2048) g = 2048;
    if (dtype == DLE_F16) hipLaunchKernelGGL(dgrad_s2_pack_kernel<DLE_F16>, dim3((unsigned)g), dim3(256), 0, stream, (const unsigned short*)w, (unsigned short*)workspace, Ko, C);
    else hipLaunchKernelGGL(dgrad_s2_pack_kernel<DLE_BF16>, dim3((unsigned)g), dim3(256), 0, stream, (const unsigned short*)w, (unsigned short*)workspace, Ko, C);
    DLE_LAUNCH_CHECK();
  }
  static const int cls_off[4] = {0, 1, 3, 5};
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) {
      const int Rp = a ? 2 : 1, Sp = b ? 2 : 1;
      Gemm2Args p = {};
      p.A = (const unsigned short*)dy;
      p.B = (const unsigned short*)workspace + (long long)C * Ko * cls_off[2 * a + b];
      p.C = (unsigned short*)dx + ((long long)a * W + b) * C;          // pixel (a, b) of image 0
      p.M = N * P * Q; p.N = C; p.K = Rp * Sp * Ko; p.lda = 0; p.ldb = (long long)Rp * Sp * Ko;
      p.ldc = 2LL * C;                                                  // next j: two pixels further
      p.row_extra = (long long)W * C;                                   // next i: one skipped image row more
      p.row_div = make_fastdiv(Q);
      p.out_dtype = dtype; p.act = ACT_NONE; p.splitk = 1; p.accumulate = 0; p.alpha = 1.f;
      // the class correlation reads dy rows i + r', columns j + s' (no padding; past the edge reads zero)
      p.cg = make_geom(P, Q, Ko, P, Q, Rp, Sp, 1, 0, C);
      if (int rc = launch_gemm(p, dtype, 2, 0, 0, stream)) return rc;
    }
  return 0;
}


// Batched GEMM over (outer, inner) slices -- the attention contractions of BERT (torch.bmm in the reference,
// LanguageModeling/BERT/modeling.py:354,373): C[z] = alpha * A[z](m,k) B[z](n,k).  Strides in elements.
extern "C" int dle_gemm_batched(const void* A, const void* B, void* C, int M, int N, int K, int64_t lda, int64_t ldb,
                                int64_t ldc, int a_kc, int b_kc, int in_dtype, int out_dtype, float alpha, int batch,
                                int batch_inner, int64_t sa_o, int64_t sa_i, int64_t sb_o, int64_t sb_i, int64_t sc_o,
                                int64_t sc_i, hipStream_t stream) {
  DLE_CHECK_ARG(in_dtype == DLE_F16 || in_dtype == DLE_BF16, "gemm_batched: 16-bit inputs only");
  DLE_CHECK_ARG(out_dtype == DLE_F32 || out_dtype == in_dtype, "gemm_batched: output fp32 or the input dtype");
  DLE_CHECK_ARG(batch >= 0 && batch_inner > 0, "gemm_batched: bad batch");
  if (batch == 0 || M == 0 || N == 0) return 0;
  DLE_CHECK_ARG(A && B && C, "gemm_batched: null pointer");
  const bool al = ((((uintptr_t)A) | ((uintptr_t)B)) & 15) == 0 && (lda & 7) == 0 && (ldb & 7) == 0 &&
                  ((sa_o | sa_i | sb_o | sb_i) & 7) == 0;
  DLE_CHECK_ARG(al && (K & 7) == 0 && (a_kc || (M & 7) == 0) && (b_kc || (N & 7) == 0) && !(a_kc == 0 && b_kc != 0),
                "gemm_batched: operands must be 16-byte aligned with K (and row-contiguous dims) multiples of 8");
  Gemm2Args p = {};
  p.A = (const unsigned short*)A; p.B = (const unsigned short*)B; p.C = C;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.out_dtype = out_dtype; p.act = ACT_NONE; p.splitk = 1; p.accumulate = 0; p.alpha = alpha;
  p.batch_inner = batch_inner; p.sa_o = sa_o; p.sa_i = sa_i; p.sb_o = sb_o; p.sb_i = sb_i; p.sc_o = sc_o; p.sc_i = sc_i;
  DLE_CHECK_ARG(batch <= 65535, "gemm_batched: batch above 65535 slices per call");
  return launch_gemm(p, in_dtype, a_kc ? 0 : 1, b_kc ? 0 : 1, batch, stream);
}

