// The persistent ping-pong 256 x 256 MFMA GEMM for gfx950 -- the kernel behind dle_gemm / dle_gemm_colsum for the big linear layers
// (LanguageModeling/BERT/modeling.py:130-160,340-384 LinearActivation / BertSelfOutput / BertIntermediate / BertOutput and
//  their backward; Recommendation/DLRM/dlrm/nn/mlps.py:38-43): forward X W^T, data gradient dY W, weight gradient dY^T X.
// (File and symbol names keep the "8" of its first version's eight phases per K tile.)
//
//   C[M,N] = epilogue( alpha * sum_k A(m,k) * B(n,k) )        (contract of gemm_dma.hip, which stays for every other shape)
//
// What is different from gemm_dma.hip's 256 x 256 tile (one barrier + vmcnt(0) per K tile, fp32 tile transposed through LDS in
// the epilogue: 0.88-0.92 PFLOP/s on the K = 1024 layers, ~29 % of a tile's time outside the K loop):
//  * PERSISTENT: one workgroup per CU walks a list of (tile, K slice) items, decoded once per workgroup into an LDS table; the
//    operand stream never drains between items -- the first K tiles of the next item are already in flight while the current
//    one's results are stored.
//  * HALF-TILE STREAM, COUNTED WAITS: a K tile (64 deep) is four 16 KiB half-tiles (A rows 0-127 / 128-255, B rows 0-127 /
//    128-255), each staged by ONE LDS-DMA piece pair per wavefront; 8 half-tile slots (two K tiles) are in flight or in use at
//    any time, and the only wait for them is s_waitcnt vmcnt(8) at the end of a load segment (never 0).
//  * PING-PONG: the eight wavefronts are two groups of four (one wavefront of each group per SIMD), staggered by one barrier:
//    while a group runs the 16 MFMAs of a segment (A half i x both B halves, K = 64) the other group issues its LDS fragment
//    reads and its DMA pieces.  A wavefront owns rows {0,128} + 64 wr .. +63 and columns {0,128} + 32 wc .. +31 of the tile, i.e.
//    one 64 x 32 block of each (A half, B half) pair.  (ktile16 below; the first version's four 8-MFMA phases per K tile
//    paid the per-segment cost of the barrier round trip twice as often: 2600 against 2216 cycles.)
//  * REGISTER EPILOGUE, ROW-CONTIGUOUS MEMORY: the B fragment rows are permuted (MFMA row 8q + 4h + e <- tile column
//    16h + 4q + e) so that a lane's 16 accumulators of a 32 x 32 block are 16 CONSECUTIVE output columns of one row: bias /
//    activation / source math runs in the accumulator layout; the rounded block then crosses a per-wavefront LDS scratch (no
//    barrier) so that every store / source-load instruction moves 64 contiguous bytes of 16 rows (128 bytes of 8 rows for fp32
//    output) -- the CU's store path is bound by requests, not bytes.  Nothing of the epilogue touches the operand stages the
//    stream is already refilling, and the two groups' epilogues run at the same time.
// LDS images, swizzles and the transpose reads of row-contiguous operands are the ones of gemm_tiles.h (half-tile = the
// TILE = 128 image).  Hazards (RAW on LDS-DMA data, WAR on restaged half-tiles) are argued next to the segments below.
#pragma once
#include "gemm_tiles.h"
#include "gemm8_walk.h"

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_RELU_BWD = 3, ACT_ADD = 4, ACT_GELU_BWD = 5, ACT_TANH = 6,
       ACT_TANH_BWD = 7, ACT_ADD_MASKED = 8, ACT_MUL = 9, ACT_GELU_DAUX = 10,
       // the ReLU of a linear layer as ONE BIT per element (this kernel only: dle_gemm8_relu_bits_try / ..._bwd_bits_try):
       // forward (EPI 1): bias + ReLU, aux RECEIVES the keep bits (bit (m N + n) & 7 of byte (m N + n) >> 3 = rounded output > 0);
       // backward (EPI 2): C = product where the bit is set, aux = those bits, NO source tensor is read
       ACT_RELU_BITS = 11, ACT_RELU_BWD_BITS = 12 };

#define G8_HALF 8192                 // 16-bit elements per half-tile image (128 rows x 64 k)
#define G8_BUF (4 * G8_HALF)         // one K tile: A0 | A1 | B0 | B1
#define G8_STAGE_BYTES (2 * G8_BUF * 2)
#define G8_TS_PITCH 80                               // bytes per row of a wavefront's transposition scratch (64 + 16: conflict-free)
#define G8_TS_BYTES (32 * G8_TS_PITCH)
#define G8_TS_BASE (G8_STAGE_BYTES + 8 * 256)
// + one 256-byte bias slot per wavefront (EPI 1) + one 32-row x 64-byte transposition scratch per wavefront (epilogue)
#define G8_TBL_BASE (G8_TS_BASE + 8 * G8_TS_BYTES)
#define G8_TBL_ITEMS 512                             // items per workgroup the LDS walk table holds (the launcher checks)
// ... + the workgroup's walk table: 16 bytes per item
#define G8_LDS_BYTES (G8_TBL_BASE + G8_TBL_ITEMS * 16)
// Lanes exchange data through the scratch without a barrier (one wavefront, in-order LDS): the compiler must be told -- thread
// by thread it may forward a lane's earlier load over another lane's store (it did: the reads of the second half-block were
// sunk into the writers' exec-masked region).  A wavefront-scope fence costs no instruction.
#define G8_WAVE_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")

struct Gemm8Args {
  const unsigned short* A;
  const unsigned short* B;
  void* C;
  void* aux;
  const float* bias;
  const unsigned short* src;
  float* ws;                 // split-K: fp32 slabs [splitk][M][N]
  float* stats;              // column sums of the rounded output: partial rows [2 * tiles_m][N] (row 2 tm + wr)
  int M, N, K;
  long long lda, ldb, ldc;
  unsigned a_bytes, b_bytes; // extent of each operand (rows x pitch x 2): the DMA's range check zero-fills rows past the end
  int out_dtype, act, splitk, accumulate;
  float alpha;
  int gm;
  unsigned c_bytes;          // extent of C (and of aux / src, same pitch) in bytes; split-K: of the slab buffer
  int nitems, grid;          // (tile, K slice) items; workgroups (= min(nitems, #CUs)): workgroup b walks items b, b + grid, ...
  int tiles_m, tiles_n, ktiles;
#ifdef G8_TIMING
  unsigned long long* dbg;   // tools/kbench/gemm8_bench.cpp: shader-clock stamps [block][group][item][4]
  int dbg_items;
#endif
};
#ifdef G8_TIMING
#define G8_STAMP(slot) do { if ((wave & 3) == 0 && lane == 0 && p.dbg && item_seq < p.dbg_items) \
    p.dbg[((((long long)blockIdx.x * 2 + wr) * p.dbg_items) + item_seq) * 4 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define G8_STAMP(slot) do { } while (0)
#endif

// (decided probe variants removed in round 6: the four-phase K tile, DMA pieces between the MFMAs, the priority modes, the 4 + 4
//  restaging split, accumulator-layout stores, store cache-policy bits -- measured numbers in DESIGN.md, section 4)
#define G8_ST_KEEP 16                  // store instructions per wavefront of an interior tile's epilogue (lower bound over all flavours)
#define G8_SB() __builtin_amdgcn_sched_barrier(0)
#define G8_BARRIER() do { G8_SB(); asm volatile("s_barrier" ::: "memory"); G8_SB(); } while (0)
#define G8_LGKM(N) do { G8_SB(); asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory"); G8_SB(); } while (0)
#define G8_VM(N) do { G8_SB(); asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); G8_SB(); } while (0)

// tile column (within a wavefront's 32-column block) that MFMA row i of the B fragment holds
__device__ __forceinline__ int g8_perm32(int i) { return 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3); }

// ---- fragment reads out of a half-tile image ---------------------------------------------------------------------
// k-contiguous image [128 rows][64 k], 16-byte chunk c of row r at slot c ^ swz_kc(r) (gemm_tiles.h).  The permuted row set of
// a ds_read_b128 lane group equals its lane set, so the B reads are as conflict-free as the A reads.
template <bool PERM>
__device__ __forceinline__ ushort8_t g8_frag_kc(const unsigned short* t, int rbase32, int ks, int lane) {
  const int row = rbase32 + (PERM ? g8_perm32(lane & 31) : (lane & 31));
  return *(const ushort8_t*)(t + row * BK + (((ks * 2 + (lane >> 5)) ^ swz_kc(row)) << 3));
}
// row-contiguous image [64 k][128 rows] read with the LDS transpose read; PERM: the four 4-row blocks a 16-lane group
// fetches are {0, 16, 4, 20} + 8 (tg & 1) instead of {0, 4, 8, 12} + 16 (tg & 1)  (output lane r of the group receives
// element r & 3 of the block fetched by the lanes with ti & 3 == r >> 2)
template <bool PERM>
__device__ __forceinline__ TrPair g8_frag_rc(const unsigned short* t, int rbase32, int ks, int lane) {
  if constexpr (!PERM) {
    return frag_issue<true, 128>(t, rbase32, ks, lane);
  } else {
    const int tg = lane >> 4, ti = lane & 15, c = ti & 3;
    const int kb = ks * 16 + (tg >> 1) * 8 + (ti >> 2);
    const int chunk = (rbase32 >> 3) + 2 * (c & 1) + (tg & 1);
    const int sub = (c >> 1) << 2;
    TrPair f;
    {
      const int cpos = (((chunk >> 1) ^ swz_rc<128>(kb)) << 1) | (chunk & 1);
      f.lo = ds_read_tr16_asm(t + kb * 128 + cpos * 8 + sub);
    }
    {
      const int k = kb + 4;
      const int cpos = (((chunk >> 1) ^ swz_rc<128>(k)) << 1) | (chunk & 1);
      f.hi = ds_read_tr16_asm(t + k * 128 + cpos * 8 + sub);
    }
    return f;
  }
}
template <int MODE, bool PERM>
__device__ __forceinline__ typename FragT<MODE == 1>::type g8_frag(const unsigned short* t, int rbase32, int ks, int lane) {
  if constexpr (MODE == 0) return g8_frag_kc<PERM>(t, rbase32, ks, lane);
  else return g8_frag_rc<PERM>(t, rbase32, ks, lane);
}

// Transpose reads with IMMEDIATE offsets: the swizzle term (k & 3) of a lane's address is the same for every k-step and for
// both 64-bit halves of a fragment, so all 8 reads of a 32-row block are lane base + {k-step * 4096 + half * 1024} bytes, and
// the half-tile's position inside the K tile buffer is a constant too: one address register per block (two for A, one for B)
// instead of one VALU add in front of every read (48 per K tile on the weight-gradient layout, whose read side is the long one).
template <bool PERM>
__device__ __forceinline__ unsigned g8_tr_lane_base(int rbase32, int lane) {
  const int tg = lane >> 4, ti = lane & 15;
  const int kb = (tg >> 1) * 8 + (ti >> 2);
  int chunk, sub;
  if (!PERM) {
    const int rbase = rbase32 + ((tg & 1) << 4);
    chunk = (rbase >> 3) + ((ti & 3) >> 1);
    sub = (ti & 1) << 2;
  } else {
    const int c = ti & 3;
    chunk = (rbase32 >> 3) + 2 * (c & 1) + (tg & 1);
    sub = (c >> 1) << 2;
  }
  const int cpos = (((chunk >> 1) ^ swz_rc<128>(kb)) << 1) | (chunk & 1);
  return (unsigned)(kb * 128 + cpos * 8 + sub) * 2u;
}
template <int OFF>
__device__ __forceinline__ short4_t g8_tr_read(unsigned addr) {
  short4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <int IMG, int KS>
__device__ __forceinline__ TrPair g8_frag_rc_imm(unsigned base) {
  TrPair f;
  f.lo = g8_tr_read<IMG + KS * 4096>(base);
  f.hi = g8_tr_read<IMG + KS * 4096 + 1024>(base);
  return f;
}

// one LDS-DMA piece (1 KiB per wavefront) to LDS byte address m0v; hidden from hipcc's wait-count pass like dma16_raw
__device__ __forceinline__ void g8_dma(int4v_t rs, unsigned m0v, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(m0v), "v"(voff), "s"(rs) : "memory");
}

// 4 bytes per lane (256 B per wavefront) by LDS-DMA: the wavefront's 64 bias values
__device__ __forceinline__ void g8_dma4(int4v_t rs, unsigned m0v, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" :: "s"(m0v), "v"(voff), "s"(rs) : "memory");
}

// per-lane constants of one operand's two DMA pieces (identical for both halves and every tile)
template <int MODE>
struct G8Lane {
  unsigned voff[2];
  int c8[2];               // row-contiguous operands: first row (inside the half-tile) of the lane's 8-row chunk
  __device__ __forceinline__ void init(int wave, int lane, long long ld) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (MODE == 0) {
        const int r = (wave * 2 + j) * 8 + (lane >> 3), cpos = lane & 7;
        const int chunk = cpos ^ swz_kc(r);
        voff[j] = (unsigned)(((long long)r * ld + chunk * 8) * 2);
        c8[j] = chunk * 8;            // k-contiguous operands: the lane's k offset inside the K tile (K tail predicate)
      } else {
        const int kr = (wave * 2 + j) * 4 + (lane >> 4), cpos = lane & 15;
        const int chunk = (((cpos >> 1) ^ swz_rc<128>(kr)) << 1) | (cpos & 1);
        voff[j] = (unsigned)(((long long)kr * ld + chunk * 8) * 2);
        c8[j] = chunk * 8;
      }
    }
  }
};

// (the same expressions, in the same order, as gelu_tanh2 / gelu_tanh2_d of gemm_dma.hip: results bit-identical to the tile kernels.
//  An algebraically shorter form -- hp = 1 - r, 1 - tanh^2 = 4 r hp, 12 instead of 22 operations -- was measured: -7 % epilogue cycles
//  on the GELU + side-output flavour, whose epilogue is bound by its two stores per block, and one rounding different, which moved
//  a 2-element gradient of the 24-layer parity test across its bar; not kept.)
// The epilogue's tanh-GELU on PAIRS of values: v_pk_mul / v_pk_add / v_pk_fma_f32 carry two fp32 lanes per instruction at the
// scalar rate (the GELU epilogue is VALU-bound: ~20 full-rate operations + exp + rcp per element, 128 elements per lane and tile);
// per element the same operations in the same order as the scalar forms below (= those of gemm_dma.hip): bit-identical results.
typedef float float2v_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2v_t g8_tanh2(float2v_t x) {
  const float2v_t a = x * 2.885390081777927f;
  const float2v_t e = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
  const float2v_t ep = e + 1.0f;
  const float2v_t r = {__builtin_amdgcn_rcpf(ep.x), __builtin_amdgcn_rcpf(ep.y)};
  return 1.0f - 2.0f * r;
}
__device__ __forceinline__ float2v_t g8_gelu2(float2v_t x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  return 0.5f * x * (1.0f + g8_tanh2(k0 * (x + k1 * x * x * x)));
}
__device__ __forceinline__ float2v_t g8_gelu_d2(float2v_t x, float2v_t& d) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float2v_t x2 = x * x;
  const float2v_t th = g8_tanh2(k0 * (x + k1 * x2 * x));
  const float2v_t hp = 0.5f * (1.0f + th);
  d = hp + 0.5f * x * (1.f - th * th) * k0 * (1.f + 3.f * k1 * x2);
  return x * hp;
}
__device__ __forceinline__ float g8_gelu(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  return 0.5f * x * (1.0f + fast_tanh(k0 * (x + k1 * x * x * x)));
}
__device__ __forceinline__ float g8_gelu_d(float x, float& d) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float x2 = x * x;
  const float th = fast_tanh(k0 * (x + k1 * x2 * x));
  const float hp = 0.5f * (1.0f + th);
  d = hp + 0.5f * x * (1.f - th * th) * k0 * (1.f + 3.f * k1 * x2);
  return x * hp;
}

// EPI: 0 = store only (16-bit, fp32 (+ accumulate), split-K slab); 1 = bias + forward activation (+ side output);
//      2 = source-tensor epilogues (ReLU mask, addend, addend under keep bits, stored derivative, GELU' / tanh' of a stored value)
//          (+ column sums);
//      3 = 16-bit store + per-(tile row, wavefront row group) column sums AND sums of squares of the rounded output: the batch
//          statistics of the BatchNorm behind a 1x1 convolution (dle_conv2d_fwd_colstats' contract, 128 rows per partial row)
// ACT: the epilogue's activation / source operation, a compile-time constant (one switch per 16 elements per block per flavour
// made the EPI 2 kernel 60 KB of code: its epilogue ran from the instruction cache misses)
template <int DT, int AM, int BMD, int EPI, int ACT>
__global__ __launch_bounds__(512, 1) void gemm8_kernel(const Gemm8Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* lds = (unsigned short*)smem_raw;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;            // group (= row block) / column block; waves w and w + 4 share a SIMD
  constexpr bool RCA = AM == 1, RCB = BMD == 1;
  constexpr int NRA = RCA ? 16 : 8;                   // LDS instructions of one A half (8 fragments)

  // item -> (tile, K slice): the workgroup decodes ITS items once, in parallel (one lane per item), into a table in LDS
  // {m0, n0, first K tile, (K slice << 16) | K tiles}; the walk reads one 16-byte record per item (same address in every lane:
  // a broadcast).  The walk order (XCD-chunked, groups of gm tile rows, K slices slowest) is g8_walk_item of gemm8_walk.h, which
  // tests/test_gemm8_walk.py compiles for the host and checks (every item exactly once, slices partition K, chunk locality).
  // Decoding inside the walk (five divisions by launch constants, even as multiply-high) kept ~18 scalars alive across the K
  // loop or re-loaded them from the argument segment inside the item switch: ~1000 cycles in the segment in which the stream
  // moves on to the next item, and SGPR spills in the flavoured instantiations.
  const int G = p.grid;
  const int my_items = __builtin_amdgcn_readfirstlane((p.nitems - (int)blockIdx.x + G - 1) / G);
  int* tbl = (int*)(smem_raw + G8_TBL_BASE);
  for (int i = tid; i < my_items; i += 512) {
    const G8WalkItem w = g8_walk_item((int)blockIdx.x + i * G, p.nitems, p.tiles_m, p.tiles_n, p.ktiles, p.splitk, p.gm);
    *(int4v_t*)(tbl + 4 * i) = (int4v_t){w.m0, w.n0, w.kt0, w.slice_and_tiles};
  }
  __syncthreads();
  auto decode = [&](int li, int& m0, int& n0, int& kt0, int& kt1, int& ky) __attribute__((always_inline)) {
    const int4v_t r = *(const int4v_t*)(tbl + 4 * li);
    m0 = __builtin_amdgcn_readfirstlane(r.x);
    n0 = __builtin_amdgcn_readfirstlane(r.y);
    kt0 = __builtin_amdgcn_readfirstlane(r.z);
    const int w = __builtin_amdgcn_readfirstlane(r.w);
    kt1 = kt0 + (w & 0xFFFF);
    ky = (int)((unsigned)w >> 16);
  };

  G8Lane<AM> la;
  G8Lane<BMD> lb;
  la.init(wave, lane, p.lda);
  lb.init(wave, lane, p.ldb);

  // ---- the operand stream: ONE cursor walks (item, K tile, half-tile) in consumption order B0, A0, B1, A1 -----------------
  int c_item = 0, c_par = 0, c_m0 = 0, c_n0 = 0, c_k0 = 0, c_kend = 0;
  bool c_valid = false;
  auto cursor_load = [&]() __attribute__((always_inline)) {
    c_valid = c_item < my_items;
    if (c_valid) {
      int kt0, kt1, ky;
      decode(c_item, c_m0, c_n0, kt0, kt1, ky);
      c_k0 = kt0 * BK;
      c_kend = kt1 * BK;
    } else {
      c_k0 = 0;
      c_kend = 0x7FFFFFFF;              // past the last item: every piece is out of range (zero-fills an idle half-tile)
    }
  };
  auto cursor_next = [&]() __attribute__((always_inline)) {
    c_par ^= 1;
    c_k0 += BK;
    if (c_k0 >= c_kend) {
      ++c_item;
      cursor_load();
    }
  };
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(lds_void*)lds;      // LDS byte address of the stages (M0 of the DMA)
  // piece J (0 / 1) of this wavefront's share of half-tile HF of operand OP at the cursor
  auto stage_piece = [&](auto OPC, auto HFC, auto JC) __attribute__((always_inline)) {
    constexpr int OP = decltype(OPC)::value, HF = decltype(HFC)::value, j = decltype(JC)::value;
    constexpr int MODE = OP == 0 ? AM : BMD;
    const int R0 = (OP == 0 ? c_m0 : c_n0) + HF * 128;
    const int nrows = OP == 0 ? p.M : p.N;
    const unsigned ld = (unsigned)(OP == 0 ? p.lda : p.ldb);
    // byte offset of the half-tile's first element; 32-bit by the launcher's extent check ((rows + 256) * pitch * 2 < 2^32)
    const unsigned sb = (MODE == 0 ? (unsigned)R0 * ld + (unsigned)c_k0 : (unsigned)c_k0 * ld + (unsigned)R0) * 2u;
    const unsigned total = OP == 0 ? p.a_bytes : p.b_bytes;
    const unsigned nrec = (c_valid && sb < total) ? total - sb : 0u;
    const unsigned long long base = (unsigned long long)(OP == 0 ? p.A : p.B) + sb;
    int4v_t rs;
    rs.x = (int)(unsigned)base;
    rs.y = (int)(unsigned)(base >> 32);
    rs.z = (int)nrec;
    rs.w = 0x00020000;
    const unsigned dst = lds0 + (unsigned)(c_par * G8_BUF + (OP == 0 ? 0 : 2 * G8_HALF) + HF * G8_HALF + wave * 1024) * 2u;
    unsigned vo;
    // row-contiguous: the lane's 8 rows are inside the operand; k-contiguous: its 8 k values are below K (rows: range check)
    if constexpr (OP == 0) vo = ((MODE == 0 ? c_k0 + la.c8[j] < p.K : R0 + la.c8[j] < nrows)) ? la.voff[j] : OOB_OFF;
    else vo = ((MODE == 0 ? c_k0 + lb.c8[j] < p.K : R0 + lb.c8[j] < nrows)) ? lb.voff[j] : OOB_OFF;
    g8_dma(rs, dst + j * 1024, vo);
  };
  auto stage = [&](auto OPC, auto HFC) __attribute__((always_inline)) {
    stage_piece(OPC, HFC, std::integral_constant<int, 0>());
    stage_piece(OPC, HFC, std::integral_constant<int, 1>());
  };
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;

  // prologue: the first two K tiles (all 8 half-tile slots); K tile 0 has landed when 8 pieces are left in flight
  cursor_load();
  stage(I1(), I0()); stage(I0(), I0()); stage(I1(), I1()); stage(I0(), I1());
  cursor_next();
  stage(I1(), I0()); stage(I0(), I0()); stage(I1(), I1()); stage(I0(), I1());
  cursor_next();
  G8_VM(8);
  G8_BARRIER();
  if (wr == 1) G8_BARRIER();            // the stagger: group 1 runs one barrier behind group 0 from here on

  float16_t acc[2][2][2];               // [A half][B half][32-row block]
  typename FragT<RCA>::type fa[2][4];   // one A half: [32-row block][k-step]
  typename FragT<RCB>::type fb0[4], fb1[4];

  // row-contiguous operands: a lane's first transpose-read address inside a half-tile image (bytes)
  unsigned trA0 = 0, trA1 = 0, trB = 0;
  if constexpr (RCA) { trA0 = g8_tr_lane_base<false>(wr * 64, lane); trA1 = g8_tr_lane_base<false>(wr * 64 + 32, lane); }
  if constexpr (RCB) trB = g8_tr_lane_base<true>(wc * 32, lane);
  // fragment reads of one half-tile: H = its index inside the K tile buffer (A0, A1, B0, B1)
  auto read_a = [&](auto HC, const unsigned short* bufc, unsigned lbase) __attribute__((always_inline)) {
    constexpr int H = decltype(HC)::value;
    if constexpr (RCA) {
      static_for<0, 4>([&](auto KS) __attribute__((always_inline)) {
        constexpr int ks = decltype(KS)::value;
        fa[0][ks] = g8_frag_rc_imm<H * G8_HALF * 2, ks>(lbase + trA0);
      });
      static_for<0, 4>([&](auto KS) __attribute__((always_inline)) {
        constexpr int ks = decltype(KS)::value;
        fa[1][ks] = g8_frag_rc_imm<H * G8_HALF * 2, ks>(lbase + trA1);
      });
    } else {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fa[b][ks] = g8_frag_kc<false>(bufc + H * G8_HALF, wr * 64 + b * 32, ks, lane);
    }
  };
  auto read_b = [&](auto HC, auto& fbx, const unsigned short* bufc, unsigned lbase) __attribute__((always_inline)) {
    constexpr int H = decltype(HC)::value;
    if constexpr (RCB) {
      static_for<0, 4>([&](auto KS) __attribute__((always_inline)) {
        constexpr int ks = decltype(KS)::value;
        fbx[ks] = g8_frag_rc_imm<H * G8_HALF * 2, ks>(lbase + trB);
      });
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fbx[ks] = g8_frag_kc<true>(bufc + H * G8_HALF, wc * 32, ks, lane);
    }
  };

  // ---- one K tile = TWO segments of 16 MFMAs per group: {C00, C01} from A0 x {B0, B1}, then {C10, C11} from A1 x {B0, B1}
  // (the B fragments stay in registers).  What a segment costs beyond its MFMA time -- the barrier round trip, the partner's
  // load segment on the same SIMD, the change of roles: ~70 cycles -- is paid 4 times per K tile instead of 8.
  //   load segment X(t): reads B0, A0, B1 of K tile t | streams A1(t + 1) into the slot A1(t - 1) left (then the cursor moves on)
  //   load segment Y(t): reads A1 of K tile t         | streams B0, A0, B1 of K tile t + 2 into the slots X(t) has just read
  // WAR: a load segment retires its reads (lgkmcnt(0)) BEFORE its closing barrier; the other group's next load segment, which
  //   is the first that can restage those slots, starts behind that barrier.
  // RAW: X(t + 1)'s half-tiles were issued in Y(t - 1), A1(t + 1) in X(t): at the end of every load segment exactly 8 younger
  //   pieces are in flight behind the half-tiles the NEXT load segment reads -- vmcnt(8), then the closing barrier (both groups'
  //   waits precede the barrier in front of the first read).
  // After an interior tile's epilogue the three following load segments wait for half-tiles that are OLDER than its stores (the
  //   hoisted A1 piece, the previous Y segment's pieces): vmcnt(8 + G8_ST_KEEP) leaves the stores in flight; the fourth needs
  //   pieces issued behind the stores and waits them out, ~2 K tiles after they were issued.
  int keep_segments = 0;
  auto seg_wait = [&](auto NC) __attribute__((always_inline)) {
    constexpr int N = decltype(NC)::value;
    if (keep_segments > 0) { --keep_segments; G8_VM(N + G8_ST_KEEP); }
    else G8_VM(N);
  };
  auto mma16 = [&](auto IC, auto ZC) __attribute__((always_inline)) {
    constexpr int i = decltype(IC)::value;
    constexpr bool zc = decltype(ZC)::value;
    const float16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
        acc[i][0][b] = Mfma32x16<DT>::run(frag_value(fb0[ks]), frag_value(fa[b][ks]), (zc && ks == 0) ? zero : acc[i][0][b]);
#pragma unroll
      for (int b = 0; b < 2; ++b)
        acc[i][1][b] = Mfma32x16<DT>::run(frag_value(fb1[ks]), frag_value(fa[b][ks]), (zc && ks == 0) ? zero : acc[i][1][b]);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto ktile16 = [&](int cpar, auto FIRSTC, auto ZEROC) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(FIRSTC)::value;
    typedef decltype(ZEROC) ZC;
    const unsigned short* bufc = lds + cpar * G8_BUF;
    const unsigned lbase = lds0 + (unsigned)cpar * (G8_BUF * 2u);
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3;
    // ---- load segment X
    read_b(I2(), fb0, bufc, lbase);
    G8_SB();
    read_a(I0(), bufc, lbase);
    G8_SB();
    read_b(I3(), fb1, bufc, lbase);
    G8_SB();
    typedef std::integral_constant<int, 8> W8;
    if constexpr (!FIRST) {
      stage(I0(), I1());
      cursor_next();
    }
    G8_LGKM(0);
    seg_wait(W8());
    G8_BARRIER();
    mma16(I0(), ZC());
    G8_BARRIER();
    // ---- load segment Y
    read_a(I1(), bufc, lbase);
    G8_SB();
    stage(I1(), I0());
    stage(I0(), I0());
    stage(I1(), I1());
    G8_LGKM(0);
    seg_wait(W8());
    G8_BARRIER();
    mma16(I1(), ZC());
    G8_BARRIER();
  };

  const int fr = lane & 31, fh = lane >> 5;
  int cpar = 0;
#ifdef G8_TIMING
  int item_seq = 0;
#endif
  for (int it = 0; it < my_items; ++it) {
    int m0, n0, kt0, kt1, ky;
    decode(it, m0, n0, kt0, kt1, ky);
    G8_STAMP(0);
    const bool interior = m0 + 256 <= p.M && n0 + 256 <= p.N;
    // (EPI 1: the bias row is added in the epilogue from SCALAR loads -- wave-uniform addresses, lgkmcnt, not the vector-memory
    //  queue that retires in order behind the stream's pieces.  Starting the accumulators from the bias instead is one rounding
    //  different from `sum + bias`: enough to move the reference-size WaveGlow loss by 1e-3, so the order of the tile kernels
    //  is kept and the results stay bit-identical to theirs.)
    constexpr bool bias_in_acc = false;
    if constexpr (EPI == 1) {
      // The bias values of this wavefront's 64 columns (2 x 32) travel with the stream: ONE 4-byte LDS-DMA piece per wavefront
      // into its own 256-byte slot behind the stages, issued here, landed (in-order vmcnt) by the second K tile's wait at the
      // latest, read back with four ds_read_b128 per column half in the epilogue.  (Vector loads in the epilogue sit behind
      // the stream's prefetch in the in-order queue: their wait drained it, +2.6 k cycles per tile; scalar loads of 32
      // values per block cost more than that in s_load latency.)  Columns past N read zero through the range check.
      if (p.bias) {
        int4v_t rsb;
        rsb.x = (int)(unsigned)(unsigned long long)p.bias;
        rsb.y = (int)(unsigned)((unsigned long long)p.bias >> 32);
        rsb.z = p.N * 4;
        rsb.w = 0x00020000;
        const unsigned col = (unsigned)(n0 + (lane >> 5) * 128 + wc * 32 + (lane & 31));
        g8_dma4(rsb, lds0 + G8_STAGE_BYTES + (unsigned)wave * 256u, col * 4u);
      }
    }
#ifdef G8_TIMING_KT
#define G8_KT_STAMP(k) do { if (item_seq == G8_TIMING_KT && (k) < 60 && (wave & 3) == 0 && lane == 0 && p.dbg) \
    p.dbg[((((long long)blockIdx.x * 2 + wr) * p.dbg_items) + p.dbg_items - 16) * 4 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define G8_KT_STAMP(k) do { } while (0)
#endif
    G8_KT_STAMP(0);
    ktile16(cpar, std::integral_constant<bool, true>(), std::integral_constant<bool, true>());
    cpar ^= 1;
    for (int kt = kt0 + 1; kt < kt1; ++kt) {
      G8_KT_STAMP(kt - kt0);
      ktile16(cpar, std::integral_constant<bool, false>(), std::integral_constant<bool, false>());
      cpar ^= 1;
    }
    G8_KT_STAMP(kt1 - kt0);
    // the one half-tile slot that is free now (A1 of the K tile just finished) is refilled BEFORE the stores below
    stage(I0(), I1());
    cursor_next();
    keep_segments = interior ? 3 : 0;                  // interior tile: every store instruction below is issued
    G8_STAMP(1);
    // The two groups run their epilogues AT THE SAME TIME: group 0 (one barrier ahead) gives group 1 the barrier its last MFMA
    // phase is waiting on before it starts storing, group 1 gives one back after its stores (below), so the stagger of the K
    // loop is the same on the other side.  Without the pair group 1 sat at that barrier through group 0's epilogue and group 0
    // sat in phase 0 of the next tile through group 1's: two epilogues back to back per item with the matrix cores idle.
    if (wr == 0) G8_BARRIER();

    // The epilogue's arguments (output / side / source pointers, pitch, bias, flags) are read from the kernel-argument segment
    // HERE, once per item, through a pointer the compiler cannot see through: kept in scalar registers across the K loop they
    // pushed the flavoured instantiations 30-50 SGPRs over the budget, and the spills (v_readlane / v_writelane and argument
    // re-loads inside the phases) cost the K loop of the bias / source flavours ~8 % against the store-only one.
    const __attribute__((address_space(4))) Gemm8Args* q = (const __attribute__((address_space(4))) Gemm8Args*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(q));
    // ... each read ONCE (pinned in a scalar register for the length of the epilogue: left to the compiler, every use inside
    // the 8 blocks became its own s_load + wait)
    unsigned long long e_C = (unsigned long long)q->C, e_aux = (unsigned long long)q->aux, e_bias = (unsigned long long)q->bias,
                       e_src = (unsigned long long)q->src;
    int e_ldc = (int)q->ldc;
    unsigned e_cbytes = q->c_bytes;
    asm volatile("" : "+s"(e_C), "+s"(e_aux), "+s"(e_bias), "+s"(e_src), "+s"(e_ldc), "+s"(e_cbytes));
    // ---- epilogue, in the accumulator layout: lane (fr, fh) of block (i, j, b) owns row m0 + 128 i + 64 wr + 32 b + fr,
    // columns n0 + 128 j + 32 wc + 16 fh .. + 15
    float st[2][16];                               // column sums of the rounded output over this lane's rows (EPI 2 / 3, q->stats)
    float st2[EPI == 3 ? 2 : 1][16];               // ... and the sums of squares (EPI 3)
    if constexpr (EPI == 2 || EPI == 3) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[j][r] = 0.f; if constexpr (EPI == 3) st2[j][r] = 0.f; }
    }
    // Interior tile, 16-bit output (every flavour) or fp32 output (plain / slab): no per-lane predicate, no 64-bit address
    // arithmetic -- a lane's byte offset inside the tile is a constant (rows fr, columns 16 fh), the block's position is a
    // scalar offset of the buffer instruction; the source tensor of EPI 2 is requested for all 8 blocks before the first use.
    constexpr bool F32FAST = EPI == 0 || (EPI == 1 && ACT == ACT_NONE);      // fp32 output: plain / slab / bias only
    const bool fastpath = interior && q->alpha == 1.0f && e_cbytes != 0 &&
                          (q->out_dtype != DLE_F32 ? true : (F32FAST && !(q->accumulate && !q->ws)));
    if (fastpath) {
      const bool f32o = q->out_dtype == DLE_F32;
      const unsigned esz = f32o ? 4u : 2u;
      const unsigned pitch = (unsigned)((EPI == 0 && q->ws) ? p.N : e_ldc) * esz;      // bytes per row
      const unsigned lane_off = (unsigned)fr * pitch + (unsigned)fh * 16u * esz;
      const void* cbase = (EPI == 0 && q->ws) ? (const void*)(q->ws + (long long)ky * p.M * p.N) : (const void*)e_C;
      __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)cbase, 0, (int)e_cbytes, 0x00020000);
      // (the block offset goes into the VECTOR offset of the stores, not into their scalar offset field: with a scalar-register
      //  soffset hipcc assumes the wide-store data hazard away and may overwrite a store's data registers in the very next
      //  instruction -- on gfx950 the last data register then reaches memory corrupted: the fp16 GELU side output had garbage in
      //  every 8th column, a few thousand different elements per run)
      auto blk_off = [&](int i, int j, int b) __attribute__((always_inline)) {
        return (unsigned)(m0 + i * 128 + wr * 64 + b * 32) * pitch + (unsigned)(n0 + j * 128 + wc * 32) * esz;
      };
      if (f32o) {
        if constexpr (F32FAST) {
          // fp32 output (split-K slab / fp32 C (+ bias: the MLM decoder)): 128 bytes per row of a block.  Same exchange as the 16-bit path below, half a
          // block (16 rows x 144-byte pitch) at a time: a store instruction then carries 8 full rows of 128 bytes instead of
          // 32 rows x two separate 16-byte pieces.
          unsigned char* ts = smem_raw + G8_TS_BASE + wave * G8_TS_BYTES;
          const unsigned tw = (unsigned)(fr & 15) * 144u + (unsigned)fh * 64u;
          const unsigned tr = (unsigned)(lane >> 3) * 144u + (unsigned)(lane & 7) * 16u;
          const unsigned lane_off_f = (unsigned)(lane >> 3) * pitch + (unsigned)(lane & 7) * 16u;
          static_for<0, 8>([&](auto BI) __attribute__((always_inline)) {
            constexpr int bi = decltype(BI)::value, i = bi >> 2, j = (bi >> 1) & 1, b = bi & 1;
            const unsigned so = blk_off(i, j, b);
            {
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                if ((fr >> 4) == h) {
#pragma unroll
                  for (int q = 0; q < 4; ++q) {
                    float4_t v = {acc[i][j][b][4 * q], acc[i][j][b][4 * q + 1], acc[i][j][b][4 * q + 2], acc[i][j][b][4 * q + 3]};
                    if constexpr (EPI == 1) {
                      if (e_bias) v += *(const float4_t*)((const float*)(smem_raw + G8_STAGE_BYTES + wave * 256) + j * 32 + fh * 16 + 4 * q);
                    }
                    *(float4_t*)(ts + tw + 16 * q) = v;
                  }
                }
                G8_WAVE_FENCE();
                const float4_t t0 = *(const float4_t*)(ts + tr), t1 = *(const float4_t*)(ts + tr + 8 * 144);
                G8_WAVE_FENCE();
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, t0), rc, lane_off_f + so + (unsigned)(16 * h) * pitch, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, t1), rc, lane_off_f + so + (unsigned)(16 * h + 8) * pitch, 0, 0);
              }
            }
          });
        }
      } else {
        // 16-bit output / side output / source tensor move through HBM in ROW-CONTIGUOUS pieces: lane l of a store (load)
        // instruction carries bytes 16 (l & 3) .. + 15 of row l >> 2 (+ 16 for the second instruction) of the wavefront's
        // 32 x 32 block -- 64 contiguous bytes per row, 16 rows per instruction.  In the accumulator layout a lane owns 32
        // bytes of ONE row in two registers quads: an instruction then touches 32 rows with two separate 16-byte pieces each,
        // 32 half-filled 64-byte requests, and the L2 request rate (not bytes) bounded the epilogue at ~15 bytes / clock / CU:
        // 8.9 k cycles for the two groups' halves of a tile, 3.6 k with this pattern.  The exchange between the two layouts goes
        // through a per-wavefront LDS scratch (32 rows x 80 bytes; written and read by this wavefront only: LDS instructions
        // of one wavefront execute in order, no barrier).
        unsigned char* ts = smem_raw + G8_TS_BASE + wave * G8_TS_BYTES;
        const unsigned ts_acc = (unsigned)fr * G8_TS_PITCH + (unsigned)fh * 32u;                 // accumulator layout
        const unsigned ts_row = (unsigned)(lane >> 2) * G8_TS_PITCH + (unsigned)(lane & 3) * 16u;  // memory layout (+ 16 rows)
        const unsigned lane_off_t = (unsigned)(lane >> 2) * pitch + (unsigned)(lane & 3) * 16u;
        const unsigned second_t = 16u * pitch;
        auto to_rows = [&](ushort8_t& a, ushort8_t& b2) __attribute__((always_inline)) {          // accumulator -> memory layout
          *(ushort8_t*)(ts + ts_acc) = a;
          *(ushort8_t*)(ts + ts_acc + 16) = b2;
          G8_WAVE_FENCE();
          a = *(const ushort8_t*)(ts + ts_row);
          b2 = *(const ushort8_t*)(ts + ts_row + 16 * G8_TS_PITCH);
          G8_WAVE_FENCE();
        };
        auto from_rows = [&](ushort8_t& a, ushort8_t& b2) __attribute__((always_inline)) {        // memory -> accumulator layout
          *(ushort8_t*)(ts + ts_row) = a;
          *(ushort8_t*)(ts + ts_row + 16 * G8_TS_PITCH) = b2;
          G8_WAVE_FENCE();
          a = *(const ushort8_t*)(ts + ts_acc);
          b2 = *(const ushort8_t*)(ts + ts_acc + 16);
          G8_WAVE_FENCE();
        };
        ushort8_t sv[EPI == 2 ? 8 : 1][2];
        constexpr bool KBITS = EPI == 2 && (ACT == ACT_ADD_MASKED || ACT == ACT_RELU_BWD_BITS);
        unsigned short kb[KBITS ? 8 : 1];      // the lane's 16 keep bits of each block (accumulator layout)
        if constexpr (EPI == 2) {
          if constexpr (ACT != ACT_RELU_BWD_BITS) {
            __amdgpu_buffer_rsrc_t rsrc_s = __builtin_amdgcn_make_buffer_rsrc((void*)e_src, 0, (int)e_cbytes, 0x00020000);
            static_for<0, 8>([&](auto BI) __attribute__((always_inline)) {
              constexpr int bi = decltype(BI)::value, i = bi >> 2, j = (bi >> 1) & 1, b = bi & 1;
              const unsigned so = blk_off(i, j, b);
              sv[bi][0] = __builtin_bit_cast(ushort8_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc_s, lane_off_t, so, 0));
              sv[bi][1] = __builtin_bit_cast(ushort8_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc_s, lane_off_t + second_t, so, 0));
            });
          }
          if constexpr (KBITS) {
            // keep bits: bit (m ldc + n) & 7 of byte (m ldc + n) >> 3 -- a lane's 16 columns are two bytes (ldc and the column
            // base are multiples of 8: the launcher checks); byte offsets are the 16-bit element offsets / 8 = byte offsets / 16
            __amdgpu_buffer_rsrc_t rsrc_k = __builtin_amdgcn_make_buffer_rsrc((void*)e_aux, 0, (int)(e_cbytes >> 4), 0x00020000);
            static_for<0, 8>([&](auto BI) __attribute__((always_inline)) {
              constexpr int bi = decltype(BI)::value, i = bi >> 2, j = (bi >> 1) & 1, b = bi & 1;
              kb[bi] = __builtin_amdgcn_raw_buffer_load_b16(rsrc_k, (lane_off + blk_off(i, j, b)) >> 4, 0, 0);
            });
          }
        }
        __amdgpu_buffer_rsrc_t ra = rc;
        if constexpr (EPI == 1 && ACT != ACT_RELU_BITS) { if (e_aux) ra = __builtin_amdgcn_make_buffer_rsrc((void*)e_aux, 0, (int)e_cbytes, 0x00020000); }
        if constexpr (EPI == 1 && ACT == ACT_RELU_BITS) ra = __builtin_amdgcn_make_buffer_rsrc((void*)e_aux, 0, (int)(e_cbytes >> 4), 0x00020000);
        static_for<0, 8>([&](auto BI) __attribute__((always_inline)) {
          constexpr int bi = decltype(BI)::value, i = bi >> 2, j = (bi >> 1) & 1, b = bi & 1;
          const unsigned so = blk_off(i, j, b);
          float v[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = acc[i][j][b][r];
          if constexpr (EPI == 1) {
            if (e_bias) {
              const float* bl = (const float*)(smem_raw + G8_STAGE_BYTES + wave * 256) + j * 32 + fh * 16;    // (see the item head)
#pragma unroll
              for (int r4 = 0; r4 < 4; ++r4) {
                const float4_t bq = *(const float4_t*)(bl + 4 * r4);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[4 * r4 + r] += bq[r];
              }
            }
            float side[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) side[r] = v[r];
            if (ACT == ACT_RELU || ACT == ACT_RELU_BITS) {
#pragma unroll
              for (int r = 0; r < 16; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
            } else if (ACT == ACT_GELU) {
#pragma unroll
              for (int r = 0; r < 16; r += 2) {
                const float2v_t y2 = g8_gelu2((float2v_t){v[r], v[r + 1]});
                v[r] = y2.x; v[r + 1] = y2.y;
              }
            } else if (ACT == ACT_TANH) {
#pragma unroll
              for (int r = 0; r < 16; r += 2) {
                const float2v_t y2 = g8_tanh2((float2v_t){v[r], v[r + 1]});
                v[r] = y2.x; v[r + 1] = y2.y;
              }
            } else if (ACT == ACT_GELU_DAUX) {
#pragma unroll
              for (int r = 0; r < 16; r += 2) {
                float2v_t d2;
                const float2v_t y2 = g8_gelu_d2((float2v_t){v[r], v[r + 1]}, d2);
                v[r] = y2.x; v[r + 1] = y2.y;
                side[r] = d2.x; side[r + 1] = d2.y;
              }
            }
            if constexpr (ACT != ACT_RELU_BITS) {
              if (e_aux) {
                ushort8_t x0 = pack8<DT>(side), x1 = pack8<DT>(side + 8);
                to_rows(x0, x1);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, x0), ra, lane_off_t + so, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, x1), ra, lane_off_t + so + second_t, 0, 0);
              }
            }
          } else if constexpr (EPI == 2 && ACT == ACT_RELU_BWD_BITS) {
            const unsigned bits = kb[KBITS ? bi : 0];
#pragma unroll
            for (int r = 0; r < 16; ++r) { if (!((bits >> r) & 1u)) v[r] = 0.f; }
          } else if constexpr (EPI == 2) {
            float y[16];
            from_rows(sv[bi][0], sv[bi][1]);
            unpack8<DT>(sv[bi][0], y);
            unpack8<DT>(sv[bi][1], y + 8);
            if (ACT == ACT_RELU_BWD) {
#pragma unroll
              for (int r = 0; r < 16; ++r) v[r] = y[r] > 0.f ? v[r] : 0.f;
            } else if (ACT == ACT_ADD) {
#pragma unroll
              for (int r = 0; r < 16; ++r) v[r] += y[r];
            } else if (ACT == ACT_ADD_MASKED) {
              const unsigned bits = kb[KBITS ? bi : 0];
#pragma unroll
              for (int r = 0; r < 16; ++r) { if ((bits >> r) & 1u) v[r] += y[r]; }
            } else if (ACT == ACT_MUL) {
#pragma unroll
              for (int r = 0; r < 16; ++r) v[r] *= y[r];
            } else if (ACT == ACT_TANH_BWD) {
#pragma unroll
              for (int r = 0; r < 16; ++r) v[r] *= (1.f - y[r] * y[r]);
            } else {
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                float d;
                g8_gelu_d(y[r], d);
                v[r] *= d;
              }
            }
          }
          const ushort8_t o0 = pack8<DT>(v), o1 = pack8<DT>(v + 8);
          if constexpr (EPI == 1 && ACT == ACT_RELU_BITS) {
            // keep bits of the ROUNDED output (what a mask derived from the stored 16-bit activation would say; v >= 0 here, so
            // "positive" = a non-zero bit pattern): two bytes per lane and block, in the accumulator layout
            unsigned kbits = 0;
#pragma unroll
            for (int r = 0; r < 8; ++r) kbits |= (o0[r] != 0 ? 1u : 0u) << r | (o1[r] != 0 ? 1u : 0u) << (8 + r);
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short)kbits, ra, (lane_off + so) >> 4, 0, 0);
          }
          {
            ushort8_t t0 = o0, t1 = o1;
            to_rows(t0, t1);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, t0), rc, lane_off_t + so, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, t1), rc, lane_off_t + so + second_t, 0, 0);
          }
          if constexpr (EPI == 2 || EPI == 3) {
            if (q->stats) {
              float vr[16];
              unpack8<DT>(o0, vr);
              unpack8<DT>(o1, vr + 8);
#pragma unroll
              for (int r = 0; r < 16; ++r) { st[j][r] += vr[r]; if constexpr (EPI == 3) st2[j][r] += vr[r] * vr[r]; }
            }
          }
        });
      }
    } else
    static_for<0, 8>([&](auto BI) __attribute__((always_inline)) {
      constexpr int bi = decltype(BI)::value, i = bi >> 2, j = (bi >> 1) & 1, b = bi & 1;
      const int m = m0 + i * 128 + wr * 64 + b * 32 + fr;
      const int n = n0 + j * 128 + wc * 32 + fh * 16;
      if (m < p.M && n < p.N) {
        const bool hi_ok = n + 8 < p.N;            // N is a multiple of 8: each 8-column group is entirely in or out
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[i][j][b][r];
        if (q->alpha != 1.0f) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] *= q->alpha;
        }
        if (EPI == 0 && q->ws) {                    // split-K partial -> fp32 slab
          float* c = q->ws + ((long long)ky * p.M + m) * p.N + n;
          *(float4_t*)c = (float4_t){v[0], v[1], v[2], v[3]};
          *(float4_t*)(c + 4) = (float4_t){v[4], v[5], v[6], v[7]};
          if (hi_ok) {
            *(float4_t*)(c + 8) = (float4_t){v[8], v[9], v[10], v[11]};
            *(float4_t*)(c + 12) = (float4_t){v[12], v[13], v[14], v[15]};
          }
        } else {
          const long long off = (long long)m * e_ldc + n;
          if (EPI == 1 && e_bias && !bias_in_acc) {
            const float4_t b0 = *(const float4_t*)((const float*)e_bias + n), b1 = *(const float4_t*)((const float*)e_bias + n + 4);
            float4_t b2 = {0.f, 0.f, 0.f, 0.f}, b3 = {0.f, 0.f, 0.f, 0.f};
            if (hi_ok) { b2 = *(const float4_t*)((const float*)e_bias + n + 8); b3 = *(const float4_t*)((const float*)e_bias + n + 12); }
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] += b0[r]; v[4 + r] += b1[r]; v[8 + r] += b2[r]; v[12 + r] += b3[r]; }
          }
          if (q->out_dtype == DLE_F32) {            // (EPI 0 / 1 with act none: checked by the launcher)
            float* c = (float*)e_C + off;
            if (q->accumulate) {
              const float4_t c0 = *(const float4_t*)c, c1 = *(const float4_t*)(c + 4);
              float4_t c2 = {0.f, 0.f, 0.f, 0.f}, c3 = {0.f, 0.f, 0.f, 0.f};
              if (hi_ok) { c2 = *(const float4_t*)(c + 8); c3 = *(const float4_t*)(c + 12); }
#pragma unroll
              for (int r = 0; r < 4; ++r) { v[r] += c0[r]; v[4 + r] += c1[r]; v[8 + r] += c2[r]; v[12 + r] += c3[r]; }
            }
            *(float4_t*)c = (float4_t){v[0], v[1], v[2], v[3]};
            *(float4_t*)(c + 4) = (float4_t){v[4], v[5], v[6], v[7]};
            if (hi_ok) {
              *(float4_t*)(c + 8) = (float4_t){v[8], v[9], v[10], v[11]};
              *(float4_t*)(c + 12) = (float4_t){v[12], v[13], v[14], v[15]};
            }
          } else {
            float side[16];
            bool has_side = false;
            if constexpr (EPI == 1) {
              has_side = e_aux != 0 && ACT != ACT_RELU_BITS;
#pragma unroll
              for (int r = 0; r < 16; ++r) side[r] = v[r];           // pre-activation
              if (ACT == ACT_RELU || ACT == ACT_RELU_BITS) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
              } else if (ACT == ACT_GELU) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = g8_gelu(v[r]);
              } else if (ACT == ACT_TANH) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = fast_tanh(v[r]);
              } else if (ACT == ACT_GELU_DAUX) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = g8_gelu_d(v[r], side[r]);   // side = the derivative
              }
            } else if constexpr (EPI == 2 && ACT == ACT_RELU_BWD_BITS) {
              const unsigned char* kbp = (const unsigned char*)e_aux + (off >> 3);
              const unsigned bits = (unsigned)kbp[0] | (hi_ok ? (unsigned)kbp[1] << 8 : 0u);
#pragma unroll
              for (int r = 0; r < 16; ++r) { if (!((bits >> r) & 1u)) v[r] = 0.f; }
            } else if constexpr (EPI == 2) {
              const unsigned short* s = (const unsigned short*)e_src + off;
              ushort8_t s0 = *(const ushort8_t*)s, s1 = {0, 0, 0, 0, 0, 0, 0, 0};
              if (hi_ok) s1 = *(const ushort8_t*)(s + 8);
              float y[16];
              unpack8<DT>(s0, y);
              unpack8<DT>(s1, y + 8);
              if (ACT == ACT_RELU_BWD) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = y[r] > 0.f ? v[r] : 0.f;
              } else if (ACT == ACT_ADD) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += y[r];
              } else if (ACT == ACT_ADD_MASKED) {
                const unsigned char* kbp = (const unsigned char*)e_aux + (off >> 3);
                const unsigned bits = (unsigned)kbp[0] | (hi_ok ? (unsigned)kbp[1] << 8 : 0u);
#pragma unroll
                for (int r = 0; r < 16; ++r) { if ((bits >> r) & 1u) v[r] += y[r]; }
              } else if (ACT == ACT_MUL) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] *= y[r];
              } else if (ACT == ACT_TANH_BWD) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] *= (1.f - y[r] * y[r]);
              } else {                                               // ACT_GELU_BWD: y = the pre-activation
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                  float d;
                  g8_gelu_d(y[r], d);
                  v[r] *= d;
                }
              }
            }
            const ushort8_t o0 = pack8<DT>(v), o1 = pack8<DT>(v + 8);
            unsigned short* c = (unsigned short*)e_C + off;
            *(ushort8_t*)c = o0;
            if (hi_ok) *(ushort8_t*)(c + 8) = o1;
            if constexpr (EPI == 1 && ACT == ACT_RELU_BITS) {
              unsigned kbits = 0;
#pragma unroll
              for (int r = 0; r < 8; ++r) kbits |= (o0[r] != 0 ? 1u : 0u) << r | (o1[r] != 0 ? 1u : 0u) << (8 + r);
              unsigned char* kbp = (unsigned char*)e_aux + (off >> 3);
              kbp[0] = (unsigned char)(kbits & 0xffu);
              if (hi_ok) kbp[1] = (unsigned char)(kbits >> 8);
            }
            if (EPI == 1 && has_side) {
              unsigned short* a = (unsigned short*)e_aux + off;
              *(ushort8_t*)a = pack8<DT>(side);
              if (hi_ok) *(ushort8_t*)(a + 8) = pack8<DT>(side + 8);
            }
            if constexpr (EPI == 2 || EPI == 3) {
              if (q->stats) {
                float vr[16];
                unpack8<DT>(o0, vr);
                unpack8<DT>(o1, vr + 8);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                  const float z = (r < 8 || hi_ok) ? vr[r] : 0.f;
                  st[j][r] += z;
                  if constexpr (EPI == 3) st2[j][r] += z * z;
                }
              }
            }
          }
        }
      }
    });
    if constexpr (EPI == 2 || EPI == 3) {
      if (q->stats) {
        // the 32 row lanes of a column group reduce-scatter their 16 sums: after the xor-16 / 8 / 4 / 2 exchanges a lane keeps
        // ONE column (index = bits 4..1 of fr), the xor-1 exchange completes it; even lanes store.  One partial row per
        // (tile row, wavefront row group), folded in a fixed order by colsum_fold_kernel (EPI 2: [row][N]) or by
        // bn_stats_from_partials (EPI 3: [row][2][N], sums then sums of squares).
        auto reduce_store = [&](float (&sx)[16], int j, int which) __attribute__((always_inline)) {
          float a8[8], a4[4], a2[2], a1;
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const bool up = (fr & 16) != 0;
            const float keep = up ? sx[8 + r] : sx[r], send = up ? sx[r] : sx[8 + r];
            a8[r] = keep + __shfl_xor(send, 16, 64);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool up = (fr & 8) != 0;
            const float keep = up ? a8[4 + r] : a8[r], send = up ? a8[r] : a8[4 + r];
            a4[r] = keep + __shfl_xor(send, 8, 64);
          }
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const bool up = (fr & 4) != 0;
            const float keep = up ? a4[2 + r] : a4[r], send = up ? a4[r] : a4[2 + r];
            a2[r] = keep + __shfl_xor(send, 4, 64);
          }
          {
            const bool up = (fr & 2) != 0;
            const float keep = up ? a2[1] : a2[0], send = up ? a2[0] : a2[1];
            a1 = keep + __shfl_xor(send, 2, 64);
          }
          a1 += __shfl_xor(a1, 1, 64);
          const int col = n0 + j * 128 + wc * 32 + fh * 16 + ((fr >> 1) & 15);
          const long long prow = (long long)((m0 >> 8) * 2 + wr);
          if ((fr & 1) == 0 && col < p.N) {
            if constexpr (EPI == 3) q->stats[(prow * 2 + which) * p.N + col] = a1;
            else q->stats[prow * p.N + col] = a1;
          }
        };
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          reduce_store(st[j], j, 0);
          if constexpr (EPI == 3) reduce_store(st2[j], j, 1);
        }
      }
    }
    if (wr == 1) G8_BARRIER();          // (the other half of the epilogue's barrier pair)
#ifdef G8_TIMING
    G8_STAMP(2);
    ++item_seq;
#endif
  }
  if (wr == 0) G8_BARRIER();            // pairs with group 1's last barrier
  G8_VM(0);                             // the idle pieces of the stream's tail
}


#define G8_MAX_DEVICES 64
// current device index, or -1 when it cannot be cached per device (the launcher then declines: gemm_dma.hip takes the shape)
static inline int g8_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= G8_MAX_DEVICES) return -1;
  return dev;
}

template <int DT, int AM, int BMD, int EPI, int ACT>
static void g8_launch(const Gemm8Args& p, int grid, hipStream_t stream) {
  // hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device only: one flag per device
  static bool attr_set[G8_MAX_DEVICES];
  const int dev = g8_current_device();
  if (dev >= 0 && !attr_set[dev]) {
    (void)hipFuncSetAttribute((const void*)gemm8_kernel<DT, AM, BMD, EPI, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS_BYTES);
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL((gemm8_kernel<DT, AM, BMD, EPI, ACT>), dim3(grid), dim3(512), G8_LDS_BYTES, stream, p);
}
