// Fused self-attention for BERT pre-training on gfx950 (64-wide heads; S = 128 keys: one kernel forward, one backward; longer
// sequences -- phase 2, S = 512 -- stream K / V in 128-key blocks: second half of this file).
//
// Replaces BertSelfAttention.forward between the QKV projection and the output projection
// (LanguageModeling/BERT/modeling.py:340-384):
//     scores = bmm(q, k^T) / sqrt(d) + mask;  probs = dropout(softmax(scores));  context = bmm(probs, v)
// and its autograd backward.  The reference materialises scores / probs / dropped probs as [B, 16, 128, 128] tensors
// (three HBM round trips forward, four backward); here nothing of that shape leaves the compute unit:
//
//  * one workgroup (4 wavefronts) per (sequence, head); the Q, K, V (and dO) slices of the [T, 3H] QKV activation go
//    HBM -> LDS once by LDS-DMA as [128][64] images whose 16-byte chunks are XOR-swizzled on the SOURCE address with a
//    function of the row that serves BOTH read shapes without bank conflicts: row fragments (ds_read_b128: Q, K as
//    contraction-over-d operands) and column fragments (ds_read_b64_tr_b16: V, K, Q, dO as contraction-over-row operands);
//  * a wavefront owns 32 query rows.  It computes S^T = K Q^T (v_mfma_f32_32x32x16, operands swapped), so a LANE holds
//    one query row's 64 scores (the other 64 sit in lane ^ 32): row max / sum / the softmax-gradient dot product are
//    in-register reductions plus one cross-half shuffle.  P^T in that accumulator layout is already the B operand of
//    the next contraction (context^T = V^T P^T, dQ^T = K^T dS^T) once the V / K fragments are read with the matching
//    key permutation -- no LDS round trip, no lane permutes;
//  * dropout is the counter-based Philox of dropout.h on the chunk index of the (never stored) [B, heads, S, S] tensor:
//    the backward pass regenerates the identical mask; softmax is recomputed from the saved row max and 1 / row sum,
//    so forward and backward use bit-identical probabilities;
//  * dK and dV contract over the query rows, which are spread over the wavefronts: dS^T and dropout(P)^T go through a
//    [query][key] LDS image (over the K | V tiles, dead by then) and every wavefront produces 32 key rows of dK, dV;
//  * outputs leave through LDS as whole 128-byte rows (16-byte stores).
// Softmax statistics and accumulation are fp32; probabilities enter the MFMAs rounded to the activation dtype after the
// dropout scaling (the reference's autocast does the same: softmax and dropout run in fp32, bmm casts its operands).
#include "gemm_tiles.h"
#include "dropout.h"

#define AT_S 128
#define AT_D 64
#define AT_TILE (AT_S * AT_D)        // elements of one [128][64] image

struct AttnArgs {
  const unsigned short* qkv;    // [T, 3H]: q | k | v, head h at columns h*64 of each third
  const unsigned short* dctx;   // [T, H]   (backward)
  const float* mask_add;        // [B, S] additive mask (0 / -10000) or NULL
  unsigned short* ctx;          // [T, H]   (forward out)
  unsigned short* dqkv;         // [T, 3H]  (backward out)
  float* stats;                 // [B * heads, S, 2]: row max of the scaled+masked scores, 1 / sum of exp
  unsigned char* mask_out;      // optional bit-packed keep mask [B * heads * S * S / 8], forward only
  const unsigned char* mask_in; // optional (backward, S = 128): the keep mask the forward pass wrote -- read instead of re-drawn
  float* colsum;                // optional (backward): [B, 3H] column sums of this sequence's rows of dqkv (QKV bias gradient partials)
  int B, nh, H;
  float scale;
  DropArgs drop;                // thr == 0: no dropout
  int S, nblk, sw;              // long sequences (S = 128 nblk > 128): sequence length, 128-row blocks, floats per stats row (4)
};

// 16-byte chunk swizzle of a [rows][64] image: the bits of (row >> 1) & 7 rotated so that rows r and r + 2 (the same
// bank half of a 128-byte row) differ in chunk bit 2.  Row fragments: the 16-lane groups of ds_read_b128 see every
// value once per bank half.  Column fragments: the four k lines of a transpose read use disjoint 4-chunk sets.
__device__ __forceinline__ int at_swz(int row) {
  const int s = (row >> 1) & 7;
  return ((s & 1) << 2) | (s >> 1);
}
// chunk swizzle of the [128 query lines][128 keys] images (256-byte lines, 16 chunks)
__device__ __forceinline__ int at_iswz(int line) { return ((line & 3) << 2) | ((line >> 2) & 3); }

// DMA one [128][64] tile (rows of a matrix with row stride ld_bytes) into a lane-linear LDS image: 16 pieces of 1 KiB
__device__ __forceinline__ void at_load_tile(__amdgpu_buffer_rsrc_t rs, unsigned base_bytes, unsigned ld_bytes,
                                             unsigned short* tile, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int pc = wave * 4 + j;
    const int row = pc * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ at_swz(row);
    dma16(rs, tile + pc * 512, base_bytes + (unsigned)row * ld_bytes + (unsigned)chunk * 16u);
  }
}

// fragment of 32 rows x 16 columns (lane: row row0 + (lane & 31), columns 16 ks + 8 (lane >> 5) .. + 7)
__device__ __forceinline__ ushort8_t at_frag_rows(const unsigned short* t, int row0, int ks, int lane) {
  const int row = row0 + (lane & 31);
  return *(const ushort8_t*)(t + row * AT_D + (((ks * 2 + (lane >> 5)) ^ at_swz(row)) << 3));
}

// fragment of 32 COLUMNS x 16 rows of the image (lane: column col0 + (lane & 31); its 8 values run over image rows).
// PERM = false: rows 16 ks + 8 hf + e.   PERM = true: rows 16 ks + 4 hf + (e & 3) + 8 (e >> 2)  (hf = lane >> 5) --
// the order in which a 32x32 accumulator holds its rows, so that accumulator registers pack straight into the other
// operand of the contraction.
template <bool PERM>
__device__ __forceinline__ ushort8_t at_frag_cols(const unsigned short* t, int col0, int ks, int lane) {
  const int tg = lane >> 4, ti = lane & 15, hf = tg >> 1;
  const int chunk = ((col0 + ((tg & 1) << 4)) >> 3) + ((ti & 3) >> 1);
  ushort8_t f;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int line = ks * 16 + (PERM ? hf * 4 + h * 8 : hf * 8 + h * 4) + (ti >> 2);
    const short4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) short4_t*)(t + line * AT_D + ((chunk ^ at_swz(line)) << 3) + ((ti & 1) << 2)));
#pragma unroll
    for (int e = 0; e < 4; ++e) f[h * 4 + e] = (unsigned short)v[e];
  }
  return f;
}

// the same out of a [128 lines][128 columns] image (rows = query lines 16 ks + 8 hf + e)
__device__ __forceinline__ ushort8_t at_img_frag_cols(const unsigned short* t, int col0, int ks, int lane) {
  const int tg = lane >> 4, ti = lane & 15, hf = tg >> 1;
  const int chunk = ((col0 + ((tg & 1) << 4)) >> 3) + ((ti & 3) >> 1);
  ushort8_t f;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int line = ks * 16 + hf * 8 + h * 4 + (ti >> 2);
    const short4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) short4_t*)(t + line * AT_S + ((chunk ^ at_iswz(line)) << 3) + ((ti & 1) << 2)));
#pragma unroll
    for (int e = 0; e < 4; ++e) f[h * 4 + e] = (unsigned short)v[e];
  }
  return f;
}

// Keep bits of this lane's 64 elements of query row chunk0 / 16: keepw[kb] bit (4 rq + j) <-> key kb*32 + 8 rq + 4 hf + j,
// i.e. element 4 hf + j of chunk chunk0 + 4 kb + rq.  One Philox call yields the 8 decisions of a chunk and the two lanes
// of a row (hf = 0 / 1) need 4 each: a lane runs the generator for the chunks c with c % 2 == hf only and trades the other
// half of its words with lane ^ 32 -- the kernels were bound by the 32-bit multiplies of 16 calls per lane.
__device__ __forceinline__ unsigned at_bits4(unsigned w0, unsigned w1, unsigned thr) {
  return ((w0 & 0xffffu) >= thr ? 1u : 0u) | ((w0 >> 16) >= thr ? 2u : 0u) | ((w1 & 0xffffu) >= thr ? 4u : 0u) |
         ((w1 >> 16) >= thr ? 8u : 0u);
}
__device__ __forceinline__ void at_keep_row(const DropArgs& d, unsigned chunk0, int hf, unsigned* keepw) {
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    unsigned bits = 0;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {                       // chunk pair (rq = 2 pr, 2 pr + 1) of this key block
      const int rq_mine = 2 * pr + hf, rq_other = 2 * pr + (hf ^ 1);
      const uint4_t r = philox4x32_10((uint4_t){chunk0 + (unsigned)(kb * 4 + rq_mine), 0u, d.off_lo, d.off_hi}, d.seed_lo, d.seed_hi);
      const unsigned mine0 = hf ? r[2] : r[0], mine1 = hf ? r[3] : r[1];          // elements 4 hf .. + 3 of my chunk
      const unsigned give0 = hf ? r[0] : r[2], give1 = hf ? r[1] : r[3];          // the partner's elements of my chunk
      const unsigned got0 = (unsigned)__shfl_xor((int)give0, 32, 64), got1 = (unsigned)__shfl_xor((int)give1, 32, 64);
      bits |= at_bits4(mine0, mine1, d.thr) << (4 * rq_mine);
      bits |= at_bits4(got0, got1, d.thr) << (4 * rq_other);
    }
    keepw[kb] = bits;
  }
}

// Accumulator block [32 values of the register-indexed dimension x 32 lanes] -> 8-byte LDS writes of a wave-private
// [32 rows][64 columns] staging tile (row = lane & 31, columns cb * 32 + 8 (r >> 2) + 4 hf + (r & 3)), then whole rows out.
template <int DT>
__device__ __forceinline__ void at_stage_block(unsigned short* stg, const float16_t& acc, int cb, int lane) {
  const int row = lane & 31, hf = lane >> 5;
#pragma unroll
  for (int rq = 0; rq < 4; ++rq) {
    const float2_t lo = {acc[rq * 4 + 0], acc[rq * 4 + 1]}, hi = {acc[rq * 4 + 2], acc[rq * 4 + 3]};
    uint2_t w;
    if (DT == DLE_F16) {
      w[0] = (unsigned)Elem<DLE_F16>::from_f32(lo[0]) | ((unsigned)Elem<DLE_F16>::from_f32(lo[1]) << 16);
      w[1] = (unsigned)Elem<DLE_F16>::from_f32(hi[0]) | ((unsigned)Elem<DLE_F16>::from_f32(hi[1]) << 16);
    } else {
      w[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2_t));
      w[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2_t));
    }
    const int chunk = cb * 4 + rq;
    *(uint2_t*)(stg + row * AT_D + ((chunk ^ (row & 7)) << 3) + (hf << 2)) = w;
  }
}
__device__ __forceinline__ void at_store_rows(const unsigned short* stg, unsigned short* out, long long ld, int lane) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = it * 8 + (lane >> 3), c = lane & 7;
    const ushort8_t v = *(const ushort8_t*)(stg + row * AT_D + ((c ^ (row & 7)) << 3));
    *(ushort8_t*)(out + row * ld + c * 8) = v;
  }
}

// scores of this lane's query row -> probabilities (fp32), given (or producing) the row statistics
// s[kb][r] <-> key kb * 32 + (r & 3) + 8 (r >> 2) + 4 hf
template <bool HAVE_STATS>
__device__ __forceinline__ void at_softmax(float16_t* s, const float* mrow, float scale, int hf, float& mx, float& inv) {
#pragma unroll
  for (int kb = 0; kb < 4; ++kb)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      float4_t m = {0.f, 0.f, 0.f, 0.f};
      if (mrow) m = *(const float4_t*)(mrow + kb * 32 + rq * 8 + hf * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) s[kb][rq * 4 + j] = s[kb][rq * 4 + j] * scale + m[j];
    }
  if (!HAVE_STATS) {
    float v = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) v = fmaxf(v, s[kb][r]);
    mx = fmaxf(v, __shfl_xor(v, 32, 64));
  }
  float sum = 0.f;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[kb][r] = __expf(s[kb][r] - mx);
      sum += s[kb][r];
    }
  if (!HAVE_STATS) {
    sum += __shfl_xor(sum, 32, 64);
    inv = 1.0f / sum;
  }
#pragma unroll
  for (int kb = 0; kb < 4; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[kb][r] *= inv;
}

template <int DT>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs p) {
  drop_resolve(p.drop);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* Qt = (unsigned short*)smem_raw;
  unsigned short* Kt = Qt + AT_TILE;
  unsigned short* Vt = Kt + AT_TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5;
  const int bh = blockIdx.x, b = bh / p.nh, h = bh - b * p.nh;
  const unsigned ld = (unsigned)p.H * 3u * 2u;
  const unsigned base = (unsigned)(((long long)b * AT_S * 3 * p.H + h * AT_D) * 2);
  {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.qkv, 0, 0xFFFFFFE0, 0x00020000);
    at_load_tile(rs, base, ld, Qt, wave, lane);
    at_load_tile(rs, base + (unsigned)p.H * 2u, ld, Kt, wave, lane);
    at_load_tile(rs, base + (unsigned)p.H * 4u, ld, Vt, wave, lane);
  }
  const int q = wave * 32 + (lane & 31);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();

  // S^T[key][q] = sum_d K[key][d] Q[q][d]
  float16_t s[4];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const ushort8_t bq = at_frag_rows(Qt, wave * 32, ks, lane);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) s[kb] = Mfma32x16<DT>::run(at_frag_rows(Kt, kb * 32, ks, lane), bq, s[kb]);
  }
  float mx, inv;
  at_softmax<false>(s, p.mask_add ? p.mask_add + (long long)b * AT_S : nullptr, p.scale, hf, mx, inv);
  if (hf == 0) *(float2_t*)(p.stats + ((long long)bh * AT_S + q) * 2) = (float2_t){mx, inv};

  // dropout(P), rounded: the B operand of context^T = V^T P^T
  ushort8_t pd[8];
  const bool drop = p.drop.thr != 0;
  const unsigned chunk0 = ((unsigned)bh * AT_S + (unsigned)q) * 16u;
  unsigned keepw[4] = {0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu};
  if (drop) at_keep_row(p.drop, chunk0, hf, keepw);
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    if (drop) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = ((keepw[kb] >> r) & 1u) ? s[kb][r] * p.drop.inv_keep : 0.f;
    }
    if (p.mask_out && drop) {
      // byte of chunk (kb, rq): elements 0-3 from the hf = 0 lane, 4-7 from the hf = 1 lane
      unsigned mbits = 0;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) mbits |= ((keepw[kb] >> (4 * rq)) & 0xFu) << (8 * rq);
      const unsigned other = __shfl_xor(mbits, 32, 64);
      if (hf == 0) *(unsigned*)(p.mask_out + chunk0 + kb * 4) = mbits | (other << 4);
    }
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = s[kb][r];
    pd[kb * 2] = pack8<DT>(v);
    pd[kb * 2 + 1] = pack8<DT>(v + 8);
  }

  // context^T[d][q] = sum_key V[key][d] P[q][key]
  float16_t o[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int db = 0; db < 2; ++db) o[db] = Mfma32x16<DT>::run(at_frag_cols<true>(Vt, db * 32, j, lane), pd[j], o[db]);

  // rows of this wavefront leave through its own (no longer read) Q rows
  unsigned short* stg = Qt + wave * 32 * AT_D;
  at_stage_block<DT>(stg, o[0], 0, lane);
  at_stage_block<DT>(stg, o[1], 1, lane);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  at_store_rows(stg, p.ctx + ((long long)b * AT_S + wave * 32) * p.H + h * AT_D, p.H, lane);
}

template <int DT>
__global__ __launch_bounds__(256, 2) void attn_bwd_kernel(AttnArgs p) {
  drop_resolve(p.drop);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* Qt = (unsigned short*)smem_raw;
  unsigned short* Ot = Qt + AT_TILE;
  unsigned short* Kt = Ot + AT_TILE;
  unsigned short* Vt = Kt + AT_TILE;
  unsigned short* img = Kt;                                    // [128 q][128 keys] over K | V once they are dead
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5;
  const int bh = blockIdx.x, b = bh / p.nh, h = bh - b * p.nh;
  const unsigned ld = (unsigned)p.H * 3u * 2u;
  const unsigned base = (unsigned)(((long long)b * AT_S * 3 * p.H + h * AT_D) * 2);
  {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.qkv, 0, 0xFFFFFFE0, 0x00020000);
    at_load_tile(rs, base, ld, Qt, wave, lane);
    at_load_tile(rs, base + (unsigned)p.H * 2u, ld, Kt, wave, lane);
    at_load_tile(rs, base + (unsigned)p.H * 4u, ld, Vt, wave, lane);
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)p.dctx, 0, 0xFFFFFFE0, 0x00020000);
    at_load_tile(ro, (unsigned)(((long long)b * AT_S * p.H + h * AT_D) * 2), (unsigned)p.H * 2u, Ot, wave, lane);
  }
  const int q = wave * 32 + (lane & 31);
  const float2_t st = *(const float2_t*)(p.stats + ((long long)bh * AT_S + q) * 2);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();

  float16_t s[4], g[4];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[kb][r] = 0.f; g[kb][r] = 0.f; }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {                             // S^T = K Q^T
    const ushort8_t bq = at_frag_rows(Qt, wave * 32, ks, lane);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) s[kb] = Mfma32x16<DT>::run(at_frag_rows(Kt, kb * 32, ks, lane), bq, s[kb]);
  }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {                             // d(dropout(P))^T = V dO^T
    const ushort8_t bo = at_frag_rows(Ot, wave * 32, ks, lane);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) g[kb] = Mfma32x16<DT>::run(at_frag_rows(Vt, kb * 32, ks, lane), bo, g[kb]);
  }
  float mx = st[0], inv = st[1];
  at_softmax<true>(s, p.mask_add ? p.mask_add + (long long)b * AT_S : nullptr, p.scale, hf, mx, inv);

  // dropout backward on g, delta = sum_key P dP; then dS = P (dP - delta) scale and dropout(P), both rounded
  const bool drop = p.drop.thr != 0;
  const unsigned chunk0 = ((unsigned)bh * AT_S + (unsigned)q) * 16u;
  unsigned keepw[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  if (drop && p.mask_in) {
    // the forward pass left the keep bits (1 bit per probability): 16 bytes per query row instead of 8 Philox calls per lane --
    // byte rq of word kb holds chunk (kb, rq): low nibble = the hf = 0 lane's four elements, high nibble = the hf = 1 lane's
    const uint4_t mw = *(const uint4_t*)(p.mask_in + chunk0);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      unsigned bits = 0;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) bits |= ((mw[kb] >> (8 * rq + 4 * hf)) & 0xFu) << (4 * rq);
      keepw[kb] = bits;
    }
  } else if (drop) at_keep_row(p.drop, chunk0, hf, keepw);
  float delta = 0.f;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float dp = ((keepw[kb] >> r) & 1u) ? g[kb][r] * p.drop.inv_keep : 0.f;
      g[kb][r] = dp;
      delta += s[kb][r] * dp;
    }
  }
  delta += __shfl_xor(delta, 32, 64);
  ushort8_t ds[8], pd[8];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    float v[16], w[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      v[r] = s[kb][r] * (g[kb][r] - delta) * p.scale;
      w[r] = ((keepw[kb] >> r) & 1u) ? s[kb][r] * p.drop.inv_keep : 0.f;
    }
    ds[kb * 2] = pack8<DT>(v);
    ds[kb * 2 + 1] = pack8<DT>(v + 8);
    pd[kb * 2] = pack8<DT>(w);
    pd[kb * 2 + 1] = pack8<DT>(w + 8);
  }

  // dQ^T[d][q] = sum_key K[key][d] dS[q][key]
  float16_t dq[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int db = 0; db < 2; ++db) dq[db] = Mfma32x16<DT>::run(at_frag_cols<true>(Kt, db * 32, j, lane), ds[j], dq[db]);

  // [q][key] image of a packed accumulator set: this lane's line q, keys kb * 32 + 8 rq + 4 hf .. + 3
  auto write_img = [&](const ushort8_t* x) __attribute__((always_inline)) {
    const int isw = at_iswz(q);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const uint4_t u = __builtin_bit_cast(uint4_t, x[kb * 2 + (rq >> 1)]);
        const uint2_t w = {u[(rq & 1) * 2], u[(rq & 1) * 2 + 1]};
        *(uint2_t*)(img + q * AT_S + (((kb * 4 + rq) ^ isw) << 3) + (hf << 2)) = w;
      }
  };
  float16_t dk[2], dv[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }

  __syncthreads();                                             // K and V tiles are dead
  write_img(ds);
  __syncthreads();
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {                             // dK^T[d][key] = sum_q Q[q][d] dS[q][key]
    const ushort8_t bs = at_img_frag_cols(img, wave * 32, ks, lane);
#pragma unroll
    for (int db = 0; db < 2; ++db) dk[db] = Mfma32x16<DT>::run(at_frag_cols<false>(Qt, db * 32, ks, lane), bs, dk[db]);
  }
  __syncthreads();
  write_img(pd);
  __syncthreads();
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {                             // dV^T[d][key] = sum_q dO[q][d] dropout(P)[q][key]
    const ushort8_t bp = at_img_frag_cols(img, wave * 32, ks, lane);
#pragma unroll
    for (int db = 0; db < 2; ++db) dv[db] = Mfma32x16<DT>::run(at_frag_cols<false>(Ot, db * 32, ks, lane), bp, dv[db]);
  }
  __syncthreads();                                             // every tile / image read is done: LDS becomes staging
  unsigned short* stg = (unsigned short*)smem_raw + wave * (3 * 32 * AT_D);
  at_stage_block<DT>(stg, dq[0], 0, lane);
  at_stage_block<DT>(stg, dq[1], 1, lane);
  at_stage_block<DT>(stg + 32 * AT_D, dk[0], 0, lane);
  at_stage_block<DT>(stg + 32 * AT_D, dk[1], 1, lane);
  at_stage_block<DT>(stg + 64 * AT_D, dv[0], 0, lane);
  at_stage_block<DT>(stg + 64 * AT_D, dv[1], 1, lane);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  unsigned short* out = p.dqkv + ((long long)b * AT_S + wave * 32) * 3 * p.H + h * AT_D;
  at_store_rows(stg, out, 3LL * p.H, lane);
  at_store_rows(stg + 32 * AT_D, out + p.H, 3LL * p.H, lane);
  at_store_rows(stg + 64 * AT_D, out + 2 * p.H, 3LL * p.H, lane);
  if (p.colsum) {
    // bias gradient of the QKV projection = column sums of dqkv: the 128 rows of this (sequence, head) are summed here from
    // the ROUNDED staged values (what a column-sum pass over dqkv would read); the host folds the B per-sequence rows
    float cs[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      const unsigned short* t = stg + o * 32 * AT_D;
      float a = 0.f;
#pragma unroll 8
      for (int row = 0; row < 32; ++row)
        a += Elem<DT>::to_f32(t[row * AT_D + ((((lane >> 3) ^ (row & 7)) << 3) | (lane & 7))]);
      cs[o] = a;
    }
    __syncthreads();                                           // every wave is done reading its staging tiles
    float* red = (float*)smem_raw;                             // [4 waves][3][64]
#pragma unroll
    for (int o = 0; o < 3; ++o) red[(wave * 3 + o) * AT_D + lane] = cs[o];
    __syncthreads();
    if (tid < 3 * AT_D) {
      const int o = tid / AT_D, c = tid - o * AT_D;
      const float v = (red[(0 * 3 + o) * AT_D + c] + red[(1 * 3 + o) * AT_D + c]) + (red[(2 * 3 + o) * AT_D + c] + red[(3 * 3 + o) * AT_D + c]);
      p.colsum[(long long)b * 3 * p.H + o * p.H + h * AT_D + c] = v;
    }
  }
}


// 1 when (S, head_dim) is inside the fused kernels' envelope (the caller otherwise uses the batched-GEMM path): 64-wide heads,
// S = 128 (one kernel each way) or a multiple of 128 up to 1024 (the key-block-streaming kernels below)
extern "C" int dle_attention_supported(int S, int head_dim) {
  return (head_dim == AT_D && S >= AT_S && S <= 1024 && (S % AT_S) == 0) ? 1 : 0;
}
// floats per row of the statistics buffer dle_attention_fwd writes / dle_attention_bwd reads: 2 (max, 1 / sum) at S = 128,
// 4 (max, 1 / sum, delta scratch of the backward pass, unused) for the long-sequence kernels
extern "C" int dle_attention_stats_floats(int S) { return S == AT_S ? 2 : 4; }

// ---------------------------------------------------------------------------------------------------------------------------
// Long sequences (S = 128 nblk, nblk = 2 .. 8: BERT phase 2 runs S = 512, run_pretraining.py --phase2; modeling.py:340-384 is
// the same module).  The [B, heads, S, S] tensors still never exist: K / V are streamed through LDS in 128-key blocks.
//   forward:  one workgroup per (sequence, head, 128-query block).  Pass 1 walks the key blocks with the ONLINE max / sum of the
//             softmax row (m, l updated per block) and leaves the row statistics (max, 1 / sum); pass 2 walks them again with the
//             FINAL statistics: probabilities, dropout, rounding and the P V contraction are then exactly those of the S = 128
//             kernel (and of the backward pass, which recomputes P from the same two numbers) -- QK^T is computed twice, 1.6 %
//             of a BERT-Large layer's forward flops at S = 512.
//   backward: two kernels, no atomics.  attn_bwd_dq_long (one workgroup per query block): first the row's delta = sum_k P dP over
//             ALL key blocks (stored beside the statistics), then dQ = sum_j dS_j K_j.  attn_bwd_dkv_long (one workgroup per key
//             block, K_j / V_j resident): walks the query blocks, dK_j += dS^T Q_i, dV_j += dropout(P)^T dO_i through the same
//             [query][key] LDS image as the S = 128 kernel.
// Statistics rows are 4 floats here: max, 1 / sum, delta, unused.  Dropout: the Philox counter is the 8-element chunk index of
// the never-stored [B, heads, S, S] tensor, as in the unfused path (dle_softmax_dropout_fwd): identical masks.
__device__ __forceinline__ void at_block_scores_scaled(float16_t* s, const float* mrow, float scale, int hf) {
#pragma unroll
  for (int kb = 0; kb < 4; ++kb)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      float4_t m = {0.f, 0.f, 0.f, 0.f};
      if (mrow) m = *(const float4_t*)(mrow + kb * 32 + rq * 8 + hf * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) s[kb][rq * 4 + j] = s[kb][rq * 4 + j] * scale + m[j];
    }
}

template <int DT>
__global__ __launch_bounds__(256, 2) void attn_fwd_long_kernel(AttnArgs p) {
  drop_resolve(p.drop);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* Qt = (unsigned short*)smem_raw;
  unsigned short* Kt = Qt + AT_TILE;
  unsigned short* Vt = Kt + AT_TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5;
  const int qb = blockIdx.x % p.nblk, bh = blockIdx.x / p.nblk, b = bh / p.nh, h = bh - b * p.nh;
  const unsigned ld = (unsigned)p.H * 3u * 2u;
  const unsigned seq_base = (unsigned)(((long long)b * p.S * 3 * p.H + h * AT_D) * 2);          // row 0 of the sequence, q columns
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.qkv, 0, 0xFFFFFFE0, 0x00020000);
  at_load_tile(rs, seq_base + (unsigned)(qb * AT_S) * ld, ld, Qt, wave, lane);
  const int ql = wave * 32 + (lane & 31), qg = qb * AT_S + ql;                                 // query row inside the block / sequence
  const float* mseq = p.mask_add ? p.mask_add + (long long)b * p.S : nullptr;
  auto scores = [&](float16_t* s) __attribute__((always_inline)) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const ushort8_t bq = at_frag_rows(Qt, wave * 32, ks, lane);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) s[kb] = Mfma32x16<DT>::run(at_frag_rows(Kt, kb * 32, ks, lane), bq, s[kb]);
    }
  };
  // ---- pass 1: online row max / sum over the key blocks
  float m_run = -INFINITY, l_run = 0.f;
  for (int j = 0; j < p.nblk; ++j) {
    if (j > 0) __syncthreads();                             // every wave is done with the previous K block
    at_load_tile(rs, seq_base + (unsigned)p.H * 2u + (unsigned)(j * AT_S) * ld, ld, Kt, wave, lane);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    float16_t s[4];
    scores(s);
    at_block_scores_scaled(s, mseq ? mseq + j * AT_S : nullptr, p.scale, hf);
    float v = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) v = fmaxf(v, s[kb][r]);
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    const float m_new = fmaxf(m_run, v);
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += __expf(s[kb][r] - m_new);
    sum += __shfl_xor(sum, 32, 64);
    l_run = l_run * __expf(m_run - m_new) + sum;
    m_run = m_new;
  }
  const float mx = m_run, inv = 1.0f / l_run;
  if (hf == 0) *(float2_t*)(p.stats + ((long long)bh * p.S + qg) * p.sw) = (float2_t){mx, inv};
  // ---- pass 2: probabilities with the final statistics, dropout, context
  float16_t o[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  const bool drop = p.drop.thr != 0;
  const unsigned row_chunk0 = ((unsigned)bh * (unsigned)p.S + (unsigned)qg) * (unsigned)(p.S >> 3);
  for (int j = 0; j < p.nblk; ++j) {
    __syncthreads();
    at_load_tile(rs, seq_base + (unsigned)p.H * 2u + (unsigned)(j * AT_S) * ld, ld, Kt, wave, lane);
    at_load_tile(rs, seq_base + (unsigned)p.H * 4u + (unsigned)(j * AT_S) * ld, ld, Vt, wave, lane);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    float16_t s[4];
    scores(s);
    float mxc = mx, invc = inv;
    at_softmax<true>(s, mseq ? mseq + j * AT_S : nullptr, p.scale, hf, mxc, invc);
    const unsigned chunk0 = row_chunk0 + (unsigned)(j * 16);
    unsigned keepw[4] = {0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu};
    if (drop) at_keep_row(p.drop, chunk0, hf, keepw);
    ushort8_t pd[8];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      if (drop) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = ((keepw[kb] >> r) & 1u) ? s[kb][r] * p.drop.inv_keep : 0.f;
      }
      if (p.mask_out && drop) {
        unsigned mbits = 0;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) mbits |= ((keepw[kb] >> (4 * rq)) & 0xFu) << (8 * rq);
        const unsigned other = __shfl_xor(mbits, 32, 64);
        if (hf == 0) *(unsigned*)(p.mask_out + chunk0 + kb * 4) = mbits | (other << 4);
      }
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = s[kb][r];
      pd[kb * 2] = pack8<DT>(v);
      pd[kb * 2 + 1] = pack8<DT>(v + 8);
    }
#pragma unroll
    for (int jj = 0; jj < 8; ++jj)
#pragma unroll
      for (int db = 0; db < 2; ++db) o[db] = Mfma32x16<DT>::run(at_frag_cols<true>(Vt, db * 32, jj, lane), pd[jj], o[db]);
  }
  __syncthreads();                                          // (a faster wave must not stage over K / V rows others still read: Q only)
  unsigned short* stg = Qt + wave * 32 * AT_D;
  at_stage_block<DT>(stg, o[0], 0, lane);
  at_stage_block<DT>(stg, o[1], 1, lane);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  at_store_rows(stg, p.ctx + ((long long)b * p.S + qb * AT_S + wave * 32) * p.H + h * AT_D, p.H, lane);
}

// per-(block, part) column sums of the rows a wavefront staged, folded over the 4 wavefronts: colsum[(b * nblk + blk), part * H + h * 64 + c]
template <int DT, int NPART>
__device__ __forceinline__ void at_long_colsum(const AttnArgs& p, const unsigned short* stg, unsigned char* smem_raw, int b, int blk,
                                               int h, int part0, int tid, int wave, int lane) {
  float cs[NPART];
#pragma unroll
  for (int o = 0; o < NPART; ++o) {
    const unsigned short* t = stg + o * 32 * AT_D;
    float a = 0.f;
#pragma unroll 8
    for (int row = 0; row < 32; ++row)
      a += Elem<DT>::to_f32(t[row * AT_D + ((((lane >> 3) ^ (row & 7)) << 3) | (lane & 7))]);
    cs[o] = a;
  }
  __syncthreads();
  float* red = (float*)smem_raw;                             // [4 waves][NPART][64]
#pragma unroll
  for (int o = 0; o < NPART; ++o) red[(wave * NPART + o) * AT_D + lane] = cs[o];
  __syncthreads();
  if (tid < NPART * AT_D) {
    const int o = tid / AT_D, c = tid - o * AT_D;
    const float v = (red[(0 * NPART + o) * AT_D + c] + red[(1 * NPART + o) * AT_D + c]) +
                    (red[(2 * NPART + o) * AT_D + c] + red[(3 * NPART + o) * AT_D + c]);
    p.colsum[((long long)b * p.nblk + blk) * 3 * p.H + (part0 + o) * p.H + h * AT_D + c] = v;
  }
}

template <int DT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_long_kernel(AttnArgs p) {
  drop_resolve(p.drop);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* Qt = (unsigned short*)smem_raw;
  unsigned short* Ot = Qt + AT_TILE;
  unsigned short* Kt = Ot + AT_TILE;
  unsigned short* Vt = Kt + AT_TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5;
  const int qb = blockIdx.x % p.nblk, bh = blockIdx.x / p.nblk, b = bh / p.nh, h = bh - b * p.nh;
  const unsigned ld = (unsigned)p.H * 3u * 2u;
  const unsigned seq_base = (unsigned)(((long long)b * p.S * 3 * p.H + h * AT_D) * 2);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.qkv, 0, 0xFFFFFFE0, 0x00020000);
  __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)p.dctx, 0, 0xFFFFFFE0, 0x00020000);
  at_load_tile(rs, seq_base + (unsigned)(qb * AT_S) * ld, ld, Qt, wave, lane);
  at_load_tile(ro, (unsigned)((((long long)b * p.S + qb * AT_S) * p.H + h * AT_D) * 2), (unsigned)p.H * 2u, Ot, wave, lane);
  const int ql = wave * 32 + (lane & 31), qg = qb * AT_S + ql;
  float* strow = p.stats + ((long long)bh * p.S + qg) * p.sw;
  const float2_t st = *(const float2_t*)strow;
  const float* mseq = p.mask_add ? p.mask_add + (long long)b * p.S : nullptr;
  const bool drop = p.drop.thr != 0;
  const unsigned row_chunk0 = ((unsigned)bh * (unsigned)p.S + (unsigned)qg) * (unsigned)(p.S >> 3);
  auto load_kv = [&](int j) __attribute__((always_inline)) {
    at_load_tile(rs, seq_base + (unsigned)p.H * 2u + (unsigned)(j * AT_S) * ld, ld, Kt, wave, lane);
    at_load_tile(rs, seq_base + (unsigned)p.H * 4u + (unsigned)(j * AT_S) * ld, ld, Vt, wave, lane);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
  };
  // the query-side fragments (this wavefront's 32 rows of Q_i and dO_i) are the same for every key block: read once
  ushort8_t bq[4], bo[4];
  bool frags_loaded = false;
  // P (fp32, final statistics) and dP of the 32 keys kb of the current key block for this lane's query row: element r of the
  // accumulators <-> key kb * 32 + (r & 3) + 8 (r >> 2) + 4 hf (one 32-key slice at a time: 32 live accumulator registers
  // instead of the 128 of a whole block; the statistics are final, so nothing crosses slices)
  auto slice = [&](int kb, const float* mrow, unsigned keep16, float16_t& sp, float16_t& gp) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { sp[r] = 0.f; gp[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) sp = Mfma32x16<DT>::run(at_frag_rows(Kt, kb * 32, ks, lane), bq[ks], sp);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) gp = Mfma32x16<DT>::run(at_frag_rows(Vt, kb * 32, ks, lane), bo[ks], gp);
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      float4_t m = {0.f, 0.f, 0.f, 0.f};
      if (mrow) m = *(const float4_t*)(mrow + kb * 32 + rq * 8 + hf * 4);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int r = rq * 4 + jj;
        const float x = sp[r] * p.scale + m[jj];              // (the expression order of at_softmax<true>: same probabilities
        sp[r] = __expf(x - st[0]) * st[1];                    //  as the forward pass)
        gp[r] = ((keep16 >> r) & 1u) ? gp[r] * p.drop.inv_keep : 0.f;
      }
    }
  };
  // ---- loop 1: delta = sum over ALL keys of P dP
  float delta = 0.f;
  for (int j = 0; j < p.nblk; ++j) {
    if (j > 0) __syncthreads();
    load_kv(j);
    if (!frags_loaded) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { bq[ks] = at_frag_rows(Qt, wave * 32, ks, lane); bo[ks] = at_frag_rows(Ot, wave * 32, ks, lane); }
      frags_loaded = true;
    }
    unsigned keepw[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    if (drop) at_keep_row(p.drop, row_chunk0 + (unsigned)(j * 16), hf, keepw);
#pragma unroll 1
    for (int kb = 0; kb < 4; ++kb) {                          // (not unrolled: the four slices' MFMA chains would be interleaved
      float16_t sp, gp;                                       //  and their accumulators all live at once)
      const unsigned keep16 = kb == 0 ? keepw[0] : kb == 1 ? keepw[1] : kb == 2 ? keepw[2] : keepw[3];
      slice(kb, mseq ? mseq + j * AT_S : nullptr, keep16, sp, gp);
#pragma unroll
      for (int r = 0; r < 16; ++r) delta += sp[r] * gp[r];
    }
  }
  delta += __shfl_xor(delta, 32, 64);
  if (hf == 0) strow[2] = delta;
  // ---- loop 2: dQ^T[d][q] = sum_j sum_key K_j[key][d] dS[q][key]
  float16_t dq[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;
  for (int j = 0; j < p.nblk; ++j) {
    __syncthreads();
    load_kv(j);
    unsigned keepw[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    if (drop) at_keep_row(p.drop, row_chunk0 + (unsigned)(j * 16), hf, keepw);
#pragma unroll 1
    for (int kb = 0; kb < 4; ++kb) {                          // (not unrolled: the four slices' MFMA chains would be interleaved
      float16_t sp, gp;                                       //  and their accumulators all live at once)
      const unsigned keep16 = kb == 0 ? keepw[0] : kb == 1 ? keepw[1] : kb == 2 ? keepw[2] : keepw[3];
      slice(kb, mseq ? mseq + j * AT_S : nullptr, keep16, sp, gp);
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = sp[r] * (gp[r] - delta) * p.scale;
      const ushort8_t d0 = pack8<DT>(v), d1 = pack8<DT>(v + 8);
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        dq[db] = Mfma32x16<DT>::run(at_frag_cols<true>(Kt, db * 32, kb * 2, lane), d0, dq[db]);
        dq[db] = Mfma32x16<DT>::run(at_frag_cols<true>(Kt, db * 32, kb * 2 + 1, lane), d1, dq[db]);
      }
    }
  }
  __syncthreads();
  unsigned short* stg = (unsigned short*)smem_raw + wave * (32 * AT_D);
  at_stage_block<DT>(stg, dq[0], 0, lane);
  at_stage_block<DT>(stg, dq[1], 1, lane);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  at_store_rows(stg, p.dqkv + ((long long)b * p.S + qb * AT_S + wave * 32) * 3 * p.H + h * AT_D, 3LL * p.H, lane);
  if (p.colsum) at_long_colsum<DT, 1>(p, stg, smem_raw + 4 * 32 * AT_D * 2, b, qb, h, 0, tid, wave, lane);
}

template <int DT>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_long_kernel(AttnArgs p) {
  drop_resolve(p.drop);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* Qt = (unsigned short*)smem_raw;
  unsigned short* Ot = Qt + AT_TILE;
  unsigned short* Kt = Ot + AT_TILE;
  unsigned short* Vt = Kt + AT_TILE;
  unsigned short* img = Vt + AT_TILE;                         // [128 q][128 keys]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5;
  const int kbj = blockIdx.x % p.nblk, bh = blockIdx.x / p.nblk, b = bh / p.nh, h = bh - b * p.nh;
  const unsigned ld = (unsigned)p.H * 3u * 2u;
  const unsigned seq_base = (unsigned)(((long long)b * p.S * 3 * p.H + h * AT_D) * 2);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.qkv, 0, 0xFFFFFFE0, 0x00020000);
  __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)p.dctx, 0, 0xFFFFFFE0, 0x00020000);
  at_load_tile(rs, seq_base + (unsigned)p.H * 2u + (unsigned)(kbj * AT_S) * ld, ld, Kt, wave, lane);
  at_load_tile(rs, seq_base + (unsigned)p.H * 4u + (unsigned)(kbj * AT_S) * ld, ld, Vt, wave, lane);
  const float* mblk = p.mask_add ? p.mask_add + (long long)b * p.S + kbj * AT_S : nullptr;
  const bool drop = p.drop.thr != 0;
  const int ql = wave * 32 + (lane & 31);
  float16_t dk[2], dv[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }
  for (int i = 0; i < p.nblk; ++i) {
    __syncthreads();                                          // Q / dO / the image of the previous query block are dead
    at_load_tile(rs, seq_base + (unsigned)(i * AT_S) * ld, ld, Qt, wave, lane);
    at_load_tile(ro, (unsigned)((((long long)b * p.S + i * AT_S) * p.H + h * AT_D) * 2), (unsigned)p.H * 2u, Ot, wave, lane);
    const int qg = i * AT_S + ql;
    const float4_t st = *(const float4_t*)(p.stats + ((long long)bh * p.S + qg) * p.sw);          // max, 1 / sum, delta
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    float16_t s[4], g[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[kb][r] = 0.f; g[kb][r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const ushort8_t bq = at_frag_rows(Qt, wave * 32, ks, lane);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) s[kb] = Mfma32x16<DT>::run(at_frag_rows(Kt, kb * 32, ks, lane), bq, s[kb]);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const ushort8_t bo = at_frag_rows(Ot, wave * 32, ks, lane);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) g[kb] = Mfma32x16<DT>::run(at_frag_rows(Vt, kb * 32, ks, lane), bo, g[kb]);
    }
    float mx = st[0], inv = st[1];
    const float delta = st[2];
    at_softmax<true>(s, mblk, p.scale, hf, mx, inv);
    unsigned keepw[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    if (drop) at_keep_row(p.drop, ((unsigned)bh * (unsigned)p.S + (unsigned)qg) * (unsigned)(p.S >> 3) + (unsigned)(kbj * 16), hf, keepw);
    ushort8_t ds[8], pd[8];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      float v[16], w[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool kp = (keepw[kb] >> r) & 1u;
        const float dp = kp ? g[kb][r] * p.drop.inv_keep : 0.f;
        v[r] = s[kb][r] * (dp - delta) * p.scale;
        w[r] = kp ? s[kb][r] * p.drop.inv_keep : 0.f;
      }
      ds[kb * 2] = pack8<DT>(v);
      ds[kb * 2 + 1] = pack8<DT>(v + 8);
      pd[kb * 2] = pack8<DT>(w);
      pd[kb * 2 + 1] = pack8<DT>(w + 8);
    }
    auto write_img = [&](const ushort8_t* x) __attribute__((always_inline)) {
      const int isw = at_iswz(ql);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const uint4_t u = __builtin_bit_cast(uint4_t, x[kb * 2 + (rq >> 1)]);
          const uint2_t w2 = {u[(rq & 1) * 2], u[(rq & 1) * 2 + 1]};
          *(uint2_t*)(img + ql * AT_S + (((kb * 4 + rq) ^ isw) << 3) + (hf << 2)) = w2;
        }
    };
    write_img(ds);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {                           // dK^T[d][key] += sum_q Q_i[q][d] dS[q][key]
      const ushort8_t bs = at_img_frag_cols(img, wave * 32, ks, lane);
#pragma unroll
      for (int db = 0; db < 2; ++db) dk[db] = Mfma32x16<DT>::run(at_frag_cols<false>(Qt, db * 32, ks, lane), bs, dk[db]);
    }
    __syncthreads();
    write_img(pd);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {                           // dV^T[d][key] += sum_q dO_i[q][d] dropout(P)[q][key]
      const ushort8_t bp = at_img_frag_cols(img, wave * 32, ks, lane);
#pragma unroll
      for (int db = 0; db < 2; ++db) dv[db] = Mfma32x16<DT>::run(at_frag_cols<false>(Ot, db * 32, ks, lane), bp, dv[db]);
    }
  }
  __syncthreads();
  unsigned short* stg = (unsigned short*)smem_raw + wave * (2 * 32 * AT_D);
  at_stage_block<DT>(stg, dk[0], 0, lane);
  at_stage_block<DT>(stg, dk[1], 1, lane);
  at_stage_block<DT>(stg + 32 * AT_D, dv[0], 0, lane);
  at_stage_block<DT>(stg + 32 * AT_D, dv[1], 1, lane);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  unsigned short* out = p.dqkv + ((long long)b * p.S + kbj * AT_S + wave * 32) * 3 * p.H + h * AT_D;
  at_store_rows(stg, out + p.H, 3LL * p.H, lane);
  at_store_rows(stg + 32 * AT_D, out + 2 * p.H, 3LL * p.H, lane);
  if (p.colsum) at_long_colsum<DT, 2>(p, stg, smem_raw + 4 * 2 * 32 * AT_D * 2, b, kbj, h, 1, tid, wave, lane);
}

static int attn_check(const char* what, int B, int S, int heads, int head_dim, int dtype, float p) {
  DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, "%s: 16-bit activations only", what);
  DLE_CHECK_ARG(dle_attention_supported(S, head_dim), "%s: built for sequence lengths 128, 256, .. 1024 and 64-wide heads (got %d, %d)", what, S, head_dim);
  DLE_CHECK_ARG(B > 0 && heads > 0, "%s: bad batch / heads", what);
  DLE_CHECK_ARG(p >= 0.f && p < 1.f, "%s: p must be in [0, 1)", what);
  DLE_CHECK_ARG((long long)B * S * 3 * heads * head_dim * 2 < 0xFFFFFFE0LL, "%s: QKV activation above 4 GiB", what);
  DLE_CHECK_ARG((long long)B * heads * S * (S / 8) < 0xFFFFFFFFLL, "%s: more than 2^35 attention probabilities (32-bit dropout chunk index)", what);
  return 0;
}

// context[T, H] = dropout(softmax(q k^T * scale + mask_add)) v per (sequence, head); qkv [T = B*S, 3H] as written by the
// fused QKV projection.  stats [B*heads, S, 2] fp32 receives (row max, 1 / row sum) for the backward pass; keep_mask
// (optional, B*heads*S*S/8 bytes) receives the dropout keep bits in the layout of dle_softmax_dropout_fwd.
extern "C" int dle_attention_fwd(const void* qkv, const float* mask_add, void* ctx, float* stats, void* keep_mask, int B,
                                 int S, int heads, int head_dim, float scale, float p, uint64_t seed, uint64_t offset,
                                 const uint64_t* offset_base, int dtype, hipStream_t stream) {
  if (int rc = attn_check("attention_fwd", B, S, heads, head_dim, dtype, p)) return rc;
  DLE_CHECK_ARG(qkv && ctx && stats, "attention_fwd: null pointer");
  DLE_CHECK_ARG(((((uintptr_t)qkv) | ((uintptr_t)ctx)) & 15) == 0, "attention_fwd: tensors must be 16-byte aligned");
  AttnArgs a = {};
  a.qkv = (const unsigned short*)qkv; a.mask_add = mask_add; a.ctx = (unsigned short*)ctx; a.stats = stats;
  a.mask_out = (unsigned char*)keep_mask; a.B = B; a.nh = heads; a.H = heads * head_dim; a.scale = scale;
  a.drop = make_drop(nullptr, p, seed, offset, offset_base);
  a.S = S; a.nblk = S / AT_S; a.sw = dle_attention_stats_floats(S);
  const size_t lds = 3 * AT_TILE * 2;
  if (S != AT_S) {
    if (dtype == DLE_F16) hipLaunchKernelGGL(attn_fwd_long_kernel<DLE_F16>, dim3(B * heads * a.nblk), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL(attn_fwd_long_kernel<DLE_BF16>, dim3(B * heads * a.nblk), dim3(256), lds, stream, a);
    DLE_LAUNCH_CHECK();
    return 0;
  }
  if (dtype == DLE_F16) hipLaunchKernelGGL(attn_fwd_kernel<DLE_F16>, dim3(B * heads), dim3(256), lds, stream, a);
  else hipLaunchKernelGGL(attn_fwd_kernel<DLE_BF16>, dim3(B * heads), dim3(256), lds, stream, a);
  DLE_LAUNCH_CHECK();
  return 0;
}

// dqkv[T, 3H] (dq | dk | dv) from dctx[T, H]; recomputes the probabilities from qkv + stats and the dropout mask from
// (seed, offset) -- the same values dle_attention_fwd was called with.  colsum_partial (optional, fp32 [B, 3H]): per-sequence
// column sums of dqkv; their sum over B is the bias gradient of the QKV projection (no column-sum pass over dqkv).
static int attention_bwd_impl(const void* qkv, const void* dctx, const float* mask_add, const float* stats, const void* keep_mask,
                              void* dqkv, float* colsum_partial, int B, int S, int heads, int head_dim, float scale, float p,
                              uint64_t seed, uint64_t offset, const uint64_t* offset_base, int dtype, hipStream_t stream) {
  if (int rc = attn_check("attention_bwd", B, S, heads, head_dim, dtype, p)) return rc;
  DLE_CHECK_ARG(qkv && dctx && stats && dqkv, "attention_bwd: null pointer");
  DLE_CHECK_ARG(((((uintptr_t)qkv) | ((uintptr_t)dctx) | ((uintptr_t)dqkv)) & 15) == 0, "attention_bwd: tensors must be 16-byte aligned");
  AttnArgs a = {};
  a.qkv = (const unsigned short*)qkv; a.dctx = (const unsigned short*)dctx; a.mask_add = mask_add;
  a.stats = (float*)stats; a.dqkv = (unsigned short*)dqkv; a.colsum = colsum_partial; a.B = B; a.nh = heads; a.H = heads * head_dim; a.scale = scale;
  a.mask_in = S == AT_S ? (const unsigned char*)keep_mask : nullptr;       // (the long-sequence kernels re-draw the mask)
  a.drop = make_drop(nullptr, p, seed, offset, offset_base);
  a.S = S; a.nblk = S / AT_S; a.sw = dle_attention_stats_floats(S);
  const size_t lds = 4 * AT_TILE * 2;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<DLE_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<DLE_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_long_kernel<DLE_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_long_kernel<DLE_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_long_kernel<DLE_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + AT_S * AT_S * 2));
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_long_kernel<DLE_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + AT_S * AT_S * 2));
    attr_set = true;
  }
  if (S != AT_S) {
    // (1) per query block: delta into the statistics rows, dQ; (2) per key block: dK, dV (reads the deltas: stream order)
    const dim3 grid(B * heads * a.nblk);
    if (dtype == DLE_F16) hipLaunchKernelGGL(attn_bwd_dq_long_kernel<DLE_F16>, grid, dim3(256), lds, stream, a);
    else hipLaunchKernelGGL(attn_bwd_dq_long_kernel<DLE_BF16>, grid, dim3(256), lds, stream, a);
    DLE_LAUNCH_CHECK();
    if (dtype == DLE_F16) hipLaunchKernelGGL(attn_bwd_dkv_long_kernel<DLE_F16>, grid, dim3(256), lds + AT_S * AT_S * 2, stream, a);
    else hipLaunchKernelGGL(attn_bwd_dkv_long_kernel<DLE_BF16>, grid, dim3(256), lds + AT_S * AT_S * 2, stream, a);
    DLE_LAUNCH_CHECK();
    return 0;
  }
  if (dtype == DLE_F16) hipLaunchKernelGGL(attn_bwd_kernel<DLE_F16>, dim3(B * heads), dim3(256), lds, stream, a);
  else hipLaunchKernelGGL(attn_bwd_kernel<DLE_BF16>, dim3(B * heads), dim3(256), lds, stream, a);
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_attention_bwd(const void* qkv, const void* dctx, const float* mask_add, const float* stats, void* dqkv,
                                 float* colsum_partial, int B, int S, int heads, int head_dim, float scale, float p,
                                 uint64_t seed, uint64_t offset, const uint64_t* offset_base, int dtype, hipStream_t stream) {
  return attention_bwd_impl(qkv, dctx, mask_add, stats, nullptr, dqkv, colsum_partial, B, S, heads, head_dim, scale, p, seed, offset,
                            offset_base, dtype, stream);
}

// dle_attention_bwd with the keep mask dle_attention_fwd wrote (keep_mask, B*heads*S*S/8 bytes, 16-byte aligned) READ instead of
// re-drawn: the S = 128 backward kernel is bound by the 32-bit multiplies of its 8 Philox calls per lane; 16 bytes per query row
// replace them.  keep_mask = NULL (or S > 128): identical to dle_attention_bwd.  Same results bit for bit.
extern "C" int dle_attention_bwd_keep(const void* qkv, const void* dctx, const float* mask_add, const float* stats,
                                      const void* keep_mask, void* dqkv, float* colsum_partial, int B, int S, int heads,
                                      int head_dim, float scale, float p, uint64_t seed, uint64_t offset,
                                      const uint64_t* offset_base, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG((((uintptr_t)keep_mask) & 15) == 0, "attention_bwd_keep: keep_mask must be 16-byte aligned");
  return attention_bwd_impl(qkv, dctx, mask_add, stats, keep_mask, dqkv, colsum_partial, B, S, heads, head_dim, scale, p, seed, offset,
                            offset_base, dtype, stream);
}
