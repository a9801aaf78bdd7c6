// Shared building blocks of the LDS-DMA fed MFMA kernels (gemm_dma.hip, conv3x3.hip): constant division, the
// implicit-GEMM geometry, the per-operand DMA tile loaders, MFMA fragment reads out of the swizzled LDS images.
#pragma once
#include "common.h"
#include <stdlib.h>
#include <type_traits>

#define BK 64

// Division by a launch-time constant as multiply-high + shift (exact for 0 <= n < 2^31): the im2col loaders
// decompose k -> (tap, channel) and pixel -> (n, p, q) for every DMA piece of every K tile; with generic integer
// division (~40 VALU instructions each) the 3x3 convolutions were bound by their address arithmetic.
struct FastDiv {
  unsigned mul, shr;
  int d;
};
static FastDiv make_fastdiv(int d) {
  FastDiv f;
  f.d = d;
  if (d <= 1) { f.mul = 0; f.shr = 0; return f; }
  unsigned lg = 0;
  while ((1u << lg) < (unsigned)d) ++lg;                 // ceil(log2 d)
  const unsigned p = 31 + lg;
  f.mul = (unsigned)(((1ull << p) + (unsigned)d - 1) / (unsigned)d);
  f.shr = p - 32;
  return f;
}
__device__ __forceinline__ int fd_div(int n, const FastDiv& f) {
  return f.d <= 1 ? n : (int)(__umulhi((unsigned)n, f.mul) >> f.shr);
}

struct ConvGeom {        // implicit-GEMM operand geometry (NHWC tensors, KRSC weights)
  int H, W, C;           // spatial size / channels of the tensor the im2col operand reads
  int P, Q;              // spatial size of the convolution OUTPUT (forward sense)
  int R, S, stride, pad;
  int Ko;                // output channels (forward sense)
  FastDiv dC, dS, dKo, dQ, dP, dW, dH, dStride;
};
static ConvGeom make_geom(int H, int W, int C, int P, int Q, int R, int S, int stride, int pad, int Ko) {
  ConvGeom g;
  g.H = H; g.W = W; g.C = C; g.P = P; g.Q = Q; g.R = R; g.S = S; g.stride = stride; g.pad = pad; g.Ko = Ko;
  g.dC = make_fastdiv(C); g.dS = make_fastdiv(S); g.dKo = make_fastdiv(Ko); g.dQ = make_fastdiv(Q);
  g.dP = make_fastdiv(P); g.dW = make_fastdiv(W); g.dH = make_fastdiv(H); g.dStride = make_fastdiv(stride);
  return g;
}

template <int DT> struct Mfma32x16;
template <> struct Mfma32x16<DLE_F16> {
  static __device__ __forceinline__ float16_t run(ushort8_t a, ushort8_t b, float16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
  }
};
template <> struct Mfma32x16<DLE_BF16> {
  static __device__ __forceinline__ float16_t run(ushort8_t a, ushort8_t b, float16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((ext_vector_type(4))) short short4_t;
#define OOB_OFF 0xFFFFFFF0u

__device__ __forceinline__ int swz_kc(int row) { return (row >> 1) & 7; }
// row-contiguous images, line = TILE rows: lines of >= 256 B all start at bank 0 -> spread the 4 k lines of a read group
// over the 32-byte pair slots with (k & 3) << 1; a 64-row image has 128-byte lines (odd k lines already sit in the other
// bank half, and only 4 pair slots per line) -> k & 2.
template <int TILE> __device__ __forceinline__ int swz_rc(int k) { return TILE >= 128 ? ((k & 3) << 1) : (k & 2); }

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, unsigned short* lds_wave_base, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)lds_wave_base, 16, voff, 0, 0, 0);
}

// The same instruction issued through inline asm, for kernels whose EVERY LDS-DMA goes this way (M0 is set here, behind the
// compiler's back).  hipcc's wait-count pass treats a builtin DMA as pending until it has seen a wait that provably covers it
// and then drains the whole queue -- s_waitcnt vmcnt(0) -- before the first overwrite of one of its address registers; a
// counted wait that leaves younger STORES in flight (the persistent tile walk of gemm_dma.hip) cannot be expressed to it.
// Hidden from that pass, every wait is the explicit one in the source.  (vmcnt retires in order, so the waits the compiler
// inserts for its own loads only become more conservative when instructions it does not know about are in the queue.)
typedef int int4v_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int4v_t rsrc_words(const void* base) {
  const unsigned long long b = (unsigned long long)base;
  int4v_t r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
  r.z = (int)0xFFFFFFE0u;
  r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ void dma16_raw(int4v_t rs, unsigned short* lds_wave_base, unsigned voff) {
  const int m0v = __builtin_amdgcn_readfirstlane((int)(unsigned)(__UINTPTR_TYPE__)(lds_void*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(m0v), "v"(voff), "s"(rs) : "memory");
}

// ---- per-operand tile loader state: 4 DMA instructions per wave per K tile --------------------------
// k-contiguous images  [128 rows][8 chunks]: instr j of wave w covers rows (4w+j)*8 .. +7     (modes 0, 2, 4)
// row-contiguous images [64 k][16 chunks]:   instr j of wave w covers k rows (4w+j)*4 .. +3   (modes 1, 3, 5)
//   0  matrix, k contiguous                1  matrix, rows contiguous
//   2  im2col, forward:   A(m=(n,p,q), k=(r,s,c))  = X[n, p*st-pad+r, q*st-pad+s, c]
//   4  im2col, data grad: A(m=(n,h,w), k=(r,s,ko)) = dY[n, (h+pad-r)/st, (w+pad-s)/st, ko]   (0 unless divisible)
//   3  im2col, weight grad B operand: B(n'=(r,s,c), k=pixel(n,p,q)) = X[n, p*st-pad+r, q*st-pad+s, c]
//   5  KRSC weights as the data-grad B operand: B(c, k=(r,s,ko)) = W[ko][r][s][c]
template <int MODE, int TILE, int NW, bool RAW = false>
struct Loader {
  // the operand tile has TILE rows (128 or 256) x 64 k = TILE/8 DMA pieces of 1 KiB; NW waves own NP pieces each
  static constexpr bool RC = (MODE == 1 || MODE == 3 || MODE == 5);
  static constexpr int NP = TILE / 8 / NW;
  static constexpr int CPL = TILE / 8;       // 16-byte chunks per k line of a row-contiguous image
  static constexpr int KPP = 64 / CPL;       // k lines per piece (row-contiguous image)
  unsigned off[NP];      // mode 0/1: byte offset inside the K-tile panel; conv modes: lane-constant part
  int kin[NP];           // KC: k element offset inside the tile; RC: k row inside the tile
  int a0[NP], a1[NP];    // mode 2: (h0, w0) of the output pixel; mode 4: (h, w); mode 3: (r, s) of the lane's tap
  unsigned fo[NP];       // modes 2 / 4 / 5, channels % 64 == 0: lane part of the offset with the k position folded in
  bool row_ok[NP];
  bool tap_uniform;      // a 64-deep K tile lies inside ONE filter tap: (r, s) are scalars, no per-lane division

  __device__ __forceinline__ void init(int wave, int lane, int row0, int nrows, long long ld, const ConvGeom& cg) {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      if (!RC) {
        const int row = (wave * NP + j) * 8 + (lane >> 3), cpos = lane & 7;
        const int chunk = cpos ^ swz_kc(row);
        const int g = row0 + row;
        row_ok[j] = g < nrows;
        kin[j] = chunk * 8;
        if (MODE == 0) {
          off[j] = (unsigned)(((long long)row * ld + chunk * 8) * 2);
        } else if (MODE == 2) {
          const int t = fd_div(g, cg.dQ), q = g - t * cg.Q;      // g = (n * P + p) * Q + q
          const int n = fd_div(t, cg.dP), p = t - n * cg.P;
          a0[j] = p * cg.stride - cg.pad;
          a1[j] = q * cg.stride - cg.pad;
          off[j] = (unsigned)((((long long)n * cg.H + a0[j]) * cg.W + a1[j]) * cg.C * 2);
        } else {                                          // MODE 4: g = (n * H + h) * W + w  (dX pixel)
          const int t = fd_div(g, cg.dW), w = g - t * cg.W;
          const int n = fd_div(t, cg.dH), h = t - n * cg.H;
          a0[j] = h + cg.pad;
          a1[j] = w + cg.pad;
          off[j] = (unsigned)n;                           // image index; pixel offset is rebuilt per tap
        }
      } else {
        const int kr = (wave * NP + j) * KPP + lane / CPL, cpos = lane % CPL;
        const int chunk = (((cpos >> 1) ^ swz_rc<TILE>(kr)) << 1) | (cpos & 1);
        const int g = row0 + chunk * 8;
        row_ok[j] = g < nrows;
        kin[j] = kr;
        if (MODE == 1) {
          off[j] = (unsigned)(((long long)kr * ld + chunk * 8) * 2);
        } else if (MODE == 3) {                           // g = (r * S + s) * C + c
          const int tap = fd_div(g, cg.dC), c = g - tap * cg.C;
          a0[j] = fd_div(tap, cg.dS);
          a1[j] = tap - a0[j] * cg.S;
          off[j] = (unsigned)(c * 2);
        } else {                                          // MODE 5: g = input channel c
          off[j] = (unsigned)(g * 2);
        }
      }
    }
    tap_uniform = (MODE == 2 && (cg.C & 63) == 0) || (MODE == 4 && (cg.Ko & 63) == 0) ||
                  (MODE == 5 && (cg.Ko & 63) == 0);
    if (MODE == 2 || MODE == 4 || MODE == 5) {
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        if (MODE == 2) fo[j] = off[j] + (unsigned)(kin[j] * 2);
        else if (MODE == 4) fo[j] = (unsigned)(((((long long)off[j] * cg.P + a0[j]) * cg.Q + a1[j]) * cg.Ko + kin[j]) * 2);
        else fo[j] = off[j] + (unsigned)(((long long)kin[j] * cg.R * cg.S * cg.C) * 2);
      }
    }
  }

  // base: (row0, k0) panel for mode 0, (k0, row0) panel for mode 1, tensor base for the conv modes
  // pieces [J0, J1) of this wave
  template <int J0, int J1>
  __device__ __forceinline__ void issue(const unsigned short* base, unsigned short* tile, int wave, int krem,
                                        int k0, const ConvGeom& cg) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0xFFFFFFE0, 0x00020000);
    if ((MODE == 2 || MODE == 4 || MODE == 5) && tap_uniform) {
      // the whole K tile belongs to one tap: (tap, r, s) and the tap's offset are wave-uniform scalars
      const int ch = MODE == 2 ? cg.C : cg.Ko;
      int tap;
      if (MODE == 2) tap = fd_div(k0, cg.dC); else tap = fd_div(k0, cg.dKo);
      const int c0 = k0 - tap * ch;
      const int r = fd_div(tap, cg.dS), s2 = tap - r * cg.S;
      const unsigned delta = MODE == 2 ? (unsigned)(((r * cg.W + s2) * cg.C + c0) * 2)
                           : MODE == 4 ? (unsigned)((c0 - (r * cg.Q + s2) * cg.Ko) * 2)
                                       : (unsigned)((((long long)c0 * cg.R * cg.S + tap) * cg.C) * 2);
#pragma unroll
      for (int j = J0; j < J1; ++j) {
        bool ok = row_ok[j] && kin[j] < krem;
        if (MODE == 2) {
          const int h = a0[j] + r, w = a1[j] + s2;
          ok = ok && h >= 0 && h < cg.H && w >= 0 && w < cg.W;
        } else if (MODE == 4) {
          const int hp = a0[j] - r, wp = a1[j] - s2;
          if (cg.stride == 1) {
            ok = ok && hp >= 0 && wp >= 0 && hp < cg.P && wp < cg.Q;
          } else {                                   // strided forward conv: only every stride-th (h, w) has a source pixel
            const int pp = fd_div(hp, cg.dStride), qq = fd_div(wp, cg.dStride);
            ok = ok && hp >= 0 && wp >= 0 && pp * cg.stride == hp && qq * cg.stride == wp && pp < cg.P && qq < cg.Q;
            const unsigned o2 = (unsigned)(((((int)off[j] * cg.P + pp) * cg.Q + qq) * cg.Ko + c0 + kin[j]) * 2);
            dma16(rs, tile + (wave * NP + j) * 512, ok ? o2 : OOB_OFF);
            continue;
          }
        }
        dma16(rs, tile + (wave * NP + j) * 512, ok ? fo[j] + delta : OOB_OFF);
      }
      return;
    }
#pragma unroll
    for (int j = J0; j < J1; ++j) {
      bool ok = row_ok[j] && kin[j] < krem;
      unsigned o = off[j];
      if (MODE == 2) {
        const int k = k0 + kin[j];
        const int tap = fd_div(k, cg.dC), c = k - tap * cg.C;
        const int r = fd_div(tap, cg.dS), s = tap - r * cg.S;
        const int h = a0[j] + r, w = a1[j] + s;
        ok = ok && h >= 0 && h < cg.H && w >= 0 && w < cg.W;
        o += (unsigned)(((r * cg.W + s) * cg.C + c) * 2);
      } else if (MODE == 4) {
        const int k = k0 + kin[j];
        const int tap = fd_div(k, cg.dKo), ko = k - tap * cg.Ko;
        const int r = fd_div(tap, cg.dS), s = tap - r * cg.S;
        const int hp = a0[j] - r, wp = a1[j] - s;
        const int p = fd_div(hp, cg.dStride), q = fd_div(wp, cg.dStride);      // (negative hp / wp are rejected below)
        ok = ok && hp >= 0 && wp >= 0 && p * cg.stride == hp && q * cg.stride == wp && p < cg.P && q < cg.Q;
        o = (unsigned)(((((long long)off[j] * cg.P + p) * cg.Q + q) * cg.Ko + ko) * 2);
      } else if (MODE == 3) {
        const int pix = k0 + kin[j];
        const int t = fd_div(pix, cg.dQ), q = pix - t * cg.Q;
        const int n = fd_div(t, cg.dP), p = t - n * cg.P;
        const int h = p * cg.stride - cg.pad + a0[j], w = q * cg.stride - cg.pad + a1[j];
        ok = ok && h >= 0 && h < cg.H && w >= 0 && w < cg.W;
        o += (unsigned)((((long long)n * cg.H + h) * cg.W + w) * cg.C * 2);
      } else if (MODE == 5) {
        const int k = k0 + kin[j];
        const int tap = fd_div(k, cg.dKo), ko = k - tap * cg.Ko;
        o += (unsigned)((((long long)ko * cg.R * cg.S + tap) * cg.C) * 2);
      }
      if constexpr (RAW) dma16_raw(rsrc_words(base), tile + (wave * NP + j) * 512, ok ? o : OOB_OFF);
      else dma16(rs, tile + (wave * NP + j) * 512, ok ? o : OOB_OFF);
    }
  }
};

// One MFMA operand fragment (32 rows x 16 k, 8 halves per lane) out of a TILE x 64 LDS image.
//
// Row-contiguous images are read with the LDS transpose read.  It is issued through INLINE ASM on purpose: hipcc (ROCm
// 7.2) treats the ds_read_tr builtin as a possible alias of every LDS-DMA in flight and puts `s_waitcnt vmcnt(0)` in front
// of it -- in a DMA-pipelined K loop that drains the prefetch of the NEXT tile before the current one is computed (the
// weight-gradient GEMMs ran with no DMA / MFMA overlap inside a workgroup, the data-gradient GEMMs drained twice per K
// tile).  The price: the compiler does not track an asm DS operation, so the caller waits for it explicitly
// (frag_wait()) before the first use; compiler-generated lgkmcnt waits for its own reads stay correct (the counter is
// in order, foreign entries only make them conservative).
struct TrPair { short4_t lo, hi; };                 // the two 64-bit transpose reads of one fragment, as issued
template <bool RC> struct FragT { typedef ushort8_t type; };
template <> struct FragT<true> { typedef TrPair type; };

__device__ __forceinline__ short4_t ds_read_tr16_asm(const unsigned short* p) {
  short4_t v;
  const unsigned addr = (unsigned)(unsigned long)(__attribute__((address_space(3))) const unsigned short*)p;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

template <bool RC, int TILE>
__device__ __forceinline__ typename FragT<RC>::type frag_issue(const unsigned short* t, int rbase32, int ks, int lane) {
  if constexpr (!RC) {
    const int row = rbase32 + (lane & 31);
    return *(const ushort8_t*)(t + row * BK + (((ks * 2 + (lane >> 5)) ^ swz_kc(row)) << 3));
  } else {
    // transpose read: lanes 0-15 / 16-31 -> rows +0..15 / +16..31 of the 32-row fragment, k group = lane >> 5
    const int tg = lane >> 4, ti = lane & 15;
    const int rbase = rbase32 + ((tg & 1) << 4);
    const int kb = ks * 16 + (tg >> 1) * 8 + (ti >> 2);
    const int chunk = (rbase >> 3) + ((ti & 3) >> 1);
    TrPair f;
    {
      const int cpos = (((chunk >> 1) ^ swz_rc<TILE>(kb)) << 1) | (chunk & 1);
      f.lo = ds_read_tr16_asm(t + kb * TILE + cpos * 8 + ((ti & 1) << 2));
    }
    {
      const int k = kb + 4;
      const int cpos = (((chunk >> 1) ^ swz_rc<TILE>(k)) << 1) | (chunk & 1);
      f.hi = ds_read_tr16_asm(t + k * TILE + cpos * 8 + ((ti & 1) << 2));
    }
    return f;
  }
}
__device__ __forceinline__ ushort8_t frag_value(const ushort8_t& f) { return f; }
__device__ __forceinline__ ushort8_t frag_value(const TrPair& f) {
  ushort8_t r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = (unsigned short)f.lo[e]; r[4 + e] = (unsigned short)f.hi[e]; }
  return r;
}
// every asm-issued fragment read of this wave has landed (call before the first frag_value() of a batch)
template <bool ANY_RC>
__device__ __forceinline__ void frag_wait() {
  if constexpr (ANY_RC) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);        // hipcc hoists register-only MFMAs over an asm wait otherwise
  }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also fences global memory (s_waitcnt vmcnt(0)):
// in an epilogue that would wait for every output store of the previous pass to be acknowledged by L2.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    static_for<I + 1, N>(f);
  }
}
