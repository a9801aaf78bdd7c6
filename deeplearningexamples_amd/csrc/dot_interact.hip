// DLRM dot-based feature interaction, forward + backward, for gfx950 (wave64 + MFMA).
//
// Replaces the reference's warp-32 WMMA kernels
//   DLRM/dlrm/cuda_src/dot_based_interact/dot_based_interact_fp16_fwd.cu:160-271
//   DLRM/dlrm/cuda_src/dot_based_interact/dot_based_interact_fp16_bwd.cu:180-332
//   (+ the fp32 FMA kernels dot_based_interact_fp32_{fwd,bwd}.cu) behind
//   dlrm.cuda_ext.interaction_*.dotBasedInteract{Fwd,Bwd}
//   (dot_based_interact_ampere/pytorch_ops.cpp:3-12).
//
// Design (MI355X-first, not a translation):
//  * one 64-lane wavefront owns one sample; 4 samples per 256-thread workgroup.
//  * forward: Z = X X^T with v_mfma_f32_32x32x16.  A and B fragments of X X^T are the SAME
//    registers (lane l holds row l&31, 8 consecutive columns) so X is loaded straight from
//    HBM into MFMA operands -- no LDS staging of the input at all.  The 480-wide output row
//    (bottom-mlp copy | strict lower triangle | zero pad) is assembled in LDS and leaves as
//    one coalesced 16 B/lane store.
//  * backward: grad = U_sym X (M=32, N=C, K=32).  U_sym (symmetric, zero diagonal) is built
//    in LDS from the flat upstream gradient, X is staged once in LDS with coalesced 16 B
//    loads and the K-strided B fragments are gathered from LDS; the result is re-staged in
//    LDS so that HBM sees only full-width coalesced stores.
//  * HBM-bound op: algorithmic bytes/sample fwd = R*C*e + OW*e, bwd = 2*R*C*e + OW*e + C*e.
//  * shapes outside the MFMA envelope (R>32, C%16 (fwd) / C%32 (bwd) != 0, fp32) take a generic
//    LDS kernel with fp32 FMA (still one workgroup per sample).
#include "common.h"

template <int DT> struct Mfma32;
template <> struct Mfma32<DLE_F16> {
  static __device__ __forceinline__ float16_t run(ushort8_t a, ushort8_t b, float16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, a),
                                                  __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
  }
};
template <> struct Mfma32<DLE_BF16> {
  static __device__ __forceinline__ float16_t run(ushort8_t a, ushort8_t b, float16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};

// strict-lower-triangle linear index t -> (i, j), i > j, row-major (interactions.py:50-53)
__device__ __forceinline__ void tril_unrank(int t, int& i, int& j) {
  int r = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)t)) * 0.5f);
  while (r * (r - 1) / 2 > t) --r;
  while ((r + 1) * r / 2 <= t) ++r;
  i = r;
  j = t - r * (r - 1) / 2;
}

// ------------------------------------------------------------------ forward, MFMA path
// requires R <= 32, C % 16 == 0, OW % 8 == 0, 16-byte aligned x/out.
template <int DT>
__global__ __launch_bounds__(256) void dot_fwd_mfma(const unsigned short* __restrict__ x,
                                                    unsigned short* __restrict__ out, int B, int R,
                                                    int C, int OW) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wave;
  unsigned short* so = (unsigned short*)smem_raw + (size_t)wave * OW;
  const int row = lane & 31, h = lane >> 5;
  if (b < B) {
    const unsigned short* xr = x + ((size_t)b * R + row) * C + h * 8;
    float16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool live = row < R;
    for (int k0 = 0; k0 < C; k0 += 16) {
      ushort8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (live) v = *(const ushort8_t*)(xr + k0);
      acc = Mfma32<DT>::run(v, v, acc);
      if (row == 0) *(ushort8_t*)(so + k0 + h * 8) = v;   // bottom-MLP slice = row 0 of X
    }
    // strict lower triangle straight from the accumulator registers
    // 32x32 C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
      const int j = row;
      if (i < R && j < i) so[C + i * (i - 1) / 2 + j] = Elem<DT>::from_f32(acc[r]);
    }
    const int ntril = R * (R - 1) / 2;
    for (int p = C + ntril + lane; p < OW; p += 64) so[p] = 0;
  }
  __syncthreads();
  if (b < B) {
    unsigned short* o = out + (size_t)b * OW;
    for (int q = lane * 8; q < OW; q += 512) *(ushort8_t*)(o + q) = *(const ushort8_t*)(so + q);
  }
}

// Persistent form (C = 16 NK known at compile time): a wavefront walks samples b = wave, wave + #waves, ... on its own LDS slice
// (no workgroup barrier: LDS operations of one wavefront execute in order) and requests the NK 16-byte pieces of sample
// b + #waves -- into registers -- before the products of sample b.  The one-shot form above is a chain per sample of NK dependent
// HBM round trips (load -> MFMA in a run-time loop) -> LDS -> store, hidden only by occupancy: 139 us for batch 65536 (3.7 TB/s).
template <int DT, int NK>
__global__ __launch_bounds__(256) void dot_fwd_mfma_walk(const unsigned short* __restrict__ x, unsigned short* __restrict__ out, int B,
                                                         int R, int OW) {
  constexpr int C = 16 * NK;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned short* so = (unsigned short*)smem_raw + (size_t)wave * OW;
  const int row = lane & 31, h = lane >> 5;
  const int nwaves = gridDim.x * 4;
  const bool live = row < R;
  const int ntril = R * (R - 1) / 2;
  ushort8_t xv[NK];
  auto issue = [&](int b) __attribute__((always_inline)) {      // unconditional (clamped) addresses, masked at use
    const unsigned short* xr = x + ((size_t)b * R + (live ? row : 0)) * C + h * 8;
#pragma unroll
    for (int k = 0; k < NK; ++k) xv[k] = *(const ushort8_t*)(xr + k * 16);
  };
  int b = blockIdx.x * 4 + wave;
  if (b < B) issue(b);
  for (; b < B; b += nwaves) {
    ushort8_t cur[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) cur[k] = live ? xv[k] : (ushort8_t){0, 0, 0, 0, 0, 0, 0, 0};
    const int bn = b + nwaves;
    if (bn < B) issue(bn);
    float16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      acc = Mfma32<DT>::run(cur[k], cur[k], acc);
      if (row == 0) *(ushort8_t*)(so + k * 16 + h * 8) = cur[k];   // bottom-MLP slice = row 0 of X
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
      const int j = row;
      if (i < R && j < i) so[C + i * (i - 1) / 2 + j] = Elem<DT>::from_f32(acc[r]);
    }
    for (int p = C + ntril + lane; p < OW; p += 64) so[p] = 0;
    // lanes exchange data through `so` without a barrier (one wavefront: its LDS instructions execute in order): the compiler
    // must see the ordering too -- thread by thread it may forward a lane's earlier load across ANOTHER lane's exec-masked
    // store (it did in gemm8_kernel.h, see G8_WAVE_FENCE).  A wavefront-scope fence costs no instruction.
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    unsigned short* o = out + (size_t)b * OW;
    for (int q = lane * 8; q < OW; q += 512) *(ushort8_t*)(o + q) = *(const ushort8_t*)(so + q);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");       // ... and before the next sample's writes
  }
}

// ------------------------------------------------------------------ backward, MFMA path
// requires R <= 32, C % 32 == 0, C <= 256, OW % 8 == 0.
#define DOT_BWD_USTRIDE 40   // halves per U row (32 + 8 pad -> 80 B, keeps 16 B alignment)
// NB = 32-wide column blocks of the gradient kept in accumulators (C <= 32 NB); sized per launch so that a C = 128
// sample costs 64 accumulator registers, not the 128 of the C = 256 maximum (which left one wave per SIMD).
// Persistent form: a wavefront walks samples b = wave, wave + #waves, ... on its OWN LDS slice (no workgroup barrier: LDS operations of
// one wavefront execute in order), and the loads of sample b + #waves are requested -- into registers -- before the products of
// sample b.  The one-shot form was a ~13 us chain per sample (HBM round trip -> LDS -> unrank -> MFMA -> stage -> store) hidden only
// by three workgroups per CU: 273 us for batch 65536 (3.5 TB/s).
template <int DT, int NB>
__global__ __launch_bounds__(256) void dot_bwd_mfma(const unsigned short* __restrict__ x,
                                                    const unsigned short* __restrict__ ug,
                                                    unsigned short* __restrict__ grad,
                                                    unsigned short* __restrict__ mlp_grad, int B,
                                                    int R, int C_rt, int OW, float* __restrict__ found_inf) {
  constexpr int C = 32 * NB;                             // (the launcher instantiates NB = C / 32: row arithmetic is shifts)
  (void)C_rt;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned bad = 0;                                      // OR of the exponent-all-ones tests of the 16-bit values written
  constexpr int XS = C + 8;                              // halves per staged X row
  const size_t per_wave = (size_t)32 * XS + 32 * DOT_BWD_USTRIDE + 256;
  unsigned short* xs = (unsigned short*)smem_raw + wave * per_wave;
  unsigned short* us = xs + 32 * XS;
  unsigned short* ms = us + 32 * DOT_BWD_USTRIDE;        // upstream[:, :C] of the sample (bottom-MLP gradient / row-0 addend)
  const int row = lane & 31, h = lane >> 5;
  const int ntril = R * (R - 1) / 2;
  constexpr int cpr = C >> 3;                            // 16-byte chunks per row
  constexpr int nb_n = NB;                               // 32-wide column blocks (<= 8)
  const int nwaves = gridDim.x * 4;
  // Every global load of a sample -- the 2 * NB 16-byte pieces of X per lane, the <= 8 pairwise-gradient values per lane and the
  // C leading entries of the upstream row -- is requested from UNCONDITIONAL (clamped) addresses and masked afterwards (loads under
  // run-time conditions are serialised by hipcc's wait-count pass).
  ushort8_t xv[2 * NB], mv;
  unsigned short tv[8];
  auto issue = [&](int b) __attribute__((always_inline)) {
    const unsigned short* xb = x + (size_t)b * R * C;
#pragma unroll
    for (int it = 0; it < 2 * NB; ++it) {
      const int q = lane + 64 * it;
      const int rr = q / cpr, cc = q - rr * cpr;
      const bool ok = q < 32 * cpr && rr < R;
      xv[it] = *(const ushort8_t*)(xb + (ok ? (size_t)rr * C + cc * 8 : 0));
    }
    const unsigned short* ub = ug + (size_t)b * OW;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int t = lane + 64 * it;
      tv[it] = ub[C + (t < ntril ? t : 0)];
    }
    mv = *(const ushort8_t*)(ub + (lane * 8 < C ? lane * 8 : 0));
  };
  int b = blockIdx.x * 4 + wave;
  if (b < B) issue(b);
  for (; b < B; b += nwaves) {
    // ---- commit this sample's operands to the wavefront's LDS slice: zero U, stage X (pad rows R..31 are zeros), upstream head
    for (int q = lane; q < 32 * DOT_BWD_USTRIDE / 8; q += 64) {
      ushort8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
      *(ushort8_t*)(us + q * 8) = z;
    }
#pragma unroll
    for (int it = 0; it < 2 * NB; ++it) {
      const int q = lane + 64 * it;
      if (q < 32 * cpr) {
        const int rr = q / cpr, cc = q - rr * cpr;
        const ushort8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
        *(ushort8_t*)(xs + rr * XS + cc * 8) = rr < R ? xv[it] : z;
      }
    }
    if (lane * 8 < C) *(ushort8_t*)(ms + lane * 8) = mv;
    if (mlp_grad && lane * 8 < C) *(ushort8_t*)(mlp_grad + (size_t)b * C + lane * 8) = mv;   // bottom-MLP gradient = upstream[:, :C]
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int t = lane + 64 * it;
      if (t < ntril) {
        int i, j;
        tril_unrank(t, i, j);
        us[i * DOT_BWD_USTRIDE + j] = tv[it];
        us[j * DOT_BWD_USTRIDE + i] = tv[it];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    // ---- the next sample's loads fly under this sample's products and stores (xv / tv / mv are free again)
    if (b + nwaves < B) issue(b + nwaves);
    // grad = U X.  U rows are k-contiguous in LDS (one 16-byte read per fragment); X is the contraction-strided
    // operand, B(n, k) = X[k][n], read with the LDS transpose read (4 k lines x 4 columns per 16-lane group) instead
    // of 16 two-byte reads.  Operands are swapped (D = X^T-fragment x U-fragment) so that a lane owns 4 CONSECUTIVE
    // columns of one gradient row: register r <-> column 8 (r >> 2) + 4 h + (r & 3), lane & 31 <-> row i.
    float16_t acc[NB];
    const ushort8_t a0 = *(const ushort8_t*)(us + row * DOT_BWD_USTRIDE + h * 8);
    const ushort8_t a1 = *(const ushort8_t*)(us + row * DOT_BWD_USTRIDE + 16 + h * 8);
    const int tg = lane >> 4, ti = lane & 15;
    typedef __attribute__((ext_vector_type(4))) short short4_t;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      if (nb < nb_n) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        ushort8_t bf[2];
        const int ncol = nb * 32 + ((tg & 1) << 4) + ((ti & 3) << 2);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const int k = ks * 16 + (tg >> 1) * 8 + (ti >> 2) + hh * 4;
            const short4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) short4_t*)(xs + k * XS + ncol));
#pragma unroll
            for (int e = 0; e < 4; ++e) bf[ks][hh * 4 + e] = (unsigned short)v[e];
          }
        acc[nb] = Mfma32<DT>::run(bf[0], a0, acc[nb]);
        acc[nb] = Mfma32<DT>::run(bf[1], a1, acc[nb]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // every B-fragment read is done: reuse xs as the output stage
    if (row < R) {
      typedef __attribute__((ext_vector_type(4))) unsigned short ushort4_t;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        if (nb < nb_n) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int n = nb * 32 + 8 * q + 4 * h;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[nb][q * 4 + e];
            // fused form (mlp_grad == NULL): row 0 also receives upstream[:, :C] (what autograd adds later)
            if (!mlp_grad && row == 0) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += Elem<DT>::to_f32(ms[n + e]);
            }
            ushort4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              o[e] = Elem<DT>::from_f32(v[e]);
              constexpr unsigned EXPM = DT == DLE_F16 ? 0x7C00u : 0x7F80u;
              bad |= ((unsigned)o[e] & EXPM) == EXPM;
            }
            *(ushort4_t*)(xs + row * XS + n) = o;
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    unsigned short* gb = grad + (size_t)b * R * C;
    for (int q = lane; q < R * cpr; q += 64) {
      const int rr = q / cpr, cc = q - rr * cpr;
      *(ushort8_t*)(gb + (size_t)rr * C + cc * 8) = *(const ushort8_t*)(xs + rr * XS + cc * 8);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the slice is re-staged at the top of the next trip
  }
  // GradScaler's inf / nan test on the gradient this kernel just produced (the values as stored, like a pass over the
  // tensor would see them): the train step's separate 450 MB sweep of the embedding gradient goes away
  if (found_inf && bad) *found_inf = 1.0f;
}

// ------------------------------------------------------------------ generic path (any R, C, dtype)
template <int DT> struct IO;
template <> struct IO<DLE_F32> {
  typedef float T;
  static __device__ __forceinline__ float ld(const T* p) { return *p; }
  static __device__ __forceinline__ void st(T* p, float v) { *p = v; }
};
template <> struct IO<DLE_F16> {
  typedef unsigned short T;
  static __device__ __forceinline__ float ld(const T* p) { return Elem<DLE_F16>::to_f32(*p); }
  static __device__ __forceinline__ void st(T* p, float v) { *p = Elem<DLE_F16>::from_f32(v); }
};
template <> struct IO<DLE_BF16> {
  typedef unsigned short T;
  static __device__ __forceinline__ float ld(const T* p) { return Elem<DLE_BF16>::to_f32(*p); }
  static __device__ __forceinline__ void st(T* p, float v) { *p = Elem<DLE_BF16>::from_f32(v); }
};

template <int DT>
__global__ __launch_bounds__(256) void dot_fwd_generic(const typename IO<DT>::T* __restrict__ x,
                                                       typename IO<DT>::T* __restrict__ out, int B,
                                                       int R, int C, int OW) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* xs = (float*)smem_raw;
  const int b = blockIdx.x;
  const typename IO<DT>::T* xb = x + (size_t)b * R * C;
  typename IO<DT>::T* ob = out + (size_t)b * OW;
  const int CP = C | 1;   // odd LDS row stride: conflict-free row-pair dot products
  for (int q = threadIdx.x; q < R * C; q += blockDim.x) {
    const int rr = q / C, cc = q - rr * C;
    xs[rr * CP + cc] = IO<DT>::ld(xb + q);
  }
  __syncthreads();
  for (int q = threadIdx.x; q < C; q += blockDim.x) IO<DT>::st(ob + q, xs[q]);
  const int ntril = R * (R - 1) / 2;
  for (int t = threadIdx.x; t < ntril; t += blockDim.x) {
    int i, j;
    tril_unrank(t, i, j);
    const float* a = xs + i * CP;
    const float* c = xs + j * CP;
    float s = 0.f;
    for (int k = 0; k < C; ++k) s = fmaf(a[k], c[k], s);
    IO<DT>::st(ob + C + t, s);
  }
  for (int p = C + ntril + threadIdx.x; p < OW; p += blockDim.x) IO<DT>::st(ob + p, 0.f);
}

template <int DT>
__global__ __launch_bounds__(256) void dot_bwd_generic(const typename IO<DT>::T* __restrict__ x,
                                                       const typename IO<DT>::T* __restrict__ ug,
                                                       typename IO<DT>::T* __restrict__ grad,
                                                       typename IO<DT>::T* __restrict__ mlp_grad,
                                                       int B, int R, int C, int OW) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* xs = (float*)smem_raw;        // [R][C]
  float* us = xs + R * C;              // [R][R] symmetric, zero diagonal
  const int b = blockIdx.x;
  const typename IO<DT>::T* xb = x + (size_t)b * R * C;
  const typename IO<DT>::T* ub = ug + (size_t)b * OW;
  for (int q = threadIdx.x; q < R * C; q += blockDim.x) xs[q] = IO<DT>::ld(xb + q);
  for (int q = threadIdx.x; q < R * R; q += blockDim.x) us[q] = 0.f;
  if (mlp_grad)
    for (int q = threadIdx.x; q < C; q += blockDim.x) IO<DT>::st(mlp_grad + (size_t)b * C + q, IO<DT>::ld(ub + q));
  __syncthreads();
  const int ntril = R * (R - 1) / 2;
  for (int t = threadIdx.x; t < ntril; t += blockDim.x) {
    int i, j;
    tril_unrank(t, i, j);
    const float v = IO<DT>::ld(ub + C + t);
    us[i * R + j] = v;
    us[j * R + i] = v;
  }
  __syncthreads();
  typename IO<DT>::T* gb = grad + (size_t)b * R * C;
  for (int q = threadIdx.x; q < R * C; q += blockDim.x) {
    const int i = q / C, c = q - i * C;
    float s = 0.f;
    for (int k = 0; k < R; ++k) s = fmaf(us[i * R + k], xs[k * C + c], s);
    if (!mlp_grad && i == 0) s += IO<DT>::ld(ub + c);
    IO<DT>::st(gb + q, s);
  }
}

// ------------------------------------------------------------------ C ABI
static int out_width(int R, int C) {
  const int raw = R * (R - 1) / 2 + C;
  return ((raw - 1) / 8 + 1) * 8;
}

extern "C" int dle_dot_interact_out_width(int rows, int cols) { return out_width(rows, cols); }

static bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

extern "C" int dle_dot_interact_fwd(const void* x, void* out, int batch, int rows, int cols,
                                    int dtype, int force_generic, hipStream_t stream) {
  DLE_CHECK_ARG(batch >= 0 && rows >= 1 && cols >= 1, "dot_interact_fwd: bad shape %d %d %d", batch, rows, cols);
  DLE_CHECK_ARG(dtype == DLE_F32 || dtype == DLE_F16 || dtype == DLE_BF16, "dot_interact_fwd: bad dtype %d", dtype);
  if (batch == 0) return 0;   // empty batch: pointers may legitimately be null
  DLE_CHECK_ARG(x && out, "dot_interact_fwd: null pointer");
  const int OW = out_width(rows, cols);
  const bool fast = !force_generic && dtype != DLE_F32 && rows <= 32 && (cols % 16) == 0 &&
                    aligned16(x) && aligned16(out);
  if (fast) {
    const size_t lds = (size_t)4 * OW * 2;
    DLE_CHECK_ARG(lds <= 64 * 1024, "dot_interact_fwd: row too wide for LDS (%d)", OW);
    dim3 grid((batch + 3) / 4), block(256);
    // persistent walk with register prefetch for the widths of the metric (C = 128) and its neighbours
    static const int walk = getenv("DLE_DOT_FWD_WALK") ? atoi(getenv("DLE_DOT_FWD_WALK")) : 1;
    static const int per_cu = getenv("DLE_DOT_FWD_WG_PER_CU") ? atoi(getenv("DLE_DOT_FWD_WG_PER_CU")) : 2;     // (65536 x 27 x 128: 92 us at 2, 97 at 3, 103 at 4-8; one-shot form 125)
    if (walk && (cols == 64 || cols == 128) && batch >= 4096) {
      dim3 wgrid((unsigned)(grid.x < 256u * per_cu ? grid.x : 256u * per_cu));
#define GOW(DT, NK) hipLaunchKernelGGL((dot_fwd_mfma_walk<DT, NK>), wgrid, block, lds, stream, (const unsigned short*)x, \
                                       (unsigned short*)out, batch, rows, OW)
      if (dtype == DLE_F16) { if (cols == 64) GOW(DLE_F16, 4); else GOW(DLE_F16, 8); }
      else { if (cols == 64) GOW(DLE_BF16, 4); else GOW(DLE_BF16, 8); }
#undef GOW
    } else if (dtype == DLE_F16)
      hipLaunchKernelGGL(dot_fwd_mfma<DLE_F16>, grid, block, lds, stream, (const unsigned short*)x,
                         (unsigned short*)out, batch, rows, cols, OW);
    else
      hipLaunchKernelGGL(dot_fwd_mfma<DLE_BF16>, grid, block, lds, stream, (const unsigned short*)x,
                         (unsigned short*)out, batch, rows, cols, OW);
  } else {
    const size_t lds = (size_t)rows * (cols | 1) * 4;
    DLE_CHECK_ARG(lds <= 64 * 1024, "dot_interact_fwd: sample does not fit LDS (%d x %d)", rows, cols);
    dim3 grid(batch), block(256);
    if (dtype == DLE_F32)
      hipLaunchKernelGGL(dot_fwd_generic<DLE_F32>, grid, block, lds, stream, (const float*)x, (float*)out, batch, rows, cols, OW);
    else if (dtype == DLE_F16)
      hipLaunchKernelGGL(dot_fwd_generic<DLE_F16>, grid, block, lds, stream, (const unsigned short*)x, (unsigned short*)out, batch, rows, cols, OW);
    else
      hipLaunchKernelGGL(dot_fwd_generic<DLE_BF16>, grid, block, lds, stream, (const unsigned short*)x, (unsigned short*)out, batch, rows, cols, OW);
  }
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_check_nonfinite(const void* x, float* found_inf, int64_t n, int dtype, hipStream_t stream);

extern "C" int dle_dot_interact_bwd_checked(const void* x, const void* upstream, void* grad, void* mlp_grad,
                                            int batch, int rows, int cols, int dtype, int force_generic,
                                            float* found_inf, hipStream_t stream) {
  DLE_CHECK_ARG(batch >= 0 && rows >= 1 && cols >= 1, "dot_interact_bwd: bad shape %d %d %d", batch, rows, cols);
  DLE_CHECK_ARG(dtype == DLE_F32 || dtype == DLE_F16 || dtype == DLE_BF16, "dot_interact_bwd: bad dtype %d", dtype);
  if (batch == 0) return 0;
  DLE_CHECK_ARG(x && upstream && grad, "dot_interact_bwd: null pointer");
  const int OW = out_width(rows, cols);
  const bool fast = !force_generic && dtype != DLE_F32 && rows <= 32 && (cols == 32 || cols == 64 || cols == 128 || cols == 256) &&
                    aligned16(x) && aligned16(upstream) && aligned16(grad) && aligned16(mlp_grad);   /* NULL is aligned */
  if (fast) {
    const size_t lds = (size_t)4 * (32 * (cols + 8) + 32 * DOT_BWD_USTRIDE + 256) * 2;
    // persistent: two workgroups per CU (180 registers at C = 128), each wavefront walks its samples
    static const int per_cu = getenv("DLE_DOT_BWD_WG_PER_CU") ? atoi(getenv("DLE_DOT_BWD_WG_PER_CU")) : 2;
    int nblk = (batch + 3) / 4;
    if (nblk > 256 * per_cu) nblk = 256 * per_cu;
    dim3 grid(nblk), block(256);
#define GO(DT, NB) hipLaunchKernelGGL((dot_bwd_mfma<DT, NB>), grid, block, lds, stream, (const unsigned short*)x, \
                         (const unsigned short*)upstream, (unsigned short*)grad, (unsigned short*)mlp_grad, batch, rows, cols, OW, found_inf)
#define PICK(DT) do { if (cols <= 32) GO(DT, 1); else if (cols <= 64) GO(DT, 2); else if (cols <= 128) GO(DT, 4); else GO(DT, 8); } while (0)
    if (dtype == DLE_F16) PICK(DLE_F16); else PICK(DLE_BF16);
#undef GO
#undef PICK
  } else {
    const size_t lds = ((size_t)rows * cols + (size_t)rows * rows) * 4;
    DLE_CHECK_ARG(lds <= 64 * 1024, "dot_interact_bwd: sample does not fit LDS (%d x %d)", rows, cols);
    dim3 grid(batch), block(256);
    if (dtype == DLE_F32)
      hipLaunchKernelGGL(dot_bwd_generic<DLE_F32>, grid, block, lds, stream, (const float*)x, (const float*)upstream, (float*)grad, (float*)mlp_grad, batch, rows, cols, OW);
    else if (dtype == DLE_F16)
      hipLaunchKernelGGL(dot_bwd_generic<DLE_F16>, grid, block, lds, stream, (const unsigned short*)x, (const unsigned short*)upstream, (unsigned short*)grad, (unsigned short*)mlp_grad, batch, rows, cols, OW);
    else
      hipLaunchKernelGGL(dot_bwd_generic<DLE_BF16>, grid, block, lds, stream, (const unsigned short*)x, (const unsigned short*)upstream, (unsigned short*)grad, (unsigned short*)mlp_grad, batch, rows, cols, OW);
  }
  DLE_LAUNCH_CHECK();
  if (!fast && found_inf)                       // generic path: the plain sweep (values as stored)
    return dle_check_nonfinite(grad, found_inf, (int64_t)batch * rows * cols, dtype, stream);
  return 0;
}

extern "C" int dle_dot_interact_bwd(const void* x, const void* upstream, void* grad, void* mlp_grad,
                                    int batch, int rows, int cols, int dtype, int force_generic,
                                    hipStream_t stream) {
  return dle_dot_interact_bwd_checked(x, upstream, grad, mlp_grad, batch, rows, cols, dtype, force_generic, nullptr, stream);
}
