// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.
// wave = 64 lanes everywhere; no CUDA-compat paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DLE_WAVE 64

// dtype codes of the C ABI (include/dle_mi355x.h)
enum { DLE_F32 = 0, DLE_F16 = 1, DLE_BF16 = 2 };

typedef __attribute__((ext_vector_type(8))) _Float16 half8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 half4_t;
typedef __attribute__((ext_vector_type(2))) _Float16 half2_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) unsigned short ushort8_t;
typedef __attribute__((ext_vector_type(4))) unsigned short ushort4_t;
typedef __attribute__((ext_vector_type(4))) float float4_t;
typedef __attribute__((ext_vector_type(16))) float float16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int uint4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int uint2_t;

extern "C" void dle_set_error(const char* fmt, ...);

#define DLE_CHECK_ARG(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      dle_set_error(__VA_ARGS__);           \
      return -1;                            \
    }                                       \
  } while (0)

#define DLE_LAUNCH_CHECK()                                        \
  do {                                                            \
    hipError_t e__ = hipGetLastError();                           \
    if (e__ != hipSuccess) {                                      \
      dle_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return (int)e__;                                            \
    }                                                             \
  } while (0)

// ---- 16-bit float <-> f32 bit helpers (storage type = unsigned short) -------------
template <int DT> struct Elem;   // DT = DLE_F16 / DLE_BF16
template <> struct Elem<DLE_F16> {
  static __device__ __forceinline__ float to_f32(unsigned short u) {
    _Float16 h = __builtin_bit_cast(_Float16, u);
    return (float)h;
  }
  static __device__ __forceinline__ unsigned short from_f32(float f) {
    _Float16 h = (_Float16)f;   // round-to-nearest-even
    return __builtin_bit_cast(unsigned short, h);
  }
};
template <> struct Elem<DLE_BF16> {
  static __device__ __forceinline__ float to_f32(unsigned short u) {
    return __builtin_bit_cast(float, ((unsigned int)u) << 16);
  }
  static __device__ __forceinline__ unsigned short from_f32(float f) {
    return __builtin_bit_cast(unsigned short, (__bf16)f);   // v_cvt_pk_bf16_f32: round-to-nearest-even in hardware
  }
};

// 8 fp32 -> 8 packed 16-bit values with the paired hardware converts (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32):
// 4 VALU instructions instead of ~50 for the bit-twiddled rounding -- the conversions, not HBM, were the limit
// of every 16-byte-per-lane epilogue / elementwise kernel before this.
typedef __attribute__((ext_vector_type(2))) float float2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
template <int DT>
__device__ __forceinline__ ushort8_t pack8(const float* v) {
  uint4_t o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float2_t f = {v[2 * i], v[2 * i + 1]};
    if (DT == DLE_F16) {
      // scalar converts + pack: the paired float2 -> half2 convert (v_cvt_pk_f16_f32 as emitted by hipcc 7.2 for
      // __builtin_convertvector) returns wrong values on gfx950 (tools/probes: half of the elements differ)
      o[i] = (unsigned int)Elem<DLE_F16>::from_f32(f[0]) | ((unsigned int)Elem<DLE_F16>::from_f32(f[1]) << 16);
    } else {
      o[i] = __builtin_bit_cast(unsigned int, __builtin_convertvector(f, bf16x2_t));
    }
  }
  return __builtin_bit_cast(ushort8_t, o);
}
// 8 packed 16-bit values -> 8 fp32 (bf16: one shift / mask per element)
template <int DT>
__device__ __forceinline__ void unpack8(ushort8_t u8, float* v) {
  const uint4_t u = __builtin_bit_cast(uint4_t, u8);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (DT == DLE_F16) {
      // element-wise on purpose (see pack8): vector half2 <-> float2 converts are miscompiled for gfx950
      v[2 * i] = Elem<DLE_F16>::to_f32((unsigned short)(u[i] & 0xffffu));
      v[2 * i + 1] = Elem<DLE_F16>::to_f32((unsigned short)(u[i] >> 16));
    } else {
      v[2 * i] = __builtin_bit_cast(float, u[i] << 16);
      v[2 * i + 1] = __builtin_bit_cast(float, u[i] & 0xffff0000u);
    }
  }
}

// tanh(x) = 1 - 2 / (exp(2x) + 1): one v_exp_f32 + one v_rcp_f32.  Absolute error ~1e-7 (saturates cleanly to
// +-1 for large |x|), far below the 16-bit rounding of every tensor it feeds.
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);      // exp(2x) = 2^(2x log2 e)
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blockDim.x = multiple of 64 (<= 1024); red = LDS scratch of >= 16 floats
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
