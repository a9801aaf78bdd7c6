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
    unsigned int x = __builtin_bit_cast(unsigned int, f);
    if ((x & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((x >> 16) | 0x40);  // quiet NaN
    unsigned int lsb = (x >> 16) & 1u;
    x += 0x7fffu + lsb;           // round-to-nearest-even
    return (unsigned short)(x >> 16);
  }
};

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
