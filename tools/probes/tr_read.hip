#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned short us4;
typedef __attribute__((ext_vector_type(4))) short s4;
__global__ void k(unsigned short* out, int pitch) {
  __shared__ __attribute__((aligned(16))) unsigned short T[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) T[i] = (unsigned short)i;   // T[r][c] = r*64+c  (pitch 64)
  __syncthreads();
  int l = threadIdx.x;
  int g = l >> 4, i = l & 15;
  // lane i of group g points at row (g*8 + i/4), cols (i%4)*4..+3
  const unsigned short* p = &T[(g * 8 + i / 4) * pitch + (i % 4) * 4];
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 64);
  unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" (%d,%d)", h[l*4+j] / 64, h[l*4+j] % 64); printf("\n"); }
  return 0;
}
