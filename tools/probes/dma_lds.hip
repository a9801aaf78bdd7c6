#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) unsigned short us8;
typedef __attribute__((ext_vector_type(4))) int i4;
__global__ void k(const unsigned short* src, unsigned short* out, int n_valid_bytes) {
  __shared__ __attribute__((aligned(16))) unsigned short T[64 * 8];
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n_valid_bytes, 0x00020000);
  int l = threadIdx.x;
  unsigned voff = (l ^ 1) * 16;            // per-lane global address, swapped neighbours
  if (l == 5) voff = 0xFFFFFFF0u;          // forced out of bounds -> must read zeros
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)T, 16, voff, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int j = 0; j < 8; ++j) out[l * 8 + j] = T[l * 8 + j];
}
int main() {
  unsigned short h[64 * 8], *d, *o;
  for (int i = 0; i < 512; ++i) h[i] = i;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(h));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 60 * 16);
  unsigned short r[512]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 1) if (l < 8 || l > 56) printf("lane %2d: %d %d ... %d\n", l, r[l*8], r[l*8+1], r[l*8+7]);
  return 0;
}
