#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
__global__ void k(const float* x, unsigned short* a, unsigned short* b, int n) {
  int i = threadIdx.x + blockIdx.x * blockDim.x;
  if (2 * i + 1 >= n) return;
  f2 v = {x[2 * i], x[2 * i + 1]};
  h2 h = __builtin_convertvector(v, h2);
  a[2 * i] = __builtin_bit_cast(unsigned short, h[0]); a[2 * i + 1] = __builtin_bit_cast(unsigned short, h[1]);
  _Float16 s0 = (_Float16)x[2 * i], s1 = (_Float16)x[2 * i + 1];
  b[2 * i] = __builtin_bit_cast(unsigned short, s0); b[2 * i + 1] = __builtin_bit_cast(unsigned short, s1);
}
int main() {
  const int n = 1 << 16;
  float* h = new float[n];
  for (int i = 0; i < n; ++i) { float m = (float)((i * 2654435761u) % 100003) / 100003.0f; h[i] = ldexpf(m * 2 - 1, (i % 40) - 26); }
  float* d; unsigned short *a, *b; hipMalloc(&d, n * 4); hipMalloc(&a, n * 2); hipMalloc(&b, n * 2);
  hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, d, a, b, n);
  unsigned short* ha = new unsigned short[n]; unsigned short* hb = new unsigned short[n];
  hipMemcpy(ha, a, n * 2, hipMemcpyDeviceToHost); hipMemcpy(hb, b, n * 2, hipMemcpyDeviceToHost);
  int diff = 0;
  for (int i = 0; i < n; ++i) if (ha[i] != hb[i]) { if (diff < 8) printf("x=%g (exp %d) pk=%04x scalar=%04x\n", h[i], (i % 40) - 26, ha[i], hb[i]); ++diff; }
  printf("differences: %d of %d\n", diff, n);
  return 0;
}
