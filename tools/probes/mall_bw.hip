// Does a re-read of a buffer that fits the 256 MB Infinity Cache run faster than an HBM stream?  Sweep of buffer sizes, each read
// 10 times back to back by a 16-byte-per-lane streaming reduction; and the REVERSED second pass over a large buffer (the tail a
// forward pass left in the cache is what a backward-ordered pass touches first).   hipcc --offload-arch=gfx950 -O3 mall_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned int u4;

template <bool REV>
__global__ __launch_bounds__(256) void rd(const u4* __restrict__ a, long long n16, unsigned* out) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned acc = 0;
  for (; i + stride < n16; i += 2 * stride) {
    const long long j0 = REV ? n16 - 1 - i : i, j1 = REV ? n16 - 1 - (i + stride) : i + stride;
    const u4 v = a[j0], w = a[j1];
    acc += v[0] ^ v[1] ^ v[2] ^ v[3] ^ w[0] ^ w[1] ^ w[2] ^ w[3];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  u4* a; unsigned* out;
  const long long maxb = 2LL << 30;
  hipMalloc(&a, maxb); hipMalloc(&out, 4);
  hipMemset(a, 1, maxb);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  const long long sizes[] = {32LL << 20, 64LL << 20, 128LL << 20, 192LL << 20, 256LL << 20, 384LL << 20, 512LL << 20, 1024LL << 20};
  for (long long b : sizes) {
    const long long n16 = b / 16;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(rd<false>, dim3(1024), dim3(256), 0, 0, a, n16, out);
    hipEventRecord(s);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(rd<false>, dim3(1024), dim3(256), 0, 0, a, n16, out);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    printf("re-read %5lld MB: %.3f ms per pass, %.2f TB/s\n", b >> 20, ms / 10, b / (ms / 10 * 1e-3) / 1e12);
  }
  // forward pass then reversed pass over 411 MB and 822 MB (the BatchNorm backward pair: reduce, then apply)
  for (long long b : {411LL << 20, 822LL << 20}) {
    const long long n16 = b / 16;
    float tf = 0, tr = 0, tf2 = 0;
    for (int it = 0; it < 6; ++it) {
      float ms;
      hipLaunchKernelGGL(rd<false>, dim3(1024), dim3(256), 0, 0, a, n16, out);
      hipEventRecord(s); hipLaunchKernelGGL(rd<true>, dim3(1024), dim3(256), 0, 0, a, n16, out); hipEventRecord(e); hipEventSynchronize(e);
      hipEventElapsedTime(&ms, s, e); if (it) tr += ms;
      hipLaunchKernelGGL(rd<false>, dim3(1024), dim3(256), 0, 0, a, n16, out);
      hipEventRecord(s); hipLaunchKernelGGL(rd<false>, dim3(1024), dim3(256), 0, 0, a, n16, out); hipEventRecord(e); hipEventSynchronize(e);
      hipEventElapsedTime(&ms, s, e); if (it) tf += ms;
    }
    printf("%4lld MB second pass: same order %.3f ms (%.2f TB/s), reversed %.3f ms (%.2f TB/s)\n", b >> 20, tf / 5, b / (tf / 5 * 1e-3) / 1e12,
           tr / 5, b / (tr / 5 * 1e-3) / 1e12);
    (void)tf2;
  }
  return 0;
}
