// Read-only HBM bandwidth probe: what does a streaming REDUCTION (no writes) reach on one MI355X, as a function of
// loads in flight per lane, workgroups per CU and the number of streams?   hipcc --offload-arch=gfx950 -O3 read_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned int u4;

template <int U, int NS>
__global__ __launch_bounds__(256) void rd(const u4* __restrict__ a, const u4* __restrict__ b, long long n16, unsigned* out) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned acc = 0;
  for (; i + (U - 1) * stride < n16; i += U * stride) {
    u4 v[U], w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { v[u] = a[i + u * stride]; if (NS == 2) w[u] = b[i + u * stride]; }
#pragma unroll
    for (int u = 0; u < U; ++u) { acc += v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3]; if (NS == 2) acc += w[u][0] ^ w[u][1] ^ w[u][2] ^ w[u][3]; }
  }
  if (acc == 0x12345678u) out[0] = acc;
}
// contiguous-per-block variant: each block sweeps its own contiguous slab (the BN reduce kernels' pattern)
template <int U>
__global__ __launch_bounds__(256) void rd_slab(const u4* __restrict__ a, long long n16, unsigned* out) {
  const long long per = (n16 + gridDim.x - 1) / gridDim.x;
  long long i = (long long)blockIdx.x * per + threadIdx.x;
  const long long end = min(n16, (long long)(blockIdx.x + 1) * per);
  unsigned acc = 0;
  for (; i + (U - 1) * 256 < end; i += U * 256) {
    u4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = a[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <class F> float timeit(F f) {
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(s);
  for (int i = 0; i < 10; ++i) f();
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e); return ms / 10;
}
int main() {
  const long long bytes = 1LL << 30;             // 1 GiB per stream (> Infinity Cache)
  u4 *a, *b; unsigned* out;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&out, 4);
  hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
  const long long n16 = bytes / 16;
  int grids[] = {512, 1024, 2048, 4096, 8192};
  for (int g : grids) {
#define RUN(U, NS) { float ms = timeit([&] { hipLaunchKernelGGL((rd<U, NS>), dim3(g), dim3(256), 0, 0, a, b, n16, out); }); \
    printf("grid %5d  U %d  streams %d : %.3f ms  %.2f TB/s\n", g, U, NS, ms, NS * bytes / ms / 1e9); }
    RUN(1, 1) RUN(2, 1) RUN(4, 1) RUN(8, 1) RUN(4, 2) RUN(8, 2)
#define RUNS(U) { float ms = timeit([&] { hipLaunchKernelGGL((rd_slab<U>), dim3(g), dim3(256), 0, 0, a, n16, out); }); \
    printf("grid %5d  U %d  slab     : %.3f ms  %.2f TB/s\n", g, U, ms, bytes / ms / 1e9); }
    RUNS(4) RUNS(8)
  }
  return 0;
}
