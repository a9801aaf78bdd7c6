// Standalone timing harness for the BatchNorm passes of the ResNet-50 step (no torch): HIP-event timings of
// dle_bn_bwd_reduce / dle_bn_bwd_apply / dle_bn_fwd_apply on the layer shapes of batch 256, for the tuning knobs of
// dle_bn_tune (workgroup target, rows in flight).   bn_bench [want_blocks ...]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

extern "C" {
int64_t dle_bn_workspace_bytes(int64_t M, int C);
int dle_bn_tune(int want_blocks, int bwd_rows_in_flight);
int dle_bn_tune_apply(int trips, int grid_cap);
int dle_bn_fwd_apply(const void* x, const void* residual, void* y, void* relu_mask, const float* mean, const float* rstd,
                     const float* gamma, const float* beta, int64_t M, int C, int relu, int dtype, hipStream_t stream);
int dle_bn_bwd_reduce(const void* dy, const void* y, const void* relu_mask, const void* x, const float* mean,
                      const float* rstd, float* dgamma, float* dbeta, int64_t M, int C, int accumulate, void* workspace,
                      int64_t workspace_bytes, int dtype, hipStream_t stream);
int dle_bn_bwd_apply(const void* dy, const void* y, const void* relu_mask, const void* x, void* dx, void* g_out,
                     const float* mean, const float* rstd, const float* gamma, const float* dgamma, const float* dbeta,
                     int64_t M, int C, int dtype, hipStream_t stream);
const char* dle_last_error(void);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

template <class F> static float timeit(F f, int iters) {
  hipEvent_t s, e; CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
  for (int i = 0; i < 3; ++i) f();
  CK(hipEventRecord(s));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(e)); CK(hipEventSynchronize(e));
  float ms; CK(hipEventElapsedTime(&ms, s, e));
  return ms / iters * 1e3f;
}

int main(int argc, char** argv) {
  struct Sh { long long M; int C; int calls; };
  const Sh shapes[] = {{802816, 256, 4}, {802816, 64, 6}, {200704, 512, 5}, {200704, 128, 7}, {50176, 1024, 7},
                       {50176, 256, 11}, {12544, 2048, 4}, {12544, 512, 5}, {3211264, 64, 1}};
  const size_t maxe = 802816ULL * 256;
  uint16_t *x, *dy, *dx, *g;
  unsigned char* mask;
  float *mean, *rstd, *gamma, *dgamma, *dbeta, *ws;
  CK(hipMalloc(&x, maxe * 2)); CK(hipMalloc(&dy, maxe * 2)); CK(hipMalloc(&dx, maxe * 2)); CK(hipMalloc(&g, maxe * 2));
  CK(hipMalloc(&mask, maxe / 8));
  CK(hipMemset(x, 0x3c, maxe * 2)); CK(hipMemset(dy, 0x3b, maxe * 2)); CK(hipMemset(mask, 0xa5, maxe / 8));
  CK(hipMalloc(&mean, 4096 * 4)); CK(hipMalloc(&rstd, 4096 * 4)); CK(hipMalloc(&gamma, 4096 * 4));
  CK(hipMalloc(&dgamma, 4096 * 4)); CK(hipMalloc(&dbeta, 4096 * 4));
  CK(hipMemset(mean, 0, 4096 * 4)); CK(hipMemset(rstd, 0, 4096 * 4)); CK(hipMemset(gamma, 0, 4096 * 4));
  const size_t wsb = 64ULL << 20;
  CK(hipMalloc(&ws, wsb));
  std::vector<int> wants;
  for (int i = 1; i < argc; ++i) wants.push_back(atoi(argv[i]));
  if (wants.empty()) wants = {1024};
  // argument >= 0: reduce sweep (want_blocks [+ cap * 100000]); argument < 0: apply sweep with grid cap = -argument
  for (int want : wants)
    for (int u : {2, 4, 8}) {
      if (want < 0) { dle_bn_tune_apply(u == 8 ? 1 : u, -want); }
      else dle_bn_tune(want, u);
      double tot_red = 0, tot_app = 0, tot_fwd = 0;
      for (const Sh& s : shapes) {
        const double bytes = (double)s.M * s.C * 2;
        const float t_red = want < 0 ? 1.f : timeit([&] { if (dle_bn_bwd_reduce(dy, nullptr, mask, x, mean, rstd, dgamma, dbeta, s.M, s.C, 0, ws, wsb, 2, 0)) { printf("%s\n", dle_last_error()); exit(3); } }, 20);
        float t_app = 0, t_fwd = 0;
        if (want < 0) {
          t_app = timeit([&] { dle_bn_bwd_apply(dy, nullptr, mask, x, dx, s.C >= 256 ? g : nullptr, mean, rstd, gamma, dgamma, dbeta, s.M, s.C, 2, 0); }, 20);
          t_fwd = timeit([&] { dle_bn_fwd_apply(x, s.C >= 256 ? dy : nullptr, dx, mask, mean, rstd, gamma, dbeta, s.M, s.C, 1, 2, 0); }, 20);
        }
        printf("want %5d U %d  M %8lld C %5d : reduce+finish %7.1f us %5.2f TB/s | bwd_apply %7.1f us %5.2f TB/s | fwd_apply %7.1f us %5.2f TB/s\n",
               want, u, s.M, s.C, t_red, 2.0625 * bytes / t_red / 1e6, t_app, (s.C >= 256 ? 4.0625 : 3.0625) * bytes / (t_app + 1e-9) / 1e6,
               t_fwd, (s.C >= 256 ? 3.0625 : 2.0625) * bytes / (t_fwd + 1e-9) / 1e6);
        tot_red += t_red * s.calls; tot_app += t_app * s.calls; tot_fwd += t_fwd * s.calls;
      }
      printf("== want %d U %d: per step reduce %.2f ms  bwd_apply %.2f ms  fwd_apply %.2f ms\n", want, u, tot_red / 1e3, tot_app / 1e3, tot_fwd / 1e3);
    }
  return 0;
}
