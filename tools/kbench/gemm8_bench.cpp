// Standalone harness for the ping-pong GEMM (gemm8) (csrc/gemm8.hip compiled INTO this file, optionally with -DG8_TIMING):
// HIP-event time per shape and, with G8_TIMING, shader-clock stamps around every item's K loop and epilogue.
//   gemm8_bench M N K layout(nt|nn|tn) [epi: plain|bias|gelu|add|mul] [splitk]
#include "../../deeplearningexamples_amd/csrc/gemm8.hip"
#include "../../deeplearningexamples_amd/csrc/gemm8_epi1.hip"
#include "../../deeplearningexamples_amd/csrc/gemm8_epi2.hip"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <algorithm>

extern "C" void dle_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static uint32_t rng_state = 12345;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xFFFF) / 32768.0f - 1.0f; }

int main(int argc, char** argv) {
  if (argc < 5) { printf("usage: gemm8_bench M N K nt|nn|tn [plain|bias|gelu|add|mul] [splitk]\n"); return 1; }
  const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]);
  const char* lay = argv[4];
  const char* epi = argc > 5 ? argv[5] : "plain";
  const int splitk = argc > 6 ? atoi(argv[6]) : 1;
  const int a_kc = lay[0] == 'n', b_kc = lay[1] == 't';
  setenv("DLE_GEMM_8PH_MIN_ITEMS", "1", 1);
  const size_t na = (size_t)M * K, nb = (size_t)N * K, nc = (size_t)M * N;
  std::vector<uint16_t> ha(na), hb(nb), hs(nc);
  for (auto& v : ha) v = f2bf(frand());
  for (auto& v : hb) v = f2bf(frand());
  for (auto& v : hs) v = f2bf(frand());
  uint16_t *dA, *dB, *dS, *dAux; void* dC; float *dBias, *dWs = nullptr;
  CK(hipMalloc(&dA, na * 2)); CK(hipMalloc(&dB, nb * 2)); CK(hipMalloc(&dS, nc * 2)); CK(hipMalloc(&dAux, nc * 2));
  CK(hipMalloc(&dC, nc * 4)); CK(hipMalloc(&dBias, N * 4));
  CK(hipMemcpy(dA, ha.data(), na * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hb.data(), nb * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dS, hs.data(), nc * 2, hipMemcpyHostToDevice)); CK(hipMemset(dBias, 0, N * 4));
  if (splitk > 1) CK(hipMalloc(&dWs, nc * 4 * splitk));
  int act = ACT_NONE; const float* bias = nullptr; void* aux = nullptr; const void* src = nullptr;
  if (!strcmp(epi, "bias")) bias = dBias;
  else if (!strcmp(epi, "gelu")) { bias = dBias; act = ACT_GELU_DAUX; aux = dAux; }
  else if (!strcmp(epi, "add")) { act = ACT_ADD; src = dS; }
  else if (!strcmp(epi, "mul")) { act = ACT_MUL; src = dS; }
  const int lda = a_kc ? K : M, ldb = b_kc ? K : N;
  const int out_dt = splitk > 1 ? DLE_F32 : DLE_BF16;
  auto run = [&]() {
    int rc = dle_gemm8_try(dA, dB, dC, aux, bias, src, M, N, K, lda, ldb, N, a_kc, b_kc, DLE_BF16, out_dt, act, splitk, 0, 1.0f, dWs,
                           nullptr, 0);
    if (rc != 1) { printf("gemm8_try rc=%d\n", rc); exit(3); }
  };
#ifdef G8_TIMING
  const int NI = 32, NB = 256;
  unsigned long long* dDbg; CK(hipMalloc(&dDbg, (size_t)NB * 2 * NI * 4 * 8)); CK(hipMemset(dDbg, 0, (size_t)NB * 2 * NI * 4 * 8));
  g8_dbg_ptr = nullptr;
#endif
  for (int i = 0; i < 3; ++i) run();
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 10;
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) run();
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters;
  printf("%s %s M=%d N=%d K=%d splitk=%d: %.1f us  %.1f TFLOP/s\n", lay, epi, M, N, K, splitk, us, 2.0 * M * N * K / us / 1e6);
#ifdef G8_TIMING
  g8_dbg_ptr = dDbg; g8_dbg_items = NI;
  run(); CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h((size_t)NB * 2 * NI * 4);
  CK(hipMemcpy(h.data(), dDbg, h.size() * 8, hipMemcpyDeviceToHost));
  // per item index: mean K-loop cycles, epilogue cycles, gap to the next item's start; per block: first start / last end
  for (int g = 0; g < 2; ++g) {
    for (int it = 0; it < NI; ++it) {
      double kl = 0, ep = 0, gap = 0; int n = 0, ng = 0;
      for (int b = 0; b < NB; ++b) {
        const unsigned long long* q = &h[(((size_t)b * 2 + g) * NI + it) * 4];
        if (!q[2]) continue;
        kl += (double)(q[1] - q[0]); ep += (double)(q[2] - q[1]); ++n;
        if (it + 1 < NI) { const unsigned long long* q2 = q + 4; if (q2[0]) { gap += (double)(q2[0] - q[2]); ++ng; } }
      }
      if (n) printf("  group %d item %2d: n=%3d  kloop %8.0f cyc  epilogue %7.0f cyc  gap-to-next %6.0f cyc\n", g, it, n, kl / n, ep / n, ng ? gap / ng : 0.0);
    }
  }
#ifdef G8_TIMING_KT
  {  // per-K-tile stamps of item G8_TIMING_KT (slots NI-16.. hold up to 60 stamps; stamp k is taken BEFORE K tile k; the last one after the loop)
    const int base = NI - 16;
    for (int g = 0; g < 2; ++g) {
      printf("  group %d, item %d: cycles per K tile (mean over blocks), K tiles 0..:", g, G8_TIMING_KT);
      for (int k = 0; k < 59; ++k) {
        double d = 0; int n = 0;
        for (int b = 0; b < NB; ++b) {
          const unsigned long long* q = &h[(((size_t)b * 2 + g) * NI + base) * 4];
          if (q[k] && q[k + 1]) { d += (double)(q[k + 1] - q[k]); ++n; }
        }
        if (n) printf(" %.0f", d / n);
      }
      printf("\n");
    }
  }
#endif
  unsigned long long tmin = ~0ull, tmax = 0;
  std::vector<double> ends;
  for (int b = 0; b < NB; ++b) { unsigned long long last = 0; for (int it = 0; it < NI; ++it) { const unsigned long long* q = &h[(((size_t)b * 2) * NI + it) * 4]; if (q[0] && q[0] < tmin) tmin = q[0]; if (q[2] > last) last = q[2]; } if (last) { ends.push_back((double)last); if (last > tmax) tmax = last; } }
  std::sort(ends.begin(), ends.end());
  if (!ends.empty()) printf("  kernel span %llu cyc; block end spread: p0 %.0f p50 %.0f p100 %.0f (relative to first start)\n", tmax - tmin, ends.front() - tmin, ends[ends.size() / 2] - tmin, ends.back() - tmin);
#endif
  return 0;
}
