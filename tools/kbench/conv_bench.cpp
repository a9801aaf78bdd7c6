// Standalone A/B harness for the 3x3 convolution kernels (no torch: starts in a second on a fresh GPU box).
//   conv_bench [quick]
// For every layer shape: the halo-tile kernel (conv3x3.hip) vs the im2col GEMM (gemm_dma.hip) on the same random
// tensors -- elementwise agreement, a sampled fp64 CPU reference, BatchNorm column statistics, and HIP-event timings.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

extern "C" {
int dle_conv2d_fwd_colstats(const void* x, const void* w, void* y, int N, int H, int W, int C, int Ko, int R, int S,
                            int stride, int pad, int dtype, float* col_partial, int64_t col_partial_bytes, int* groups,
                            hipStream_t stream);
int dle_conv2d_fwd(const void* x, const void* w, void* y, const float* bias, int N, int H, int W, int C, int Ko, int R,
                   int S, int stride, int pad, int dtype, int out_dtype, int act, hipStream_t stream);
int dle_conv2d_dgrad(const void* dy, const void* w, void* dx, const void* addend, int N, int H, int W, int C, int Ko,
                     int R, int S, int stride, int pad, int dtype, hipStream_t stream);
int dle_conv3x3_mode(int mode);
const char* dle_last_error(void);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint32_t rng_state = 12345;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xFFFF) / 32768.0f - 1.0f; }

struct Shape { int N, H, W, C, Ko; };

static int run_shape(const Shape& s, int iters, bool check_all) {
  const int N = s.N, H = s.H, W = s.W, C = s.C, Ko = s.Ko, DT = 2;
  const size_t nx = (size_t)N * H * W * C, ny = (size_t)N * H * W * Ko, nw = (size_t)Ko * 9 * C;
  std::vector<uint16_t> hx(nx), hw(nw), hdy(ny);
  for (auto& v : hx) v = f2bf(frand());
  for (auto& v : hw) v = f2bf(frand() * 0.1f);
  for (auto& v : hdy) v = f2bf(frand());
  uint16_t *dx, *dw, *dy0, *dy1, *ddy, *ddx0, *ddx1;
  CK(hipMalloc(&dx, nx * 2)); CK(hipMalloc(&dw, nw * 2)); CK(hipMalloc(&dy0, ny * 2)); CK(hipMalloc(&dy1, ny * 2));
  CK(hipMalloc(&ddy, ny * 2)); CK(hipMalloc(&ddx0, nx * 2)); CK(hipMalloc(&ddx1, nx * 2));
  CK(hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(ddy, hdy.data(), ny * 2, hipMemcpyHostToDevice));
  const long long M = (long long)N * H * W;
  const int gmax = (int)((M + 127) / 128);
  const size_t sbytes = (size_t)gmax * 2 * Ko * 4;
  float *st0, *st1;
  CK(hipMalloc(&st0, sbytes)); CK(hipMalloc(&st1, sbytes));
  CK(hipMemset(dy0, 0xFF, ny * 2)); CK(hipMemset(dy1, 0xFF, ny * 2));
  CK(hipMemset(ddx0, 0xFF, nx * 2)); CK(hipMemset(ddx1, 0xFF, nx * 2));
  int g0 = 0, g1 = 0, bad = 0;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float t[4] = {0, 0, 0, 0};
  for (int mode = 0; mode < 2; ++mode) {
    dle_conv3x3_mode(mode);
    uint16_t* y = mode ? dy1 : dy0;
    uint16_t* gx = mode ? ddx1 : ddx0;
    float* st = mode ? st1 : st0;
    int* g = mode ? &g1 : &g0;
    for (int it = 0; it < iters + 2; ++it) {
      if (it == 2) CK(hipEventRecord(e0, 0));
      int rc = dle_conv2d_fwd_colstats(dx, dw, y, N, H, W, C, Ko, 3, 3, 1, 1, DT, st, (int64_t)sbytes, g, 0);
      if (rc) { printf("fwd rc %d: %s\n", rc, dle_last_error()); return 1; }
    }
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t[mode], e0, e1));
    for (int it = 0; it < iters + 2; ++it) {
      if (it == 2) CK(hipEventRecord(e0, 0));
      int rc = dle_conv2d_dgrad(ddy, dw, gx, nullptr, N, H, W, C, Ko, 3, 3, 1, 1, DT, 0);
      if (rc) { printf("dgrad rc %d: %s\n", rc, dle_last_error()); return 1; }
    }
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t[2 + mode], e0, e1));
  }
  std::vector<uint16_t> y0(ny), y1(ny), x0(nx), x1(nx);
  CK(hipMemcpy(y0.data(), dy0, ny * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(y1.data(), dy1, ny * 2, hipMemcpyDeviceToHost));
  CK(hipMemcpy(x0.data(), ddx0, nx * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(x1.data(), ddx1, nx * 2, hipMemcpyDeviceToHost));
  // new vs old, every element
  double maxd_f = 0, maxd_b = 0; size_t nbad_f = 0, nbad_b = 0;
  for (size_t i = 0; i < ny; ++i) {
    const float a = bf2f(y0[i]), b = bf2f(y1[i]);
    const double d = fabs(a - b), tol = 0.02 + 0.01 * fabs(a);
    if (!(d <= tol)) ++nbad_f;
    if (d > maxd_f) maxd_f = d;
  }
  for (size_t i = 0; i < nx; ++i) {
    const float a = bf2f(x0[i]), b = bf2f(x1[i]);
    const double d = fabs(a - b), tol = 0.02 + 0.01 * fabs(a);
    if (!(d <= tol)) ++nbad_b;
    if (d > maxd_b) maxd_b = d;
  }
  // sampled (or complete) fp64 reference
  const size_t nsamp = check_all ? ny : 4000;
  double maxr_f = 0, maxr_b = 0; size_t rbad = 0;
  for (size_t si = 0; si < nsamp; ++si) {
    size_t idx = check_all ? si : (size_t)((double)((rng_state = rng_state * 1664525u + 1013904223u) >> 4) / 268435456.0 * ny);
    if (idx >= ny) idx = ny - 1;
    const int ko = idx % Ko; size_t pix = idx / Ko; const int w_ = pix % W; pix /= W; const int h_ = pix % H; const int n = pix / H;
    double acc = 0;
    for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) {
      const int hh = h_ + r - 1, ww = w_ + q - 1;
      if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
      const uint16_t* xp = &hx[(((size_t)n * H + hh) * W + ww) * C];
      const uint16_t* wp = &hw[(((size_t)ko * 3 + r) * 3 + q) * C];
      for (int c = 0; c < C; ++c) acc += (double)bf2f(xp[c]) * bf2f(wp[c]);
    }
    const double d = fabs(acc - bf2f(y1[idx])), tol = 0.02 + 0.01 * fabs(acc);
    if (!(d <= tol)) ++rbad;
    if (d > maxr_f) maxr_f = d;
  }
  const size_t nsampb = check_all ? nx : 4000;
  for (size_t si = 0; si < nsampb; ++si) {
    size_t idx = check_all ? si : (size_t)((double)((rng_state = rng_state * 1664525u + 1013904223u) >> 4) / 268435456.0 * nx);
    if (idx >= nx) idx = nx - 1;
    const int c = idx % C; size_t pix = idx / C; const int w_ = pix % W; pix /= W; const int h_ = pix % H; const int n = pix / H;
    double acc = 0;
    for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) {
      const int pp = h_ + 1 - r, qq = w_ + 1 - q;
      if (pp < 0 || pp >= H || qq < 0 || qq >= W) continue;
      const uint16_t* gp = &hdy[(((size_t)n * H + pp) * W + qq) * Ko];
      for (int ko = 0; ko < Ko; ++ko) acc += (double)bf2f(gp[ko]) * bf2f(hw[(((size_t)ko * 3 + r) * 3 + q) * C + c]);
    }
    const double d = fabs(acc - bf2f(x1[idx])), tol = 0.02 + 0.01 * fabs(acc);
    if (!(d <= tol)) ++rbad;
    if (d > maxr_b) maxr_b = d;
    if (getenv("CONV_DUMP") && nx <= 64) printf("  c %2d ref %8.4f new %8.4f old %8.4f\n", c, acc, bf2f(x1[idx]), bf2f(x0[idx]));
  }
  // column statistics: fold the per-tile partials on the host
  std::vector<float> s0((size_t)g0 * 2 * Ko), s1((size_t)g1 * 2 * Ko);
  CK(hipMemcpy(s0.data(), st0, s0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(s1.data(), st1, s1.size() * 4, hipMemcpyDeviceToHost));
  double maxs = 0; size_t sbad = 0;
  for (int which = 0; which < 2; ++which) for (int k = 0; k < Ko; ++k) {
    double a = 0, b = 0;
    for (int g = 0; g < g0; ++g) a += s0[((size_t)g * 2 + which) * Ko + k];
    for (int g = 0; g < g1; ++g) b += s1[((size_t)g * 2 + which) * Ko + k];
    const double d = fabs(a - b), tol = 1e-3 * (fabs(a) + (double)M * 0.01);
    if (!(d <= tol)) ++sbad;
    if (d / (fabs(a) + 1) > maxs) maxs = d / (fabs(a) + 1);
  }
  const double gf = 2.0 * M * Ko * 9.0 * C * 1e-9;
  printf("N%d %dx%d C%d K%d | fwd old %.1f us (%.0f TF) new %.1f us (%.0f TF) x%.2f | dgrad old %.1f us new %.1f us (%.0f TF) x%.2f | "
         "new-vs-old maxdiff %.3g/%.3g bad %zu/%zu | vs fp64 maxerr %.3g/%.3g bad %zu | stats rel %.2g bad %zu (groups %d/%d)\n",
         N, H, W, C, Ko, t[0] / iters * 1e3, gf / (t[0] / iters), t[1] / iters * 1e3, gf / (t[1] / iters), t[0] / t[1],
         t[2] / iters * 1e3, t[3] / iters * 1e3, gf / (t[3] / iters), t[2] / t[3], maxd_f, maxd_b, nbad_f, nbad_b, maxr_f, maxr_b, rbad, maxs,
         sbad, g0, g1);
  bad = (nbad_f || nbad_b || rbad || sbad) ? 1 : 0;
  hipFree(dx); hipFree(dw); hipFree(dy0); hipFree(dy1); hipFree(ddy); hipFree(ddx0); hipFree(ddx1); hipFree(st0); hipFree(st1);
  return bad;
}

int main(int argc, char** argv) {
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  int bad = 0;
  // small / ragged shapes, every output element against the fp64 reference
  if (getenv("CONV_DUMP")) { Shape d1 = {1, 1, 1, 64, 64}; run_shape(d1, 1, true); return 0; }
  const Shape small[] = {{3, 5, 7, 64, 64}, {2, 9, 6, 128, 64}, {2, 7, 7, 64, 128}, {1, 14, 14, 128, 256}, {5, 3, 3, 64, 64}, {1, 1, 1, 64, 64},
                         {2, 20, 33, 64, 64}};
  for (const Shape& s : small) bad |= run_shape(s, 2, true);
  if (!quick) {
    const Shape big[] = {{256, 56, 56, 64, 64}, {256, 28, 28, 128, 128}, {256, 14, 14, 256, 256}, {256, 7, 7, 512, 512}};
    for (const Shape& s : big) bad |= run_shape(s, 20, false);
  }
  printf(bad ? "CONV_BENCH FAILED\n" : "CONV_BENCH OK\n");
  return bad;
}
