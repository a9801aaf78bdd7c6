// PROBE (not part of the library): how fast is a 256x256 GEMM main loop on one MI355X when the operands are REGISTER-STAGED
// (global_load -> VGPR -> ds_write_b128) instead of LDS-DMA, with ONE wavefront per SIMD and the accumulators in AGPRs?
//
// Why: DESIGN.md section 7 ("what comes next", item 1).  The library's 8-wave 256x256 kernel (csrc/gemm_dma.hip) feeds LDS with
// `buffer_load ... lds`; measured in round 3 it tops out at ~1.23 PFLOP/s in the loop (0.93 in BERT's step): a CU sustains only
// 55-60 GB/s through the DMA path and every 1 KiB piece costs the issuing wave 100+ cycles, and at 256 VGPRs (8 spilled) the
// kernel has no registers to stage operands.  The shape with registers to spare is 4 waves x (128 x 128 outputs): 256 accumulators
// in AGPRs, 256 VGPRs for fragments (64) + two K tiles of staging (128).
//
// C[M, N] (bf16) = A[M, K] B[N, K]^T, both k-contiguous (the "NT" forward layout); M, N multiples of 256, K of 64.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/gemm4w.hip -o /tmp/gemm4w && /tmp/gemm4w
// prints TFLOP/s for a few BERT / DLRM shapes next to a correctness check against a scalar reference on sampled outputs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) unsigned short ushort8_t;
typedef __attribute__((ext_vector_type(4))) unsigned short ushort4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float float16_t;

#define TM 256
#define TN 256
#define BK 64
#define STAGE_HALVES ((TM + TN) * BK)            // 32768 halves = 64 KiB per stage

// LDS image of an operand tile: [256 rows][8 chunks of 8 halves]; chunk c of row r sits in slot c ^ ((r >> 1) & 7): a fragment
// read (32 rows x one chunk) and a staging write (8 rows x 8 chunks per wave) are both conflict-free
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * BK + ((chunk ^ ((row >> 1) & 7)) << 3); }

__device__ __forceinline__ float bf2f(unsigned short u) { return __builtin_bit_cast(float, ((unsigned)u) << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }

struct Stage8 { ushort8_t a[8], b[8]; };        // one K tile of one thread: 8 + 8 chunks of 16 bytes (64 VGPRs)

// VAR bit 0: no output stores (main loop alone); VAR >= 2: staging writes / loads spread over k steps 1..3.  The loop body is branch-free (the last two K tiles are peeled), so that the staging
// writes / loads sit in the same basic block as the MFMAs of k step 3 (first version, one `if` per tile: 908 TFLOP/s at 8192^3).
template <int VAR>
__global__ __launch_bounds__(256) void gemm4w_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B,
                                                     unsigned short* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* lds = (unsigned short*)smem_raw;          // [2 stages][A tile | B tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;                   // wave grid 2 x 2, 128 x 128 outputs each
  const int tiles_n = N / TN;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int m0 = tm * TM, n0 = tn * TN;
  // staging assignment: chunk j of this thread = row (tid >> 3) + 32 j, k chunk tid & 7 (8 lanes cover one 128-byte row segment).
  // Addresses = a wave-uniform base (SGPRs: tile origin + 32 j rows + K tile) + ONE 32-bit per-thread byte offset.
  const int srow = tid >> 3, skc = tid & 7;
  const unsigned voff = (unsigned)(((long long)srow * K + skc * 8) * 2);
  const char* baseA = (const char*)(A + (long long)m0 * K);
  const char* baseB = (const char*)(B + (long long)n0 * K);
  const long long rstep = 64LL * K;                           // bytes per 32 rows
  auto gload = [&](Stage8& s, int kt) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s.a[j] = *(const ushort8_t*)(baseA + j * rstep + (long long)kt * (BK * 2) + voff);
      s.b[j] = *(const ushort8_t*)(baseB + j * rstep + (long long)kt * (BK * 2) + voff);
    }
  };
  const int loff = lds_off(srow, skc);                        // (row + 32 j keeps (row >> 1) & 7: the slot is the same for every j)
  auto lstore = [&](const Stage8& s, int stage) __attribute__((always_inline)) {
    unsigned short* ta = lds + stage * STAGE_HALVES + loff;
    unsigned short* tb = ta + TM * BK;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      *(ushort8_t*)(ta + j * 32 * BK) = s.a[j];
      *(ushort8_t*)(tb + j * 32 * BK) = s.b[j];
    }
  };
  float16_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = K / BK;
  Stage8 st;                                                  // one K tile in flight in registers (64 VGPRs)
  gload(st, 0);
  lstore(st, 0);                                              // (the compiler waits for the loads here)
  if (nk > 1) gload(st, 1);
  const int fr = lane & 31, fh = lane >> 5;
  auto body = [&](int kt, auto STORE, auto LOAD) __attribute__((always_inline)) {
    __syncthreads();                                          // stage kt & 1 is complete; stage (kt + 1) & 1 is no longer read
    const unsigned short* ta = lds + (kt & 1) * STAGE_HALVES + (wm * 128 + fr) * BK;
    const unsigned short* tb = lds + (kt & 1) * STAGE_HALVES + TM * BK + (wn * 128 + fr) * BK;
    const int sw = (fr >> 1) & 7;                             // (the 32-row blocks start at multiples of 32: same swizzle term)
    ushort8_t fa[2][4], fb[2][4];                             // fragments of k step ks + 1 are requested before the MFMAs of ks
    auto fread = [&](int buf, int ks) __attribute__((always_inline)) {
      const int slot = ((2 * ks + fh) ^ sw) << 3;
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[buf][i] = *(const ushort8_t*)(ta + i * 32 * BK + slot);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[buf][j] = *(const ushort8_t*)(tb + j * 32 * BK + slot);
    };
    fread(0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < 3) fread((ks + 1) & 1, ks + 1);
      if (VAR < 2) {
        if (ks == 3) {                                        // next tile: registers -> the other stage (its loads have had three
          if (decltype(STORE)::value) lstore(st, (kt + 1) & 1); // k steps to land), then the registers take the tile after it
          if (decltype(LOAD)::value) gload(st, kt + 2);
        }
      } else if (ks >= 1) {                                   // VAR >= 2: the same traffic spread over k steps 1..3 (3 + 3 + 2 chunk pairs)
        const int j0 = ks == 1 ? 0 : ks == 2 ? 3 : 6, j1 = ks == 1 ? 3 : ks == 2 ? 6 : 8;
        unsigned short* wa = lds + ((kt + 1) & 1) * STAGE_HALVES + loff;
        unsigned short* wb = wa + TM * BK;
#pragma unroll
        for (int j = j0; j < j1; ++j) {
          if (decltype(STORE)::value) { *(ushort8_t*)(wa + j * 32 * BK) = st.a[j]; *(ushort8_t*)(wb + j * 32 * BK) = st.b[j]; }
          if (decltype(LOAD)::value) {
            st.a[j] = *(const ushort8_t*)(baseA + j * rstep + (long long)(kt + 2) * (BK * 2) + voff);
            st.b[j] = *(const ushort8_t*)(baseB + j * rstep + (long long)(kt + 2) * (BK * 2) + voff);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[ks & 1][j]),
                                                              __builtin_bit_cast(bf16x8_t, fa[ks & 1][i]), acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);                     // one k step at a time (the unrolled loop would hoist every read)
    }
  };
  typedef std::integral_constant<bool, true> Yes;
  typedef std::integral_constant<bool, false> No;
  {                                                           // branch-free body: the last two K tiles are peeled
    int kt = 0;
    for (; kt + 2 < nk; ++kt) body(kt, Yes(), Yes());
    if (kt + 1 < nk) { body(kt, Yes(), No()); ++kt; }
    body(kt, No(), No());
  }
  if ((VAR & 1) && acc[0][0][0] != 12345.678f) return;        // main loop alone: nothing is stored (the test keeps the loop alive)
  // ---- probe epilogue: straight from the accumulator layout (lane: row fr of the block, columns 8 (r >> 2) + 4 fh + (r & 3))
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long long m = m0 + wm * 128 + i * 32 + fr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int nb = n0 + wn * 128 + j * 32;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        ushort4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(acc[i][j][4 * q + e]);
        *(ushort4_t*)(C + m * N + nb + 8 * q + 4 * fh) = o;
      }
      __builtin_amdgcn_sched_barrier(0);                     // one block at a time out of the AGPRs
    }
  }
}

static unsigned short h_f2bf(float f) {
  unsigned u;
  std::memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static float h_bf2f(unsigned short h) {
  unsigned u = ((unsigned)h) << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

int main() {
  const int shapes[][3] = {{8192, 8192, 8192}, {32768, 4096, 1024}, {32768, 1024, 4096}, {32768, 1024, 1024}, {65536, 1024, 1024},
                           {10240, 1024, 1536}};
  hipFuncSetAttribute((const void*)gemm4w_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_HALVES * 2);
  hipFuncSetAttribute((const void*)gemm4w_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_HALVES * 2);
  hipFuncSetAttribute((const void*)gemm4w_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_HALVES * 2);
  hipFuncSetAttribute((const void*)gemm4w_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_HALVES * 2);
  for (auto& sh : shapes) {
    const int M = sh[0], N = sh[1], K = sh[2];
    std::vector<unsigned short> ha((size_t)M * K), hb((size_t)N * K);
    unsigned seed = 12345u + M + N + K;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 9) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : ha) v = h_f2bf(rnd());
    for (auto& v : hb) v = h_f2bf(rnd() * 0.25f);
    unsigned short *da, *db, *dc;
    hipMalloc(&da, ha.size() * 2); hipMalloc(&db, hb.size() * 2); hipMalloc(&dc, (size_t)M * N * 2);
    hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    const dim3 grid((M / TM) * (N / TN)), block(256);
    const size_t ldsb = 2 * STAGE_HALVES * 2;
    float msv[4];
    for (int var = 3; var >= 0; --var) {                       // variant 0 last: its output is the one that is checked
      auto launch = [&]() {
        if (var == 0) hipLaunchKernelGGL(gemm4w_kernel<0>, grid, block, ldsb, 0, da, db, dc, M, N, K);
        else if (var == 1) hipLaunchKernelGGL(gemm4w_kernel<1>, grid, block, ldsb, 0, da, db, dc, M, N, K);
        else if (var == 2) hipLaunchKernelGGL(gemm4w_kernel<2>, grid, block, ldsb, 0, da, db, dc, M, N, K);
        else hipLaunchKernelGGL(gemm4w_kernel<3>, grid, block, ldsb, 0, da, db, dc, M, N, K);
      };
      for (int i = 0; i < 3; ++i) launch();
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      const int iters = 10;
      hipEventRecord(e0, 0);
      for (int i = 0; i < iters; ++i) launch();
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&msv[var], e0, e1);
      msv[var] /= iters;
      hipEventDestroy(e0); hipEventDestroy(e1);
    }
    const float ms = msv[0];
    std::vector<unsigned short> hc((size_t)M * N);
    hipMemcpy(hc.data(), dc, hc.size() * 2, hipMemcpyDeviceToHost);
    double worst = 0.0;
    for (int t = 0; t < 64; ++t) {
      const int m = (int)((1103515245u * (t + 1) + 12345u) % (unsigned)M), n = (int)((22695477u * (t + 7) + 1u) % (unsigned)N);
      double ref = 0.0;
      for (int k = 0; k < K; ++k) ref += (double)h_bf2f(ha[(size_t)m * K + k]) * (double)h_bf2f(hb[(size_t)n * K + k]);
      const double got = h_bf2f(hc[(size_t)m * N + n]);
      const double err = fabs(got - ref) / (fabs(ref) + 1e-2 * sqrt((double)K));
      if (err > worst) worst = err;
    }
    const double fl = 2.0 * M * N * K / 1e9;
    printf("%6d x %5d x %5d   %7.1f TFLOP/s  | main loop alone %7.1f | staging spread over k steps: %7.1f, alone %7.1f |  worst sampled rel. error %.2e %s\n",
           M, N, K, fl / ms, fl / msv[1], fl / msv[2], fl / msv[3], worst, worst < 2e-2 ? "" : "  <-- WRONG");
    hipFree(da); hipFree(db); hipFree(dc);
  }
  return 0;
}
