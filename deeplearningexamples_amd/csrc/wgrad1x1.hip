// Weight gradient of a 1x1 convolution whose WHOLE output fits one workgroup's accumulators, as a STREAMING kernel for gfx950:
//     dw[ko][c] = sum over rows m  dy[m][ko] * x[m][c]        (dy [M, Ko], x [M, C] 16-bit row-major, dw fp32)
// The ResNet-50 layers with the long contraction and the small output (Classification/ConvNets/image_classification/models/
// resnet.py:148-175: the 1x1 convolutions of the 56x56 and 28x28 stages at batch 256: M = 802,816 / 200,704, Ko x C <= 512 x 128)
// sit at 51-102 flop/B, far on the HBM side of the ridge.  The split-K tile GEMM of gemm_dma.hip ran them at 2.8-4.4 TB/s: a
// 128 x 128 (256 x 256) tile sees half (a quarter) of the output, so every operand row is fetched by 2-4 workgroups, 512 K slices
// write and re-read 32-64 MB of fp32 slabs (684 MB of fabric traffic for 514 MB of operands on 256 x 64 x 802816).  Here
//  * ONE workgroup holds the whole Ko x C block in registers (8 wavefronts x up to 8 blocks of 32 x 32) and walks 32 / 64-row
//    tiles of the contraction: every operand byte leaves HBM once;
//  * both operands go HBM -> LDS by LDS-DMA in their own row-major layout (double buffered, 40 KiB per tile) and are read with the
//    LDS transpose read (the contraction runs over rows); 64-byte channel blocks are XOR-swizzled with the row (on the DMA's
//    source address) so that the 4 rows a read group touches cover all banks;
//  * one fp32 partial block per workgroup (256 / 512 of them: 16-64 MB instead of the operands again), folded in a fixed order.
#include "gemm_tiles.h"

struct W1Args {
  const unsigned short* G;     // dy [M, Ko]
  const unsigned short* X;     // x  [M, C]
  float* ws;                   // [workgroups][Ko][C]
  int M, Ko, C, ntiles;
};

template <int OFF>
__device__ __forceinline__ short4_t w1_tr(unsigned addr) {
  short4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}

// 64-byte block b of row r sits at block position b ^ swz(r): rows of 128 bytes (2 blocks) swap halves on odd row PAIRS, longer rows
// rotate within groups of 4 blocks -- the 4 consecutive rows x 64 bytes of a 32-lane transpose-read group then hit 8 distinct
// 32-byte bank slots.
template <int L> __device__ __forceinline__ int w1_swz(int r) { return L == 128 ? ((r >> 1) & 1) : (r & 3); }

// KO, CC: the output block; TG rows per tile; the 8 wavefronts form PH (row halves) x WGM x WGN, each WM x WN blocks of 32 x 32.
template <int DT, int KO, int CC, int TG, int PH, int WGM, int WGN, int WM, int WN>
__global__ __launch_bounds__(512) void wgrad1x1_kernel(W1Args p) {
  static_assert(PH * WGM * WGN == 8 && WGM * WM * 32 == KO && WGN * WN * 32 == CC, "wave grid must tile the output");
  constexpr int LG = KO * 2, LX = CC * 2;                               // row bytes
  constexpr int GBYTES = TG * LG, XBYTES = TG * LX, BUF = GBYTES + XBYTES;
  constexpr int NPG = GBYTES / 1024 / 8, NPX = XBYTES / 1024 / 8;       // DMA pieces per wavefront
  static_assert(GBYTES % 8192 == 0 && XBYTES % 8192 == 0, "whole pieces per wavefront");
  constexpr int KSTEPS = TG / 16 / PH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int ph = wave / (WGM * WGN), wq = wave % (WGM * WGN), wm = wq / WGN, wn = wq % WGN;

  auto issue_tile = [&](int t, int buf) __attribute__((always_inline)) {
    unsigned char* base = smem_raw + buf * BUF;
    __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)p.G, 0, 0xFFFFFFE0, 0x00020000);
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, 0xFFFFFFE0, 0x00020000);
    const long long row0 = (long long)t * TG;
#pragma unroll
    for (int k = 0; k < NPG; ++k) {
      const int pc = wave_u + 8 * k;
      const int ob = pc * 1024 + lane * 16, r = ob / LG, cb = ob % LG;  // LDS row, byte inside the row
      const int blk = (cb >> 6) ^ w1_swz<LG>(r);
      const long long m = row0 + r;
      dma16(rg, (unsigned short*)(base + pc * 1024), m < p.M ? (unsigned)(m * LG + blk * 64 + (cb & 63)) : OOB_OFF);
    }
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
      const int pc = wave_u + 8 * k;
      const int ob = pc * 1024 + lane * 16, r = ob / LX, cb = ob % LX;
      const int blk = (cb >> 6) ^ w1_swz<LX>(r);
      const long long m = row0 + r;
      dma16(rx, (unsigned short*)(base + GBYTES + pc * 1024), m < p.M ? (unsigned)(m * LX + blk * 64 + (cb & 63)) : OOB_OFF);
    }
  };

  float16_t acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // per-lane byte offsets of the first read of k step 0 (tile- and k-step-invariant: + 16 rows / + 4 rows keep row & 3 and
  // bit 0 of row >> 1, so every other read is an immediate offset)
  const int tg = lane >> 4, ti = lane & 15;
  const int r0 = ph * (TG / PH) + (tg >> 1) * 8 + (ti >> 2);
  const int chanb = ((tg & 1) * 16 + (ti & 3) * 4) * 2;
  unsigned aoff[WM], boff[WN];
#pragma unroll
  for (int i = 0; i < WM; ++i) aoff[i] = (unsigned)(r0 * LG + (((wm * WM + i) ^ w1_swz<LG>(r0)) << 6) + chanb);
#pragma unroll
  for (int j = 0; j < WN; ++j) boff[j] = (unsigned)(GBYTES + r0 * LX + (((wn * WN + j) ^ w1_swz<LX>(r0)) << 6) + chanb);
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)smem_raw;

  int t = blockIdx.x, it = 0;
  if (t < p.ntiles) issue_tile(t, 0);
  for (; t < p.ntiles; t += gridDim.x, ++it) {
    const int buf = it & 1;
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): this wave's pieces of tile t have landed
    __syncthreads();                                       // ... everybody's; the other buffer is no longer read
    if (t + (int)gridDim.x < p.ntiles) issue_tile(t + gridDim.x, buf ^ 1);
    const unsigned bb = lds0 + (unsigned)(buf * BUF);
    static_for<0, KSTEPS>([&](auto KS) __attribute__((always_inline)) {
      constexpr int ks = decltype(KS)::value;
      TrPair fa[WM], fb[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) { fa[i].lo = w1_tr<ks * 16 * LG>(bb + aoff[i]); fa[i].hi = w1_tr<ks * 16 * LG + 4 * LG>(bb + aoff[i]); }
#pragma unroll
      for (int j = 0; j < WN; ++j) { fb[j].lo = w1_tr<ks * 16 * LX>(bb + boff[j]); fb[j].hi = w1_tr<ks * 16 * LX + 4 * LX>(bb + boff[j]); }
      frag_wait<true>();
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const ushort8_t va = frag_value(fa[i]);
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = Mfma32x16<DT>::run(va, frag_value(fb[j]), acc[i][j]);
      }
    });
  }

  // ---- row halves meet in LDS (PH == 2 only), then ONE partial block per workgroup
  if constexpr (PH == 2) {
    float* red = (float*)smem_raw;                         // [4 waves][WM * WN][16][64]
    __syncthreads();
    if (ph == 1) {
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[((wq * (WM * WN) + i * WN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
    }
    __syncthreads();
    if (ph == 0) {
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] += red[((wq * (WM * WN) + i * WN + j) * 16 + r) * 64 + lane];
    }
  }
  if (ph == 0) {
    // D[row = ko][col = c]: lane holds c = 32 cblk + (lane & 31), ko = 32 kblk + 8 (reg >> 2) + 4 (lane >> 5) + (reg & 3)
    float* out = p.ws + (long long)blockIdx.x * (KO * CC);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int c = (wn * WN + j) * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ko = (wm * WM + i) * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
          out[ko * CC + c] = acc[i][j][r];
        }
      }
  }
}

// dw[e] (+)= sum over the G partial blocks (16 float4 elements x 16 group slices per workgroup; fixed order)
__global__ __launch_bounds__(256) void wgrad1x1_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, long long total4,
                                                              int G, int accumulate) {
  __shared__ float4_t red[256];
  const int el = threadIdx.x & 15, gs = threadIdx.x >> 4;
  const long long e = (long long)blockIdx.x * 16 + el;
  float4_t s = {0.f, 0.f, 0.f, 0.f};
  if (e < total4) {
    const float4_t* src = (const float4_t*)ws + e;
    int g = gs;
    for (; g + 48 < G; g += 64) {
      const float4_t a = src[(long long)g * total4], b = src[(long long)(g + 16) * total4];
      const float4_t c = src[(long long)(g + 32) * total4], d = src[(long long)(g + 48) * total4];
      s += (a + b) + (c + d);
    }
    for (; g < G; g += 16) s += src[(long long)g * total4];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (gs == 0 && e < total4) {
    float4_t t = red[el];
#pragma unroll
    for (int q = 1; q < 16; ++q) t += red[q * 16 + el];
    if (accumulate) t += ((float4_t*)dw)[e];
    ((float4_t*)dw)[e] = t;
  }
}

static int g_w1_mode = -1;
extern "C" int dle_wgrad1x1_mode(int mode) {
  const int old = g_w1_mode;
  g_w1_mode = mode;
  return old;
}

// largest workspace any supported shape needs: 512 workgroups x 256 x 64 x 4 B = 32 MB, 256 x 512 x 128 x 4 B = 64 MB
extern "C" int64_t dle_wgrad1x1_workspace(void) { return 64LL << 20; }

// workspace of ONE shape (bytes), 0 when (M, Ko, C) is outside the kernel's envelope -- so that a caller does not grow its
// per-stream scratch to the 64 MB maximum for shapes the kernel declines
extern "C" int64_t dle_wgrad1x1_workspace_for(int M, int Ko, int C) {
  static const int env_mode = getenv("DLE_WGRAD1X1") ? atoi(getenv("DLE_WGRAD1X1")) : -1;
  if (g_w1_mode == 0 || (g_w1_mode < 0 && env_mode == 0)) return 0;
  if (M < 8192 || (long long)M * (Ko > C ? Ko : C) * 2 >= 0xFFFFFFE0LL) return 0;
  int tg = 0, wgs = 0;
#define W1_CFG(KOv, CCv, TGv, WGSv) if (Ko == KOv && C == CCv) { tg = TGv; wgs = WGSv; }
  W1_CFG(256, 64, 64, 512) W1_CFG(64, 256, 64, 512) W1_CFG(64, 64, 128, 512) W1_CFG(128, 256, 64, 256) W1_CFG(256, 128, 64, 256)
  W1_CFG(512, 128, 32, 256) W1_CFG(128, 512, 32, 256)
#undef W1_CFG
  if (!tg) return 0;
  const int ntiles = (M + tg - 1) / tg;
  if (wgs > ntiles) wgs = ntiles;
  return (int64_t)wgs * Ko * C * 4;
}

// 1: launched; 0: outside the envelope (the caller uses the split-K tile GEMM); > 1: error.
extern "C" int dle_wgrad1x1_try(const void* dy, const void* x, float* dw, int M, int Ko, int C, int dtype, int accumulate,
                                void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  static const int env_mode = getenv("DLE_WGRAD1X1") ? atoi(getenv("DLE_WGRAD1X1")) : -1;
  if (g_w1_mode == 0 || (g_w1_mode < 0 && env_mode == 0)) return 0;
  if (dtype != DLE_F16 && dtype != DLE_BF16) return 0;
  if (M < 8192 || (long long)M * (Ko > C ? Ko : C) * 2 >= 0xFFFFFFE0LL) return 0;
  if (((((uintptr_t)dy) | ((uintptr_t)x) | ((uintptr_t)dw) | ((uintptr_t)workspace)) & 15) != 0 || !workspace) return 0;
  W1Args p;
  p.G = (const unsigned short*)dy; p.X = (const unsigned short*)x; p.ws = (float*)workspace; p.M = M; p.Ko = Ko; p.C = C;
  int tg = 0, wgs = 0;
  size_t lds = 0;
  // (Ko, C) -> rows per tile, resident workgroups (two per CU while the block is <= 64 accumulator registers per wavefront)
#define W1_CFG(KOv, CCv, TGv, WGSv) if (Ko == KOv && C == CCv) { tg = TGv; wgs = WGSv; lds = (size_t)2 * TGv * (KOv + CCv) * 2; }
  W1_CFG(256, 64, 64, 512) W1_CFG(64, 256, 64, 512) W1_CFG(64, 64, 128, 512) W1_CFG(128, 256, 64, 256) W1_CFG(256, 128, 64, 256)
  W1_CFG(512, 128, 32, 256) W1_CFG(128, 512, 32, 256)
#undef W1_CFG
  if (!tg) return 0;
  if (lds < 64 * 1024 && Ko == 64 && C == 64) lds = 64 * 1024;          // the meeting buffer of the row halves
  p.ntiles = (M + tg - 1) / tg;
  if (wgs > p.ntiles) wgs = p.ntiles;
  if (workspace_bytes < (long long)wgs * Ko * C * 4) return 0;
  const dim3 grid(wgs), block(512);
#define W1_GO(DT, KOv, CCv, TGv, PHv, WGMv, WGNv, WMv, WNv) do { static bool attr_set = false; \
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)wgrad1x1_kernel<DT, KOv, CCv, TGv, PHv, WGMv, WGNv, WMv, WNv>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; } \
    hipLaunchKernelGGL((wgrad1x1_kernel<DT, KOv, CCv, TGv, PHv, WGMv, WGNv, WMv, WNv>), grid, block, lds, stream, p); } while (0)
#define W1_PICK(DT) do { \
    if (Ko == 256 && C == 64) W1_GO(DT, 256, 64, 64, 1, 8, 1, 1, 2); \
    else if (Ko == 64 && C == 256) W1_GO(DT, 64, 256, 64, 1, 2, 4, 1, 2); \
    else if (Ko == 64 && C == 64) W1_GO(DT, 64, 64, 128, 2, 2, 2, 1, 1); \
    else if (Ko == 128 && C == 256) W1_GO(DT, 128, 256, 64, 1, 2, 4, 2, 2); \
    else if (Ko == 256 && C == 128) W1_GO(DT, 256, 128, 64, 1, 4, 2, 2, 2); \
    else if (Ko == 512 && C == 128) W1_GO(DT, 512, 128, 32, 1, 8, 1, 2, 4); \
    else W1_GO(DT, 128, 512, 32, 1, 2, 4, 2, 4); } while (0)
  if (dtype == DLE_F16) W1_PICK(DLE_F16); else W1_PICK(DLE_BF16);
#undef W1_PICK
#undef W1_GO
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { dle_set_error("wgrad1x1 launch failed: %s", hipGetErrorString(e)); return (int)e + 1000; }
  const long long total4 = (long long)Ko * C / 4;
  hipLaunchKernelGGL(wgrad1x1_reduce_kernel, dim3((unsigned)((total4 + 15) / 16)), dim3(256), 0, stream, (const float*)workspace, dw, total4,
                     wgs, accumulate);
  e = hipGetLastError();
  if (e != hipSuccess) { dle_set_error("wgrad1x1 reduce launch failed: %s", hipGetErrorString(e)); return (int)e + 1000; }
  return 1;
}
