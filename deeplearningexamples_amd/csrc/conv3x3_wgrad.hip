// Weight gradient of the 3x3 / stride 1 / pad 1 convolution as a HALO-TILE product for gfx950.
//
// Replaces cuDNN's bwd-filter behind the 3x3 nn.Conv2d(bias=False) of the ResNet bottleneck
// (Classification/ConvNets/image_classification/models/resnet.py:126,148-175, models/common.py:31-60):
//     dw[ko][r][s][c] = sum over pixels  dy[n, h, w, ko] * x[n, h + r - 1, w + s - 1, c]
//
// Why not the split-K implicit GEMM of gemm_dma.hip (its B operand gathers im2col pixels per tap): there each of the nine taps is
// a separate stretch of the contraction, i.e. the activation goes L2 -> LDS nine times and dy nine times with it, on a 128 x 128
// tile whose LDS-DMA issue rate caps it at ~0.55 PFLOP/s; the 56x56x64 layers (Ko = 64: half of the tile's rows empty) ran at
// 0.33 PFLOP/s, 176 us for 59 GFLOP.  Here, in the flat padded pixel space of conv3x3.hip (image n = (H + 1) rows of W + 2 slots,
// zero padding slots; the input of slot g for tap (r, s) is slot g + (r - 1)(W + 2) + (s - 1)):
//  * a workgroup owns a 64 (ko) x 64 (c) x 9 (taps) block of dw and walks 128-slot pixel tiles: per tile ONE dy tile [128 slots x
//    64 ko] and ONE activation patch [128 + 2 (W + 2) + 2 slots x 64 c] go to LDS by LDS-DMA (double buffered, the next tile's DMA
//    flies under this tile's MFMAs); the nine taps read the SAME patch at shifted slots;
//  * the contraction runs over pixels, so both operands are read with the LDS transpose read (ds_read_b64_tr_b16) from their
//    pixel-major images: a dy fragment (32 ko x 16 slots) is read once per k step and meets nine activation fragments
//    (v_mfma_f32_32x32x16): 10 KiB of LDS reads per 9 MFMAs.  The 64-byte halves of a slot's 128-byte row are swapped on odd slot
//    PAIRS (on the DMA's source address): any 4 consecutive slots -- whatever the tap shift -- then cover all 64 banks once;
//  * 8 wavefronts = 2 pixel halves x (2 x 2) blocks of 32 ko x 32 c, 144 accumulator registers each; the two pixel halves meet
//    through LDS at the end, ONE fp32 partial block per workgroup (256 workgroups: 37.7 MB whatever the layer), folded in a fixed
//    order by a second launch (deterministic, no atomics).
#include "gemm_tiles.h"

#define W3_TG 128

struct W3Args {
  const unsigned short* gt;    // dy [N, H, W, Ko]
  const unsigned short* x;     // x  [N, H, W, C]
  float* ws;                   // [workgroups][64 ko][9][64 c]
  int N, H, W, C, Ko;
  int Wp, IMG, G;              // padded row length, slots per image, total slots
  int ntiles, nsub_c, nsub, npg;
  int ppieces;                 // 1 KiB DMA pieces of the activation patch
  FastDiv dIMG, dWp;
};

// LDS transpose read at byte address `addr` + an immediate offset (inline asm: see ds_read_tr16_asm in gemm_tiles.h)
template <int OFF>
__device__ __forceinline__ short4_t w3_tr(unsigned addr) {
  short4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}

template <int DT>
__global__ __launch_bounds__(512, 1) void conv3x3_wgrad_kernel(W3Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ph = wave >> 2, kob = (wave >> 1) & 1, cb = wave & 1;
  const int sub = blockIdx.x % p.nsub, pg = blockIdx.x / p.nsub;
  const int kosub = sub / p.nsub_c, csub = sub - kosub * p.nsub_c;
  const int bufbytes = (16 + p.ppieces) * 1024;

  // one 1 KiB piece = 8 slots x 8 chunks of 16 bytes; LDS chunk j of slot sl holds source chunk j ^ (((sl >> 1) & 1) << 2).
  // Two straight-line loops (dy pieces: 2 per wave, patch pieces: <= 4 per wave), each with its OWN wave-uniform descriptor: one
  // loop that picked the descriptor per piece made it a VGPR value and hipcc wrapped every DMA in a waterfall loop.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto issue_tile = [&](int t, int buf) __attribute__((always_inline)) {
    unsigned char* base = smem_raw + buf * bufbytes;
    __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)p.gt, 0, 0xFFFFFFE0, 0x00020000);
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0xFFFFFFE0, 0x00020000);
    const int sl8 = lane >> 3, j = lane & 7;
    auto pixel_of = [&](int f, bool& ok) __attribute__((always_inline)) {
      const int fc = f < 0 ? 0 : f;
      const int n = fd_div(fc, p.dIMG), rem = fc - n * p.IMG;
      const int hp = fd_div(rem, p.dWp), wp = rem - hp * p.Wp;
      ok = f >= 0 && f < p.G && hp >= 1 && wp >= 1 && wp <= p.W;
      return ((long long)n * p.H + (hp - 1)) * p.W + (wp - 1);
    };
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int pc = wave_u + 8 * k, slot = pc * 8 + sl8;
      bool ok;
      const long long pix = pixel_of(t * W3_TG + slot, ok);
      const int chunk = j ^ (((slot >> 1) & 1) << 2);
      const unsigned off = (unsigned)((pix * p.Ko + kosub * 64 + chunk * 8) * 2);
      dma16(rg, (unsigned short*)(base + pc * 1024), ok ? off : OOB_OFF);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int pc = wave_u + 8 * k;
      if (pc < p.ppieces) {
        const int slot = pc * 8 + sl8;
        bool ok;
        const long long pix = pixel_of(t * W3_TG + slot - p.Wp - 1, ok);
        const int chunk = j ^ (((slot >> 1) & 1) << 2);
        const unsigned off = (unsigned)((pix * p.C + csub * 64 + chunk * 8) * 2);
        dma16(rx, (unsigned short*)(base + (16 + pc) * 1024), ok ? off : OOB_OFF);
      }
    }
  };

  float16_t acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int tg = lane >> 4, ti = lane & 15;
  const int kline = (tg >> 1) * 8 + (ti >> 2);             // slot of the first read inside the 16-slot k step (+ 4: second read)
  const int chanb = ((tg & 1) * 16 + (ti & 3) * 4) * 2;    // byte offset of this lane's 4 channels inside the 32-channel block
  // Per-lane LDS byte offsets of the first read of k step 0: tile- and k-step-invariant (a k step moves 16 slots, the second read
  // 4 slots: neither changes bit 0 of slot >> 1, so the half swap stays put and the other 7 reads are immediate offsets).
  const int s00 = ph * 64 + kline;
  const unsigned aoff = (unsigned)(s00 * 128 + ((kob ^ ((s00 >> 1) & 1)) << 6) + chanb);
  unsigned boff[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int sx = 0; sx < 3; ++sx) {
      const int ps = s00 + r * p.Wp + sx;                  // activation slot of dy slot s00 for tap (r, sx)
      boff[r * 3 + sx] = (unsigned)(16 * 1024 + ps * 128 + ((cb ^ ((ps >> 1) & 1)) << 6) + chanb);
    }
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)smem_raw;

  int t = pg, it = 0;
  if (t < p.ntiles) issue_tile(t, 0);
  for (; t < p.ntiles; t += p.npg, ++it) {
    const int buf = it & 1;
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): this wave's pieces of tile t have landed
    __syncthreads();                                       // ... everybody's; the other buffer is no longer read
    if (t + p.npg < p.ntiles) issue_tile(t + p.npg, buf ^ 1);
    const unsigned bb = lds0 + (unsigned)(buf * bufbytes);
    // Software pipeline over (k step, tap group): the fragment reads of the NEXT group are in flight under the MFMAs of the current
    // one.  Group 0 = the dy fragment + taps 0-4 (12 reads), group 1 = taps 5-8 (8 reads); the reads are asm-issued and the LDS
    // counter retires in order, so "the older group has landed" = lgkmcnt(<reads of the younger group>).  (No scalar memory load
    // sits in this loop -- SMEM shares the counter and returns out of order.)
    TrPair fa[2], fb[9];
    fa[0].lo = w3_tr<0>(bb + aoff);
    fa[0].hi = w3_tr<512>(bb + aoff);
#pragma unroll
    for (int tap = 0; tap < 5; ++tap) { fb[tap].lo = w3_tr<0>(bb + boff[tap]); fb[tap].hi = w3_tr<512>(bb + boff[tap]); }
    static_for<0, 4>([&](auto KS) __attribute__((always_inline)) {
      constexpr int ks = decltype(KS)::value, cur = ks & 1, nxt = cur ^ 1;
#pragma unroll
      for (int tap = 5; tap < 9; ++tap) {
        fb[tap].lo = w3_tr<ks * 2048>(bb + boff[tap]);
        fb[tap].hi = w3_tr<ks * 2048 + 512>(bb + boff[tap]);
      }
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");   // group 0 of this k step has landed (8 younger reads in flight)
      __builtin_amdgcn_sched_barrier(0);
      const ushort8_t va = frag_value(fa[cur]);
#pragma unroll
      for (int tap = 0; tap < 5; ++tap) acc[tap] = Mfma32x16<DT>::run(va, frag_value(fb[tap]), acc[tap]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ks < 3) {
        fa[nxt].lo = w3_tr<(ks + 1) * 2048>(bb + aoff);
        fa[nxt].hi = w3_tr<(ks + 1) * 2048 + 512>(bb + aoff);
#pragma unroll
        for (int tap = 0; tap < 5; ++tap) {
          fb[tap].lo = w3_tr<(ks + 1) * 2048>(bb + boff[tap]);
          fb[tap].hi = w3_tr<(ks + 1) * 2048 + 512>(bb + boff[tap]);
        }
        asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");  // group 1 has landed (the 12 reads of the next k step stay in flight)
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tap = 5; tap < 9; ++tap) acc[tap] = Mfma32x16<DT>::run(va, frag_value(fb[tap]), acc[tap]);
      __builtin_amdgcn_sched_barrier(0);
    });
  }

  // ---- the two pixel halves meet in LDS (taps 0-4, then 5-8), then ONE partial block per workgroup
  float* red = (float*)smem_raw;                           // [4 wave slots][<= 5 taps][16 regs][64 lanes]
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    __syncthreads();
    if (ph == 1) {
#pragma unroll
      for (int tl = 0; tl < 5; ++tl) {
        const int tap = round * 5 + tl;
        if (tap < 9) {
#pragma unroll
          for (int r = 0; r < 16; ++r) red[(((wave & 3) * 5 + tl) * 16 + r) * 64 + lane] = acc[tap][r];
        }
      }
    }
    __syncthreads();
    if (ph == 0) {
#pragma unroll
      for (int tl = 0; tl < 5; ++tl) {
        const int tap = round * 5 + tl;
        if (tap < 9) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[tap][r] += red[(((wave & 3) * 5 + tl) * 16 + r) * 64 + lane];
        }
      }
    }
  }
  if (ph == 0) {
    // D[row = ko][col = c]: lane holds c = cb * 32 + (lane & 31), ko = kob * 32 + 8 (reg >> 2) + 4 (lane >> 5) + (reg & 3)
    float* out = p.ws + (long long)blockIdx.x * (64 * 9 * 64);
    const int c = cb * 32 + (lane & 31);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ko = kob * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        out[(ko * 9 + tap) * 64 + c] = acc[tap][r];
      }
  }
}

// dw[ko][tap][c] (+)= sum over the pixel groups of the partial blocks.  A workgroup = EL float4 elements x SL group slices
// (SL = 16 for the many-group layers: 256 partial blocks per element would otherwise be 16 dependent batches per thread).
template <int SL>
__global__ __launch_bounds__(256) void conv3x3_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Ko, int C,
                                                                   int nsub_c, int nsub, int npg, int accumulate) {
  constexpr int EL = 256 / SL;
  __shared__ float4_t red[256];
  const int c4n = C >> 2;
  const long long total4 = (long long)Ko * 9 * c4n;
  const int el = threadIdx.x % EL, gs = threadIdx.x / EL;
  const long long e = (long long)blockIdx.x * EL + el;
  float4_t s = {0.f, 0.f, 0.f, 0.f};
  long long dst = 0;
  if (e < total4) {
    const int c4 = (int)(e % c4n);
    const long long kt = e / c4n;
    const int tap = (int)(kt % 9), ko = (int)(kt / 9);
    const int c = c4 * 4;
    const int sub = (ko >> 6) * nsub_c + (c >> 6);
    const float* src = ws + ((long long)sub * 64 + (ko & 63)) * (9 * 64) + tap * 64 + (c & 63);
    const long long gstride = (long long)nsub * (64 * 9 * 64);
    int g = gs;
    for (; g + 3 * SL < npg; g += 4 * SL) {
      const float4_t a = *(const float4_t*)(src + (long long)g * gstride), b = *(const float4_t*)(src + (long long)(g + SL) * gstride);
      const float4_t cc = *(const float4_t*)(src + (long long)(g + 2 * SL) * gstride), d = *(const float4_t*)(src + (long long)(g + 3 * SL) * gstride);
      s += (a + b) + (cc + d);
    }
    for (; g < npg; g += SL) s += *(const float4_t*)(src + (long long)g * gstride);
    dst = ((long long)ko * 9 + tap) * C + c;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (gs == 0 && e < total4) {
    float4_t t = red[el];
#pragma unroll
    for (int q = 1; q < SL; ++q) t += red[q * EL + el];
    if (accumulate) t += *(float4_t*)(dw + dst);
    *(float4_t*)(dw + dst) = t;
  }
}

static int g_w3_mode = -1;               // -1 / 1: on where it applies, 0: off (A/B harnesses, DLE_CONV3X3_WGRAD=0)
extern "C" int dle_conv3x3_wgrad_mode(int mode) {
  const int old = g_w3_mode;
  g_w3_mode = mode;
  return old;
}

// Workspace the halo-tile weight gradient needs (one 64 x 9 x 64 fp32 block per workgroup, 256 workgroups).
extern "C" int64_t dle_conv3x3_wgrad_workspace(void) { return 256LL * 64 * 9 * 64 * 4; }

// 1: launched; 0: outside the envelope (the caller uses the split-K implicit GEMM); > 1: error.
extern "C" int dle_conv3x3_wgrad_try(const void* dy, const void* x, float* dw, int N, int H, int W, int C, int Ko, int dtype,
                                     int accumulate, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  static const int env_mode = getenv("DLE_CONV3X3_WGRAD") ? atoi(getenv("DLE_CONV3X3_WGRAD")) : -1;
  if (g_w3_mode == 0 || (g_w3_mode < 0 && env_mode == 0)) return 0;
  if ((C & 63) || (Ko & 63) || (dtype != DLE_F16 && dtype != DLE_BF16)) return 0;
  if (((((uintptr_t)dy) | ((uintptr_t)x) | ((uintptr_t)dw) | ((uintptr_t)workspace)) & 15) != 0) return 0;
  const int nsub = (Ko / 64) * (C / 64);
  if (nsub > 256 || (256 % nsub) != 0) return 0;
  if (!workspace || workspace_bytes < dle_conv3x3_wgrad_workspace()) return 0;
  const long long Wp = W + 2, IMG = (long long)(H + 1) * Wp, G = (long long)N * IMG;
  if (G + 2 * Wp + W3_TG >= 0x7FFFFFFFLL || (long long)N * H * W * (C > Ko ? C : Ko) * 2 >= 0xFFFFFFE0LL) return 0;
  W3Args p;
  p.gt = (const unsigned short*)dy; p.x = (const unsigned short*)x; p.ws = (float*)workspace;
  p.N = N; p.H = H; p.W = W; p.C = C; p.Ko = Ko;
  p.Wp = (int)Wp; p.IMG = (int)IMG; p.G = (int)G;
  p.ntiles = (int)((G + W3_TG - 1) / W3_TG);
  p.nsub_c = C / 64; p.nsub = nsub; p.npg = 256 / nsub;
  p.ppieces = (int)((W3_TG + 2 * Wp + 2 + 7) / 8);
  p.dIMG = make_fastdiv(p.IMG); p.dWp = make_fastdiv(p.Wp);
  size_t lds = (size_t)2 * (16 + p.ppieces) * 1024;
  if (lds < 4 * 5 * 16 * 64 * 4) lds = 4 * 5 * 16 * 64 * 4;          // the meeting buffer of the pixel halves (80 KiB)
  if (lds > 160 * 1024 || p.ppieces > 32) return 0;
  const dim3 grid(256), block(512);
#define W3_GO(DT) do { static bool attr_set = false; \
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_kernel<DT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; } \
    hipLaunchKernelGGL((conv3x3_wgrad_kernel<DT>), grid, block, lds, stream, p); } while (0)
  if (dtype == DLE_F16) W3_GO(DLE_F16); else W3_GO(DLE_BF16);
#undef W3_GO
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { dle_set_error("conv3x3_wgrad launch failed: %s", hipGetErrorString(e)); return (int)e + 1000; }
  const long long total4 = (long long)Ko * 9 * (C / 4);
  if (p.npg >= 64) hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel<16>, dim3((unsigned)((total4 + 15) / 16)), dim3(256), 0, stream,
                                      (const float*)workspace, dw, Ko, C, p.nsub_c, nsub, p.npg, accumulate);
  else hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel<4>, dim3((unsigned)((total4 + 63) / 64)), dim3(256), 0, stream,
                          (const float*)workspace, dw, Ko, C, p.nsub_c, nsub, p.npg, accumulate);
  e = hipGetLastError();
  if (e != hipSuccess) { dle_set_error("conv3x3_wgrad reduce launch failed: %s", hipGetErrorString(e)); return (int)e + 1000; }
  return 1;
}
