// 3x3 / stride 1 / pad 1 convolution (forward and data gradient) as a HALO-TILE implicit GEMM for gfx950.
//
// Replaces cuDNN's conv fwd / bwd-data behind the 3x3 nn.Conv2d(bias=False) of the ResNet bottleneck
// (Classification/ConvNets/image_classification/models/resnet.py:126,148-175, models/common.py:31-60).
//
// Why not the im2col loader of gemm_dma.hip: there the [128 pixels x 64 k] operand tile of EVERY filter tap is a fresh
// LDS-DMA, i.e. the activation goes L2 -> LDS nine times (the 56x56x64 layers ran at ~1 TB/s algorithmic, 10 % of the
// matrix peak, bound by the DMA issue / address work and L2, not by MFMA or HBM).  Here the activation tile is loaded
// ONCE per 64-channel chunk, halo included, and the nine taps read it at shifted LDS addresses:
//
//  * pixels are indexed in a FLAT PADDED space: image n occupies (H+1) rows of Wp = W+2 slots, row 0 and columns
//    0 / W+1 being zero padding (the zero row of image n+1 doubles as the bottom padding of image n).  The input of
//    output slot g for tap (r, s) is slot g + (r-1)*Wp + (s-1) -- a constant shift, whatever the image borders.
//    A workgroup owns 256 consecutive slots; its LDS patch holds slots [g0 - Wp - 1, g0 + 256 + Wp + 1) x 64 channels
//    (128 B per slot), zero-filled where the slot is padding or outside the tensor (buffer range check).  The price is
//    the padding slots that are computed and dropped: 5 % at 56x56, 10 % at 28x28, 18 % at 14x14.
//  * the patch image is lane-linear (LDS-DMA), 16-byte chunks XOR-swizzled by (slot >> 1) & 7 on the SOURCE address; a
//    32-pixel MFMA fragment of any tap reads 32 consecutive slots with ds_read_b128, conflict free for every shift.
//  * the weights of one (tap, channel chunk) are a [NT x 64] tile, double buffered, by the loaders of gemm_tiles.h:
//    forward  B(ko, k = c)  = W[ko][r][s][c]        (k contiguous)
//    dgrad    B(c,  k = ko) = W[ko][2-r][2-s][c]    (rows contiguous -> LDS transpose reads; no flipped copy)
//  * 4 wavefronts, each 64 pixels x NT output channels (NT = 64 / 128: 64 / 128 accumulator VGPRs), two workgroups
//    per CU.  Epilogue: each wavefront transposes its own 32 x NT fp32 block through LDS (no workgroup barrier),
//    16-byte stores of whole channel rows, padding slots skipped; optional per-tile column sums / sums of squares of
//    the rounded output for the BatchNorm that follows (same contract as dle_conv2d_fwd_colstats).
#include "gemm_tiles.h"

#define C3_MT 256

struct C3Args {
  const unsigned short* x;     // input  [N, H, W, CI]
  const unsigned short* w;     // weights [Ko, 3, 3, C] (forward: CI = C, CO = Ko; dgrad: CI = Ko, CO = C)
  unsigned short* y;           // output [N, H, W, CO]
  float* stats;                // [tiles_m][2][CO] or NULL
  int N, H, W, CI, CO;
  int Wp, IMG, G;              // padded row length, slots per image, total slots
  int tiles_m, tiles_n, pieces;
  FastDiv dIMG, dWp;
};

template <int DT, bool DGRAD, int NT>
__global__ __launch_bounds__(256, 2) void conv3x3_kernel(C3Args p) {
  constexpr int WTN = NT / 32;
  typedef Loader<DGRAD ? 1 : 0, NT, 4> LB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* patch = (unsigned short*)smem_raw;
  unsigned short* bst = patch + p.pieces * 512;            // 2 weight stages of NT x 64 halves
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;

  int bid = blockIdx.x;
  {
    const int ntiles = p.tiles_m * p.tiles_n;
    const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tm = bid / p.tiles_n, tn = bid - tm * p.tiles_n;
  const int g0 = tm * C3_MT, n0 = tn * NT;
  const int f0 = g0 - p.Wp - 1;                            // flat slot held by patch slot 0

  ConvGeom nog;                                            // the matrix loader modes ignore the geometry
  nog.H = nog.W = nog.C = nog.P = nog.Q = nog.R = nog.S = nog.stride = nog.Ko = 1; nog.pad = 0;
  LB lb;
  const long long ldb = 9LL * (DGRAD ? p.CO : p.CI);
  lb.init(wave, lane, n0, p.CO, ldb, nog);

  // the activation patch of channel chunk cc: one 1 KiB piece = 8 slots x 8 chunks, pieces dealt round-robin to waves
  auto issue_patch = [&](int cc) __attribute__((always_inline)) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0xFFFFFFE0, 0x00020000);
    for (int pc = wave; pc < p.pieces; pc += 4) {
      const int slot = pc * 8 + (lane >> 3), chunk = (lane & 7) ^ swz_kc(slot);
      const int f = f0 + slot;
      const int fc = f < 0 ? 0 : f;
      const int n = fd_div(fc, p.dIMG), rem = fc - n * p.IMG;
      const int hp = fd_div(rem, p.dWp), wp = rem - hp * p.Wp;
      const bool ok = f >= 0 && f < p.G && hp >= 1 && wp >= 1 && wp <= p.W;
      const unsigned off = (unsigned)(((((long long)n * p.H + (hp - 1)) * p.W + (wp - 1)) * p.CI + cc * 64 + chunk * 8) * 2);
      dma16(rs, patch + pc * 512, ok ? off : OOB_OFF);
    }
  };
  // weight tile of K tile kt = (chunk cc, tap): 64 contraction channels of one filter tap
  auto issue_b = [&](int cc, int tap, int stage) __attribute__((always_inline)) {
    const unsigned short* base = DGRAD ? p.w + (long long)cc * 64 * ldb + (8 - tap) * p.CO + n0
                                       : p.w + (long long)n0 * ldb + tap * p.CI + cc * 64;
    lb.template issue<0, LB::NP>(base, bst + stage * (NT * 64), wave, 64, 0, nog);
  };

  float16_t acc[2][WTN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nchunks = p.CI >> 6;
  issue_patch(0);
  issue_b(0, 0, 0);
  int stage = 0;
  for (int cc = 0; cc < nchunks; ++cc) {
    if (cc > 0) {
      // every wave is past tap 8 of the previous chunk (barrier), then reload the patch
      __syncthreads();
      issue_patch(cc);
    }
    int tapoff = 0, tapcol = 0;                           // r * Wp + s, walked incrementally
    for (int tap = 0; tap < 9; ++tap) {
      __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0): this wave's DMA pieces have landed
      __syncthreads();                                    // ... everybody's; the other weight stage is free again
      if (tap < 8) issue_b(cc, tap + 1, stage ^ 1);
      else if (cc + 1 < nchunks) issue_b(cc + 1, 0, stage ^ 1);
      const unsigned short* tb = bst + stage * (NT * 64);
      int abase[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int slot = wave * 64 + i * 32 + fr + tapoff;
        abase[i] = slot * 64 + ((fh ^ swz_kc(slot)) << 3);
      }
      if constexpr (!DGRAD) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          ushort8_t fa[2], fb[WTN];
#pragma unroll
          for (int i = 0; i < 2; ++i) fa[i] = *(const ushort8_t*)(patch + (abase[i] ^ (ks << 4)));
#pragma unroll
          for (int j = 0; j < WTN; ++j) fb[j] = frag_issue<false, NT>(tb, j * 32, ks, lane);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j) acc[i][j] = Mfma32x16<DT>::run(fb[j], fa[i], acc[i][j]);
        }
      } else {
        // weights through asm-issued transpose reads (gemm_tiles.h): requested one k-step ahead, one explicit wait each
        TrPair fb[2][WTN];
#pragma unroll
        for (int j = 0; j < WTN; ++j) fb[0][j] = frag_issue<true, NT>(tb, j * 32, 0, lane);
        static_for<0, 4>([&](auto KS) __attribute__((always_inline)) {
          constexpr int ks = decltype(KS)::value, cur = ks & 1, nxt = cur ^ 1;
          ushort8_t fa[2], vb[WTN];
#pragma unroll
          for (int i = 0; i < 2; ++i) fa[i] = *(const ushort8_t*)(patch + (abase[i] ^ (ks << 4)));
          frag_wait<true>();
          if constexpr (ks < 3) {
#pragma unroll
            for (int j = 0; j < WTN; ++j) fb[nxt][j] = frag_issue<true, NT>(tb, j * 32, ks + 1, lane);
          }
#pragma unroll
          for (int j = 0; j < WTN; ++j) vb[j] = frag_value(fb[cur][j]);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j) acc[i][j] = Mfma32x16<DT>::run(vb[j], fa[i], acc[i][j]);
        });
      }
      stage ^= 1;
      if (++tapcol == 3) { tapcol = 0; tapoff += p.Wp - 2; } else ++tapoff;
    }
  }

  // ---- epilogue.  D = (B A^T) block: lane owns pixel fr, channels 8*(r>>2) + 4*fh + (r&3) of each 32x32 block.
  __syncthreads();                                         // patch / weight stages are no longer read
  constexpr int LPR = NT / 8, RPT = 64 / LPR, TRIPS = 32 / RPT, MASK = NT / 4 - 1;
  float* epi = (float*)smem_raw + wave * (32 * NT);
  const int erow = lane / LPR, ecg = lane % LPR;
  float st0[8], st1[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) { st0[r] = 0.f; st1[r] = 0.f; }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4_t v = {acc[i][j][qd * 4 + 0], acc[i][j][qd * 4 + 1], acc[i][j][qd * 4 + 2], acc[i][j][qd * 4 + 3]};
        const int c4 = j * 8 + qd * 2 + fh;
        *(float4_t*)(epi + fr * NT + ((c4 ^ (fr & MASK)) << 2)) = v;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
      const int row = t * RPT + erow;
      const float4_t lo = *(const float4_t*)(epi + row * NT + (((2 * ecg) ^ (row & MASK)) << 2));
      const float4_t hi = *(const float4_t*)(epi + row * NT + (((2 * ecg + 1) ^ (row & MASK)) << 2));
      const int g = g0 + wave * 64 + i * 32 + row;
      const int n = fd_div(g, p.dIMG), rem = g - n * p.IMG;
      const int hp = fd_div(rem, p.dWp), wp = rem - hp * p.Wp;
      const bool ok = g < p.G && hp >= 1 && wp >= 1 && wp <= p.W;
      if (ok) {
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const ushort8_t ov = pack8<DT>(v);
        *(ushort8_t*)(p.y + (((long long)n * p.H + (hp - 1)) * p.W + (wp - 1)) * p.CO + n0 + ecg * 8) = ov;
        if (p.stats) {
          float vr[8];
          unpack8<DT>(ov, vr);
#pragma unroll
          for (int r = 0; r < 8; ++r) { st0[r] += vr[r]; st1[r] += vr[r] * vr[r]; }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (p.stats) {
    // lanes that share a channel group (same lane % LPR) fold by shuffles, the 4 waves meet in LDS; one plain store
    // per (tile, statistic, channel): deterministic, no atomics
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) {
        st0[r] += __shfl_xor(st0[r], o, 64);
        st1[r] += __shfl_xor(st1[r], o, 64);
      }
    }
    __syncthreads();                                       // every wave is done with its transposition block
    float* red = (float*)smem_raw;                         // [4 waves][2][NT]
    if (lane < LPR) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        red[(wave * 2 + 0) * NT + lane * 8 + r] = st0[r];
        red[(wave * 2 + 1) * NT + lane * 8 + r] = st1[r];
      }
    }
    __syncthreads();
    if (tid < 2 * NT) {
      const int which = tid / NT, col = tid - which * NT;
      const float t = (red[(0 * 2 + which) * NT + col] + red[(1 * 2 + which) * NT + col]) +
                      (red[(2 * 2 + which) * NT + col] + red[(3 * 2 + which) * NT + col]);
      p.stats[((long long)tm * 2 + which) * p.CO + n0 + col] = t;
    }
  }
}

static int g_conv3x3_mode = -1;      // -1: on where it applies, 0: off, 1: on (same as -1; kept for A/B harnesses)

// 0 / 1 switch for A/B measurements (tools/kbench); returns the previous value.
extern "C" int dle_conv3x3_mode(int mode) {
  const int old = g_conv3x3_mode;
  g_conv3x3_mode = mode;
  return old;
}

// Number of column-statistics tile rows this kernel writes for an [N, H, W] output (256 padded slots per tile).
extern "C" int dle_conv3x3_tiles(int N, int H, int W) {
  const long long G = (long long)N * (H + 1) * (W + 2);
  return (int)((G + C3_MT - 1) / C3_MT);
}

// Returns 1 when this kernel took the launch, 0 when the shape is outside its envelope (the caller then uses the
// im2col GEMM), > 1 on a launch error.  x/y channel counts in the FORWARD sense of the call: dgrad = 1 swaps roles
// (x = dy [N,H,W,Ko], y = dx [N,H,W,C], w [Ko,3,3,C]).
extern "C" int dle_conv3x3_try(const void* x, const void* w, void* y, float* stats, long long stats_bytes, int N, int H,
                               int W, int C, int Ko, int dgrad, int dtype, hipStream_t stream) {
  if (g_conv3x3_mode == 0) return 0;
  const int CI = dgrad ? Ko : C, CO = dgrad ? C : Ko;
  if ((CI & 63) || (CO & 63) || (dtype != DLE_F16 && dtype != DLE_BF16)) return 0;
  // 7x7 images: a third of the 256-slot tile is padding and the layer is bound by the weight traffic per tile; the
  // forward im2col GEMM measured 7 % faster there (tools/kbench/conv_bench), the data gradient 26 % slower
  if (!dgrad && g_conv3x3_mode < 1 && H * W < 100) return 0;
  if ((((uintptr_t)x) | ((uintptr_t)w) | ((uintptr_t)y)) & 15) return 0;
  const long long Wp = W + 2, IMG = (long long)(H + 1) * Wp, G = (long long)N * IMG;
  if (G + 2 * Wp + C3_MT >= 0x7FFFFFFFLL) return 0;
  const int NT = (CO & 127) == 0 ? 128 : 64;
  const int pieces = (int)((C3_MT + 2 * Wp + 2 + 7) / 8);
  const size_t lds = (size_t)pieces * 1024 + 2 * NT * 64 * 2;
  if (lds > 80 * 1024) return 0;                          // two workgroups per CU
  C3Args p;
  p.x = (const unsigned short*)x; p.w = (const unsigned short*)w; p.y = (unsigned short*)y; p.stats = stats;
  p.N = N; p.H = H; p.W = W; p.CI = CI; p.CO = CO;
  p.Wp = (int)Wp; p.IMG = (int)IMG; p.G = (int)G;
  p.tiles_m = (int)((G + C3_MT - 1) / C3_MT); p.tiles_n = CO / NT; p.pieces = pieces;
  p.dIMG = make_fastdiv(p.IMG); p.dWp = make_fastdiv(p.Wp);
  if (stats && stats_bytes < (long long)p.tiles_m * 2 * CO * 4) {
    dle_set_error("conv3x3: statistics buffer too small (%d tile rows)", p.tiles_m);
    return 2;
  }
  dim3 grid((unsigned)(p.tiles_m * p.tiles_n)), block(256);
#define C3_GO(DT, DG, NTV) do { static bool attr_set = false; \
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)conv3x3_kernel<DT, DG, NTV>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attr_set = true; } \
    hipLaunchKernelGGL((conv3x3_kernel<DT, DG, NTV>), grid, block, lds, stream, p); } while (0)
#define C3_PICK(DT) do { if (dgrad) { if (NT == 128) C3_GO(DT, true, 128); else C3_GO(DT, true, 64); } \
                         else { if (NT == 128) C3_GO(DT, false, 128); else C3_GO(DT, false, 64); } } while (0)
  if (dtype == DLE_F16) C3_PICK(DLE_F16); else C3_PICK(DLE_BF16);
#undef C3_GO
#undef C3_PICK
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { dle_set_error("conv3x3 launch failed: %s", hipGetErrorString(e)); return (int)e + 1000; }
  return 1;
}
