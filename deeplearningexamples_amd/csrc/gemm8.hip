// Launcher of the persistent ping-pong GEMM (kernel and design notes: gemm8_kernel.h) + its store-only instantiations.
#include "gemm8_kernel.h"

// ---- launcher ----------------------------------------------------------------------------------------------------------------
static int g8_mode_value = -2;          // -2: read DLE_GEMM_8PH on first use; 0 off; 1 on
extern "C" int dle_gemm8_mode(int mode) {
  if (g8_mode_value == -2) g8_mode_value = getenv("DLE_GEMM_8PH") ? atoi(getenv("DLE_GEMM_8PH")) : 1;
  const int prev = g8_mode_value;
  if (mode >= 0) g8_mode_value = mode;
  return prev;
}

static int g8_min_items_value = -1;     // -1: read DLE_GEMM_8PH_MIN_ITEMS on first use
extern "C" int dle_gemm8_min_items(int n) {
  if (g8_min_items_value < 0) g8_min_items_value = getenv("DLE_GEMM_8PH_MIN_ITEMS") ? atoi(getenv("DLE_GEMM_8PH_MIN_ITEMS")) : 128;
  const int prev = g8_min_items_value;
  if (n >= 0) g8_min_items_value = n;
  return prev;
}
static long long g8_launches = 0;
extern "C" int64_t dle_gemm8_launch_count(void) { return __atomic_load_n(&g8_launches, __ATOMIC_RELAXED); }

#ifdef G8_TIMING
unsigned long long* g8_dbg_ptr = nullptr;
int g8_dbg_items = 0;
#endif
// instantiation units: gemm8.hip (store-only epilogues), gemm8_epi1.hip (bias / forward activations), gemm8_epi2.hip
// (source-tensor epilogues); each returns 0 for a combination it does not carry
extern "C" int g8_launch_epi1(const Gemm8Args* p, int dt, int am, int bm, int act, int grid, hipStream_t stream);
extern "C" int g8_launch_epi2(const Gemm8Args* p, int dt, int am, int bm, int act, int grid, hipStream_t stream);
static int g8_launch_epi0(const Gemm8Args* p, int dt, int am, int bm, int grid, hipStream_t stream) {
#define G8_E0(DT) do { if (am == 0 && bm == 0) g8_launch<DT, 0, 0, 0, ACT_NONE>(*p, grid, stream); \
    else if (am == 0) g8_launch<DT, 0, 1, 0, ACT_NONE>(*p, grid, stream); else g8_launch<DT, 1, 1, 0, ACT_NONE>(*p, grid, stream); } while (0)
  if (dt == DLE_F16) G8_E0(DLE_F16); else G8_E0(DLE_BF16);
#undef G8_E0
  return 1;
}

// 1: launched; 0: outside the envelope (the caller continues with the kernels of gemm_dma.hip); > 1: error.
// stats != NULL: column sums of the rounded output into stats[2 * ceil(M / 256)][N] (act = ReLU mask / stored derivative only).
static int g8_try(const void* A, const void* B, void* C, void* aux, const float* bias, const void* src, int M, int N,
                  int K, int64_t lda, int64_t ldb, int64_t ldc, int a_kc, int b_kc, int in_dtype, int out_dtype,
                  int act, int splitk, int accumulate, float alpha, float* ws, float* stats, int stats_sq, hipStream_t stream) {
  if (dle_gemm8_mode(-1) <= 0) return 0;
  if (in_dtype != DLE_F16 && in_dtype != DLE_BF16) return 0;
  if (!a_kc && b_kc) return 0;
  if (M < 256 || N < 256 || K < 2 * BK || (K & 7) != 0 || (N & 7) != 0) return 0;
  if ((K % BK) != 0 && !(a_kc && lda >= K && (!b_kc || ldb >= K))) return 0;      // (K tail: per-lane predicate of k-contiguous operands)
  if (!a_kc && (M & 7) != 0) return 0;
  const bool al = ((((uintptr_t)A) | ((uintptr_t)B) | ((uintptr_t)C) | ((uintptr_t)src) | ((uintptr_t)bias) |
                    ((uintptr_t)ws)) & 15) == 0 && (lda & 7) == 0 && (ldb & 7) == 0 && (ldc & 7) == 0 &&
                  (((uintptr_t)aux) & ((act == ACT_ADD_MASKED || act == ACT_RELU_BITS || act == ACT_RELU_BWD_BITS) ? 1 : 15)) == 0;
  if (!al) return 0;
  // operand extents (range-checked DMA) and 32-bit lane offsets
  const long long a_rows = a_kc ? M : K, b_rows = b_kc ? N : K;
  const long long a_bytes = a_rows * lda * 2, b_bytes = b_rows * ldb * 2;
  // (a half-tile may start up to 256 rows past the end of an operand: its 32-bit byte offset must not wrap)
  if ((a_rows + 256) * lda * 2 >= 0xFFFFFFFFLL || (b_rows + 256) * ldb * 2 >= 0xFFFFFFFFLL) return 0;
  if ((long long)M * ldc * 4 >= 0x7FFFFFFFFFFFLL) return 0;
  if (splitk < 1) splitk = 1;
  const int ktiles = (K + BK - 1) / BK;
  if (splitk > ktiles) return 0;
  int epi;
  if (splitk > 1) {
    if (!ws || out_dtype != DLE_F32 || bias || act != ACT_NONE || aux || stats) return 0;
    epi = 0;
  } else if (act == ACT_RELU_BWD_BITS) {
    // C = product under keep bits (aux), no source tensor; optional column sums (the bias gradient of the layer below)
    if (src || !aux || bias || out_dtype != in_dtype || accumulate || alpha != 1.0f || ldc != N || (N & 15) != 0 || b_kc) return 0;
    if ((long long)(M + 256) * ldc * 2 >= 0xFFFFFFFFLL) return 0;
    if (stats && (M & 255) != 0) return 0;
    epi = 2;
  } else if (act == ACT_RELU_BITS) {
    // bias + ReLU forward that also leaves the keep bits of its (dense [M, N]) output in aux
    if (!aux || stats || out_dtype != in_dtype || accumulate || ldc != N || (N & 15) != 0 || !a_kc || !b_kc) return 0;
    if ((long long)(M + 256) * ldc * 2 >= 0xFFFFFFFFLL) return 0;
    epi = 1;
  } else if (act == ACT_RELU_BWD || act == ACT_ADD || act == ACT_ADD_MASKED || act == ACT_MUL || act == ACT_GELU_BWD || act == ACT_TANH_BWD) {
    if (!src || bias || (aux != nullptr) != (act == ACT_ADD_MASKED) || out_dtype != in_dtype || accumulate || alpha != 1.0f) return 0;
    if (act == ACT_ADD_MASKED && (ldc != N || (N & 15) != 0 || b_kc)) return 0;   // (keep bits of a dense [M, N] addend: two aligned bytes per lane)
    if ((long long)(M + 256) * ldc * 2 >= 0xFFFFFFFFLL) return 0;      // 32-bit byte offsets of the range-checked source loads
    if (stats && !(act == ACT_RELU_BWD || act == ACT_MUL)) return 0;
    if (stats && (M & 255) != 0) return 0;
    epi = 2;
  } else if (stats && act == ACT_NONE && stats_sq) {
    // column sums + sums of squares of the rounded 16-bit output (a 1x1 convolution forward with its BatchNorm statistics)
    if (bias || aux || out_dtype != in_dtype || accumulate || alpha != 1.0f || !a_kc || !b_kc || (M & 255) != 0) return 0;
    epi = 3;
  } else if (act == ACT_NONE || act == ACT_RELU || act == ACT_GELU || act == ACT_TANH || act == ACT_GELU_DAUX) {
    if (stats) return 0;
    if (out_dtype == DLE_F32) {
      if (act != ACT_NONE || aux) return 0;
    } else if (out_dtype != in_dtype || accumulate) return 0;
    if (act == ACT_GELU_DAUX && !aux) return 0;
    epi = (bias || act != ACT_NONE || aux) ? 1 : 0;
  } else return 0;
  // the three operand layouts of a linear layer, each with the epilogues its pass uses (the instantiation units decide)
  const int am = a_kc ? 0 : 1, bm = b_kc ? 0 : 1;
  static const int tn_mode = getenv("DLE_GEMM_8PH_TN") ? atoi(getenv("DLE_GEMM_8PH_TN")) : 1;
  if (am == 1 && !tn_mode) return 0;
  const long long tiles = (long long)((M + 255) / 256) * ((N + 255) / 256);
  const long long nitems = tiles * splitk;
  const int min_items = dle_gemm8_min_items(-1);
  if (nitems < min_items || nitems > (1 << 22) || splitk > 0x7FFF || ktiles > 0xFFFF) return 0;      // (walk table fields)
  // device properties cached PER DEVICE (a process that drives a second GPU must not launch with the first one's CU count /
  // LDS limit; g8_launch keys its MaxDynamicSharedMemorySize attribute the same way)
  static int ncu_of[G8_MAX_DEVICES], lds_of[G8_MAX_DEVICES];
  const int dev = g8_current_device();
  if (dev < 0) return 0;
  if (ncu_of[dev] == 0) {
    int v = 0, l = 0;
    hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
    hipDeviceGetAttribute(&l, hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
    lds_of[dev] = l;
    __atomic_store_n(&ncu_of[dev], v > 0 ? v : 256, __ATOMIC_RELEASE);
  }
  const int ncu = ncu_of[dev], lds_max = lds_of[dev];
  if (lds_max < G8_LDS_BYTES) return 0;
  Gemm8Args p = {};
  p.A = (const unsigned short*)A; p.B = (const unsigned short*)B; p.C = C; p.aux = aux; p.bias = bias;
  p.src = (const unsigned short*)src; p.ws = splitk > 1 ? ws : nullptr; p.stats = stats;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.a_bytes = (unsigned)a_bytes; p.b_bytes = (unsigned)b_bytes;
  p.out_dtype = out_dtype; p.act = act; p.splitk = splitk; p.accumulate = accumulate; p.alpha = alpha;
  static const int gm_env = getenv("DLE_GEMM_GM") ? atoi(getenv("DLE_GEMM_GM")) : 8;
  p.gm = gm_env > 0 ? gm_env : 8;
  {
    const long long cb = splitk > 1 ? (long long)M * N * 4 : (long long)M * ldc * (out_dtype == DLE_F32 ? 4 : 2);
    p.c_bytes = cb < 0xFFFFFFFFLL ? (unsigned)cb : 0u;           // 0: the interior fast path is off (c_bytes == 0 below)
  }
  p.nitems = (int)nitems;
  p.tiles_m = (M + 255) / 256; p.tiles_n = (N + 255) / 256; p.ktiles = ktiles;
#ifdef G8_TIMING
  p.dbg = g8_dbg_ptr; p.dbg_items = g8_dbg_items;
#endif
  const int grid = (int)(nitems < ncu ? nitems : ncu);
  p.grid = grid;
  if ((nitems + grid - 1) / grid > G8_TBL_ITEMS) return 0;         // (the workgroup's walk table in LDS)
  int launched;
  if (epi == 3) {
    if (in_dtype == DLE_F16) g8_launch<DLE_F16, 0, 0, 3, ACT_NONE>(p, grid, stream); else g8_launch<DLE_BF16, 0, 0, 3, ACT_NONE>(p, grid, stream);
    launched = 1;
  } else if (epi == 0) launched = g8_launch_epi0(&p, in_dtype, am, bm, grid, stream);
  else if (epi == 1) launched = g8_launch_epi1(&p, in_dtype, am, bm, act, grid, stream);
  else launched = g8_launch_epi2(&p, in_dtype, am, bm, act, grid, stream);
  if (!launched) return 0;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { dle_set_error("gemm8 launch failed: %s", hipGetErrorString(e)); return (int)e + 1000; }
  __atomic_fetch_add(&g8_launches, 1, __ATOMIC_RELAXED);
  return 1;
}

extern "C" int dle_gemm8_try(const void* A, const void* B, void* C, void* aux, const float* bias, const void* src, int M, int N,
                             int K, int64_t lda, int64_t ldb, int64_t ldc, int a_kc, int b_kc, int in_dtype, int out_dtype,
                             int act, int splitk, int accumulate, float alpha, float* ws, float* stats, hipStream_t stream) {
  return g8_try(A, B, C, aux, bias, src, M, N, K, lda, ldb, ldc, a_kc, b_kc, in_dtype, out_dtype, act, splitk, accumulate, alpha, ws,
                stats, 0, stream);
}

// C [M, N] = A [M, K] B [N, K]^T (16-bit, both k-contiguous) AND, per 128 output rows, the column sums and sums of squares of the
// ROUNDED output: stats [M / 128][2][N] (row r = rows 128 r .. 128 r + 127) -- the 1x1 convolution forward with the batch
// statistics of the BatchNorm behind it (dle_conv2d_fwd_colstats; fold with dle_bn_stats_from_partials, groups = M / 128).
// M a multiple of 256.  1: launched; 0: outside the envelope; > 1: error.
extern "C" int dle_gemm8_colstats_try(const void* A, const void* B, void* C, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                                      int dtype, float* stats, hipStream_t stream) {
  if (!stats) return 0;
  return g8_try(A, B, C, nullptr, nullptr, nullptr, M, N, K, lda, ldb, ldc, 1, 1, dtype, dtype, ACT_NONE, 1, 0, 1.0f, nullptr, stats, 1,
                stream);
}

// Y [M, N] = relu(X [M, K] W [N, K]^T + bias) AND the keep bits of Y (bits [M N / 8]: bit (m N + n) & 7 of byte (m N + n) >> 3 =
// rounded Y > 0) -- the forward of one (Linear + ReLU) layer of an MLP (Recommendation/DLRM/dlrm/nn/mlps.py:38-43,106-114) whose
// backward then reads 1 bit per element instead of the 16-bit activation.  Dense Y (ldc = N), N a multiple of 16.
// 1: launched; 0: outside the envelope (the caller runs dle_gemm with DLE_ACT_RELU and keeps Y as the mask source); > 1: error.
extern "C" int dle_gemm8_relu_bits_try(const void* X, const void* W, void* Y, void* bits, const float* bias, int M, int N, int K,
                                       int64_t ldx, int64_t ldw, int dtype, hipStream_t stream) {
  return g8_try(X, W, Y, bits, bias, nullptr, M, N, K, ldx, ldw, N, 1, 1, dtype, dtype, ACT_RELU_BITS, 1, 0, 1.0f, nullptr, nullptr, 0,
                stream);
}

// dX [M, N] = (dY [M, K] W [K, N]) under the keep bits of the layer below (bits as written by dle_gemm8_relu_bits_try for ITS
// output [M, N]) AND, when colsum_partial != NULL, the per-128-row column sums of the rounded dX (fold with colsum_fold_kernel via
// dle_gemm_colsum_bits) -- the data gradient through the ReLU of the layer below with that layer's bias gradient, no 16-bit mask
// source read.  M a multiple of 256 with column sums.  Same return convention.
extern "C" int dle_gemm8_relu_bwd_bits_try(const void* dY, const void* W, void* dX, const void* bits, float* colsum_partial, int M,
                                           int N, int K, int64_t lddy, int64_t ldw, int dtype, hipStream_t stream) {
  return g8_try(dY, W, dX, (void*)bits, nullptr, nullptr, M, N, K, lddy, ldw, N, 1, 0, dtype, dtype, ACT_RELU_BWD_BITS, 1, 0, 1.0f,
                nullptr, colsum_partial, 0, stream);
}
