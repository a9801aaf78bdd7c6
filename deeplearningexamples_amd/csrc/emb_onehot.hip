// Sparse SGD on the TINY embedding tables (<= 128 rows, dim 128) as a streaming MFMA kernel for gfx950.
// Reference: Recommendation/DLRM/dlrm/cuda_src/gather_gpu_fused.cu:161-202 (atomicAdd of every lookup's gradient row into its
// table row) -- a table of 4..104 rows takes all 65536 lookups of the batch on a handful of rows.
//
// The per-row sums of a table are a matrix product:   dW[r, :] = sum_b [id_b == r] * g[b, :] = OneHot(ids)^T  G.
// G ([batch, 128] 16-bit, the gradient rows as they lie in the interaction gradient) streams HBM -> LDS by LDS-DMA in its own
// row-major layout and is read with the LDS transpose read (the contraction runs over samples); the one-hot operand never
// exists in memory: a lane builds its 8-sample fragment from 8 id bytes with compares (1.0 / 0.0 are exact in fp16 / bf16, the
// products are exact, the sums are fp32 -- the arithmetic of the register / LDS forms it replaces).  8 wavefronts = 4 row blocks
// of 32 table rows x 2 sample halves, each owning all 4 column blocks (64 accumulator registers).  One fp32 partial block per
// workgroup (table, batch slice), folded in slice order by a second launch that does ONE read-modify-write per touched table
// element: no float atomics, bit-reproducible.  (The register form it replaces -- emb_sgd_tiny, one gradient row in flight per
// wavefront -- ran the eight tiny tables of criteo_f15 in 110-150 us at batch 65536: 134 MB at ~1 TB/s.)
#include "gemm_tiles.h"

#define OH_TG 64                          // samples per tile
#define OH_D 128                          // embedding dim (columns of G): 4 blocks of 32
#define OH_ROWS 128                       // table rows per workgroup: 4 blocks of 32

struct OhArgs {
  float* weight;
  const long long* rows;                  // [batch, T] global row ids
  const unsigned short* grad;             // sample b, table t at grad + b * gstride + t * 128
  const float* lr_dev;
  float lr_host;
  const float* scale;
  const float* skip;
  float* ws;                              // [n][slices][128][128]
  long long batch, gstride;
  int T, slices, n;
  int t[64];
  long long base[64];
  int nrows[64];
};

template <int OFF>
__device__ __forceinline__ short4_t oh_tr(unsigned addr) {
  short4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}

// the id bytes go through asm LDS instructions as well: a compiler-visible LDS access next to an LDS-DMA in flight gets an
// s_waitcnt vmcnt(0) in front of it (possible alias), which would drain the prefetch of the next tile before this one is computed
__device__ __forceinline__ uint2_t oh_read_ids(unsigned addr) {
  uint2_t v;
  asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void oh_write_id(unsigned addr, unsigned v) {
  asm volatile("ds_write_b8 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

template <int DT>
__global__ __launch_bounds__(512) void emb_onehot_kernel(OhArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (p.skip && *p.skip != 0.0f) return;
  constexpr int LX = OH_D * 2, XBYTES = OH_TG * LX;                      // 256-byte rows, 16 KiB per tile
  constexpr unsigned IDOFF = 2 * XBYTES;                                 // [2][OH_TG] local row ids (255 = no row) behind the tiles
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int ph = wave >> 2, rb = wave & 3;
  const int k = (int)blockIdx.x / p.slices, sl = (int)blockIdx.x - k * p.slices;
  const int t = p.t[k];
  const long long base = p.base[k];
  const long long per = ((p.batch + p.slices - 1) / p.slices + OH_TG - 1) / OH_TG * OH_TG;
  const long long b0 = (long long)sl * per;
  const long long b1 = b0 + per < p.batch ? b0 + per : p.batch;
  const int ntiles = b0 < b1 ? (int)((b1 - b0 + OH_TG - 1) / OH_TG) : 0;
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)p.grad, 0, 0xFFFFFFE0, 0x00020000);

  // 64-byte block q of row r sits at block position q ^ (r & 3) (on the DMA's source address): the 4 consecutive rows x 64 bytes
  // of a 32-lane transpose-read group then hit 8 distinct 32-byte bank slots
  auto issue_tile = [&](int ti, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int pc = wave_u + 8 * q;
      const int ob = pc * 1024 + lane * 16, r = ob >> 8, cb = ob & 255;
      const int blk = (cb >> 6) ^ (r & 3);
      const long long b = b0 + (long long)ti * OH_TG + r;
      dma16(rg, (unsigned short*)(smem_raw + buf * XBYTES + pc * 1024),
            b < b1 ? (unsigned)((b * p.gstride + (long long)t * OH_D) * 2 + blk * 64 + (cb & 63)) : OOB_OFF);
    }
  };
  // raw low word of the global row id (converted where it is written to LDS: the first use of a load is where hipcc waits for
  // it, and vmcnt retires in order -- a use right behind the load would wait for the DMA of the next tile as well)
  const unsigned base_lo = (unsigned)base;
  auto load_id = [&](int ti) __attribute__((always_inline)) -> unsigned {
    const long long b = b0 + (long long)ti * OH_TG + tid;
    return (tid < OH_TG && b < b1) ? (unsigned)p.rows[b * p.T + t] : base_lo + 255u;
  };

  float16_t acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // per-lane byte offset of the first transpose read of k step 0 (+ 16 rows / + 4 rows keep row & 3: immediates)
  const int tg = lane >> 4, ti16 = lane & 15;
  const int r0 = ph * (OH_TG / 2) + (tg >> 1) * 8 + (ti16 >> 2);
  const int chanb = ((tg & 1) * 16 + (ti16 & 3) * 4) * 2;
  unsigned boff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) boff[j] = (unsigned)(r0 * LX + ((j ^ (r0 & 3)) << 6) + chanb);
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)smem_raw;
  const unsigned rmine = (unsigned)(32 * rb + (lane & 31));
  constexpr unsigned ONE = DT == DLE_F16 ? 0x3C00u : 0x3F80u;

  if (ntiles > 0) {
    issue_tile(0, 0);
    const unsigned i0 = load_id(0);
    if (tid < OH_TG) oh_write_id(lds0 + IDOFF + tid, (i0 - base_lo) & 0xFFu);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  for (int it = 0; it < ntiles; ++it) {
    const int buf = it & 1;
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): this wave's pieces of tile `it` have landed
    __syncthreads();                                       // ... everybody's, and the ids; the other buffer is no longer read
    unsigned idn = base_lo + 255u;
    if (it + 1 < ntiles) {
      issue_tile(it + 1, buf ^ 1);
      idn = load_id(it + 1);
    }
    const unsigned bb = lds0 + (unsigned)(buf * XBYTES);
    static_for<0, 2>([&](auto KS) __attribute__((always_inline)) {
      constexpr int ks = decltype(KS)::value;
      TrPair fb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { fb[j].lo = oh_tr<ks * 16 * LX>(bb + boff[j]); fb[j].hi = oh_tr<ks * 16 * LX + 4 * LX>(bb + boff[j]); }
      // one-hot fragment: lane (row rmine, k group lane >> 5) x samples ph * 32 + ks * 16 + 8 (lane >> 5) + 0..7
      const uint2_t ids = oh_read_ids(lds0 + IDOFF + buf * OH_TG + ph * (OH_TG / 2) + ks * 16 + (lane >> 5) * 8);
      frag_wait<true>();
      uint4_t a;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned w = q < 2 ? ids[0] : ids[1];
        const unsigned ia = (w >> (16 * (q & 1))) & 0xFFu, ib = (w >> (16 * (q & 1) + 8)) & 0xFFu;
        a[q] = (ia == rmine ? ONE : 0u) | (ib == rmine ? ONE << 16 : 0u);
      }
      const ushort8_t va = __builtin_bit_cast(ushort8_t, a);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = Mfma32x16<DT>::run(va, frag_value(fb[j]), acc[j]);
    });
    if (tid < OH_TG) oh_write_id(lds0 + IDOFF + (buf ^ 1) * OH_TG + tid, (idn - base_lo) & 0xFFu);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  // ---- the sample halves meet in LDS, then ONE partial block per workgroup (rows of the table only)
  float* red = (float*)smem_raw;                           // [4 row blocks][4][16][64]
  __syncthreads();
  if (ph == 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((rb * 4 + j) * 16 + r) * 64 + lane] = acc[j][r];
  }
  __syncthreads();
  if (ph == 0) {
    const int nrows = p.nrows[k];
    float* out = p.ws + ((long long)k * p.slices + sl) * (OH_ROWS * OH_D);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] += red[((rb * 4 + j) * 16 + r) * 64 + lane];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // D[row][col]: lane holds col = 32 j + (lane & 31), row = 32 rb + 8 (r >> 2) + 4 (lane >> 5) + (r & 3)
        const int row = rb * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        if (row < nrows) out[row * OH_D + c] = acc[j][r];
      }
    }
  }
}

// w[base + r][d] -= lr * scale * sum over slices (index order); rows no sample touched keep their bits
__global__ __launch_bounds__(128) void emb_onehot_fold_kernel(OhArgs p) {
  if (p.skip && *p.skip != 0.0f) return;
  const int k = (int)blockIdx.x / OH_ROWS, r = (int)blockIdx.x - k * OH_ROWS;
  if (r >= p.nrows[k]) return;
  const int d = threadIdx.x;
  const float* src = p.ws + ((long long)k * p.slices * OH_ROWS + r) * OH_D + d;
  const long long step = (long long)OH_ROWS * OH_D;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int s = 0;
  for (; s + 4 <= p.slices; s += 4) {
    s0 += src[(long long)(s + 0) * step]; s1 += src[(long long)(s + 1) * step];
    s2 += src[(long long)(s + 2) * step]; s3 += src[(long long)(s + 3) * step];
  }
  for (; s < p.slices; ++s) s0 += src[(long long)s * step];
  const float v = (s0 + s1) + (s2 + s3);
  if (v != 0.f) {
    const float lr = p.lr_dev ? *p.lr_dev : p.lr_host;
    const float alpha = -lr * (p.scale ? *p.scale : 1.0f);
    float* w = p.weight + (p.base[k] + r) * OH_D + d;
    *w += alpha * v;
  }
}

static int oh_slices(int n, long long batch) {
  long long s = 512 / (n > 0 ? n : 1);
  const long long cap = (batch + 4 * OH_TG - 1) / (4 * OH_TG);           // >= 4 tiles per slice
  if (s > cap) s = cap;
  return (int)(s < 1 ? 1 : s);
}

// workspace for n tiny tables at this batch (0: nothing to do)
extern "C" int64_t dle_emb_onehot_workspace_bytes(int n_tables, int64_t batch) {
  if (n_tables <= 0 || batch <= 0) return 0;
  return (int64_t)n_tables * oh_slices(n_tables, batch) * OH_ROWS * OH_D * 4;
}

// 1: launched; 0: outside the envelope (the caller keeps its register / LDS forms); > 1: error.
extern "C" int dle_emb_onehot_try(float* weight, const int64_t* rows, const void* grad, const float* lr_dev, float lr_host,
                                  const float* scale_dev, const float* skip_flag_dev, const int* tab_t, const int64_t* tab_base,
                                  const int* tab_rows, int n_tab, int64_t batch, int tables, int dim, int64_t grad_batch_stride,
                                  int grad_dtype, void* ws, int64_t ws_bytes, hipStream_t stream) {
  static const int mode = getenv("DLE_EMB_ONEHOT") ? atoi(getenv("DLE_EMB_ONEHOT")) : 1;
  if (!mode || n_tab <= 0 || n_tab > 64 || dim != OH_D || !ws) return 0;
  // NON-FINITE GRADIENTS: dW = OneHot^T G multiplies EVERY gradient row into every row of the tiny table (by 0 or 1), so one
  // inf / NaN gradient row poisons the whole table (0 * inf = NaN), where the register / LDS forms and the reference's atomicAdd
  // only touch the row that was looked up.  The fp16 path is protected by ordering: the GradScaler's found_inf is final BEFORE
  // this launch (dle_check_nonfinite runs on the gradient first) and arrives here as skip_flag_dev, which drops the whole step.
  // A caller without a scaler (bf16, skip_flag_dev == NULL) gets a poisoned table instead of a poisoned row from a non-finite
  // gradient -- the run is lost either way; DLE_EMB_ONEHOT=0 keeps the row-local forms.
  if (grad_dtype != DLE_F16 && grad_dtype != DLE_BF16) return 0;
  if ((grad_batch_stride % 8) != 0 || ((((uintptr_t)grad) | ((uintptr_t)ws)) & 15) != 0) return 0;
  if (batch * grad_batch_stride * 2 >= 0xFFFFFFE0LL || batch < OH_TG) return 0;
  OhArgs p;
  p.weight = weight; p.rows = (const long long*)rows; p.grad = (const unsigned short*)grad; p.lr_dev = lr_dev; p.lr_host = lr_host;
  p.scale = scale_dev; p.skip = skip_flag_dev; p.ws = (float*)ws; p.batch = batch; p.gstride = grad_batch_stride; p.T = tables;
  p.n = n_tab; p.slices = oh_slices(n_tab, batch);
  if (ws_bytes < dle_emb_onehot_workspace_bytes(n_tab, batch)) return 0;
  for (int i = 0; i < n_tab; ++i) {
    if (tab_rows[i] > OH_ROWS || tab_rows[i] > 255) return 0;
    p.t[i] = tab_t[i]; p.base[i] = tab_base[i]; p.nrows[i] = tab_rows[i];
  }
  const size_t lds = 64 * 1024;                                          // 2 x 16 KiB tiles + ids; the meeting buffer of the halves
#define GO(DT)                                                                                                           \
  do {                                                                                                                   \
    static bool attr_set = false;                                                                                        \
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)emb_onehot_kernel<DT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; } \
    hipLaunchKernelGGL((emb_onehot_kernel<DT>), dim3(n_tab * p.slices), dim3(512), lds, stream, p);                      \
  } while (0)
  if (grad_dtype == DLE_F16) GO(DLE_F16); else GO(DLE_BF16);
#undef GO
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { dle_set_error("emb_onehot launch failed: %s", hipGetErrorString(e)); return (int)e + 1000; }
  hipLaunchKernelGGL(emb_onehot_fold_kernel, dim3(n_tab * OH_ROWS), dim3(128), 0, stream, p);
  e = hipGetLastError();
  if (e != hipSuccess) { dle_set_error("emb_onehot fold launch failed: %s", hipGetErrorString(e)); return (int)e + 1000; }
  return 1;
}
