// DLRM embedding path for gfx950: multi-table gather (forward), index+offset arithmetic,
// sparse-gradient values, and the fused sparse SGD row update.
//
// Replaces (DLRM/dlrm/cuda_src/...):
//   gather_gpu_fused.cu:107-159  lookupEmbeddings      (26 tables x dim128, warp-32 shuffle)
//   gather_gpu_fused.cu:161-202  indices_offset_addition + gradient_copy_kernel
//   sparse_gather/gather_gpu.cu:15-49   GatherKernel (joint table, pre-offset indices)
//   sparse_gather/gather_gpu.cu:53-75   GatherBackwardFuseSgdKernel (atomicAdd(W[idx], -lr*g))
//
// Design: these are pure HBM gather/scatter ops (512 B fp32 rows at dim 128).
//  * one half-wavefront (32 lanes x 16 B) streams one 128-float row, so a wave64 keeps two
//    independent rows in flight per instruction and UNROLL x 2 rows per loop trip; rows are
//    walked grid-stride so the launch is a few workgroups per CU regardless of batch.
//  * any table count / any dim that is a multiple of 4 (the reference hard-codes 26 x 128).
//  * index arithmetic is int64 end to end (bit-exact with idx + offsets[t], `%` hashing).
//  * the SGD update uses the hardware fp32 atomic add (global_atomic_add_f32); duplicates of a
//    row inside a batch accumulate exactly like the reference's atomicAdd.
// Algorithmic bytes / looked-up row (dim D, fp32 table): fwd D*4 read + D*e write + 8 B index;
// update: D*e grad read + 2*D*4 row read-modify-write + 8 B index.
#include "common.h"

template <int ODT> struct Out;
template <> struct Out<DLE_F32> {
  typedef float4_t V;   // 4 elements
  static __device__ __forceinline__ V pack(float4_t v) { return v; }
};
template <> struct Out<DLE_F16> {
  typedef ushort4_t V;
  static __device__ __forceinline__ V pack(float4_t v) {
    V o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = Elem<DLE_F16>::from_f32(v[i]);
    return o;
  }
};
template <> struct Out<DLE_BF16> {
  typedef ushort4_t V;
  static __device__ __forceinline__ V pack(float4_t v) {
    V o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = Elem<DLE_BF16>::from_f32(v[i]);
    return o;
  }
};

// rows_total = B*T looked-up rows.  offsets == nullptr -> indices already address the joint table.
// hash_sizes != nullptr -> idx = idx mod size[t] first (embeddings.py:132-134, python floor-mod).
template <int ODT, int UNROLL>
__global__ __launch_bounds__(256) void emb_gather_fwd(const float* __restrict__ weight,
                                                      const long long* __restrict__ indices,
                                                      const long long* __restrict__ offsets,
                                                      const long long* __restrict__ hash_sizes,
                                                      typename Out<ODT>::V* __restrict__ out,
                                                      long long rows_total, int T, int D4,
                                                      long long out_bstride4) {
  // lanes are split in groups of G = min(32, pow2 >= D4) lanes per row
  int G = 32;
  while (G > 1 && (G >> 1) >= D4) G >>= 1;
  const int rows_per_wave = 64 / G;
  const int sub = (threadIdx.x & 63) / G, l = (threadIdx.x & 63) % G;
  const long long wave_id = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
  const long long step = n_waves * rows_per_wave * UNROLL;
  for (long long base = wave_id * rows_per_wave * UNROLL; base < rows_total; base += step) {
    long long row[UNROLL];
    long long src[UNROLL];
    long long orow[UNROLL];   // output offset in units of 4 elements: b * out_bstride4 + t * D4
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      row[u] = base + (long long)u * rows_per_wave + sub;
      src[u] = -1;
      orow[u] = 0;
      if (row[u] < rows_total) {
        long long ix = indices[row[u]];
        const long long bb = row[u] / T;
        const int t = (int)(row[u] - bb * T);
        orow[u] = bb * out_bstride4 + (long long)t * D4;
        if (hash_sizes) {
          const long long m = hash_sizes[t];
          ix %= m;
          if (ix < 0) ix += m;
        }
        if (offsets) ix += offsets[t];
        src[u] = ix;
      }
    }
    for (int c = l; c < D4; c += G) {
      float4_t v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
        if (src[u] >= 0) v[u] = *(const float4_t*)(weight + src[u] * (long long)(D4 * 4) + c * 4);
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
        if (src[u] >= 0) out[orow[u] + c] = Out<ODT>::pack(v[u]);
    }
  }
}

// rows_out[b,t] = (hash? idx mod size[t] : idx) + offsets[t]       (int64, bit exact)
__global__ void emb_offset_indices(const long long* __restrict__ indices,
                                   const long long* __restrict__ offsets,
                                   const long long* __restrict__ hash_sizes,
                                   long long* __restrict__ rows_out, long long n, int T) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    long long ix = indices[i];
    if (hash_sizes) {
      const long long m = hash_sizes[t];
      ix %= m;
      if (ix < 0) ix += m;
    }
    rows_out[i] = ix + (offsets ? offsets[t] : 0);
  }
}

template <int IDT> struct In4;
template <> struct In4<DLE_F32> {
  typedef float4_t V;
  static __device__ __forceinline__ float4_t up(V v) { return v; }
};
template <> struct In4<DLE_F16> {
  typedef ushort4_t V;
  static __device__ __forceinline__ float4_t up(V v) {
    float4_t o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = Elem<DLE_F16>::to_f32(v[i]);
    return o;
  }
};
template <> struct In4<DLE_BF16> {
  typedef ushort4_t V;
  static __device__ __forceinline__ float4_t up(V v) {
    float4_t o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = Elem<DLE_BF16>::to_f32(v[i]);
    return o;
  }
};

// values[i] = (float) grad[i] * (*scale)   -- the fp32 COO "values" of the sparse gradient
// (gather_gpu_fused.cu:177-202 gradient_copy_kernel)
template <int IDT>
__global__ void emb_grad_values(const typename In4<IDT>::V* __restrict__ grad, float4_t* __restrict__ values,
                                const float* __restrict__ scale, long long n4) {
  const float s = scale ? *scale : 1.0f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (long long)gridDim.x * blockDim.x) {
    float4_t v = In4<IDT>::up(grad[i]);
    values[i] = v * s;
  }
}

// W[rows[i], :] += alpha * grad[i, :]  with alpha = -lr * (*scale)   (duplicates accumulate)
template <int IDT>
__global__ __launch_bounds__(256) void emb_sparse_sgd(float* __restrict__ weight,
                                                      const long long* __restrict__ rows,
                                                      const typename In4<IDT>::V* __restrict__ grad,
                                                      const float* __restrict__ lr_dev, float lr_host,
                                                      const float* __restrict__ scale,
                                                      const float* __restrict__ skip_flag,
                                                      long long n_rows, int D4, int T, long long g_bstride4) {
  if (skip_flag && *skip_flag != 0.0f) return;   // found_inf -> skip the whole update
  const float lr = lr_dev ? *lr_dev : lr_host;
  const float alpha = -lr * (scale ? *scale : 1.0f);
  int G = 32;
  while (G > 1 && (G >> 1) >= D4) G >>= 1;
  const int rows_per_wave = 64 / G;
  const int sub = (threadIdx.x & 63) / G, l = (threadIdx.x & 63) % G;
  const long long wave_id = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
  for (long long base = wave_id * rows_per_wave; base < n_rows; base += n_waves * rows_per_wave) {
    const long long r = base + sub;
    if (r >= n_rows) continue;
    const long long dst = rows[r];
    const long long gb = r / T;
    const long long goff = gb * g_bstride4 + (r - gb * T) * D4;
    for (int c = l; c < D4; c += G) {
      const float4_t g = In4<IDT>::up(grad[goff + c]);
      float* w = weight + dst * (long long)(D4 * 4) + c * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) unsafeAtomicAdd(w + k, alpha * g[k]);
    }
  }
}

static int grid_for(long long work_items, int per_block) {
  long long g = (work_items + per_block - 1) / per_block;
  const long long cap = 256 * 8;   // <= 8 workgroups per CU, grid-stride beyond
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int dle_emb_gather_fwd(const float* weight, const int64_t* indices, const int64_t* offsets,
                                  const int64_t* hash_sizes, void* out, int64_t batch, int tables,
                                  int dim, int out_dtype, int64_t out_batch_stride, hipStream_t stream) {
  DLE_CHECK_ARG(batch >= 0 && tables > 0, "emb_gather_fwd: bad shape");
  if (batch == 0) return 0;
  DLE_CHECK_ARG(weight && indices && out, "emb_gather_fwd: null pointer");
  DLE_CHECK_ARG(dim > 0 && dim % 4 == 0, "emb_gather_fwd: dim %d must be a multiple of 4", dim);
  DLE_CHECK_ARG(tables > 0 && batch >= 0, "emb_gather_fwd: bad shape");
  DLE_CHECK_ARG((((uintptr_t)weight) & 15) == 0 && (((uintptr_t)out) & 7) == 0, "emb_gather_fwd: misaligned");
  const long long rows_total = (long long)batch * tables;
  if (rows_total == 0) return 0;
  const int D4 = dim / 4;
  if (out_batch_stride == 0) out_batch_stride = (int64_t)tables * dim;
  DLE_CHECK_ARG(out_batch_stride % 4 == 0 && out_batch_stride >= (int64_t)tables * dim,
                "emb_gather_fwd: bad output batch stride %lld", (long long)out_batch_stride);
  const long long obs4 = out_batch_stride / 4;
  const int grid = grid_for(rows_total, 4 * 2 * 4);
  dim3 block(256);
#define GO(ODT)                                                                                    \
  hipLaunchKernelGGL((emb_gather_fwd<ODT, 4>), dim3(grid), block, 0, stream, weight,                \
                     (const long long*)indices, (const long long*)offsets, (const long long*)hash_sizes, \
                     (typename Out<ODT>::V*)out, rows_total, tables, D4, obs4)
  if (out_dtype == DLE_F32) GO(DLE_F32);
  else if (out_dtype == DLE_F16) GO(DLE_F16);
  else if (out_dtype == DLE_BF16) GO(DLE_BF16);
  else { dle_set_error("emb_gather_fwd: bad out dtype %d", out_dtype); return -1; }
#undef GO
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_emb_offset_indices(const int64_t* indices, const int64_t* offsets,
                                      const int64_t* hash_sizes, int64_t* rows_out, int64_t batch,
                                      int tables, hipStream_t stream) {
  const long long n = (long long)batch * tables;
  if (n == 0) return 0;
  DLE_CHECK_ARG(indices && rows_out, "emb_offset_indices: null pointer");
  hipLaunchKernelGGL(emb_offset_indices, dim3(grid_for(n, 256)), dim3(256), 0, stream,
                     (const long long*)indices, (const long long*)offsets, (const long long*)hash_sizes,
                     (long long*)rows_out, n, tables);
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_emb_grad_values(const void* grad, float* values, const float* scale_dev,
                                   int64_t n_elems, int grad_dtype, hipStream_t stream) {
  DLE_CHECK_ARG(grad && values, "emb_grad_values: null pointer");
  DLE_CHECK_ARG(n_elems % 4 == 0, "emb_grad_values: element count must be a multiple of 4");
  const long long n4 = n_elems / 4;
  if (n4 == 0) return 0;
  const int grid = grid_for(n4, 256);
  if (grad_dtype == DLE_F32)
    hipLaunchKernelGGL(emb_grad_values<DLE_F32>, dim3(grid), dim3(256), 0, stream, (const float4_t*)grad, (float4_t*)values, scale_dev, n4);
  else if (grad_dtype == DLE_F16)
    hipLaunchKernelGGL(emb_grad_values<DLE_F16>, dim3(grid), dim3(256), 0, stream, (const ushort4_t*)grad, (float4_t*)values, scale_dev, n4);
  else if (grad_dtype == DLE_BF16)
    hipLaunchKernelGGL(emb_grad_values<DLE_BF16>, dim3(grid), dim3(256), 0, stream, (const ushort4_t*)grad, (float4_t*)values, scale_dev, n4);
  else { dle_set_error("emb_grad_values: bad dtype %d", grad_dtype); return -1; }
  DLE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Duplicate-free sparse SGD (the fast path of the train step).
//
// fp32 atomics run at ~0.77 TB/s on this part, 10x under the HBM roofline, and tiny tables (4..100 rows,
// hit by every sample) serialise on a handful of L2 lines.  Instead, by table size (the paths of dle_emb_sgd_dedup_ws):
//   * <= 128 rows at dim 128, 16-bit gradients, scratch given: OneHot(ids)^T G on the matrix pipe (emb_onehot.hip);
//     without scratch / for other dims: the register form (emb_sgd_tiny) or the LDS form (emb_sgd_small: each workgroup reduces
//     a slice of the batch for one table into an LDS-resident dense [rows, dim] fp32 image and flushes it once);
//   * every other table: one pass threads the batch's lookups into per-row linked lists
//     (head[row] <- atomicExch, next[i] <- previous head: one 4 B atomic per lookup), a second pass lets
//     the list head of each touched row sum its duplicates in fp32 and do ONE plain read-modify-write of
//     the 512 B row -- no float atomics, HBM sees row read + row write + one grad read per lookup;
//   * of those, tables of <= 4096 rows (with scratch): EIGHT lists per row and a third pass that adds their partial sums
//     (MidMap below) -- their ~70 duplicates per row were one ~95-hop dependent chain per row otherwise.
// head[] (int32 per table row, all -1 between calls) is persistent workspace owned by the caller.
struct SmallTables {
  int n;
  int t[64];          // table (column) index
  long long base[64]; // first row in the joint table
  int rows[64];       // row count
};

template <int IDT>
__global__ __launch_bounds__(256) void emb_sgd_small(float* __restrict__ weight, const long long* __restrict__ rows,
                                                     const typename In4<IDT>::V* __restrict__ grad,
                                                     const float* __restrict__ lr_dev, float lr_host,
                                                     const float* __restrict__ scale,
                                                     const float* __restrict__ skip_flag, SmallTables st,
                                                     long long batch, int T, int D4, long long g_bstride4,
                                                     int slices) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (skip_flag && *skip_flag != 0.0f) return;
  float* acc = (float*)smem_raw;
  const int k = blockIdx.x / slices, sl = blockIdx.x - k * slices;
  const int t = st.t[k], nrows = st.rows[k];
  const long long base = st.base[k];
  const int D = D4 * 4;
  for (int q = threadIdx.x; q < nrows * D; q += 256) acc[q] = 0.f;
  __syncthreads();
  const long long per = (batch + slices - 1) / slices;
  const long long b0 = sl * per;
  long long b1 = b0 + per;
  if (b1 > batch) b1 = batch;
  const int hw = threadIdx.x >> 5, l = threadIdx.x & 31;   // 8 half-waves, UB samples each per trip
  // the row ids and the gradient rows of UB samples are requested BEFORE the first LDS add: with one sample in flight per
  // half-wave (24 per CU) the pass was bound by the two dependent global loads of every trip (0.38 TB/s)
  constexpr int UB = 4;
  long long b = b0 + (long long)hw * UB;
  if (D4 <= 32) {
    for (; b + UB <= b1; b += 8 * UB) {
      long long r[UB];
      typename In4<IDT>::V gv[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        r[u] = rows[(b + u) * T + t] - base;
        if (l < D4) gv[u] = grad[(b + u) * g_bstride4 + (long long)t * D4 + l];
      }
      if (l < D4) {
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const float4_t g = In4<IDT>::up(gv[u]);
          float* a = acc + r[u] * D + l * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j) atomicAdd(a + j, g[j]);   // ds_add_f32
        }
      }
    }
  }
  // tail of the slice (and rows wider than 32 float4 chunks): one sample per half-wave trip
  for (long long bb = (D4 <= 32 ? b : b0 + (long long)hw * UB), e = bb + UB; bb < b1; ) {
    const long long r = rows[bb * T + t] - base;
    const long long goff = bb * g_bstride4 + (long long)t * D4;
    for (int c = l; c < D4; c += 32) {
      const float4_t g = In4<IDT>::up(grad[goff + c]);
      float* a = acc + r * D + c * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(a + j, g[j]);   // ds_add_f32
    }
    if (++bb == e) { bb += 7 * UB; e = bb + UB; }            // this half-wave's next group of UB samples
  }
  __syncthreads();
  const float lr = lr_dev ? *lr_dev : lr_host;
  const float alpha = -lr * (scale ? *scale : 1.0f);
  for (int q = threadIdx.x; q < nrows * D; q += 256) {
    const float v = acc[q];
    if (v != 0.f) unsafeAtomicAdd(weight + base * D + q, alpha * v);
  }
}

// Tiny tables, second form (<= 128 rows, dim <= 128): the per-row sums of a batch slice are accumulated in REGISTERS.
// A wavefront takes one sample at a time (lane = 2 consecutive dims of the 128-wide gradient row: one coalesced 256-byte
// load per sample), so the row id is wave-uniform and selects the accumulator pair by register index (v_movrel) -- no LDS
// read-modify-write per lookup, no same-address serialisation (a 4-row table takes every lookup of the batch on 4 rows).
// 16 wavefronts per workgroup; tables with more than 32 / 64 rows split their rows over 2 / 4 wavefront groups, each
// wavefront scans 64 row ids at a time (one per lane), ballots the samples whose row it owns and requests only those
// gradient rows, DLE_EMB_TINY_U at a time -- every load in flight is a useful one.  The 16 register images meet in LDS once
// per workgroup (plain read-modify-writes in turns, see below); one fp32 atomic per (slice, row, dim) reaches the table (32 slices: 1.4 M adds for the eight tiny tables of
// criteo_f15, where the LDS form issued 67 M ds_add_f32 on a handful of rows and 5.4 M global adds: 341 us per step).
#define DLE_EMB_TINY_U 16
template <int IDT> struct In2;
template <> struct In2<DLE_F32> {
  typedef float2_t V;
  static __device__ __forceinline__ void up(V v, float& a, float& b) { a = v[0]; b = v[1]; }
};
template <> struct In2<DLE_F16> {
  typedef unsigned V;
  static __device__ __forceinline__ void up(V v, float& a, float& b) {
    a = Elem<DLE_F16>::to_f32((unsigned short)(v & 0xFFFFu)); b = Elem<DLE_F16>::to_f32((unsigned short)(v >> 16));
  }
};
template <> struct In2<DLE_BF16> {
  typedef unsigned V;
  static __device__ __forceinline__ void up(V v, float& a, float& b) {
    a = __builtin_bit_cast(float, v << 16); b = __builtin_bit_cast(float, v & 0xFFFF0000u);
  }
};

template <int IDT>
__global__ __launch_bounds__(1024) void emb_sgd_tiny(float* __restrict__ weight, const long long* __restrict__ rows,
                                                     const void* __restrict__ grad_v,
                                                     const float* __restrict__ lr_dev, float lr_host,
                                                     const float* __restrict__ scale,
                                                     const float* __restrict__ skip_flag, SmallTables st,
                                                     long long batch, int T, int D, long long g_bstride, int slices) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (skip_flag && *skip_flag != 0.0f) return;
  typedef typename In2<IDT>::V V2;
  const V2* __restrict__ grad = (const V2*)grad_v;                 // pairs of elements
  float* sum = (float*)smem_raw;                                   // [nrows][D]
  const int k = blockIdx.x / slices, sl = blockIdx.x - k * slices;
  const int t = st.t[k], nrows = st.rows[k];
  const long long base = st.base[k];
  for (int q = threadIdx.x; q < nrows * D; q += 1024) sum[q] = 0.f;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;        // 16 wavefronts
  const int split = nrows <= 32 ? 1 : (nrows <= 64 ? 2 : 4);
  const int nrw = (nrows + split - 1) / split;                     // rows per wavefront group (<= 32)
  const int r0 = (w % split) * nrw;
  const int phase = w / split, nphase = 16 / split;
  const bool active = 2 * lane < D;
  const long long per = (batch + slices - 1) / slices;
  const long long b0 = sl * per;
  long long b1 = b0 + per;
  if (b1 > batch) b1 = batch;
  typedef float f32x32_t __attribute__((ext_vector_type(32)));
  f32x32_t acc0 = 0.f, acc1 = 0.f;                                 // [row of this wavefront's group]: even / odd dim of the lane
  const long long half_stride = g_bstride >> 1;                    // gradient row stride in element pairs
  const long long toff = ((long long)t * D) >> 1;
  // The row ids of the NEXT chunk are requested together with this chunk's first gradient rows.  The explicit vmcnt(0) at the
  // head of every batch costs nothing (the previous batch has been consumed) and is there for the COMPILER: without it hipcc
  // assumes a load of an earlier batch may still target gv[u] (a batch can end early) and puts s_waitcnt vmcnt(0) in front of
  // EVERY load of the batch -- one gradient row in flight per wavefront (221 us for the eight tiny tables).
  long long rid_next = -1;
  {
    const long long bl = b0 + (long long)phase * 64 + lane;
    if (bl < b1) rid_next = rows[bl * T + t];
  }
  for (long long bb = b0 + (long long)phase * 64; bb < b1; bb += (long long)nphase * 64) {
    const long long rid_raw = rid_next;
    int rid = -1;
    if (bb + lane < b1) rid = (int)(rid_raw - base) - r0;
    unsigned long long mask = __ballot(rid >= 0 && rid < nrw);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    {
      const long long bl = bb + (long long)nphase * 64 + lane;
      rid_next = -1;
      if (bl < b1) rid_next = rows[bl * T + t];
    }
    while (mask) {
      int ri[DLE_EMB_TINY_U];
      __builtin_amdgcn_s_waitcnt(0x0F70);      // (see above; rid_next may be waited for here: it was issued a batch ago)
      V2 gv[DLE_EMB_TINY_U];
#pragma unroll
      for (int u = 0; u < DLE_EMB_TINY_U; ++u) {
        ri[u] = -1;
        if (mask) {
          const int sidx = __builtin_ctzll(mask);
          mask &= mask - 1;
          ri[u] = __builtin_amdgcn_readlane(rid, sidx);
          if (active) gv[u] = grad[(bb + sidx) * half_stride + toff + lane];
        }
      }
      // (unrolled: with register-indexed accumulators a copy is ~15 instructions; a uniform select of (ri[u], gv[u]) by a
      //  run-time u cost 15 compares + selects per sample -- 90 scalar instructions per sample in the PMC counts)
#pragma unroll
      for (int u = 0; u < DLE_EMB_TINY_U; ++u) {
        const int r = ri[u];
        if (r < 0) break;
        float g0 = 0.f, g1 = 0.f;
        if (active) In2<IDT>::up(gv[u], g0, g1);
        acc0[r] += g0;                     // r is wave-uniform (v_readlane): register-indexed, no LDS, no branch
        acc1[r] += g1;
      }
    }
  }
  // The 16 register images meet in LDS WITHOUT float atomics: ds_add_f32 retires about one lane per clock per CU on this part
  // (PMC: 336 stall cycles per LDS instruction; the 65 K lane-adds of this merge were ~85 % of the kernel's 190 us, and the 67 M
  // of the LDS form its 341 us).  Wavefronts that share rows take turns: in turn `ph` the wavefronts of sample phase ph -- one
  // per row group, disjoint rows -- add their pairs with a plain 8-byte read-modify-write.
  for (int ph = 0; ph < nphase; ++ph) {
    __syncthreads();                                               // sum[] is zeroed / the previous turn is done
    if (phase == ph && active) {
#pragma unroll
      for (int c = 0; c < 32; ++c) {
        const int r = r0 + c;
        if (c < nrw && r < nrows) {
          float2_t* dst = (float2_t*)(sum + r * D + 2 * lane);
          float2_t v = *dst;
          v[0] += acc0[c]; v[1] += acc1[c];
          *dst = v;
        }
      }
    }
  }
  __syncthreads();
  const float lr = lr_dev ? *lr_dev : lr_host;
  const float alpha = -lr * (scale ? *scale : 1.0f);
  for (int q = threadIdx.x; q < nrows * D; q += 1024) {
    const float v = sum[q];
    if (v != 0.f) unsafeAtomicAdd(weight + base * D + q, alpha * v);
  }
}

// pass 1: thread every lookup of a "large" table into its row's list
// Lookups of the LARGE tables only: i in [0, batch * nl) -> sample b = i / nl, table t[i % nl], lookup index b * T + t (the value
// threaded through head[] / next[]).  The walk used to cover all batch * T lookups and skip the small tables' ones after
// loading their row id, next pointer and GRADIENT ROW (8 of criteo_f15's 26 tables: 134 MB of reads nobody used), with a 64-bit
// division by T per lookup; 32-bit multiply-shift divisions here (batch * T < 2^31 is checked by the launcher).
struct LookupMap {
  int nl, T;                 // nl == 0: every table, small ones skipped through is_small[] (more than 128 tables)
  unsigned mul_nl, shr_nl, mul_t, shr_t;
  unsigned char t[128];
};
static void emb_make_div(int d, unsigned& mul, unsigned& shr) {
  if (d <= 1) { mul = 0; shr = 0; return; }
  unsigned lg = 0;
  while ((1u << lg) < (unsigned)d) ++lg;
  const unsigned p = 31 + lg;
  mul = (unsigned)(((1ull << p) + (unsigned)d - 1) / (unsigned)d);
  shr = p - 32;
}
__device__ __forceinline__ int emb_div(int n, int d, unsigned mul, unsigned shr) {
  return d <= 1 ? n : (int)(__umulhi((unsigned)n, mul) >> shr);
}
// -> lookup index (or -1 for a small table's lookup in the every-table form), its (sample, table) and its slot k in the map
__device__ __forceinline__ int emb_lookup(const LookupMap& m, const unsigned char* t_lds, const unsigned char* is_small, int i,
                                          int& b, int& t, int& k) {
  const int nl = m.nl ? m.nl : m.T;
  b = emb_div(i, nl, m.nl ? m.mul_nl : m.mul_t, m.nl ? m.shr_nl : m.shr_t);
  k = i - b * nl;
  t = m.nl ? (int)t_lds[k] : k;
  if (!m.nl && is_small && is_small[t]) return -1;
  return b * m.T + t;
}

// MID tables (more rows than the LDS / one-hot forms take, at most EMB_MID_ROWS): 65536 lookups on 1-2 k rows make lists of ~70
// duplicates, and a list is walked by ONE half-wavefront, one dependent hop (next pointer + gradient row) at a time: ~95 hops
// x 2.5 us for the longest one, which no amount of parallelism elsewhere shortens (tools/probes/emb_chain_probe.py: -92 us when
// those tables have no duplicates).  Their rows get EMB_MID_S lists each, one per residue of the sample index: 8 x shorter
// chains, walked by 8 half-wavefronts; a list's head writes its fp32 partial sum into scratch instead of updating the row,
// and a small third pass adds a row's partials in residue order and does the ONE read-modify-write.
#define EMB_MID_S 8
#define EMB_MID_ROWS 4096
#define EMB_MID_TABLES 16
struct MidMap {
  int n;                               // mid tables (0: the feature is off)
  int rows_total;                      // sum of their row counts
  int* head;                           // [rows_total * S]: -1 on entry (memset per call); a head that has written its partial: -2
  float* partial;                      // [rows_total * S][dim]
  int off[128];                        // per slot k of the LookupMap: first row of that table in head / partial, -1: not a mid table
  long long base[128];                 // per slot k: first joint row of the table
};
struct MidFold {                       // the same tables, listed for the fold pass
  int n, rows_total;
  int off[EMB_MID_TABLES], rows[EMB_MID_TABLES];
  long long base[EMB_MID_TABLES];
};

__global__ __launch_bounds__(256) void emb_link(const long long* __restrict__ rows, int* __restrict__ head,
                                                int* __restrict__ next, const unsigned char* __restrict__ is_small,
                                                const float* __restrict__ skip_flag, int n_lookups, LookupMap map, MidMap mid) {
  __shared__ unsigned char t_lds[128];
  __shared__ int moff_lds[128];
  __shared__ long long mbase_lds[128];
  if (skip_flag && *skip_flag != 0.0f) return;
  if (threadIdx.x < 128) {
    t_lds[threadIdx.x] = map.t[threadIdx.x];
    moff_lds[threadIdx.x] = mid.n ? mid.off[threadIdx.x] : -1;
    mbase_lds[threadIdx.x] = mid.n ? mid.base[threadIdx.x] : 0;
  }
  __syncthreads();
  for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < n_lookups; i += (int)(gridDim.x * blockDim.x)) {
    int b, t, k;
    const int li = emb_lookup(map, t_lds, is_small, i, b, t, k);
    if (li < 0) continue;
    const long long r = rows[li];
    const int mo = map.nl ? moff_lds[k] : -1;
    int* slot = mo >= 0 ? mid.head + ((long long)(mo + (int)(r - mbase_lds[k])) * EMB_MID_S + (b & (EMB_MID_S - 1))) : head + r;
    next[li] = atomicExch(slot, li);
  }
}

// pass 2: the head of each list applies the summed update with a plain read-modify-write.
// The first form walked a chain of FOUR dependent loads per lookup (row id -> list head -> gradient / next -> table row), two
// lookups in flight per wavefront: 3.6 TB/s, bound by the chain's latency, not by bytes.  Here the depth is two:
//   level 0 (depends on the lookup index only): row id, next pointer, the lookup's own gradient row;
//   level 1 (depends on the row id): list head AND the table row, requested together (speculatively: a lookup that turns out not
//           to be its list's head wastes one row read, which duplicates of the small and medium tables find in L2);
//   then the store.  Duplicates behind the head are walked as before (gradient + next per step).
// Two lookups per half-wavefront (four per wavefront) are in flight, every load unconditional on a clamped address.
template <int IDT, bool SPEC>
__global__ __launch_bounds__(256) void emb_sgd_lists(float* __restrict__ weight, const long long* __restrict__ rows,
                                                     const typename In4<IDT>::V* __restrict__ grad,
                                                     int* __restrict__ head, const int* __restrict__ next,
                                                     const unsigned char* __restrict__ is_small,
                                                     const float* __restrict__ lr_dev, float lr_host,
                                                     const float* __restrict__ scale,
                                                     const float* __restrict__ skip_flag, int n, LookupMap map,
                                                     int D4, long long g_bstride4, MidMap mid) {
  __shared__ unsigned char t_lds[128];
  __shared__ int moff_lds[128];
  __shared__ long long mbase_lds[128];
  if (skip_flag && *skip_flag != 0.0f) return;
  if (threadIdx.x < 128) {
    t_lds[threadIdx.x] = map.t[threadIdx.x];
    moff_lds[threadIdx.x] = mid.n ? mid.off[threadIdx.x] : -1;
    mbase_lds[threadIdx.x] = mid.n ? mid.base[threadIdx.x] : 0;
  }
  __syncthreads();
  const int T = map.T;
  const float lr = lr_dev ? *lr_dev : lr_host;
  const float alpha = -lr * (scale ? *scale : 1.0f);
  const int sub = (threadIdx.x & 63) >> 5, l = threadIdx.x & 31;
  const int lc = l < D4 ? l : D4 - 1;             // (dim < 128: the upper lanes shadow the last column and do not store)
  const int wave_id = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  const int n_waves = (int)(gridDim.x * (blockDim.x >> 6));
  constexpr int U = 2;
  for (int base = wave_id * (2 * U); base < n; base += n_waves * (2 * U)) {
    long long r[U];
    int iu[U], nx[U], h[U], ms[U];                  // ms: sub-list slot of a mid table's lookup, -1 otherwise
    bool ok[U];
    typename In4<IDT>::V g[U];
    float4_t wv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // LAST lookup first: the head of a row's list is the lookup that was linked last, i.e. (nearly always) the one with the
      // highest sample index -- walking the batch from its end starts the long duplicate chains of the 1-2 k-row tables (~70
      // dependent hops each) at the START of the kernel, under everything else, instead of as its tail
      const int i = n - 1 - (base + 2 * u + sub);
      int jb, t, k;
      const int li = emb_lookup(map, t_lds, is_small, i >= 0 ? i : 0, jb, t, k);
      ok[u] = i >= 0 && li >= 0;
      iu[u] = li >= 0 ? li : jb * T + t;            // (a small table's lookup: loads stay in range, nothing is stored)
      r[u] = rows[iu[u]];
      nx[u] = next[iu[u]];
      g[u] = grad[jb * g_bstride4 + (long long)t * D4 + lc];
      const int mo = map.nl ? moff_lds[k] : -1;
      ms[u] = mo >= 0 ? (mo + (int)(r[u] - mbase_lds[k])) * EMB_MID_S + (jb & (EMB_MID_S - 1)) : -1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      h[u] = *(ms[u] >= 0 ? (const int*)mid.head + ms[u] : (const int*)head + r[u]);
      if (SPEC) wv[u] = ((const float4_t*)(weight + r[u] * (long long)(D4 * 4)))[lc];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!ok[u] || h[u] != iu[u]) continue;        // small-table lookup / not the list head: the head does the work
      if (!SPEC) wv[u] = ((const float4_t*)(weight + r[u] * (long long)(D4 * 4)))[lc];     // (DLE_EMB_SPEC=0: heads only, depth 3)
      float4_t sacc = In4<IDT>::up(g[u]);
      int j = nx[u];
      while (j >= 0) {
        const int jb = emb_div(j, T, map.mul_t, map.shr_t);
        sacc += In4<IDT>::up(grad[jb * g_bstride4 + (long long)(j - jb * T) * D4 + lc]);
        j = next[j];
      }
      if (ms[u] >= 0) {                             // mid table: this list's share of the row's sum goes to scratch (D4 <= 32)
        if (l < D4) ((float4_t*)(mid.partial + (long long)ms[u] * (D4 * 4)))[l] = sacc;
        if (l == 0) mid.head[ms[u]] = -2;
        continue;
      }
      float4_t* w = (float4_t*)(weight + r[u] * (long long)(D4 * 4));
      if (l < D4) w[l] = wv[u] + alpha * sacc;
      for (int c = l + 32; c < D4; c += 32) {       // dim > 128: the remaining columns, chain walked again per trip
        float4_t s2 = {0.f, 0.f, 0.f, 0.f};
        int j2 = iu[u];
        while (j2 >= 0) {
          const int jb = emb_div(j2, T, map.mul_t, map.shr_t);
          s2 += In4<IDT>::up(grad[jb * g_bstride4 + (long long)(j2 - jb * T) * D4 + c]);
          j2 = next[j2];
        }
        w[c] = w[c] + alpha * s2;
      }
      if (l == 0) head[r[u]] = -1;                  // restore the workspace invariant
    }
  }
}

// pass 3 (mid tables): a half-wavefront per row adds the partial sums of its EMB_MID_S lists in residue order and updates the row
__global__ __launch_bounds__(256) void emb_mid_fold(float* __restrict__ weight, const int* __restrict__ mhead,
                                                    const float* __restrict__ partial, const float* __restrict__ lr_dev,
                                                    float lr_host, const float* __restrict__ scale,
                                                    const float* __restrict__ skip_flag, MidFold mf, int D4) {
  if (skip_flag && *skip_flag != 0.0f) return;
  const int q = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 5), l = threadIdx.x & 31;
  if (q >= mf.rows_total) return;
  int j = 0;
  while (j + 1 < mf.n && q >= mf.off[j + 1]) ++j;
  const long long row = mf.base[j] + (q - mf.off[j]);
  const int lc = l < D4 ? l : D4 - 1;
  int f[EMB_MID_S];
  float4_t v[EMB_MID_S];
#pragma unroll
  for (int s2 = 0; s2 < EMB_MID_S; ++s2) {
    f[s2] = mhead[q * EMB_MID_S + s2];
    v[s2] = ((const float4_t*)(partial + ((long long)q * EMB_MID_S + s2) * (D4 * 4)))[lc];   // (unwritten scratch is never USED)
  }
  float4_t sum = {0.f, 0.f, 0.f, 0.f};
  bool any = false;
#pragma unroll
  for (int s2 = 0; s2 < EMB_MID_S; ++s2)
    if (f[s2] == -2) { sum += v[s2]; any = true; }
  if (!any || l >= D4) return;
  const float lr = lr_dev ? *lr_dev : lr_host;
  const float alpha = -lr * (scale ? *scale : 1.0f);
  float4_t* w = (float4_t*)(weight + row * (long long)(D4 * 4));
  w[l] = w[l] + alpha * sum;
}

extern "C" int dle_emb_sparse_sgd(float* weight, const int64_t* rows, const void* grad,
                                  const float* lr_dev, float lr_host, const float* scale_dev,
                                  const float* skip_flag_dev, int64_t n_rows, int tables, int dim,
                                  int64_t grad_batch_stride, int grad_dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dim > 0 && dim % 4 == 0, "emb_sparse_sgd: dim %d must be a multiple of 4", dim);
  if (n_rows == 0) return 0;
  DLE_CHECK_ARG(weight && rows && grad, "emb_sparse_sgd: null pointer");
  if (tables <= 0) tables = 1;
  if (grad_batch_stride == 0) grad_batch_stride = (int64_t)tables * dim;
  DLE_CHECK_ARG(grad_batch_stride % 4 == 0, "emb_sparse_sgd: grad batch stride must be a multiple of 4");
  const int D4 = dim / 4;
  const long long gs4 = grad_batch_stride / 4;
  const int grid = grid_for(n_rows, 4 * 2);
#define GO(IDT, VT) hipLaunchKernelGGL(emb_sparse_sgd<IDT>, dim3(grid), dim3(256), 0, stream, weight, (const long long*)rows, (const VT*)grad, lr_dev, lr_host, scale_dev, skip_flag_dev, (long long)n_rows, D4, tables, gs4)
  if (grad_dtype == DLE_F32) GO(DLE_F32, float4_t);
  else if (grad_dtype == DLE_F16) GO(DLE_F16, ushort4_t);
  else if (grad_dtype == DLE_BF16) GO(DLE_BF16, ushort4_t);
  else { dle_set_error("emb_sparse_sgd: bad dtype %d", grad_dtype); return -1; }
#undef GO
  DLE_LAUNCH_CHECK();
  return 0;
}

// rows[batch, tables] int64 joint-table row ids; table_offsets_host[tables+1] (HOST memory) = first row of
// each table; head[total_rows] int32 device workspace, all -1 on entry and on exit; next[batch*tables]
// int32 device scratch; is_small[tables] uint8 device array consistent with the rule
// rows_t * dim * 4 <= DLE_EMB_SMALL_LDS_BYTES (build it with dle_emb_small_table_mask).
#define DLE_EMB_SMALL_LDS_BYTES (64 * 1024)
extern "C" int dle_emb_small_table_mask(const int64_t* table_offsets_host, int tables, int dim,
                                        unsigned char* mask_host) {
  DLE_CHECK_ARG(table_offsets_host && mask_host && tables >= 0 && dim > 0, "emb_small_table_mask: bad args");
  int n_small = 0;
  for (int t = 0; t < tables; ++t) {
    const long long r = table_offsets_host[t + 1] - table_offsets_host[t];
    const bool sm = r * dim * 4 <= DLE_EMB_SMALL_LDS_BYTES && n_small < 64;
    mask_host[t] = sm ? 1 : 0;
    n_small += sm;
  }
  return 0;
}

extern "C" int dle_emb_onehot_try(float* weight, const int64_t* rows, const void* grad, const float* lr_dev, float lr_host,
                                  const float* scale_dev, const float* skip_flag_dev, const int* tab_t, const int64_t* tab_base,
                                  const int* tab_rows, int n_tab, int64_t batch, int tables, int dim, int64_t grad_batch_stride,
                                  int grad_dtype, void* ws, int64_t ws_bytes, hipStream_t stream);

extern "C" int64_t dle_emb_onehot_workspace_bytes(int n_tables, int64_t batch);

// Scratch layout of dle_emb_sgd_dedup_ws for 16-bit gradients: [one-hot partial blocks of the tiny tables | sub-list heads of the mid
// tables | their partial sums], each part 256-byte aligned.
struct EmbScratchPlan {
  int n_tiny, n_mid, mid_rows;
  long long onehot_bytes, mid_head_off, mid_partial_off, total;
};
static EmbScratchPlan emb_scratch_plan(const int64_t* table_offsets_host, int tables, int dim, int64_t batch) {
  EmbScratchPlan pl = {};
  static const int mid_mode = getenv("DLE_EMB_MID") ? atoi(getenv("DLE_EMB_MID")) : 1;
  int n_small = 0;
  for (int t = 0; t < tables; ++t) {
    const long long r = table_offsets_host[t + 1] - table_offsets_host[t];
    if (r * dim * 4 <= DLE_EMB_SMALL_LDS_BYTES && n_small < 64) {
      ++n_small;
      if (r <= 128 && dim <= 128 && (dim & 1) == 0) ++pl.n_tiny;
    } else if (mid_mode && r <= EMB_MID_ROWS && dim <= 128 && tables <= 128 && pl.n_mid < EMB_MID_TABLES) {
      ++pl.n_mid;
      pl.mid_rows += (int)r;
    }
  }
  pl.onehot_bytes = dim == 128 ? dle_emb_onehot_workspace_bytes(pl.n_tiny, batch) : 0;
  pl.mid_head_off = (pl.onehot_bytes + 255) / 256 * 256;
  pl.mid_partial_off = (pl.mid_head_off + (long long)pl.mid_rows * EMB_MID_S * 4 + 255) / 256 * 256;
  pl.total = pl.mid_partial_off + (long long)pl.mid_rows * EMB_MID_S * dim * 4;
  return pl;
}
extern "C" int64_t dle_emb_sgd_workspace_bytes(const int64_t* table_offsets_host, int tables, int dim, int64_t batch) {
  if (!table_offsets_host || tables <= 0 || dim <= 0 || batch <= 0) return 0;
  return emb_scratch_plan(table_offsets_host, tables, dim, batch).total;
}

// ws (optional, dle_emb_sgd_workspace_bytes() bytes, 16-bit gradients): the tiny tables run as the one-hot MFMA segment sum of
// emb_onehot.hip instead of the register form below, the mid tables (<= 4096 rows) thread 8 lists per row (MidMap above)
extern "C" int dle_emb_sgd_dedup_ws(float* weight, const int64_t* rows, const void* grad, int32_t* head,
                                    int32_t* next, const unsigned char* is_small_dev,
                                    const int64_t* table_offsets_host, const float* lr_dev, float lr_host,
                                    const float* scale_dev, const float* skip_flag_dev, int64_t batch, int tables,
                                    int dim, int64_t grad_batch_stride, int grad_dtype, void* ws, int64_t ws_bytes,
                                    hipStream_t stream) {
  DLE_CHECK_ARG(dim > 0 && dim % 4 == 0 && tables > 0, "emb_sgd_dedup: bad shape");
  if (batch == 0) return 0;
  DLE_CHECK_ARG(weight && rows && grad && head && next && is_small_dev && table_offsets_host, "emb_sgd_dedup: null pointer");
  DLE_CHECK_ARG(batch * tables < 2147483647LL, "emb_sgd_dedup: more than 2^31 lookups per call");
  DLE_CHECK_ARG(grad_dtype == DLE_F32 || grad_dtype == DLE_F16 || grad_dtype == DLE_BF16, "emb_sgd_dedup: bad dtype %d", grad_dtype);
  if (grad_batch_stride == 0) grad_batch_stride = (int64_t)tables * dim;
  DLE_CHECK_ARG(grad_batch_stride % 4 == 0, "emb_sgd_dedup: grad batch stride must be a multiple of 4");
  const int D4 = dim / 4;
  const long long gs4 = grad_batch_stride / 4;
  const long long n = (long long)batch * tables;
  SmallTables st;
  st.n = 0;
  int n_large = 0;
  for (int t = 0; t < tables; ++t) {
    const long long r = table_offsets_host[t + 1] - table_offsets_host[t];
    if (r * dim * 4 <= DLE_EMB_SMALL_LDS_BYTES && st.n < 64) {
      st.t[st.n] = t; st.base[st.n] = table_offsets_host[t]; st.rows[st.n] = (int)r;
      ++st.n;
    } else {
      ++n_large;
    }
  }
  // tiny tables (<= 128 rows, dim <= 128, 16-bit gradients, even element-pair strides): register form; the other "small" ones: LDS form
  SmallTables tiny, lds_t;
  tiny.n = 0; lds_t.n = 0;
  int max_rows = 0, max_rows_tiny = 0;
  static const int tiny_mode = getenv("DLE_EMB_TINY") ? atoi(getenv("DLE_EMB_TINY")) : 1;
  for (int i = 0; i < st.n; ++i) {
    const bool is_tiny = tiny_mode && grad_dtype != DLE_F32 && st.rows[i] <= 128 && dim <= 128 && (dim & 1) == 0 &&
                         (grad_batch_stride & 1) == 0;        // (fp32 gradients: the LDS form -- not the train step's path)
    SmallTables& dst = is_tiny ? tiny : lds_t;
    dst.t[dst.n] = st.t[i]; dst.base[dst.n] = st.base[i]; dst.rows[dst.n] = st.rows[i];
    ++dst.n;
    if (is_tiny) { if (st.rows[i] > max_rows_tiny) max_rows_tiny = st.rows[i]; }
    else if (st.rows[i] > max_rows) max_rows = st.rows[i];
  }
  const bool plan_ok = ws && grad_dtype != DLE_F32 && (((uintptr_t)ws) & 255) == 0;
  const EmbScratchPlan pl = plan_ok ? emb_scratch_plan(table_offsets_host, tables, dim, batch) : EmbScratchPlan{};
  const bool ws_ok = plan_ok && ws_bytes >= pl.total;
  if (tiny.n > 0 && ws_ok && pl.onehot_bytes > 0 && tiny.n == pl.n_tiny) {
    static_assert(sizeof(long long) == sizeof(int64_t), "table bases are passed as int64");
    const int rc = dle_emb_onehot_try(weight, rows, grad, lr_dev, lr_host, scale_dev, skip_flag_dev, tiny.t,
                                      (const int64_t*)tiny.base, tiny.rows, tiny.n, batch, tables, dim, grad_batch_stride,
                                      grad_dtype, ws, pl.onehot_bytes, stream);
    if (rc > 1) return rc;
    if (rc == 1) tiny.n = 0;
  }
  if (tiny.n > 0) {
    // one 16-wavefront workgroup per CU and table slice; >= 2048 samples per workgroup
    int slices = (int)((256 + tiny.n - 1) / tiny.n);
    const long long max_slices = (batch + 2047) / 2048;
    if (slices > max_slices) slices = (int)max_slices;
    if (slices < 1) slices = 1;
    const size_t lds = (size_t)max_rows_tiny * dim * 4;
#define GO(IDT) hipLaunchKernelGGL(emb_sgd_tiny<IDT>, dim3(tiny.n * slices), dim3(1024), lds, stream, weight, (const long long*)rows, grad, lr_dev, lr_host, scale_dev, skip_flag_dev, tiny, (long long)batch, tables, dim, (long long)grad_batch_stride, slices)
    if (grad_dtype == DLE_F16) GO(DLE_F16);
    else GO(DLE_BF16);
#undef GO
    DLE_LAUNCH_CHECK();
  }
  if (lds_t.n > 0) {
    int slices = (int)((batch + 511) / 512);
    if (slices > 128) slices = 128;
    if (slices < 1) slices = 1;
    const size_t lds = (size_t)max_rows * dim * 4;
#define GO(IDT, VT) hipLaunchKernelGGL(emb_sgd_small<IDT>, dim3(lds_t.n * slices), dim3(256), lds, stream, weight, (const long long*)rows, (const VT*)grad, lr_dev, lr_host, scale_dev, skip_flag_dev, lds_t, (long long)batch, tables, D4, gs4, slices)
    if (grad_dtype == DLE_F32) GO(DLE_F32, float4_t);
    else if (grad_dtype == DLE_F16) GO(DLE_F16, ushort4_t);
    else GO(DLE_BF16, ushort4_t);
#undef GO
    DLE_LAUNCH_CHECK();
  }
  if (n_large > 0) {
    LookupMap map;
    map.T = tables; map.nl = 0;
    emb_make_div(tables, map.mul_t, map.shr_t);
    map.mul_nl = map.mul_t; map.shr_nl = map.shr_t;
    for (int i = 0; i < 128; ++i) map.t[i] = 0;
    if (tables <= 128) {
      int k = 0, q = 0;
      for (int t = 0; t < tables; ++t) {
        if (q < st.n && st.t[q] == t) { ++q; continue; }                 // (st lists the small tables in ascending order)
        map.t[k++] = (unsigned char)t;
      }
      map.nl = k;
      emb_make_div(k, map.mul_nl, map.shr_nl);
    }
    const long long n_lk = (long long)batch * (map.nl ? map.nl : tables);
    // mid tables: 8 lists per row in scratch (emptied here: nothing persists between calls)
    MidMap mid;
    MidFold mf;
    mid.n = 0; mid.rows_total = 0; mid.head = nullptr; mid.partial = nullptr;
    mf.n = 0; mf.rows_total = 0;
    for (int i = 0; i < 128; ++i) { mid.off[i] = -1; mid.base[i] = 0; }
    if (ws_ok && pl.n_mid > 0 && map.nl > 0) {
      int n_small = 0;
      for (int k = 0; k < map.nl; ++k) {
        const int t = map.t[k];
        const long long r = table_offsets_host[t + 1] - table_offsets_host[t];
        if (r <= EMB_MID_ROWS && mf.n < EMB_MID_TABLES) {
          mid.off[k] = mf.rows_total; mid.base[k] = table_offsets_host[t];
          mf.off[mf.n] = mf.rows_total; mf.rows[mf.n] = (int)r; mf.base[mf.n] = table_offsets_host[t];
          ++mf.n;
          mf.rows_total += (int)r;
        }
      }
      (void)n_small;
      if (mf.n == pl.n_mid && mf.rows_total == pl.mid_rows) {
        mid.n = mf.n; mid.rows_total = mf.rows_total;
        mid.head = (int*)((char*)ws + pl.mid_head_off);
        mid.partial = (float*)((char*)ws + pl.mid_partial_off);
        hipError_t e = hipMemsetAsync(mid.head, 0xFF, (size_t)mid.rows_total * EMB_MID_S * 4, stream);
        if (e != hipSuccess) { dle_set_error("emb_sgd_dedup memset: %s", hipGetErrorString(e)); return (int)e; }
      } else {
        for (int i = 0; i < 128; ++i) mid.off[i] = -1;
        mf.n = 0;
      }
    }
    hipLaunchKernelGGL(emb_link, dim3(grid_for(n_lk, 256)), dim3(256), 0, stream, (const long long*)rows, head, next,
                       is_small_dev, skip_flag_dev, (int)n_lk, map, mid);
    DLE_LAUNCH_CHECK();
    const int grid = grid_for(n_lk, 4 * 4);
    static const int spec = getenv("DLE_EMB_SPEC") ? atoi(getenv("DLE_EMB_SPEC")) : 1;
#define GO(IDT, VT) do { if (spec) hipLaunchKernelGGL((emb_sgd_lists<IDT, true>), dim3(grid), dim3(256), 0, stream, weight, (const long long*)rows, (const VT*)grad, head, (const int*)next, is_small_dev, lr_dev, lr_host, scale_dev, skip_flag_dev, (int)n_lk, map, D4, gs4, mid); \
    else hipLaunchKernelGGL((emb_sgd_lists<IDT, false>), dim3(grid), dim3(256), 0, stream, weight, (const long long*)rows, (const VT*)grad, head, (const int*)next, is_small_dev, lr_dev, lr_host, scale_dev, skip_flag_dev, (int)n_lk, map, D4, gs4, mid); } while (0)
    if (grad_dtype == DLE_F32) GO(DLE_F32, float4_t);
    else if (grad_dtype == DLE_F16) GO(DLE_F16, ushort4_t);
    else GO(DLE_BF16, ushort4_t);
#undef GO
    DLE_LAUNCH_CHECK();
    if (mid.n > 0) {
      hipLaunchKernelGGL(emb_mid_fold, dim3((mid.rows_total * 32 + 255) / 256), dim3(256), 0, stream, weight, (const int*)mid.head,
                         (const float*)mid.partial, lr_dev, lr_host, scale_dev, skip_flag_dev, mf, D4);
      DLE_LAUNCH_CHECK();
    }
  }
  return 0;
}

extern "C" int dle_emb_sgd_dedup(float* weight, const int64_t* rows, const void* grad, int32_t* head,
                                 int32_t* next, const unsigned char* is_small_dev,
                                 const int64_t* table_offsets_host, const float* lr_dev, float lr_host,
                                 const float* scale_dev, const float* skip_flag_dev, int64_t batch, int tables,
                                 int dim, int64_t grad_batch_stride, int grad_dtype, hipStream_t stream) {
  return dle_emb_sgd_dedup_ws(weight, rows, grad, head, next, is_small_dev, table_offsets_host, lr_dev, lr_host, scale_dev,
                              skip_flag_dev, batch, tables, dim, grad_batch_stride, grad_dtype, nullptr, 0, stream);
}
