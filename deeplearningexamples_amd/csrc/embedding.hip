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
                                                      long long rows_total, int T, int D4) {
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
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      row[u] = base + (long long)u * rows_per_wave + sub;
      src[u] = -1;
      if (row[u] < rows_total) {
        long long ix = indices[row[u]];
        const int t = (int)(row[u] % T);
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
        if (src[u] >= 0) out[row[u] * D4 + c] = Out<ODT>::pack(v[u]);
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
                                                      long long n_rows, int D4) {
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
    for (int c = l; c < D4; c += G) {
      const float4_t g = In4<IDT>::up(grad[r * D4 + c]);
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
                                  int dim, int out_dtype, hipStream_t stream) {
  DLE_CHECK_ARG(batch >= 0 && tables > 0, "emb_gather_fwd: bad shape");
  if (batch == 0) return 0;
  DLE_CHECK_ARG(weight && indices && out, "emb_gather_fwd: null pointer");
  DLE_CHECK_ARG(dim > 0 && dim % 4 == 0, "emb_gather_fwd: dim %d must be a multiple of 4", dim);
  DLE_CHECK_ARG(tables > 0 && batch >= 0, "emb_gather_fwd: bad shape");
  DLE_CHECK_ARG((((uintptr_t)weight) & 15) == 0 && (((uintptr_t)out) & 7) == 0, "emb_gather_fwd: misaligned");
  const long long rows_total = (long long)batch * tables;
  if (rows_total == 0) return 0;
  const int D4 = dim / 4;
  const int grid = grid_for(rows_total, 4 * 2 * 4);
  dim3 block(256);
#define GO(ODT)                                                                                    \
  hipLaunchKernelGGL((emb_gather_fwd<ODT, 4>), dim3(grid), block, 0, stream, weight,                \
                     (const long long*)indices, (const long long*)offsets, (const long long*)hash_sizes, \
                     (typename Out<ODT>::V*)out, rows_total, tables, D4)
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

extern "C" int dle_emb_sparse_sgd(float* weight, const int64_t* rows, const void* grad,
                                  const float* lr_dev, float lr_host, const float* scale_dev,
                                  const float* skip_flag_dev, int64_t n_rows, int dim, int grad_dtype,
                                  hipStream_t stream) {
  DLE_CHECK_ARG(dim > 0 && dim % 4 == 0, "emb_sparse_sgd: dim %d must be a multiple of 4", dim);
  if (n_rows == 0) return 0;
  DLE_CHECK_ARG(weight && rows && grad, "emb_sparse_sgd: null pointer");
  const int D4 = dim / 4;
  const int grid = grid_for(n_rows, 4 * 2);
  if (grad_dtype == DLE_F32)
    hipLaunchKernelGGL(emb_sparse_sgd<DLE_F32>, dim3(grid), dim3(256), 0, stream, weight, (const long long*)rows, (const float4_t*)grad, lr_dev, lr_host, scale_dev, skip_flag_dev, (long long)n_rows, D4);
  else if (grad_dtype == DLE_F16)
    hipLaunchKernelGGL(emb_sparse_sgd<DLE_F16>, dim3(grid), dim3(256), 0, stream, weight, (const long long*)rows, (const ushort4_t*)grad, lr_dev, lr_host, scale_dev, skip_flag_dev, (long long)n_rows, D4);
  else if (grad_dtype == DLE_BF16)
    hipLaunchKernelGGL(emb_sparse_sgd<DLE_BF16>, dim3(grid), dim3(256), 0, stream, weight, (const long long*)rows, (const ushort4_t*)grad, lr_dev, lr_host, scale_dev, skip_flag_dev, (long long)n_rows, D4);
  else { dle_set_error("emb_sparse_sgd: bad dtype %d", grad_dtype); return -1; }
  DLE_LAUNCH_CHECK();
  return 0;
}
