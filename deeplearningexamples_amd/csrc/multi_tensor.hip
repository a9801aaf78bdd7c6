// Multi-tensor optimizer kernels for gfx950: L2 norm, LAMB stage 1/2, SGD(+momentum).
//
// Replaces BERT/lamb_amp_opt/csrc/{multi_tensor_apply.cuh:16-133, multi_tensor_l2norm_kernel.cu:28-216,
// multi_tensor_lamb.cu:43-500} behind fused_lamb_CUDA.{multi_tensor_l2norm, multi_tensor_lamb}
// (lamb_amp_opt/csrc/frontend.cpp:3-32), plus the apex FusedSGD / torch.optim.SGD dense steps of
// RN50 (CN/image_classification/optimizers.py:34-56) and DLRM (dlrm/scripts/main.py:468-471).
//
// Design: the reference packs <=110 tensor pointers into a ~3.7 KB by-value kernel argument and
// relaunches until the list is exhausted.  Here the tensor list lives in a DEVICE-SIDE descriptor
// table that the host builds once and reuses while the addresses are unchanged:
//
//   int64 table[] = { size[0..n) | chunk_start[0..n] | ptr_list0[0..n) | ptr_list1[0..n) | ... }
//
// One launch covers every chunk of every tensor (grid = chunk_start[n]); a workgroup finds its tensor
// with a binary search over chunk_start (scalar loads).  All per-step scalars (lr, step, norms,
// found_inf, inv_scale) are device pointers, so the optimizer step never synchronises the host.
// These are pure HBM-streaming kernels: 16 B per lane accesses, wave64 shuffle reductions.
// Algorithmic bytes / parameter: l2norm e_g; stage1 reads g+p+m+v, writes g+m+v; stage2 reads g+p, writes p(+p16).
#include "common.h"

#define MT_BLOCK 512

struct MtTable {
  const long long* size;
  const long long* chunk_start;
  const long long* ptr;   // ptr[list * n + t]
  int n;
};

__device__ __forceinline__ MtTable mt_view(const long long* table, int n) {
  MtTable t;
  t.size = table;
  t.chunk_start = table + n;
  t.ptr = table + n + (n + 1);
  t.n = n;
  return t;
}

// tensor owning global chunk `c`:  chunk_start[t] <= c < chunk_start[t+1]
__device__ __forceinline__ int mt_find(const MtTable& t, long long c) {
  int lo = 0, hi = t.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (t.chunk_start[mid] <= c) lo = mid; else hi = mid - 1;
  }
  return lo;
}

template <int DT> struct MtIO;
template <> struct MtIO<DLE_F32> {
  typedef float T;
  static __device__ __forceinline__ float ld(const T* p) { return *p; }
  static __device__ __forceinline__ void st(T* p, float v) { *p = v; }
};
template <> struct MtIO<DLE_F16> {
  typedef unsigned short T;
  static __device__ __forceinline__ float ld(const T* p) { return Elem<DLE_F16>::to_f32(*p); }
  static __device__ __forceinline__ void st(T* p, float v) { *p = Elem<DLE_F16>::from_f32(v); }
};
template <> struct MtIO<DLE_BF16> {
  typedef unsigned short T;
  static __device__ __forceinline__ float ld(const T* p) { return Elem<DLE_BF16>::to_f32(*p); }
  static __device__ __forceinline__ void st(T* p, float v) { *p = Elem<DLE_BF16>::from_f32(v); }
};

// load/store 4 consecutive elements (vector when aligned)
template <int DT> struct Cvt {
  static __device__ __forceinline__ float up(unsigned short u) { return Elem<DT>::to_f32(u); }
  static __device__ __forceinline__ unsigned short down(float f) { return Elem<DT>::from_f32(f); }
};
template <> struct Cvt<DLE_F32> {
  static __device__ __forceinline__ float up(unsigned short) { return 0.f; }
  static __device__ __forceinline__ unsigned short down(float) { return 0; }
};
template <int DT>
__device__ __forceinline__ float4_t ld4(const typename MtIO<DT>::T* p, bool vec) {
  float4_t o;
  if (vec) {
    if (DT == DLE_F32) {
      o = *(const float4_t*)p;
    } else {
      const ushort4_t u = *(const ushort4_t*)p;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = Cvt<DT>::up(u[i]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = MtIO<DT>::ld(p + i);
  }
  return o;
}
template <int DT>
__device__ __forceinline__ void st4(typename MtIO<DT>::T* p, float4_t v, bool vec) {
  if (vec) {
    if (DT == DLE_F32) {
      *(float4_t*)p = v;
    } else {
      ushort4_t u;
#pragma unroll
      for (int i = 0; i < 4; ++i) u[i] = Cvt<DT>::down(v[i]);
      *(ushort4_t*)p = u;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) MtIO<DT>::st(p + i, v[i]);
  }
}

// s + (v0^2 + v1^2 + v2^2 + v3^2) with the roundings pinned (explicit fused multiply-adds): the l2norm sweep and the LAMB stage that
// takes the same sums on the fly must agree to the bit, whatever hipcc's contraction makes of the code around them
__device__ __forceinline__ float mt_sumsq4(float s, float4_t v) {
  float t = v[0] * v[0];
  t = __builtin_fmaf(v[1], v[1], t);
  t = __builtin_fmaf(v[2], v[2], t);
  t = __builtin_fmaf(v[3], v[3], t);
  return s + t;
}

// -------------------------------------------------------------------- L2 norm
// partial[c] = sum over chunk c of x^2  (multi_tensor_l2norm_kernel.cu:28-110)
template <int DT>
__global__ __launch_bounds__(MT_BLOCK) void mt_l2norm_partial(const long long* __restrict__ table, int n,
                                                              int chunk, float* __restrict__ partial,
                                                              int* __restrict__ noop) {
  if (noop && *noop) return;
  __shared__ float red[16];
  const MtTable t = mt_view(table, n);
  const long long c = blockIdx.x;
  const int ti = mt_find(t, c);
  const long long ci = c - t.chunk_start[ti];
  const typename MtIO<DT>::T* x = (const typename MtIO<DT>::T*)t.ptr[ti] + ci * chunk;
  long long len = t.size[ti] - ci * chunk;
  if (len > chunk) len = chunk;
  const bool vec = (((uintptr_t)x) & 15) == 0;
  float s = 0.f;
  const long long len4 = len & ~3LL;
  for (long long i = (long long)threadIdx.x * 4; i < len4; i += MT_BLOCK * 4) {
    s = mt_sumsq4(s, ld4<DT>(x + i, vec));
  }
  for (long long i = len4 + threadIdx.x; i < len; i += MT_BLOCK) {
    const float v = MtIO<DT>::ld(x + i);
    s += v * v;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    if (!isfinite(s) && noop) *noop = 1;   // l2norm_kernel.cu:103-104
    partial[c] = s;
  }
}

// ret[0] = sqrt(sum partial); ret_per_tensor[t] = sqrt(sum of tensor t's chunks)  (cleanup, :112-151)
__global__ __launch_bounds__(MT_BLOCK) void mt_l2norm_finish(const float* __restrict__ partial,
                                                             const long long* __restrict__ table, int n,
                                                             float* __restrict__ ret,
                                                             float* __restrict__ ret_per_tensor,
                                                             const int* __restrict__ noop) {
  if (noop && *noop) {
    // the reference returns its ZERO-INITIALISED outputs untouched when the flag is set -- before the launch (an overflow step:
    // fused_lamb.py copies found_inf into the flag) or by the partial pass that met a non-finite value (multi_tensor_l2norm_kernel.cu:
    // 40-42, 103-104, 121-123; the outputs are at::zeros).  The callers here hand over uninitialised buffers: write the zeros.
    if (threadIdx.x == 0) {
      if (blockIdx.x == 0) ret[0] = 0.f;
      else ret_per_tensor[blockIdx.x - 1] = 0.f;
    }
    return;
  }
  __shared__ float red[16];
  const MtTable t = mt_view(table, n);
  if (blockIdx.x == 0) {
    const long long total = t.chunk_start[n];
    float s = 0.f;
    for (long long i = threadIdx.x; i < total; i += MT_BLOCK) s += partial[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) ret[0] = sqrtf(s);
  } else {
    const int ti = blockIdx.x - 1;
    float s = 0.f;
    for (long long i = t.chunk_start[ti] + threadIdx.x; i < t.chunk_start[ti + 1]; i += MT_BLOCK) s += partial[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) ret_per_tensor[ti] = sqrtf(s);
  }
}

// -------------------------------------------------------------------- LAMB stage 1
// lists: 0 = g (GT, overwritten with the update), 1 = p, 2 = m, 3 = v (fp32)
// multi_tensor_lamb.cu:43-245
// NORMS: the per-chunk sums of squares of p (as read) and of the update (as stored) leave with the pass -- the two per-tensor
// l2norm sweeps multi_tensor_lamb_cuda runs around this stage (multi_tensor_lamb.cu:380-420: param norms before, update norms
// after; 2 x 1.34 GB re-read for BERT-Large) become a fold over `partial_p` / `partial_u` (mt_lamb_norms_finish, which also raises
// the noop flag on a non-finite sum, as the l2norm pass does).
template <int GT, bool NORMS>
__global__ __launch_bounds__(MT_BLOCK) void mt_lamb_stage1(const long long* __restrict__ table, int n, int chunk,
                                                           int* noop, float beta1, float beta2,
                                                           float beta3, const int* __restrict__ step_ptr,
                                                           int bias_correction, float eps, int mode, float decay,
                                                           const float* __restrict__ global_grad_norm,
                                                           const float* __restrict__ max_global_grad_norm,
                                                           const float* __restrict__ inv_scale,
                                                           float* __restrict__ partial_p, float* __restrict__ partial_u) {
  if (noop && *noop) return;   // :63-65
  __shared__ float red[16];
  float sp = 0.f, su = 0.f;
  const MtTable t = mt_view(table, n);
  const long long c = blockIdx.x;
  const int ti = mt_find(t, c);
  const long long off = (c - t.chunk_start[ti]) * chunk;
  long long len = t.size[ti] - off;
  if (len > chunk) len = chunk;
  typename MtIO<GT>::T* g = (typename MtIO<GT>::T*)t.ptr[0 * n + ti] + off;
  float* p = (float*)t.ptr[1 * n + ti] + off;
  float* m = (float*)t.ptr[2 * n + ti] + off;
  float* v = (float*)t.ptr[3 * n + ti] + off;

  float b1c = 1.0f, b2c = 1.0f;
  if (bias_correction == 1) {   // :67-73 -- std::pow(float,int) evaluates in double
    const int step = *step_ptr;
    b1c = (float)(1.0 - pow((double)beta1, (double)step));
    b2c = (float)(1.0 - pow((double)beta2, (double)step));
  }
  const float ggn = *global_grad_norm, mgn = *max_global_grad_norm;
  const float clip = ggn > mgn ? ggn / mgn : 1.0f;   // :79
  const float is = *inv_scale;
  const bool vec = ((((uintptr_t)g) | ((uintptr_t)p) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
  const long long len4 = len & ~3LL;
  for (long long i = (long long)threadIdx.x * 4; i < len; i += MT_BLOCK * 4) {
    const bool full = i < len4;
    const int cnt = full ? 4 : (int)(len - i);
    float4_t rg, rp, rm, rv;
    if (full) {
      rg = ld4<GT>(g + i, vec);
      rp = decay != 0.f ? ld4<DLE_F32>(p + i, vec) : (float4_t){0.f, 0.f, 0.f, 0.f};
      rm = ld4<DLE_F32>(m + i, vec);
      rv = ld4<DLE_F32>(v + i, vec);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool ok = k < cnt;
        rg[k] = ok ? MtIO<GT>::ld(g + i + k) : 0.f;
        rp[k] = (ok && decay != 0.f) ? p[i + k] : 0.f;
        rm[k] = ok ? m[i + k] : 0.f;
        rv[k] = ok ? v[i + k] : 0.f;
      }
    }
    float4_t upd;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float sg = (rg[k] * is) / clip;
      if (mode == 0) {   // L2 regularisation mode (:130-139)
        sg = sg + decay * rp[k];
        rm[k] = rm[k] * beta1 + beta3 * sg;
        rv[k] = rv[k] * beta2 + (1.f - beta2) * sg * sg;
        const float mh = rm[k] / b1c, vh = rv[k] / b2c;
        upd[k] = mh / (sqrtf(vh) + eps);
      } else {           // AdamW mode (:141-149)
        rm[k] = rm[k] * beta1 + beta3 * sg;
        rv[k] = rv[k] * beta2 + (1.f - beta2) * sg * sg;
        const float mh = rm[k] / b1c, vh = rv[k] / b2c;
        upd[k] = mh / (sqrtf(vh) + eps) + decay * rp[k];
      }
    }
    if (NORMS) {
      float4_t us = upd;                       // the value the update-norm sweep would read back
      if (GT != DLE_F32) {
#pragma unroll
        for (int k = 0; k < 4; ++k) us[k] = Cvt<GT>::up(Cvt<GT>::down(upd[k]));
      }
      if (full) {
        sp = mt_sumsq4(sp, rp);
        su = mt_sumsq4(su, us);
      } else {
        for (int k = 0; k < cnt; ++k) { sp += rp[k] * rp[k]; su += us[k] * us[k]; }
      }
    }
    if (full) {
      st4<GT>(g + i, upd, vec);   // the update goes back into the grad buffer (:163,168)
      st4<DLE_F32>(m + i, rm, vec);
      st4<DLE_F32>(v + i, rv, vec);
    } else {
      for (int k = 0; k < cnt; ++k) {
        MtIO<GT>::st(g + i + k, upd[k]);
        m[i + k] = rm[k];
        v[i + k] = rv[k];
      }
    }
  }
  if (NORMS) {
    sp = block_sum(sp, red);
    su = block_sum(su, red);
    if (threadIdx.x == 0) {            // (a non-finite partial raises the flag in the FOLD: raised here, workgroups that start
      partial_p[c] = sp;               //  later would leave at the top and their m / v stay behind)
      partial_u[c] = su;
    }
  }
}

// -------------------------------------------------------------------- LAMB stage 2
// lists: 0 = update (GT), 1 = p (fp32), [2 = low-precision model copy (CT)]   multi_tensor_lamb.cu:251-368
// CT = -1: no copy.  (The reference's copy always has the gradient's dtype -- fp16 params with fp16 grads,
// multi_tensor_lamb.cu:320-327; here fp32 gradients may feed a 16-bit working copy, so CT is independent.)
template <int GT, int CT>
__global__ __launch_bounds__(MT_BLOCK) void mt_lamb_stage2(const long long* __restrict__ table, int n, int chunk,
                                                           const int* __restrict__ noop,
                                                           const float* __restrict__ param_norm,
                                                           const float* __restrict__ update_norm,
                                                           const float* __restrict__ lr_ptr, float decay,
                                                           int use_nvlamb) {
  if (noop && *noop) return;
  const MtTable t = mt_view(table, n);
  const long long c = blockIdx.x;
  const int ti = mt_find(t, c);
  const long long off = (c - t.chunk_start[ti]) * chunk;
  long long len = t.size[ti] - off;
  if (len > chunk) len = chunk;
  const float lr = *lr_ptr;
  float ratio = lr;
  if (use_nvlamb || decay != 0.0f) {   // :277-282
    const float pn = param_norm[ti], un = update_norm[ti];
    ratio = (un != 0.0f && pn != 0.0f) ? lr * (pn / un) : lr;
  }
  const typename MtIO<GT>::T* u = (const typename MtIO<GT>::T*)t.ptr[0 * n + ti] + off;
  float* p = (float*)t.ptr[1 * n + ti] + off;
  constexpr bool HAS_OUT = CT >= 0;
  constexpr int OT = CT >= 0 ? CT : DLE_F32;
  typename MtIO<OT>::T* o = HAS_OUT ? (typename MtIO<OT>::T*)t.ptr[2 * n + ti] + off : nullptr;
  const bool vec = ((((uintptr_t)u) | ((uintptr_t)p)) & 15) == 0 && (((uintptr_t)o) & (OT == DLE_F32 ? 15 : 7)) == 0;
  const long long len4 = len & ~3LL;
  for (long long i = (long long)threadIdx.x * 4; i < len; i += MT_BLOCK * 4) {
    if (i < len4) {
      const float4_t ru = ld4<GT>(u + i, vec);
      float4_t rp = ld4<DLE_F32>(p + i, vec);
#pragma unroll
      for (int k = 0; k < 4; ++k) rp[k] = rp[k] - ratio * ru[k];
      st4<DLE_F32>(p + i, rp, vec);
      if (HAS_OUT) st4<OT>(o + i, rp, vec);
    } else {
      for (long long k = i; k < len; ++k) {
        const float np = p[k] - ratio * MtIO<GT>::ld(u + k);
        p[k] = np;
        if (HAS_OUT) MtIO<OT>::st(o + k, np);
      }
    }
  }
}

// -------------------------------------------------------------------- SGD (+momentum, nesterov, wd)
// lists: 0 = g (GT), 1 = p (fp32), [2 = momentum buffer (fp32)], [last = low-precision model copy (CT)]
// torch.optim.SGD semantics (CN/image_classification/optimizers.py:34-56): d = g*inv_scale + wd*p;
// buf = first ? d : mom*buf + (1-damp)*d; d = nesterov ? d + mom*buf : buf; p -= lr*d.
template <int GT, bool HAS_MOM, int CT>
__global__ __launch_bounds__(MT_BLOCK) void mt_sgd(const long long* __restrict__ table, int n, int chunk,
                                                   const float* __restrict__ skip_flag,
                                                   const float* __restrict__ lr_ptr, float lr_host, float momentum,
                                                   float dampening, float wd, int nesterov, int first_step,
                                                   const float* __restrict__ inv_scale) {
  if (skip_flag && *skip_flag != 0.f) return;   // GradScaler found_inf -> skip
  const MtTable t = mt_view(table, n);
  const long long c = blockIdx.x;
  const int ti = mt_find(t, c);
  const long long off = (c - t.chunk_start[ti]) * chunk;
  long long len = t.size[ti] - off;
  if (len > chunk) len = chunk;
  const float lr = lr_ptr ? *lr_ptr : lr_host;
  const float is = inv_scale ? *inv_scale : 1.0f;
  const typename MtIO<GT>::T* g = (const typename MtIO<GT>::T*)t.ptr[0 * n + ti] + off;
  float* p = (float*)t.ptr[1 * n + ti] + off;
  float* mb = HAS_MOM ? (float*)t.ptr[2 * n + ti] + off : nullptr;
  // optional low-precision working copy of the parameter (the AMP "model weights"), last list
  // (a tensor without one -- a bias next to weights in one table -- carries pointer 0)
  unsigned short* pc0 = CT >= 0 ? (unsigned short*)t.ptr[(HAS_MOM ? 3 : 2) * n + ti] : nullptr;
  const bool has_pc = CT >= 0 && pc0 != nullptr;
  unsigned short* pc = has_pc ? pc0 + off : nullptr;
  const bool vec = ((((uintptr_t)g) | ((uintptr_t)p) | ((uintptr_t)mb)) & 15) == 0 && ((((uintptr_t)pc) & 7) == 0);
  const long long len4 = len & ~3LL;
  for (long long i = (long long)threadIdx.x * 4; i < len; i += MT_BLOCK * 4) {
    const bool full = i < len4;
    const int cnt = full ? 4 : (int)(len - i);
    float4_t rg, rp, rb = {0.f, 0.f, 0.f, 0.f};
    if (full) {
      rg = ld4<GT>(g + i, vec);
      rp = ld4<DLE_F32>(p + i, vec);
      if (HAS_MOM && !first_step) rb = ld4<DLE_F32>(mb + i, vec);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool ok = k < cnt;
        rg[k] = ok ? MtIO<GT>::ld(g + i + k) : 0.f;
        rp[k] = ok ? p[i + k] : 0.f;
        rb[k] = (ok && HAS_MOM && !first_step) ? mb[i + k] : 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float d = rg[k] * is + wd * rp[k];
      if (HAS_MOM) {
        rb[k] = first_step ? d : momentum * rb[k] + (1.f - dampening) * d;
        d = nesterov ? d + momentum * rb[k] : rb[k];
      }
      rp[k] = rp[k] - lr * d;
    }
    if (full) {
      st4<DLE_F32>(p + i, rp, vec);
      if (HAS_MOM) st4<DLE_F32>(mb + i, rb, vec);
      if (has_pc) st4<(CT >= 0 ? CT : DLE_F16)>(pc + i, rp, vec);
    } else {
      for (int k = 0; k < cnt; ++k) {
        p[i + k] = rp[k];
        if (HAS_MOM) mb[i + k] = rb[k];
        if (has_pc) MtIO<(CT >= 0 ? CT : DLE_F16)>::st(pc + i + k, rp[k]);
      }
    }
  }
}

// -------------------------------------------------------------------- Adam (torch.optim.Adam, fp32 state)
// SpeechSynthesis/Tacotron2/train.py:400-401,487-497: GradScaler.unscale_ + clip_grad_norm_ + Adam.step in one pass.
// lists: g (fp32, scaled by the loss scale), p, exp_avg, exp_avg_sq.  grad = g * inv_scale * clip,
// clip = min(1, max_norm / (||g|| * inv_scale + 1e-6)) (torch.nn.utils.clip_grad_norm_), then torch's Adam:
//   grad += wd * p;  m = b1 m + (1 - b1) grad;  v = b2 v + (1 - b2) grad^2;
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps),  t = *step (already advanced by the caller).
__global__ __launch_bounds__(MT_BLOCK) void mt_adam(const long long* __restrict__ table, int n, int chunk,
                                                    const float* __restrict__ skip_flag, const float* __restrict__ lr_ptr,
                                                    float lr_host, float beta1, float beta2, float eps, float wd,
                                                    const int* __restrict__ step_ptr, const float* __restrict__ inv_scale,
                                                    const float* __restrict__ gnorm, float max_norm) {
  if (skip_flag && *skip_flag != 0.f) return;   // GradScaler found_inf -> the step is skipped
  const MtTable t = mt_view(table, n);
  const long long c = blockIdx.x;
  const int ti = mt_find(t, c);
  const long long off = (c - t.chunk_start[ti]) * chunk;
  long long len = t.size[ti] - off;
  if (len > chunk) len = chunk;
  const float lr = lr_ptr ? *lr_ptr : lr_host;
  const float is = inv_scale ? *inv_scale : 1.0f;
  float gs = is;
  if (gnorm && max_norm > 0.f) {
    const float coef = max_norm / (*gnorm * is + 1e-6f);
    if (coef < 1.0f) gs = is * coef;
  }
  const int step = *step_ptr;
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2 = (float)(1.0 - pow((double)beta2, (double)step));
  const float step_size = lr / bc1;
  const float rsq_bc2 = 1.0f / sqrtf(bc2);
  const float* g = (const float*)t.ptr[0 * n + ti] + off;
  float* p = (float*)t.ptr[1 * n + ti] + off;
  float* m = (float*)t.ptr[2 * n + ti] + off;
  float* v = (float*)t.ptr[3 * n + ti] + off;
  for (long long i = threadIdx.x; i < len; i += MT_BLOCK) {
    float gr = g[i] * gs;
    const float pi = p[i];
    gr += wd * pi;
    const float mi = beta1 * m[i] + (1.f - beta1) * gr;
    const float vi = beta2 * v[i] + (1.f - beta2) * gr * gr;
    m[i] = mi;
    v[i] = vi;
    p[i] = pi - step_size * mi / (sqrtf(vi) * rsq_bc2 + eps);
  }
}

// param_norm[t] / update_norm[t] = sqrt of tensor t's chunk partials (the fold of mt_l2norm_finish, both arrays in one launch)
__global__ __launch_bounds__(MT_BLOCK) void mt_lamb_norms_finish(const float* __restrict__ partial_p, const float* __restrict__ partial_u,
                                                                 const long long* __restrict__ table, int n,
                                                                 float* __restrict__ param_norm, float* __restrict__ update_norm,
                                                                 int* noop) {
  const int ti = blockIdx.x;
  __shared__ float red[16];
  const MtTable t = mt_view(table, n);
  float sp = 0.f, su = 0.f;
  // (the flag may have been set before stage 1: its workgroups then left at the top and the partials are stale -- the norms are
  //  not read in that case, stage 2 leaves at the top too; they are written as zeros like the l2norm pass's outputs)
  const bool skipped = noop && *noop;
  if (!skipped)
    for (long long i = t.chunk_start[ti] + threadIdx.x; i < t.chunk_start[ti + 1]; i += MT_BLOCK) { sp += partial_p[i]; su += partial_u[i]; }
  sp = block_sum(sp, red);
  su = block_sum(su, red);
  if (threadIdx.x == 0) {
    const bool bad = !isfinite(sp) || !isfinite(su);
    if (bad && noop) *noop = 1;
    param_norm[ti] = (skipped || bad) ? 0.f : sqrtf(sp);
    update_norm[ti] = (skipped || bad) ? 0.f : sqrtf(su);
  }
}

// -------------------------------------------------------------------- C ABI
extern "C" int64_t dle_mt_table_len(int n_tensors, int n_lists) {
  return (int64_t)n_tensors + (n_tensors + 1) + (int64_t)n_lists * n_tensors;
}

// Host helper: fill a table image (host memory) from pointer lists; returns total chunk count.
extern "C" int64_t dle_mt_table_fill(int64_t* table_host, int n_tensors, int n_lists, const int64_t* sizes,
                                     const void* const* ptrs /* [n_lists][n_tensors] */, int chunk) {
  int64_t acc = 0;
  for (int t = 0; t < n_tensors; ++t) {
    table_host[t] = sizes[t];
    table_host[n_tensors + t] = acc;
    acc += (sizes[t] + chunk - 1) / chunk;
  }
  table_host[n_tensors + n_tensors] = acc;
  int64_t* p = table_host + n_tensors + (n_tensors + 1);
  for (int l = 0; l < n_lists; ++l)
    for (int t = 0; t < n_tensors; ++t) p[(int64_t)l * n_tensors + t] = (int64_t)(uintptr_t)ptrs[(int64_t)l * n_tensors + t];
  return acc;
}

// partial: device scratch of >= total_chunks floats.
extern "C" int dle_mt_l2norm(const int64_t* table_dev, int n_tensors, int64_t total_chunks, int chunk, int dtype,
                             float* partial, float* ret, float* ret_per_tensor, int per_tensor, int* noop_flag,
                             hipStream_t stream) {
  DLE_CHECK_ARG(table_dev && partial && ret, "mt_l2norm: null pointer");
  DLE_CHECK_ARG(!per_tensor || ret_per_tensor, "mt_l2norm: per_tensor needs ret_per_tensor");
  DLE_CHECK_ARG(chunk > 0 && chunk % 4 == 0, "mt_l2norm: chunk must be a positive multiple of 4");
  if (n_tensors == 0 || total_chunks == 0) {
    hipError_t e = hipMemsetAsync(ret, 0, 4, stream);
    return e == hipSuccess ? 0 : (int)e;
  }
  dim3 grid((unsigned)total_chunks), block(MT_BLOCK);
  if (dtype == DLE_F32) hipLaunchKernelGGL(mt_l2norm_partial<DLE_F32>, grid, block, 0, stream, (const long long*)table_dev, n_tensors, chunk, partial, noop_flag);
  else if (dtype == DLE_F16) hipLaunchKernelGGL(mt_l2norm_partial<DLE_F16>, grid, block, 0, stream, (const long long*)table_dev, n_tensors, chunk, partial, noop_flag);
  else if (dtype == DLE_BF16) hipLaunchKernelGGL(mt_l2norm_partial<DLE_BF16>, grid, block, 0, stream, (const long long*)table_dev, n_tensors, chunk, partial, noop_flag);
  else { dle_set_error("mt_l2norm: bad dtype %d", dtype); return -1; }
  DLE_LAUNCH_CHECK();
  hipLaunchKernelGGL(mt_l2norm_finish, dim3(per_tensor ? n_tensors + 1 : 1), block, 0, stream, partial,
                     (const long long*)table_dev, n_tensors, ret, ret_per_tensor, noop_flag);
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_mt_lamb_stage1(const int64_t* table_dev, int n_tensors, int64_t total_chunks, int chunk,
                                  int grad_dtype, const int* noop_flag, float beta1, float beta2, float beta3,
                                  const int* step_dev, int bias_correction, float eps, int mode, float weight_decay,
                                  const float* global_grad_norm, const float* max_grad_norm,
                                  const float* inv_scale, hipStream_t stream) {
  DLE_CHECK_ARG(table_dev && step_dev && global_grad_norm && max_grad_norm && inv_scale, "mt_lamb_stage1: null pointer");
  if (n_tensors == 0 || total_chunks == 0) return 0;
  dim3 grid((unsigned)total_chunks), block(MT_BLOCK);
#define GO(GT) hipLaunchKernelGGL((mt_lamb_stage1<GT, false>), grid, block, 0, stream, (const long long*)table_dev, n_tensors, chunk, (int*)noop_flag, beta1, beta2, beta3, step_dev, bias_correction, eps, mode, weight_decay, global_grad_norm, max_grad_norm, inv_scale, (float*)nullptr, (float*)nullptr)
  if (grad_dtype == DLE_F32) GO(DLE_F32);
  else if (grad_dtype == DLE_F16) GO(DLE_F16);
  else if (grad_dtype == DLE_BF16) GO(DLE_BF16);
  else { dle_set_error("mt_lamb_stage1: bad dtype %d", grad_dtype); return -1; }
#undef GO
  DLE_LAUNCH_CHECK();
  return 0;
}

// Stage 1 + the per-tensor norms multi_tensor_lamb_cuda takes in two l2norm sweeps around it (multi_tensor_lamb.cu:380-420): param_norm[t]
// = ||p_t|| BEFORE the step, update_norm[t] = ||update_t||, both fp32 [n_tensors], from per-chunk partial sums the stage leaves
// (partial: fp32 scratch of >= 2 * total_chunks).  weight_decay != 0 (the stage reads p only then; without decay and without
// NVLAMB stage 2 ignores the norms: call dle_mt_lamb_stage1).  A non-finite partial sets noop_flag and the norms come back 0.
extern "C" int dle_mt_lamb_stage1_norms(const int64_t* table_dev, int n_tensors, int64_t total_chunks, int chunk,
                                        int grad_dtype, int* noop_flag, float beta1, float beta2, float beta3,
                                        const int* step_dev, int bias_correction, float eps, int mode, float weight_decay,
                                        const float* global_grad_norm, const float* max_grad_norm, const float* inv_scale,
                                        float* partial, float* param_norm, float* update_norm, hipStream_t stream) {
  DLE_CHECK_ARG(table_dev && step_dev && global_grad_norm && max_grad_norm && inv_scale && partial && param_norm && update_norm,
                "mt_lamb_stage1_norms: null pointer");
  DLE_CHECK_ARG(weight_decay != 0.f, "mt_lamb_stage1_norms: the stage reads the parameters only under weight decay");
  if (n_tensors == 0 || total_chunks == 0) return 0;
  dim3 grid((unsigned)total_chunks), block(MT_BLOCK);
  float* pp = partial;
  float* pu = partial + total_chunks;
#define GO(GT) hipLaunchKernelGGL((mt_lamb_stage1<GT, true>), grid, block, 0, stream, (const long long*)table_dev, n_tensors, chunk, noop_flag, beta1, beta2, beta3, step_dev, bias_correction, eps, mode, weight_decay, global_grad_norm, max_grad_norm, inv_scale, pp, pu)
  if (grad_dtype == DLE_F32) GO(DLE_F32);
  else if (grad_dtype == DLE_F16) GO(DLE_F16);
  else if (grad_dtype == DLE_BF16) GO(DLE_BF16);
  else { dle_set_error("mt_lamb_stage1_norms: bad dtype %d", grad_dtype); return -1; }
#undef GO
  DLE_LAUNCH_CHECK();
  hipLaunchKernelGGL(mt_lamb_norms_finish, dim3(n_tensors), block, 0, stream, pp, pu, (const long long*)table_dev, n_tensors,
                     param_norm, update_norm, noop_flag);
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_mt_lamb_stage2(const int64_t* table_dev, int n_tensors, int64_t total_chunks, int chunk,
                                  int grad_dtype, int copy_dtype, const int* noop_flag, const float* param_norm,
                                  const float* update_norm, const float* lr_dev, float weight_decay, int use_nvlamb,
                                  hipStream_t stream) {
  DLE_CHECK_ARG(copy_dtype >= -1 && copy_dtype <= DLE_BF16, "mt_lamb_stage2: bad copy dtype %d", copy_dtype);
  DLE_CHECK_ARG(table_dev && param_norm && update_norm && lr_dev, "mt_lamb_stage2: null pointer");
  if (n_tensors == 0 || total_chunks == 0) return 0;
  dim3 grid((unsigned)total_chunks), block(MT_BLOCK);
#define GO2(GT, CT) hipLaunchKernelGGL((mt_lamb_stage2<GT, CT>), grid, block, 0, stream, (const long long*)table_dev, n_tensors, chunk, noop_flag, param_norm, update_norm, lr_dev, weight_decay, use_nvlamb)
#define GO(GT) do { if (copy_dtype == -1) GO2(GT, -1); else if (copy_dtype == DLE_F32) GO2(GT, DLE_F32); else if (copy_dtype == DLE_F16) GO2(GT, DLE_F16); else GO2(GT, DLE_BF16); } while (0)
  if (grad_dtype == DLE_F32) GO(DLE_F32);
  else if (grad_dtype == DLE_F16) GO(DLE_F16);
  else if (grad_dtype == DLE_BF16) GO(DLE_BF16);
  else { dle_set_error("mt_lamb_stage2: bad dtype %d", grad_dtype); return -1; }
#undef GO2
#undef GO
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_mt_sgd(const int64_t* table_dev, int n_tensors, int64_t total_chunks, int chunk, int grad_dtype,
                          int has_momentum, const float* skip_flag_dev, const float* lr_dev, float lr_host,
                          float momentum, float dampening, float weight_decay, int nesterov, int first_step,
                          const float* inv_scale_dev, int copy_dtype, hipStream_t stream) {
  DLE_CHECK_ARG(table_dev, "mt_sgd: null table");
  DLE_CHECK_ARG(copy_dtype == -1 || copy_dtype == DLE_F16 || copy_dtype == DLE_BF16, "mt_sgd: bad copy dtype %d", copy_dtype);
  if (n_tensors == 0 || total_chunks == 0) return 0;
  dim3 grid((unsigned)total_chunks), block(MT_BLOCK);
#define GO3(GT, HM, CT) hipLaunchKernelGGL((mt_sgd<GT, HM, CT>), grid, block, 0, stream, (const long long*)table_dev, n_tensors, chunk, skip_flag_dev, lr_dev, lr_host, momentum, dampening, weight_decay, nesterov, first_step, inv_scale_dev)
#define GO(GT, HM) do { if (copy_dtype == DLE_F16) GO3(GT, HM, DLE_F16); else if (copy_dtype == DLE_BF16) GO3(GT, HM, DLE_BF16); else GO3(GT, HM, -1); } while (0)
  if (grad_dtype == DLE_F32) { if (has_momentum) GO(DLE_F32, true); else GO(DLE_F32, false); }
  else if (grad_dtype == DLE_F16) { if (has_momentum) GO(DLE_F16, true); else GO(DLE_F16, false); }
  else if (grad_dtype == DLE_BF16) { if (has_momentum) GO(DLE_BF16, true); else GO(DLE_BF16, false); }
  else { dle_set_error("mt_sgd: bad dtype %d", grad_dtype); return -1; }
#undef GO
#undef GO3
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_mt_adam(const int64_t* table_dev, int n_tensors, int64_t total_chunks, int chunk,
                           const float* skip_flag_dev, const float* lr_dev, float lr_host, float beta1, float beta2,
                           float eps, float weight_decay, const int* step_dev, const float* inv_scale_dev,
                           const float* grad_norm_dev, float max_grad_norm, hipStream_t stream) {
  DLE_CHECK_ARG(table_dev && step_dev, "mt_adam: null table / step");
  if (n_tensors == 0 || total_chunks == 0) return 0;
  hipLaunchKernelGGL(mt_adam, dim3((unsigned)total_chunks), dim3(MT_BLOCK), 0, stream, (const long long*)table_dev, n_tensors,
                     chunk, skip_flag_dev, lr_dev, lr_host, beta1, beta2, eps, weight_decay, step_dev, inv_scale_dev,
                     grad_norm_dev, max_grad_norm);
  DLE_LAUNCH_CHECK();
  return 0;
}
