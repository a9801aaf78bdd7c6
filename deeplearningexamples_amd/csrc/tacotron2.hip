// Tacotron2 training-step kernels for gfx950 (second half; first correct path).
//
// Replace, around the dense contractions that go through dle_gemm (paths relative to
// /root/reference/PyTorch/SpeechSynthesis/Tacotron2/):
//   tacotron2/model.py:205-214,425-444  the pointwise part of nn.LSTM / nn.LSTMCell (cuDNN / ATen fused cell) + the F.dropout on the
//                                        hidden state that follows it in the decoder                    -> t2_lstm_{fwd,bwd}
//   tacotron2/model.py:79-121           Attention.forward of one decoder step: v . tanh(query + location + memory terms), masked
//                                        softmax over the text positions, context = weights x memory     -> t2_attention_{fwd,bwd}
//   tacotron2/model.py:170              torch.tanh of the postnet                                        -> t2_tanh_fwd
//   tacotron2/loss_function.py:42-44    the two MSE terms of Tacotron2Loss and their gradient            -> t2_mel_loss
// The decoder state is one row per sample (B ~ 50-100 rows): these are latency-bound kernels of a sequential loop, one thread per
// (sample, unit) for the cells, one workgroup per sample for the attention (wave64 shuffle reductions over channels).
#include "common.h"

#define T2_BLOCK 256

template <int DT> __device__ __forceinline__ float t2_ld(const unsigned short* p) { return Elem<DT>::to_f32(*p); }
template <int DT> __device__ __forceinline__ void t2_st(unsigned short* p, float v) { *p = Elem<DT>::from_f32(v); }
__device__ __forceinline__ float t2_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }

template <int DT>
__global__ __launch_bounds__(T2_BLOCK) void t2_tanh_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y,
                                                           long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] = Elem<DT>::from_f32(tanhf(Elem<DT>::to_f32(x[i])));
}

// gates [B, 4H] (i, f, g, o pre-activations, biases included) -> activations in place; c = f c_prev + i g; h = o tanh(c),
// dropped by the bit-packed keep mask (bit e of the mask <-> element keep_index + b*H + j); live[b] == 0: the row keeps
// (h_prev, c_prev) as its state and writes 0 to out_dst (packed-sequence semantics).
template <int DT>
__global__ __launch_bounds__(T2_BLOCK) void t2_lstm_fwd_kernel(unsigned short* gates, long long ld_g, const float* __restrict__ c_prev,
                                                               float* __restrict__ c_out, unsigned short* d0, long long ld0,
                                                               unsigned short* d1, long long ld1, unsigned short* d2, long long ld2,
                                                               const unsigned char* __restrict__ keep, long long keep_index,
                                                               float inv_keep, const float* __restrict__ live,
                                                               const unsigned short* __restrict__ h_prev, long long ld_hp,
                                                               unsigned short* out_dst, long long ld_out, int B, int H) {
  const long long total = (long long)B * H;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(idx / H), j = (int)(idx - (long long)b * H);
    unsigned short* gr = gates + b * ld_g + j;
    const float gi = t2_sigmoid(t2_ld<DT>(gr)), gf = t2_sigmoid(t2_ld<DT>(gr + H));
    const float gg = tanhf(t2_ld<DT>(gr + 2 * H)), go = t2_sigmoid(t2_ld<DT>(gr + 3 * H));
    t2_st<DT>(gr, gi); t2_st<DT>(gr + H, gf); t2_st<DT>(gr + 2 * H, gg); t2_st<DT>(gr + 3 * H, go);
    const float cp = c_prev[idx];
    float c = gf * cp + gi * gg;
    float h = go * tanhf(c);
    if (keep) {
      const long long e = keep_index + idx;
      h = ((keep[e >> 3] >> (e & 7)) & 1) ? h * inv_keep : 0.f;
    }
    if (live) {
      const float lv = live[b];
      if (out_dst) t2_st<DT>(out_dst + b * ld_out + j, lv != 0.f ? h : 0.f);
      if (lv == 0.f) { h = t2_ld<DT>(h_prev + b * ld_hp + j); c = cp; }
    }
    c_out[idx] = c;
    if (d0) t2_st<DT>(d0 + b * ld0 + j, h);
    if (d1) t2_st<DT>(d1 + b * ld1 + j, h);
    if (d2) t2_st<DT>(d2 + b * ld2 + j, h);
  }
}

// dh fp32 [B, H] (row stride ld_dh) = gradient wrt the (dropped) h; act = the saved activations; dgates may alias act.
template <int DT>
__global__ __launch_bounds__(T2_BLOCK) void t2_lstm_bwd_kernel(const float* __restrict__ dh, long long ld_dh,
                                                               const float* __restrict__ dc_next, const unsigned short* act,
                                                               long long ld_act, const float* __restrict__ c_prev,
                                                               unsigned short* dgates, long long ld_dg, float* __restrict__ dc_prev,
                                                               const unsigned char* __restrict__ keep, long long keep_index,
                                                               float inv_keep, const float* __restrict__ live,
                                                               float* __restrict__ dh_prev, int B, int H) {
  const long long total = (long long)B * H;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(idx / H), j = (int)(idx - (long long)b * H);
    const unsigned short* ar = act + b * ld_act + j;
    const float gi = t2_ld<DT>(ar), gf = t2_ld<DT>(ar + H), gg = t2_ld<DT>(ar + 2 * H), go = t2_ld<DT>(ar + 3 * H);
    const float dh_in = dh[b * ld_dh + j];
    float g = dh_in;
    if (keep) {
      const long long e = keep_index + idx;
      g = ((keep[e >> 3] >> (e & 7)) & 1) ? g * inv_keep : 0.f;
    }
    const float cp = c_prev[idx], dcn = dc_next[idx];
    const float tc = tanhf(gf * cp + gi * gg);
    const float d_o = g * tc;
    const float dc = dcn + g * go * (1.f - tc * tc);
    float di = dc * gg * gi * (1.f - gi), df = dc * cp * gf * (1.f - gf), dg = dc * gi * (1.f - gg * gg), dog = d_o * go * (1.f - go);
    float dcp = dc * gf;
    if (live) {
      const float lv = live[b];
      if (lv == 0.f) { di = df = dg = dog = 0.f; dcp = dcn; }
      dh_prev[idx] = lv == 0.f ? dh_in : 0.f;          // the carried state's gradient; the recurrent GEMM adds the live part
    }
    unsigned short* dr = dgates + b * ld_dg + j;
    t2_st<DT>(dr, di); t2_st<DT>(dr + H, df); t2_st<DT>(dr + 2 * H, dg); t2_st<DT>(dr + 3 * H, dog);
    dc_prev[idx] = dcp;
  }
}

// One workgroup per sample.  q fp32 [B, A]; pl [B*Ti, A] (processed memory + location term); v fp32 [A]; memory [B*Ti, E].
// LDS: Ti floats (energies -> weights) + 16 (reductions).
template <int DT>
__global__ __launch_bounds__(T2_BLOCK) void t2_attention_fwd_kernel(const float* __restrict__ q, const unsigned short* __restrict__ pl,
                                                                    const float* __restrict__ v, const unsigned short* __restrict__ memory,
                                                                    const long long* __restrict__ lengths,
                                                                    const unsigned short* __restrict__ awc_prev,
                                                                    unsigned short* __restrict__ tanh_out, float* __restrict__ aw_out,
                                                                    unsigned short* __restrict__ awc_next, unsigned short* d0, long long ld0,
                                                                    unsigned short* d1, long long ld1, unsigned short* d2, long long ld2,
                                                                    int Ti, int A, int E) {
  extern __shared__ float sm[];
  float* en = sm;                 // [Ti]
  float* red = sm + Ti;           // [16]
  const int b = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  int len = (int)lengths[b];
  if (len > Ti) len = Ti;
  const float* qb = q + (long long)b * A;
  for (int t = wave; t < Ti; t += nw) {
    const long long row = (long long)b * Ti + t;
    float acc = 0.f;
    for (int a = lane; a < A; a += 64) {
      const float th = tanhf(qb[a] + t2_ld<DT>(pl + row * A + a));
      const unsigned short ts = Elem<DT>::from_f32(th);
      tanh_out[row * A + a] = ts;
      acc += v[a] * Elem<DT>::to_f32(ts);              // the saved (rounded) tanh is what the backward pass sees
    }
    acc = wave_sum(acc);
    if (lane == 0) en[t] = acc;
  }
  __syncthreads();
  float mx = -3.0e38f;
  for (int t = threadIdx.x; t < len; t += blockDim.x) mx = fmaxf(mx, en[t]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = red[0];
  for (int i = 1; i < nw; ++i) mx = fmaxf(mx, red[i]);
  __syncthreads();
  float s = 0.f;
  for (int t = threadIdx.x; t < Ti; t += blockDim.x) {
    const float e = t < len ? __expf(en[t] - mx) : 0.f;
    en[t] = e;
    s += e;
  }
  s = block_sum(s, red);
  const float inv = 1.0f / s;
  __syncthreads();
  for (int t = threadIdx.x; t < Ti; t += blockDim.x) {
    const float w = en[t] * inv;
    en[t] = w;
    const long long row = (long long)b * Ti + t;
    aw_out[row] = w;
    const float cum = (awc_prev ? t2_ld<DT>(awc_prev + row * 8 + 1) : 0.f) + w;
    ushort8_t o = {0, 0, 0, 0, 0, 0, 0, 0};
    o[0] = Elem<DT>::from_f32(w);
    o[1] = Elem<DT>::from_f32(cum);
    *(ushort8_t*)(awc_next + row * 8) = o;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < E; c += blockDim.x) {
    float acc = 0.f;
    const unsigned short* mb = memory + (long long)b * Ti * E + c;
    for (int t = 0; t < len; ++t) acc += en[t] * t2_ld<DT>(mb + (long long)t * E);
    if (d0) t2_st<DT>(d0 + b * ld0 + c, acc);
    if (d1) t2_st<DT>(d1 + b * ld1 + c, acc);
    if (d2) t2_st<DT>(d2 + b * ld2 + c, acc);
  }
}

// One workgroup per sample.  LDS: Ti floats (d weights -> d energies) + A floats x waves (dq / dv partials) + 16.
template <int DT>
__global__ __launch_bounds__(T2_BLOCK) void t2_attention_bwd_kernel(const float* __restrict__ d_ctx, const float* __restrict__ d_aw_in,
                                                                    const float* __restrict__ aw, const unsigned short* __restrict__ tanh_out,
                                                                    const float* __restrict__ v, const unsigned short* __restrict__ memory,
                                                                    float* __restrict__ d_memory, unsigned short* __restrict__ d_pl,
                                                                    float* __restrict__ dq, float* __restrict__ dv_acc,
                                                                    float* __restrict__ d_pm_acc, int Ti, int A, int E) {
  extern __shared__ float sm[];
  float* de = sm;                           // [Ti]
  float* red = sm + Ti;                     // [16]
  float* part = red + 16;                   // [2][nw][A]
  const int b = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const float* dcb = d_ctx + (long long)b * E;
  for (int t = wave; t < Ti; t += nw) {
    const long long row = (long long)b * Ti + t;
    const float w = aw[row];
    float acc = 0.f;
    for (int c = lane; c < E; c += 64) {
      const float dc = dcb[c];
      acc += t2_ld<DT>(memory + row * E + c) * dc;
      if (d_memory) d_memory[row * E + c] += w * dc;        // NULL: the caller forms sum_t weights_t (x) d_ctx_t as one batched GEMM
    }
    acc = wave_sum(acc);
    if (lane == 0) de[t] = acc + d_aw_in[row];
  }
  __syncthreads();
  float s = 0.f;
  for (int t = threadIdx.x; t < Ti; t += blockDim.x) s += aw[(long long)b * Ti + t] * de[t];
  s = block_sum(s, red);
  __syncthreads();
  for (int t = threadIdx.x; t < Ti; t += blockDim.x) de[t] = aw[(long long)b * Ti + t] * (de[t] - s);
  __syncthreads();
  for (int a = lane; a < A; a += 64) { part[wave * A + a] = 0.f; part[(nw + wave) * A + a] = 0.f; }
  for (int t = wave; t < Ti; t += nw) {
    const long long row = (long long)b * Ti + t;
    const float e = de[t];
    for (int a = lane; a < A; a += 64) {
      const float th = t2_ld<DT>(tanh_out + row * A + a);
      const float dpre = e * v[a] * (1.f - th * th);
      d_pl[row * A + a] = Elem<DT>::from_f32(dpre);
      d_pm_acc[row * A + a] += dpre;
      part[wave * A + a] += dpre;
      part[(nw + wave) * A + a] += e * th;
    }
  }
  __syncthreads();
  for (int a = threadIdx.x; a < A; a += blockDim.x) {
    float sq = 0.f, sv = 0.f;
    for (int w = 0; w < nw; ++w) { sq += part[w * A + a]; sv += part[(nw + w) * A + a]; }
    dq[(long long)b * A + a] = sq;
    atomicAdd(dv_acc + a, sv);
  }
}

// MSE(mel_out, target) + MSE(mel_out + post, target), mean over R * n_mel elements; gradients scaled by *scale.
template <int DT>
__global__ __launch_bounds__(T2_BLOCK) void t2_mel_loss_kernel(const float* __restrict__ out_all, long long ld_out,
                                                               const unsigned short* __restrict__ post, const float* __restrict__ target,
                                                               const float* __restrict__ scale, unsigned short* __restrict__ d_out,
                                                               long long ld_dout, unsigned short* __restrict__ d_post,
                                                               float* __restrict__ partial, long long R, int n_mel) {
  __shared__ float red[16];
  const long long total = R * n_mel;
  const float invn = 1.0f / (float)total;
  const float sc = scale ? *scale : 1.0f;
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / n_mel;
    const int c = (int)(i - r * n_mel);
    const float mo = out_all[r * ld_out + c], tg = target[i];
    const float mp = mo + Elem<DT>::to_f32(post[i]);
    const float e1 = mo - tg, e2 = mp - tg;
    s += e1 * e1 + e2 * e2;
    const float g2 = 2.0f * e2 * invn * sc;
    d_post[i] = Elem<DT>::from_f32(g2);
    d_out[r * ld_dout + c] = Elem<DT>::from_f32(2.0f * e1 * invn * sc + g2);
  }
  const float t = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = t * invn;
}

__global__ __launch_bounds__(T2_BLOCK) void t2_sum_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += partial[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) *out = s;
}

// =============================================================================================== C ABI
static int t2_grid(long long items, int cap = 1024) {
  long long g = (items + T2_BLOCK - 1) / T2_BLOCK;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}
#define T2_DT_CHECK(what) DLE_CHECK_ARG(dtype == DLE_F16 || dtype == DLE_BF16, what ": 16-bit dtypes only (got %d)", dtype)
#define T2_GO(KERNEL, GRID, LDS, ...)                                                                        \
  do {                                                                                                       \
    if (dtype == DLE_F16) hipLaunchKernelGGL(KERNEL<DLE_F16>, dim3(GRID), dim3(T2_BLOCK), LDS, stream, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL<DLE_BF16>, dim3(GRID), dim3(T2_BLOCK), LDS, stream, __VA_ARGS__);                 \
    DLE_LAUNCH_CHECK();                                                                                      \
  } while (0)

extern "C" int dle_t2_tanh_fwd(const void* x, void* y, int64_t n, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(x && y && n > 0, "t2_tanh_fwd: bad args");
  T2_DT_CHECK("t2_tanh_fwd");
  T2_GO(t2_tanh_kernel, t2_grid(n, 4096), 0, (const unsigned short*)x, (unsigned short*)y, (long long)n);
  return 0;
}

extern "C" int dle_t2_lstm_fwd(void* gates, int64_t ld_g, const float* c_prev, float* c_out, void* d0, int64_t ld0, void* d1,
                               int64_t ld1, void* d2, int64_t ld2, const void* keep, int64_t keep_index, float inv_keep,
                               const float* live, const void* h_prev, int64_t ld_hp, void* out_dst, int64_t ld_out, int B, int H,
                               int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(gates && c_prev && c_out && B > 0 && H > 0 && ld_g >= 4LL * H, "t2_lstm_fwd: bad args");
  DLE_CHECK_ARG(!live || h_prev, "t2_lstm_fwd: live rows need h_prev");
  T2_DT_CHECK("t2_lstm_fwd");
  T2_GO(t2_lstm_fwd_kernel, t2_grid((long long)B * H), 0, (unsigned short*)gates, (long long)ld_g, c_prev, c_out, (unsigned short*)d0,
        (long long)ld0, (unsigned short*)d1, (long long)ld1, (unsigned short*)d2, (long long)ld2, (const unsigned char*)keep,
        (long long)keep_index, inv_keep, live, (const unsigned short*)h_prev, (long long)ld_hp, (unsigned short*)out_dst,
        (long long)ld_out, B, H);
  return 0;
}

extern "C" int dle_t2_lstm_bwd(const float* dh, int64_t ld_dh, const float* dc_next, const void* act, int64_t ld_act,
                               const float* c_prev, void* dgates, int64_t ld_dg, float* dc_prev, const void* keep, int64_t keep_index,
                               float inv_keep, const float* live, float* dh_prev, int B, int H, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dh && dc_next && act && c_prev && dgates && dc_prev && B > 0 && H > 0, "t2_lstm_bwd: bad args");
  DLE_CHECK_ARG(!live || dh_prev, "t2_lstm_bwd: live rows need dh_prev");
  T2_DT_CHECK("t2_lstm_bwd");
  T2_GO(t2_lstm_bwd_kernel, t2_grid((long long)B * H), 0, dh, (long long)ld_dh, dc_next, (const unsigned short*)act, (long long)ld_act,
        c_prev, (unsigned short*)dgates, (long long)ld_dg, dc_prev, (const unsigned char*)keep, (long long)keep_index, inv_keep, live,
        dh_prev, B, H);
  return 0;
}

extern "C" int dle_t2_attention_fwd(const float* q, const void* pl, const float* v, const void* memory, const int64_t* lengths,
                                    const void* awc_prev, void* tanh_out, float* aw_out, void* awc_next, void* d0, int64_t ld0,
                                    void* d1, int64_t ld1, void* d2, int64_t ld2, int B, int Ti, int A, int E, int dtype,
                                    hipStream_t stream) {
  DLE_CHECK_ARG(q && pl && v && memory && lengths && tanh_out && aw_out && awc_next && B > 0 && Ti > 0 && A > 0 && E > 0,
                "t2_attention_fwd: bad args");
  DLE_CHECK_ARG(Ti <= 8192 && ((((uintptr_t)awc_next) | ((uintptr_t)awc_prev)) & 15) == 0, "t2_attention_fwd: Ti <= 8192, aligned weights rows");
  T2_DT_CHECK("t2_attention_fwd");
  const size_t lds = (size_t)(Ti + 16) * 4;
  T2_GO(t2_attention_fwd_kernel, B, lds, q, (const unsigned short*)pl, v, (const unsigned short*)memory, (const long long*)lengths,
        (const unsigned short*)awc_prev, (unsigned short*)tanh_out, aw_out, (unsigned short*)awc_next, (unsigned short*)d0,
        (long long)ld0, (unsigned short*)d1, (long long)ld1, (unsigned short*)d2, (long long)ld2, Ti, A, E);
  return 0;
}

extern "C" int dle_t2_attention_bwd(const float* d_ctx, const float* d_aw_in, const float* aw, const void* tanh_out, const float* v,
                                    const void* memory, float* d_memory, void* d_pl, float* dq, float* dv_acc, float* d_pm_acc,
                                    int B, int Ti, int A, int E, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(d_ctx && d_aw_in && aw && tanh_out && v && memory && d_pl && dq && dv_acc && d_pm_acc && B > 0 && Ti > 0 &&
                A > 0 && E > 0, "t2_attention_bwd: bad args");
  const size_t lds = ((size_t)Ti + 16 + 2 * (T2_BLOCK / 64) * (size_t)A) * 4;
  DLE_CHECK_ARG(lds <= 60000, "t2_attention_bwd: Ti / attention_dim too large for one workgroup's LDS");
  T2_DT_CHECK("t2_attention_bwd");
  T2_GO(t2_attention_bwd_kernel, B, lds, d_ctx, d_aw_in, aw, (const unsigned short*)tanh_out, v, (const unsigned short*)memory,
        d_memory, (unsigned short*)d_pl, dq, dv_acc, d_pm_acc, Ti, A, E);
  return 0;
}

// workspace: >= 1024 floats
extern "C" int dle_t2_mel_loss(const float* out_all, int64_t ld_out, const void* post, const float* target, const float* scale_dev,
                               void* d_out, int64_t ld_dout, void* d_post, float* loss, float* workspace, int64_t R, int n_mel,
                               int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(out_all && post && target && d_out && d_post && loss && workspace && R > 0 && n_mel > 0, "t2_mel_loss: bad args");
  T2_DT_CHECK("t2_mel_loss");
  const int G = t2_grid(R * n_mel, 1024);
  T2_GO(t2_mel_loss_kernel, G, 0, out_all, (long long)ld_out, (const unsigned short*)post, target, scale_dev, (unsigned short*)d_out,
        (long long)ld_dout, (unsigned short*)d_post, workspace, (long long)R, n_mel);
  hipLaunchKernelGGL(t2_sum_kernel, dim3(1), dim3(T2_BLOCK), 0, stream, (const float*)workspace, G, loss);
  DLE_LAUNCH_CHECK();
  return 0;
}
