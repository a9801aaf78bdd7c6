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
#include <type_traits>
#include "gemm_tiles.h"

#define T2_BLOCK 256

template <int DT> __device__ __forceinline__ float t2_ld(const unsigned short* p) { return Elem<DT>::to_f32(*p); }
template <int DT> __device__ __forceinline__ void t2_st(unsigned short* p, float v) { *p = Elem<DT>::from_f32(v); }
__device__ __forceinline__ float t2_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ void t2_ld8f(const float* p, float* v) {
  const float4_t a = *(const float4_t*)p, b = *(const float4_t*)(p + 4);
#pragma unroll
  for (int r = 0; r < 4; ++r) { v[r] = a[r]; v[4 + r] = b[r]; }
}

template <int DT>
__global__ __launch_bounds__(T2_BLOCK) void t2_tanh_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y,
                                                           long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] = Elem<DT>::from_f32(tanhf(Elem<DT>::to_f32(x[i])));
}

// gates [B, 4H] (i, f, g, o pre-activations, biases included) -> activations in place; c = f c_prev + i g; h = o tanh(c),
// dropped by the bit-packed keep mask (bit e of the mask <-> element keep_index + b*H + j); live[b] == 0: the row keeps
// (h_prev, c_prev) as its state and writes 0 to out_dst (packed-sequence semantics).
// One thread per (sample, 8 consecutive units): 16-byte loads / stores of every 16-bit row segment (the first version moved 2 bytes
// per access: ~4.5 us per call, 4,300 calls per iteration); VEC = false is the scalar form for H % 8 != 0 or unaligned views.
template <int DT, bool VEC>
__global__ __launch_bounds__(128) void t2_lstm_fwd_kernel(unsigned short* gates, long long ld_g, const float* __restrict__ c_prev,
                                                          float* __restrict__ c_out, unsigned short* d0, long long ld0,
                                                          unsigned short* d1, long long ld1, unsigned short* d2, long long ld2,
                                                          const unsigned char* __restrict__ keep, long long keep_index,
                                                          float inv_keep, const float* __restrict__ live,
                                                          const unsigned short* __restrict__ h_prev, long long ld_hp,
                                                          unsigned short* out_dst, long long ld_out, int B, int H) {
  constexpr int W = VEC ? 8 : 1;
  const int hw = H / W;
  const long long total = (long long)B * hw;
  for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(it / hw), j = (int)(it - (long long)b * hw) * W;
    const long long idx = (long long)b * H + j;
    unsigned short* gr = gates + b * ld_g + j;
    float gi[W], gf[W], gg[W], go[W], cp[W], c[W], h[W];
    if constexpr (VEC) {
      unpack8<DT>(*(const ushort8_t*)gr, gi); unpack8<DT>(*(const ushort8_t*)(gr + H), gf);
      unpack8<DT>(*(const ushort8_t*)(gr + 2 * H), gg); unpack8<DT>(*(const ushort8_t*)(gr + 3 * H), go);
      t2_ld8f(c_prev + idx, cp);
    } else {
      gi[0] = t2_ld<DT>(gr); gf[0] = t2_ld<DT>(gr + H); gg[0] = t2_ld<DT>(gr + 2 * H); go[0] = t2_ld<DT>(gr + 3 * H);
      cp[0] = c_prev[idx];
    }
    unsigned kbits = 0xffu;
    if (keep) {
      const long long e = keep_index + idx;
      kbits = VEC ? keep[e >> 3] : ((keep[e >> 3] >> (e & 7)) & 1u);           // (VEC: keep_index % 8 == 0, checked by the launcher)
    }
#pragma unroll
    for (int r = 0; r < W; ++r) {
      gi[r] = t2_sigmoid(gi[r]); gf[r] = t2_sigmoid(gf[r]); gg[r] = fast_tanh(gg[r]); go[r] = t2_sigmoid(go[r]);
      c[r] = gf[r] * cp[r] + gi[r] * gg[r];
      h[r] = go[r] * fast_tanh(c[r]);
      if (keep) h[r] = ((kbits >> r) & 1u) ? h[r] * inv_keep : 0.f;
    }
    float lv = 1.f;
    if (live) {
      lv = live[b];
      if (out_dst) {
        float o[W];
#pragma unroll
        for (int r = 0; r < W; ++r) o[r] = lv != 0.f ? h[r] : 0.f;
        if constexpr (VEC) *(ushort8_t*)(out_dst + b * ld_out + j) = pack8<DT>(o);
        else t2_st<DT>(out_dst + b * ld_out + j, o[0]);
      }
      if (lv == 0.f) {
        if constexpr (VEC) unpack8<DT>(*(const ushort8_t*)(h_prev + b * ld_hp + j), h);
        else h[0] = t2_ld<DT>(h_prev + b * ld_hp + j);
#pragma unroll
        for (int r = 0; r < W; ++r) c[r] = cp[r];
      }
    }
    if constexpr (VEC) {
      *(ushort8_t*)gr = pack8<DT>(gi); *(ushort8_t*)(gr + H) = pack8<DT>(gf);
      *(ushort8_t*)(gr + 2 * H) = pack8<DT>(gg); *(ushort8_t*)(gr + 3 * H) = pack8<DT>(go);
      *(float4_t*)(c_out + idx) = (float4_t){c[0], c[1], c[2], c[3]};
      *(float4_t*)(c_out + idx + 4) = (float4_t){c[4], c[5], c[6], c[7]};
      const ushort8_t hv = pack8<DT>(h);
      if (d0) *(ushort8_t*)(d0 + b * ld0 + j) = hv;
      if (d1) *(ushort8_t*)(d1 + b * ld1 + j) = hv;
      if (d2) *(ushort8_t*)(d2 + b * ld2 + j) = hv;
    } else {
      t2_st<DT>(gr, gi[0]); t2_st<DT>(gr + H, gf[0]); t2_st<DT>(gr + 2 * H, gg[0]); t2_st<DT>(gr + 3 * H, go[0]);
      c_out[idx] = c[0];
      if (d0) t2_st<DT>(d0 + b * ld0 + j, h[0]);
      if (d1) t2_st<DT>(d1 + b * ld1 + j, h[0]);
      if (d2) t2_st<DT>(d2 + b * ld2 + j, h[0]);
    }
  }
}

// dh fp32 [B, H] (row stride ld_dh) = gradient wrt the (dropped) h (+ dh1 + dh2 when given: the pieces that reach the hidden state
// through different consumers -- projection, next step's gates, query -- are summed here, not by separate passes); act = the
// saved activations; dgates may alias act.  Same 8-units-per-thread layout as the forward cell.
template <int DT, bool VEC>
__global__ __launch_bounds__(128) void t2_lstm_bwd_kernel(const float* __restrict__ dh, long long ld_dh,
                                                          const float* __restrict__ dh1, long long ld_dh1,
                                                          const float* __restrict__ dh2, long long ld_dh2,
                                                          const float* __restrict__ dc_next, const unsigned short* act,
                                                          long long ld_act, const float* __restrict__ c_prev,
                                                          unsigned short* dgates, long long ld_dg, float* __restrict__ dc_prev,
                                                          const unsigned char* __restrict__ keep, long long keep_index,
                                                          float inv_keep, const float* __restrict__ live,
                                                          float* __restrict__ dh_prev, int B, int H) {
  constexpr int W = VEC ? 8 : 1;
  const int hw = H / W;
  const long long total = (long long)B * hw;
  for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(it / hw), j = (int)(it - (long long)b * hw) * W;
    const long long idx = (long long)b * H + j;
    const unsigned short* ar = act + b * ld_act + j;
    float gi[W], gf[W], gg[W], go[W], dhv[W], cp[W], dcn[W], t8[W];
    if constexpr (VEC) {
      unpack8<DT>(*(const ushort8_t*)ar, gi); unpack8<DT>(*(const ushort8_t*)(ar + H), gf);
      unpack8<DT>(*(const ushort8_t*)(ar + 2 * H), gg); unpack8<DT>(*(const ushort8_t*)(ar + 3 * H), go);
      t2_ld8f(dh + b * ld_dh + j, dhv);
      if (dh1) { t2_ld8f(dh1 + b * ld_dh1 + j, t8);
#pragma unroll
        for (int r = 0; r < W; ++r) dhv[r] += t8[r]; }
      if (dh2) { t2_ld8f(dh2 + b * ld_dh2 + j, t8);
#pragma unroll
        for (int r = 0; r < W; ++r) dhv[r] += t8[r]; }
      t2_ld8f(c_prev + idx, cp);
      t2_ld8f(dc_next + idx, dcn);
    } else {
      gi[0] = t2_ld<DT>(ar); gf[0] = t2_ld<DT>(ar + H); gg[0] = t2_ld<DT>(ar + 2 * H); go[0] = t2_ld<DT>(ar + 3 * H);
      dhv[0] = dh[b * ld_dh + j];
      if (dh1) dhv[0] += dh1[b * ld_dh1 + j];
      if (dh2) dhv[0] += dh2[b * ld_dh2 + j];
      cp[0] = c_prev[idx]; dcn[0] = dc_next[idx];
    }
    unsigned kbits = 0xffu;
    if (keep) {
      const long long e = keep_index + idx;
      kbits = VEC ? keep[e >> 3] : ((keep[e >> 3] >> (e & 7)) & 1u);
    }
    const float lv = live ? live[b] : 1.f;
    float di[W], df[W], dg[W], dog[W], dcp[W], dhp[W];
#pragma unroll
    for (int r = 0; r < W; ++r) {
      float g = dhv[r];
      if (keep) g = ((kbits >> r) & 1u) ? g * inv_keep : 0.f;
      const float tc = fast_tanh(gf[r] * cp[r] + gi[r] * gg[r]);
      const float d_o = g * tc;
      const float dc = dcn[r] + g * go[r] * (1.f - tc * tc);
      di[r] = dc * gg[r] * gi[r] * (1.f - gi[r]); df[r] = dc * cp[r] * gf[r] * (1.f - gf[r]);
      dg[r] = dc * gi[r] * (1.f - gg[r] * gg[r]); dog[r] = d_o * go[r] * (1.f - go[r]);
      dcp[r] = dc * gf[r];
      if (live) {
        if (lv == 0.f) { di[r] = df[r] = dg[r] = dog[r] = 0.f; dcp[r] = dcn[r]; }
        dhp[r] = lv == 0.f ? dhv[r] : 0.f;        // the carried state's gradient; the recurrent GEMM adds the live part
      }
    }
    unsigned short* dr = dgates + b * ld_dg + j;
    if constexpr (VEC) {
      *(ushort8_t*)dr = pack8<DT>(di); *(ushort8_t*)(dr + H) = pack8<DT>(df);
      *(ushort8_t*)(dr + 2 * H) = pack8<DT>(dg); *(ushort8_t*)(dr + 3 * H) = pack8<DT>(dog);
      *(float4_t*)(dc_prev + idx) = (float4_t){dcp[0], dcp[1], dcp[2], dcp[3]};
      *(float4_t*)(dc_prev + idx + 4) = (float4_t){dcp[4], dcp[5], dcp[6], dcp[7]};
      if (live) {
        *(float4_t*)(dh_prev + idx) = (float4_t){dhp[0], dhp[1], dhp[2], dhp[3]};
        *(float4_t*)(dh_prev + idx + 4) = (float4_t){dhp[4], dhp[5], dhp[6], dhp[7]};
      }
    } else {
      t2_st<DT>(dr, di[0]); t2_st<DT>(dr + H, df[0]); t2_st<DT>(dr + 2 * H, dg[0]); t2_st<DT>(dr + 3 * H, dog[0]);
      dc_prev[idx] = dcp[0];
      if (live) dh_prev[idx] = dhp[0];
    }
  }
}

// ---- location-sensitive attention of one decoder step -------------------------------------------------------------------------
// One workgroup of 8 wavefronts per sample (the softmax runs over the sample's text positions).  Round 2 ran 4 wavefronts with
// 2-byte scalar loads and a serial per-channel context loop: 72 / 139 us per step for ~26 MB (0.4 TB/s), a third of the Tacotron2
// iteration.  Here every global access is 16 bytes per lane: a row of A (or E) 16-bit values is covered by LP = pow2 >= A / 8 lanes,
// a wavefront holds 64 / LP rows per pass, row sums are xor-shuffles inside the LP-lane group, per-channel sums stay in 8
// registers per lane across the wavefront's rows and meet in LDS once.  Rows past the end are read at a clamped address with a
// zero weight (unconditional loads: hipcc can then keep several in flight).
#define T2A_BLOCK 1024
#define T2A_NW (T2A_BLOCK / 64)

// q fp32 [B, A]; pl [B*Ti, A] (processed memory + location term); v fp32 [A]; memory [B*Ti, E].
// wloc != NULL: `pl` is the processed memory ALONE and the location term (model.py:40-76: Conv1d(2 -> F, k = KL) on (previous,
// cumulative) weights + Linear(F -> A), pre-multiplied by the caller into ONE [A, KK] matrix, k = tap * 2 + channel, KK = KL * 2
// rounded up to 16, zero padded) is formed here on the matrix cores: the sample's (Ti + KL) x 2 weights sit in LDS, the im2col
// operand of v_mfma_f32_32x32x16 is four consecutive LDS dwords per lane, the [Ti, A] result goes through LDS (fp32) to the
// energy pass.  That removes, per decoder step, the row gather (10 MB written), a [B*Ti, A, KL*8] GEMM over 3/4 zero padding and
// the round trip of the summed term through HBM (taps 4.5 us + GEMM 12 us -> ~2 us inside this kernel).
// LDS: Ti floats (energies -> weights) + 16 (reductions) + T2A_NW x E (context partials) [+ Ti32 x (A + 4) + Ti32 + KK / 2 + 8].
template <int DT>
__global__ __launch_bounds__(T2A_BLOCK) void t2_attention_fwd_kernel(const float* __restrict__ q, const unsigned short* __restrict__ pl,
                                                                     const float* __restrict__ v, const unsigned short* __restrict__ memory,
                                                                     const long long* __restrict__ lengths,
                                                                     const unsigned short* __restrict__ awc_prev,
                                                                     unsigned short* __restrict__ tanh_out, float* __restrict__ aw_out,
                                                                     unsigned short* __restrict__ awc_next, unsigned short* d0, long long ld0,
                                                                     unsigned short* d1, long long ld1, unsigned short* d2, long long ld2,
                                                                     int Ti, int A, int E, int lpa, int lpe,
                                                                     const unsigned short* __restrict__ wloc, int KL, int KK) {
  extern __shared__ float sm[];
  float* en = sm;                 // [Ti] (padded to a multiple of 4: what follows is read 16 bytes at a time)
  float* red = sm + ((Ti + 3) & ~3);   // [16]
  float* part = red + 16;         // [T2A_NW][E]
  const int Ti32 = (Ti + 31) & ~31, LS = A + 4;
  float* loc = part + T2A_NW * E; // [Ti32][A + 4]  (wloc only)
  unsigned* seq = (unsigned*)(loc + (size_t)Ti32 * LS);     // [Ti32 + KK / 2 + 8]: (previous, cumulative) weights, zero padded
  const int b = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int len = (int)lengths[b];
  if (len > Ti) len = Ti;
  if (wloc) {
    const int pad = KL / 2;
    for (int p = threadIdx.x; p < Ti32 + KK / 2 + 8; p += T2A_BLOCK) {
      const int t = p - pad;
      unsigned x = 0;
      if (t >= 0 && t < Ti && awc_prev) x = *(const unsigned*)(awc_prev + ((long long)b * Ti + t) * 8);
      seq[p] = x;
    }
    __syncthreads();
    const int nab = A >> 5, ntb = Ti32 >> 5, fr = lane & 31, fh = lane >> 5;
    for (int job = wave; job < nab * ntb; job += T2A_NW) {
      const int ab = job % nab, tb = job / nab;
      float16_t acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      for (int ks = 0; ks < KK / 16; ++ks) {
        const ushort8_t fa = *(const ushort8_t*)(wloc + (long long)(ab * 32 + fr) * KK + ks * 16 + fh * 8);
        const unsigned* sp = seq + tb * 32 + fr + ks * 8 + fh * 4;         // taps ks*8 + fh*4 .. + 3, both channels
        const uint4_t xb = {sp[0], sp[1], sp[2], sp[3]};
        acc = Mfma32x16<DT>::run(fa, __builtin_bit_cast(ushort8_t, xb), acc);
      }
      // lane: position t = tb*32 + fr, channels ab*32 + 8 g + 4 fh + {0..3}
      float* lp = loc + (size_t)(tb * 32 + fr) * LS + ab * 32 + 4 * fh;
#pragma unroll
      for (int g = 0; g < 4; ++g) *(float4_t*)(lp + 8 * g) = (float4_t){acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    }
    __syncthreads();
  }
  {
    const int sub = lane & (lpa - 1), rin = lane / lpa, rpp = 64 / lpa;
    const bool act = sub * 8 < A;
    const int col = act ? sub * 8 : 0;
    float qv[8], vv[8];
    t2_ld8f(q + (long long)b * A + col, qv);
    t2_ld8f(v + col, vv);
#pragma unroll 2
    for (int t0 = wave * rpp; t0 < Ti; t0 += T2A_NW * rpp) {
      const int t = t0 + rin, tc = t < Ti ? t : Ti - 1;
      const long long row = (long long)b * Ti + tc;
      float xf[8], th[8], tr[8];
      unpack8<DT>(*(const ushort8_t*)(pl + row * A + col), xf);
      if (wloc) {
        const float* lp = loc + (size_t)tc * LS + col;
        const float4_t l0 = *(const float4_t*)lp, l1 = *(const float4_t*)(lp + 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { xf[r] += l0[r]; xf[4 + r] += l1[r]; }
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) th[r] = fast_tanh(qv[r] + xf[r]);
      const ushort8_t o = pack8<DT>(th);
      if (t < Ti && act) *(ushort8_t*)(tanh_out + row * A + col) = o;
      unpack8<DT>(o, tr);                                   // the saved (rounded) tanh is what the backward pass sees
      float acc = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) acc += vv[r] * tr[r];
      if (!act) acc = 0.f;
      for (int s = lpa >> 1; s > 0; s >>= 1) acc += __shfl_xor(acc, s, 64);
      if (t < Ti && sub == 0) en[t] = acc;
    }
  }
  __syncthreads();
  float mx = -3.0e38f;
  for (int t = threadIdx.x; t < len; t += T2A_BLOCK) mx = fmaxf(mx, en[t]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = red[0];
  for (int i = 1; i < T2A_NW; ++i) mx = fmaxf(mx, red[i]);
  __syncthreads();
  float s = 0.f;
  for (int t = threadIdx.x; t < Ti; t += T2A_BLOCK) {
    const float e = t < len ? __expf(en[t] - mx) : 0.f;
    en[t] = e;
    s += e;
  }
  s = block_sum(s, red);
  const float inv = 1.0f / s;
  __syncthreads();
  for (int t = threadIdx.x; t < Ti; t += T2A_BLOCK) {
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
  {
    const int sub = lane & (lpe - 1), rin = lane / lpe, rpp = 64 / lpe;
    const bool act = sub * 8 < E;
    const int col = act ? sub * 8 : 0;
    const unsigned short* mb = memory + (long long)b * Ti * E + col;
    float acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = 0.f;
#pragma unroll 4
    for (int t0 = wave * rpp; t0 < len; t0 += T2A_NW * rpp) {
      const int t = t0 + rin, tc = t < len ? t : (len > 0 ? len - 1 : 0);
      const float w = t < len ? en[tc] : 0.f;
      float xf[8];
      unpack8<DT>(*(const ushort8_t*)(mb + (long long)tc * E), xf);
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[r] += w * xf[r];
    }
    for (int o = 32; o >= lpe; o >>= 1) {
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[r] += __shfl_xor(acc[r], o, 64);
    }
    if (rin == 0 && act) {
#pragma unroll
      for (int r = 0; r < 8; ++r) part[wave * E + col + r] = acc[r];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < E; c += T2A_BLOCK) {
    float acc = 0.f;
#pragma unroll
    for (int w = 0; w < T2A_NW; ++w) acc += part[w * E + c];
    if (d0) t2_st<DT>(d0 + b * ld0 + c, acc);
    if (d1) t2_st<DT>(d1 + b * ld1 + c, acc);
    if (d2) t2_st<DT>(d2 + b * ld2 + c, acc);
  }
}

// Backward of the step above.  The gradient of the context arrives as the SUM of up to three fp32 row-strided pieces (projection
// of this step, the decoder LSTM's gates of this step, the attention LSTM's gates of the next step) and the gradient of the
// weights as the sum of two (through the location convolution's two input channels): summed on load, not by separate passes.
// Writes d_pl (16-bit), dq (fp32 and / or 16-bit = the query GEMM's operand), the summed context gradient (16-bit, the operand of
// the hoisted memory-gradient GEMM); accumulates d_pm (fp32) and the per-SAMPLE partials of dv (dv_acc [B, A]: one owner per row,
// no atomics -- deterministic; the caller folds the B rows after the sweep).
// LDS: Ti floats + 16 + T2A_NW x 2 x A.
template <int DT>
__global__ __launch_bounds__(T2A_BLOCK) void t2_attention_bwd_kernel(const float* __restrict__ dc0, long long ldc0,
                                                                     const float* __restrict__ dc1, long long ldc1,
                                                                     const float* __restrict__ dc2, long long ldc2,
                                                                     const float* daw0, const float* daw1,
                                                                     const float* __restrict__ aw, const unsigned short* __restrict__ tanh_out,
                                                                     const float* __restrict__ v, const unsigned short* __restrict__ memory,
                                                                     float* __restrict__ d_memory, unsigned short* __restrict__ d_pl,
                                                                     float* __restrict__ dq, unsigned short* __restrict__ dq16,
                                                                     unsigned short* __restrict__ dctx16, float* __restrict__ dv_acc,
                                                                     float* __restrict__ d_pm_acc, int Ti, int A, int E, int lpa, int lpe,
                                                                     const unsigned short* __restrict__ wlocT, int KL, int KK,
                                                                     float* d_prev, float* d_cum) {
  extern __shared__ float sm[];
  float* de = sm;                           // [Ti] (padded to a multiple of 4)
  float* red = sm + ((Ti + 3) & ~3);        // [16]
  float* part = red + 16;                   // [T2A_NW][2][A]
  // wlocT != NULL: the backward of the location term (the transposed convolution) runs here too: d_pl stays in LDS (16-bit),
  // dcol = d_pl x W_loc on the matrix cores, its anti-diagonal sums are the gradients of the previous / cumulative weights
  const int Ti32 = (Ti + 31) & ~31, PS = A + 8, CS = KK + 4;
  unsigned short* dpl_l = (unsigned short*)(part + T2A_NW * 2 * A);        // [Ti32][A + 8] 16-bit
  float* dcol_l = (float*)(dpl_l + (size_t)Ti32 * PS);                     // [Ti32][KK + 4] fp32
  if (wlocT) {
    for (int i = Ti * PS / 8 + threadIdx.x; i < Ti32 * PS / 8; i += T2A_BLOCK)
      ((ushort8_t*)dpl_l)[i] = (ushort8_t){0, 0, 0, 0, 0, 0, 0, 0};        // rows past the sample: zero operand rows
  }
  const int b = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  {
    const int sub = lane & (lpe - 1), rin = lane / lpe, rpp = 64 / lpe;
    const bool act = sub * 8 < E;
    const int col = act ? sub * 8 : 0;
    float dcv[8];
    t2_ld8f(dc0 + b * ldc0 + col, dcv);
    if (dc1) { float t8[8]; t2_ld8f(dc1 + b * ldc1 + col, t8);
#pragma unroll
      for (int r = 0; r < 8; ++r) dcv[r] += t8[r]; }
    if (dc2) { float t8[8]; t2_ld8f(dc2 + b * ldc2 + col, t8);
#pragma unroll
      for (int r = 0; r < 8; ++r) dcv[r] += t8[r]; }
    if (dctx16 && wave == 0 && rin == 0 && act) *(ushort8_t*)(dctx16 + (long long)b * E + col) = pack8<DT>(dcv);
    const unsigned short* mb = memory + (long long)b * Ti * E + col;
#pragma unroll 4
    for (int t0 = wave * rpp; t0 < Ti; t0 += T2A_NW * rpp) {
      const int t = t0 + rin, tc = t < Ti ? t : Ti - 1;
      const long long row = (long long)b * Ti + tc;
      float xf[8];
      unpack8<DT>(*(const ushort8_t*)(mb + (long long)tc * E), xf);
      float acc = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) acc += xf[r] * dcv[r];
      if (!act) acc = 0.f;
      if (d_memory && t < Ti && act) {                     // NULL: the caller forms sum_t weights_t (x) d_ctx_t as one batched GEMM
        const float w = aw[row];
        float* dm = d_memory + row * E + col;
#pragma unroll
        for (int r = 0; r < 8; ++r) dm[r] += w * dcv[r];
      }
      for (int s = lpe >> 1; s > 0; s >>= 1) acc += __shfl_xor(acc, s, 64);
      if (t < Ti && sub == 0) de[t] = acc + daw0[row] + (daw1 ? daw1[row] : 0.f);
    }
  }
  __syncthreads();
  float s = 0.f;
  for (int t = threadIdx.x; t < Ti; t += T2A_BLOCK) s += aw[(long long)b * Ti + t] * de[t];
  s = block_sum(s, red);
  __syncthreads();
  for (int t = threadIdx.x; t < Ti; t += T2A_BLOCK) de[t] = aw[(long long)b * Ti + t] * (de[t] - s);
  __syncthreads();
  {
    const int sub = lane & (lpa - 1), rin = lane / lpa, rpp = 64 / lpa;
    const bool act = sub * 8 < A;
    const int col = act ? sub * 8 : 0;
    float vv[8], accq[8], accv[8];
    t2_ld8f(v + col, vv);
#pragma unroll
    for (int r = 0; r < 8; ++r) { accq[r] = 0.f; accv[r] = 0.f; }
#pragma unroll 2
    for (int t0 = wave * rpp; t0 < Ti; t0 += T2A_NW * rpp) {
      const int t = t0 + rin, tc = t < Ti ? t : Ti - 1;
      const long long row = (long long)b * Ti + tc;
      const float e = t < Ti ? de[tc] : 0.f;
      float th[8], dpre[8];
      unpack8<DT>(*(const ushort8_t*)(tanh_out + row * A + col), th);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        dpre[r] = e * vv[r] * (1.f - th[r] * th[r]);
        accq[r] += dpre[r];
        accv[r] += e * th[r];
      }
      if (t < Ti && act) {
        const ushort8_t pv = pack8<DT>(dpre);
        *(ushort8_t*)(d_pl + row * A + col) = pv;
        if (wlocT) *(ushort8_t*)(dpl_l + (size_t)t * PS + col) = pv;
      }
      if (d_pm_acc && t < Ti && act) {            // NULL: the caller sums the kept d_pl over the steps itself (dle_t2_sum_steps)
        float* pm = d_pm_acc + row * A + col;
        float4_t p0 = *(float4_t*)pm, p1 = *(float4_t*)(pm + 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { p0[r] += dpre[r]; p1[r] += dpre[4 + r]; }
        *(float4_t*)pm = p0;
        *(float4_t*)(pm + 4) = p1;
      }
    }
    for (int o = 32; o >= lpa; o >>= 1) {
#pragma unroll
      for (int r = 0; r < 8; ++r) { accq[r] += __shfl_xor(accq[r], o, 64); accv[r] += __shfl_xor(accv[r], o, 64); }
    }
    if (rin == 0 && act) {
#pragma unroll
      for (int r = 0; r < 8; ++r) { part[(wave * 2) * A + col + r] = accq[r]; part[(wave * 2 + 1) * A + col + r] = accv[r]; }
    }
  }
  __syncthreads();
  for (int a = threadIdx.x; a < A; a += T2A_BLOCK) {
    float sq = 0.f, sv = 0.f;
#pragma unroll
    for (int w = 0; w < T2A_NW; ++w) { sq += part[(w * 2) * A + a]; sv += part[(w * 2 + 1) * A + a]; }
    if (dq) dq[(long long)b * A + a] = sq;
    if (dq16) dq16[(long long)b * A + a] = Elem<DT>::from_f32(sq);
    dv_acc[(long long)b * A + a] += sv;
  }
  if (wlocT) {
    // dcol[t, k] = sum_a d_pl[t, a] W_loc[a, k]: first operand = rows k of W_loc^T (a contiguous), second = rows t of d_pl
    const int nkb = KK >> 5, ntb = Ti32 >> 5, fr = lane & 31, fh = lane >> 5;
    for (int job = wave; job < nkb * ntb; job += T2A_NW) {
      const int kb = job % nkb, tb = job / nkb;
      float16_t acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      for (int ks = 0; ks < A / 16; ++ks) {
        const ushort8_t fa = *(const ushort8_t*)(wlocT + (long long)(kb * 32 + fr) * A + ks * 16 + fh * 8);
        const ushort8_t fb = *(const ushort8_t*)(dpl_l + (size_t)(tb * 32 + fr) * PS + ks * 16 + fh * 8);
        acc = Mfma32x16<DT>::run(fa, fb, acc);
      }
      float* cp = dcol_l + (size_t)(tb * 32 + fr) * CS + kb * 32 + 4 * fh;
#pragma unroll
      for (int g = 0; g < 4; ++g) *(float4_t*)(cp + 8 * g) = (float4_t){acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    }
    __syncthreads();
    // col[(t), j*2 + c] = weights[t + j - pad][c]  =>  d weights[s][c] = sum_j dcol[s - j + pad][j*2 + c]
    const int pad = KL / 2;
    for (int i = threadIdx.x; i < 2 * Ti; i += T2A_BLOCK) {
      const int sx = i >> 1, c = i & 1;
      float acc = 0.f;
      for (int j = 0; j < KL; ++j) {
        const int t = sx - j + pad;
        if (t >= 0 && t < Ti) acc += dcol_l[(size_t)t * CS + j * 2 + c];
      }
      const long long o = (long long)b * Ti + sx;
      if (c == 0) d_prev[o] = acc;           // gradient of weights_{t-1} through the "previous weights" channel
      else d_cum[o] += acc;                  // cumulative weights feed every later step: accumulated
    }
  }
}

// Transpose of the 2-channel location convolution's row gather (model.py:40-76 backward): dcol [B*Ti, KL*8] 16-bit = gradient of
// the gathered rows col[(b, t), j*8 + c] = awc[b, t + j - KL/2, c].  Channel 0 (the previous step's weights) -> d_prev (written),
// channel 1 (the cumulative weights) -> d_cum (accumulated: cumulative_t feeds every later step).  fp32 [B, Ti] both.
template <int DT>
__global__ __launch_bounds__(T2_BLOCK) void t2_location_bwd_kernel(const unsigned short* __restrict__ dcol, float* __restrict__ d_prev,
                                                                   float* __restrict__ d_cum, int B, int Ti, int KL) {
  const long long total = (long long)B * Ti;
  const int pad = KL / 2, ld = KL * 8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int s = (int)(i % Ti);
    const long long b = i / Ti;
    float a0 = 0.f, a1 = 0.f;
    for (int j = 0; j < KL; ++j) {
      const int t = s - j + pad;
      if (t < 0 || t >= Ti) continue;
      const unsigned int x = *(const unsigned int*)(dcol + (b * Ti + t) * ld + j * 8);
      a0 += Elem<DT>::to_f32((unsigned short)(x & 0xffffu));
      a1 += Elem<DT>::to_f32((unsigned short)(x >> 16));
    }
    d_prev[i] = a0;
    d_cum[i] += a1;
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

// model.py:648-655 parse_output under --mask-padding: rows (b, t) with t >= lengths[b] of a [B * To, cols] block are overwritten
// with one value (0 for the mel outputs and -- gradient side -- for what flows back into them, 1e3 for the gate energies).
template <int DT>   // DLE_F32: T = float; else 16-bit storage
__global__ __launch_bounds__(T2_BLOCK) void t2_mask_rows_kernel(void* __restrict__ xv, long long ld, int cols,
                                                                const long long* __restrict__ lengths, long long R, int To, float fill) {
  using T = typename std::conditional<DT == DLE_F32, float, unsigned short>::type;
  T* x = (T*)xv;
  T value;
  if constexpr (DT == DLE_F32) value = fill;
  else value = Elem<DT>::from_f32(fill);
  const long long total = R * cols;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cols;
    const int c = (int)(i - r * cols);
    const long long b = r / To;
    if ((long long)(r - b * To) >= lengths[b]) x[r * ld + c] = value;
  }
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
  {
    auto al = [](const void* p, int64_t ld) { return !p || (((((uintptr_t)p) & 15) == 0) && (ld & 7) == 0); };
    const bool vec = (H & 7) == 0 && (keep_index & 7) == 0 && al(gates, ld_g) && al(d0, ld0) && al(d1, ld1) && al(d2, ld2) &&
                     al(h_prev, ld_hp) && al(out_dst, ld_out) && ((((uintptr_t)c_prev) | ((uintptr_t)c_out)) & 15) == 0;
    const long long items = (long long)B * (vec ? H / 8 : H);
    long long grid = (items + 127) / 128;
    if (grid > 2048) grid = 2048;
#define T2_CELL(DT, V) hipLaunchKernelGGL((t2_lstm_fwd_kernel<DT, V>), dim3((unsigned)grid), dim3(128), 0, stream, (unsigned short*)gates, \
        (long long)ld_g, c_prev, c_out, (unsigned short*)d0, (long long)ld0, (unsigned short*)d1, (long long)ld1, (unsigned short*)d2,       \
        (long long)ld2, (const unsigned char*)keep, (long long)keep_index, inv_keep, live, (const unsigned short*)h_prev,                    \
        (long long)ld_hp, (unsigned short*)out_dst, (long long)ld_out, B, H)
    if (dtype == DLE_F16) { if (vec) T2_CELL(DLE_F16, true); else T2_CELL(DLE_F16, false); }
    else { if (vec) T2_CELL(DLE_BF16, true); else T2_CELL(DLE_BF16, false); }
#undef T2_CELL
    DLE_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int dle_t2_lstm_bwd(const float* dh, int64_t ld_dh, const float* dh1, int64_t ld_dh1, const float* dh2, int64_t ld_dh2,
                               const float* dc_next, const void* act, int64_t ld_act,
                               const float* c_prev, void* dgates, int64_t ld_dg, float* dc_prev, const void* keep, int64_t keep_index,
                               float inv_keep, const float* live, float* dh_prev, int B, int H, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dh && dc_next && act && c_prev && dgates && dc_prev && B > 0 && H > 0, "t2_lstm_bwd: bad args");
  DLE_CHECK_ARG(!live || dh_prev, "t2_lstm_bwd: live rows need dh_prev");
  T2_DT_CHECK("t2_lstm_bwd");
  {
    auto al = [](const void* p, int64_t ld, int mask) { return !p || (((((uintptr_t)p) & 15) == 0) && (ld & mask) == 0); };
    const bool vec = (H & 7) == 0 && (keep_index & 7) == 0 && al(dh, ld_dh, 3) && al(dh1, ld_dh1, 3) && al(dh2, ld_dh2, 3) &&
                     al(act, ld_act, 7) && al(dgates, ld_dg, 7) &&
                     ((((uintptr_t)c_prev) | ((uintptr_t)dc_next) | ((uintptr_t)dc_prev) | ((uintptr_t)dh_prev)) & 15) == 0;
    const long long items = (long long)B * (vec ? H / 8 : H);
    long long grid = (items + 127) / 128;
    if (grid > 2048) grid = 2048;
#define T2_CELL(DT, V) hipLaunchKernelGGL((t2_lstm_bwd_kernel<DT, V>), dim3((unsigned)grid), dim3(128), 0, stream, dh, (long long)ld_dh, dh1, \
        (long long)ld_dh1, dh2, (long long)ld_dh2, dc_next, (const unsigned short*)act, (long long)ld_act, c_prev, (unsigned short*)dgates,     \
        (long long)ld_dg, dc_prev, (const unsigned char*)keep, (long long)keep_index, inv_keep, live, dh_prev, B, H)
    if (dtype == DLE_F16) { if (vec) T2_CELL(DLE_F16, true); else T2_CELL(DLE_F16, false); }
    else { if (vec) T2_CELL(DLE_BF16, true); else T2_CELL(DLE_BF16, false); }
#undef T2_CELL
    DLE_LAUNCH_CHECK();
  }
  return 0;
}

static int t2_pow2_ge(int x) { int p = 1; while (p < x) p <<= 1; return p; }

extern "C" int dle_t2_attention_fwd(const float* q, const void* pl, const float* v, const void* memory, const int64_t* lengths,
                                    const void* awc_prev, void* tanh_out, float* aw_out, void* awc_next, void* d0, int64_t ld0,
                                    void* d1, int64_t ld1, void* d2, int64_t ld2, const void* wloc, int KL, int KK, int B, int Ti,
                                    int A, int E, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(q && pl && v && memory && lengths && tanh_out && aw_out && awc_next && B > 0 && Ti > 0 && A > 0 && E > 0,
                "t2_attention_fwd: bad args");
  DLE_CHECK_ARG(Ti <= 8192 && ((((uintptr_t)awc_next) | ((uintptr_t)awc_prev)) & 15) == 0, "t2_attention_fwd: Ti <= 8192, aligned weights rows");
  DLE_CHECK_ARG(A % 8 == 0 && E % 8 == 0 && A <= 512 && E <= 512, "t2_attention_fwd: A, E multiples of 8, at most 512 (got %d, %d)", A, E);
  DLE_CHECK_ARG(((((uintptr_t)q) | ((uintptr_t)pl) | ((uintptr_t)v) | ((uintptr_t)memory) | ((uintptr_t)tanh_out) | ((uintptr_t)wloc)) & 15) == 0,
                "t2_attention_fwd: 16-byte aligned tensors");
  T2_DT_CHECK("t2_attention_fwd");
  const int Ti32 = (Ti + 31) & ~31;
  if (!wloc) KK = 0;
  DLE_CHECK_ARG(!wloc || (KL > 0 && (KL & 1) && A % 32 == 0 && KK % 16 == 0 && KK >= 2 * KL),
                "t2_attention_fwd: the fused location term needs an odd kernel size, A %% 32 == 0 and rows of KK >= 2 KL, KK %% 16 == 0");
  size_t lds = ((size_t)((Ti + 3) & ~3) + 16 + (size_t)T2A_NW * E) * 4;
  if (wloc) lds += ((size_t)Ti32 * (A + 4) + Ti32 + KK / 2 + 8) * 4;
  DLE_CHECK_ARG(lds <= 160 * 1024, "t2_attention_fwd: Ti too large for one workgroup's LDS");
  const int lpa = t2_pow2_ge(A / 8), lpe = t2_pow2_ge(E / 8);
#define T2_ATT_FWD(DT)                                                                                                              \
  do {                                                                                                                              \
    static size_t lds_set = 0;                                                                                                      \
    if (lds > 65536 && lds > lds_set) {                                                                                             \
      (void)hipFuncSetAttribute((const void*)t2_attention_fwd_kernel<DT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
      lds_set = lds;                                                                                                                \
    }                                                                                                                               \
    hipLaunchKernelGGL(t2_attention_fwd_kernel<DT>, dim3(B), dim3(T2A_BLOCK), lds, stream, q, (const unsigned short*)pl, v,         \
                       (const unsigned short*)memory, (const long long*)lengths, (const unsigned short*)awc_prev,                   \
                       (unsigned short*)tanh_out, aw_out, (unsigned short*)awc_next, (unsigned short*)d0, (long long)ld0,           \
                       (unsigned short*)d1, (long long)ld1, (unsigned short*)d2, (long long)ld2, Ti, A, E, lpa, lpe,                \
                       (const unsigned short*)wloc, KL, KK);                                                                        \
  } while (0)
  if (dtype == DLE_F16) T2_ATT_FWD(DLE_F16); else T2_ATT_FWD(DLE_BF16);
#undef T2_ATT_FWD
  DLE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dle_t2_attention_bwd(const float* d_ctx0, int64_t ld_c0, const float* d_ctx1, int64_t ld_c1, const float* d_ctx2,
                                    int64_t ld_c2, const float* d_aw0, const float* d_aw1, const float* aw, const void* tanh_out,
                                    const float* v, const void* memory, float* d_memory, void* d_pl, float* dq, void* dq16,
                                    void* dctx16, float* dv_acc, float* d_pm_acc, const void* wlocT, int KL, int KK, float* d_prev,
                                    float* d_cum, int B, int Ti, int A, int E, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(d_ctx0 && d_aw0 && aw && tanh_out && v && memory && d_pl && (dq || dq16) && dv_acc && B > 0 && Ti > 0 &&
                A > 0 && E > 0, "t2_attention_bwd: bad args");
  DLE_CHECK_ARG(A % 8 == 0 && E % 8 == 0 && A <= 512 && E <= 512, "t2_attention_bwd: A, E multiples of 8, at most 512 (got %d, %d)", A, E);
  DLE_CHECK_ARG((ld_c0 & 3) == 0 && (!d_ctx1 || (ld_c1 & 3) == 0) && (!d_ctx2 || (ld_c2 & 3) == 0) &&
                ((((uintptr_t)d_ctx0) | ((uintptr_t)d_ctx1) | ((uintptr_t)d_ctx2) | ((uintptr_t)tanh_out) | ((uintptr_t)memory) |
                  ((uintptr_t)d_pl) | ((uintptr_t)d_pm_acc) | ((uintptr_t)v) | ((uintptr_t)dctx16) | ((uintptr_t)wlocT)) & 15) == 0,
                "t2_attention_bwd: 16-byte aligned tensors, row strides of the context gradients multiples of 4");
  const int Ti32 = (Ti + 31) & ~31;
  if (!wlocT) KK = 0;
  DLE_CHECK_ARG(!wlocT || (KL > 0 && (KL & 1) && A % 16 == 0 && KK % 32 == 0 && KK >= 2 * KL && d_prev && d_cum),
                "t2_attention_bwd: the fused location backward needs an odd kernel size, A %% 16 == 0, KK >= 2 KL with KK %% 32 == 0, d_prev and d_cum");
  size_t lds = ((size_t)((Ti + 3) & ~3) + 16 + 2 * T2A_NW * (size_t)A) * 4;
  if (wlocT) lds += (size_t)Ti32 * (A + 8) * 2 + (size_t)Ti32 * (KK + 4) * 4;
  DLE_CHECK_ARG(lds <= 160 * 1024, "t2_attention_bwd: Ti / attention_dim too large for one workgroup's LDS");
  T2_DT_CHECK("t2_attention_bwd");
  const int lpa = t2_pow2_ge(A / 8), lpe = t2_pow2_ge(E / 8);
#define T2_ATT_BWD(DT)                                                                                                              \
  do {                                                                                                                              \
    static size_t lds_set = 0;                                                                                                      \
    if (lds > 65536 && lds > lds_set) {                                                                                             \
      (void)hipFuncSetAttribute((const void*)t2_attention_bwd_kernel<DT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
      lds_set = lds;                                                                                                                \
    }                                                                                                                               \
    hipLaunchKernelGGL(t2_attention_bwd_kernel<DT>, dim3(B), dim3(T2A_BLOCK), lds, stream, d_ctx0, (long long)ld_c0, d_ctx1,        \
                       (long long)ld_c1, d_ctx2, (long long)ld_c2, d_aw0, d_aw1, aw, (const unsigned short*)tanh_out, v,            \
                       (const unsigned short*)memory, d_memory, (unsigned short*)d_pl, dq, (unsigned short*)dq16,                   \
                       (unsigned short*)dctx16, dv_acc, d_pm_acc, Ti, A, E, lpa, lpe, (const unsigned short*)wlocT, KL, KK,         \
                       d_prev, d_cum);                                                                                              \
  } while (0)
  if (dtype == DLE_F16) T2_ATT_BWD(DLE_F16); else T2_ATT_BWD(DLE_BF16);
#undef T2_ATT_BWD
  DLE_LAUNCH_CHECK();
  return 0;
}

// out[r] += sum_s x[s][r] (16-bit x, fp32 out, R % 8 == 0): the per-step gradients of a tensor every decoder step reads (the
// processed memory) are kept in 16 bits for a chunk of steps and folded here -- one pass over the chunk instead of an fp32
// read-modify-write of the accumulator inside every step's attention kernel.
template <int DT>
__global__ __launch_bounds__(T2_BLOCK) void t2_sum_steps_kernel(const unsigned short* __restrict__ x, float* __restrict__ out, int n,
                                                                long long R8) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < R8; i += (long long)gridDim.x * blockDim.x) {
    float acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = 0.f;
    int s = 0;
    for (; s + 4 <= n; s += 4) {
      ushort8_t v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = ((const ushort8_t*)x)[(long long)(s + u) * R8 + i];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float f[8];
        unpack8<DT>(v[u], f);
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[r] += f[r];
      }
    }
    for (; s < n; ++s) {
      float f[8];
      unpack8<DT>(((const ushort8_t*)x)[(long long)s * R8 + i], f);
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[r] += f[r];
    }
    float4_t* o = (float4_t*)(out + i * 8);
    float4_t a = o[0], b = o[1];
#pragma unroll
    for (int r = 0; r < 4; ++r) { a[r] += acc[r]; b[r] += acc[4 + r]; }
    o[0] = a;
    o[1] = b;
  }
}

extern "C" int dle_t2_sum_steps(const void* x, float* out, int n_steps, int64_t R, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(x && out && n_steps > 0 && R > 0 && R % 8 == 0 && ((((uintptr_t)x) | ((uintptr_t)out)) & 15) == 0, "t2_sum_steps: bad args");
  T2_DT_CHECK("t2_sum_steps");
  T2_GO(t2_sum_steps_kernel, t2_grid(R / 8, 4096), 0, (const unsigned short*)x, out, n_steps, (long long)(R / 8));
  return 0;
}

extern "C" int dle_t2_location_bwd(const void* dcol, float* d_prev, float* d_cum, int B, int Ti, int KL, int dtype, hipStream_t stream) {
  DLE_CHECK_ARG(dcol && d_prev && d_cum && B > 0 && Ti > 0 && KL > 0 && (((uintptr_t)dcol) & 3) == 0, "t2_location_bwd: bad args");
  T2_DT_CHECK("t2_location_bwd");
  T2_GO(t2_location_bwd_kernel, t2_grid((long long)B * Ti), 0, (const unsigned short*)dcol, d_prev, d_cum, B, Ti, KL);
  return 0;
}

extern "C" int dle_t2_mask_rows(void* x, int64_t ld, int cols, const int64_t* lengths, int64_t B, int To, float value, int dtype,
                                hipStream_t stream) {
  DLE_CHECK_ARG(x && lengths && B > 0 && To > 0 && cols > 0 && ld >= cols, "t2_mask_rows: bad args");
  DLE_CHECK_ARG(dtype == DLE_F32 || dtype == DLE_F16 || dtype == DLE_BF16, "t2_mask_rows: unknown dtype %d", dtype);
  const long long R = (long long)B * To;
  const int G = t2_grid(R * cols, 1024);
  if (dtype == DLE_F32)
    hipLaunchKernelGGL(t2_mask_rows_kernel<DLE_F32>, dim3(G), dim3(T2_BLOCK), 0, stream, x, (long long)ld, cols,
                       (const long long*)lengths, R, To, value);
  else if (dtype == DLE_F16)
    hipLaunchKernelGGL(t2_mask_rows_kernel<DLE_F16>, dim3(G), dim3(T2_BLOCK), 0, stream, x, (long long)ld, cols,
                       (const long long*)lengths, R, To, value);
  else
    hipLaunchKernelGGL(t2_mask_rows_kernel<DLE_BF16>, dim3(G), dim3(T2_BLOCK), 0, stream, x, (long long)ld, cols,
                       (const long long*)lengths, R, To, value);
  DLE_LAUNCH_CHECK();
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
