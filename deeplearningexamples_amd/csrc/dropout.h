// Counter-based dropout shared by the BERT kernels (transformer.hip, attention.hip).
#pragma once
#include "common.h"

// ------------------------------------------------------------------ dropout (nn.Dropout, modeling.py:276,320,392,428)
// Counter-based RNG: Philox4x32-10 keyed by the 64-bit seed, counter = (16-byte chunk index of the tensor, 64-bit
// call offset).  One call yields the 8 keep decisions of one 8-element chunk: a 16-bit uniform per element is
// compared with thr = round(p * 65536) (so the drop probability is quantised to 1/65536: p = 0.1 -> 0.100006).
// The mask is stored bit-packed, bit k of byte i <-> element 8 i + k, 1 = kept; kept values are scaled by
// 1 / (1 - thr / 65536).  Torch draws from its own Philox stream, so masks differ from the reference's bit for bit;
// the step is checked against the oracle with THESE masks (tests/test_gpu_bert_step.py).
struct DropArgs {
  unsigned char* mask;     // NULL: no dropout
  unsigned thr;
  float inv_keep;
  unsigned seed_lo, seed_hi, off_lo, off_hi;
  const unsigned long long* off_dev;   // optional DEVICE word added to the call offset (graph-safe RNG advance)
};

__device__ __forceinline__ uint4_t philox4x32_10(uint4_t c, unsigned k0, unsigned k1) {
  const unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0], hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    c = (uint4_t){hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0};
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

// keep bits of the 8 elements of chunk `chunk`
__device__ __forceinline__ unsigned drop_bits(const DropArgs& d, long long chunk) {
  const uint4_t r = philox4x32_10((uint4_t){(unsigned)chunk, (unsigned)((unsigned long long)chunk >> 32), d.off_lo, d.off_hi},
                                  d.seed_lo, d.seed_hi);
  unsigned bits = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    bits |= ((r[k] & 0xffffu) >= d.thr ? 1u : 0u) << (2 * k);
    bits |= ((r[k] >> 16) >= d.thr ? 1u : 0u) << (2 * k + 1);
  }
  return bits;
}

// The call offset a captured HIP graph would freeze in its kernel arguments: with off_dev the kernels add a counter that
// lives in device memory (the captured step advances it itself), the way torch's graph-safe Philox state does.
__device__ __forceinline__ void drop_resolve(DropArgs& d) {
  if (d.off_dev) {
    const unsigned long long o = (((unsigned long long)d.off_hi << 32) | d.off_lo) + *d.off_dev;
    d.off_lo = (unsigned)o;
    d.off_hi = (unsigned)(o >> 32);
  }
}

static inline DropArgs make_drop(void* mask, float p, unsigned long long seed, unsigned long long offset,
                                 const void* offset_base = nullptr) {
  DropArgs d;
  d.mask = (unsigned char*)mask;
  long long thr = (long long)(p * 65536.0f + 0.5f);
  if (thr < 0) thr = 0;
  if (thr > 65535) thr = 65535;
  d.thr = (unsigned)thr;
  d.inv_keep = 65536.0f / (float)(65536 - thr);
  d.seed_lo = (unsigned)seed; d.seed_hi = (unsigned)(seed >> 32);
  d.off_lo = (unsigned)offset; d.off_hi = (unsigned)(offset >> 32);
  d.off_dev = (const unsigned long long*)offset_base;
  return d;
}

