// Eight-phase GEMM (gemm8_kernel.h): the data-gradient epilogues that read a source tensor -- ReLU mask, addend, stored
// activation derivative (+ the column sums = the bias gradient of the layer below) -- on the layout of a linear layer's data
// gradient with the weight as stored (B row-contiguous) or pre-transposed (B k-contiguous).
#include "gemm8_kernel.h"

extern "C" int g8_launch_epi2(const Gemm8Args* p, int dt, int am, int bm, int act, int grid, hipStream_t stream) {
  if (am != 0) return 0;
#define G8_E2(DT, BM_) do { switch (act) { \
    case ACT_RELU_BWD: g8_launch<DT, 0, BM_, 2, ACT_RELU_BWD>(*p, grid, stream); break; \
    case ACT_ADD: g8_launch<DT, 0, BM_, 2, ACT_ADD>(*p, grid, stream); break; \
    case ACT_ADD_MASKED: if (BM_ != 1) return 0; g8_launch<DT, 0, 1, 2, ACT_ADD_MASKED>(*p, grid, stream); break; \
    case ACT_RELU_BWD_BITS: if (BM_ != 1) return 0; g8_launch<DT, 0, 1, 2, ACT_RELU_BWD_BITS>(*p, grid, stream); break; \
    case ACT_MUL: g8_launch<DT, 0, BM_, 2, ACT_MUL>(*p, grid, stream); break; \
    default: return 0; } } while (0)
  if (dt == DLE_F16) { if (bm == 0) G8_E2(DLE_F16, 0); else G8_E2(DLE_F16, 1); }
  else { if (bm == 0) G8_E2(DLE_BF16, 0); else G8_E2(DLE_BF16, 1); }
#undef G8_E2
  return 1;
}
