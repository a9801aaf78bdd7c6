// Eight-phase GEMM (gemm8_kernel.h): the forward epilogues -- bias (+ ReLU / tanh-GELU / tanh, + side output) on the
// k-contiguous layout of a linear layer's forward pass.
#include "gemm8_kernel.h"

extern "C" int g8_launch_epi1(const Gemm8Args* p, int dt, int am, int bm, int act, int grid, hipStream_t stream) {
  if (am != 0 || bm != 0) return 0;
#define G8_E1(DT) do { switch (act) { \
    case ACT_NONE: g8_launch<DT, 0, 0, 1, ACT_NONE>(*p, grid, stream); break; \
    case ACT_RELU: g8_launch<DT, 0, 0, 1, ACT_RELU>(*p, grid, stream); break; \
    case ACT_RELU_BITS: g8_launch<DT, 0, 0, 1, ACT_RELU_BITS>(*p, grid, stream); break; \
    case ACT_GELU: g8_launch<DT, 0, 0, 1, ACT_GELU>(*p, grid, stream); break; \
    case ACT_GELU_DAUX: g8_launch<DT, 0, 0, 1, ACT_GELU_DAUX>(*p, grid, stream); break; \
    case ACT_TANH: g8_launch<DT, 0, 0, 1, ACT_TANH>(*p, grid, stream); break; \
    default: return 0; } } while (0)
  if (dt == DLE_F16) G8_E1(DLE_F16); else G8_E1(DLE_BF16);
#undef G8_E1
  return 1;
}
