"""3x3 weight gradients of the batch-256 ResNet-50 step: halo-tile kernel (csrc/conv3x3_wgrad.hip) vs the split-K implicit GEMM."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deeplearningexamples_amd import functional as F
from deeplearningexamples_amd import _cabi as C


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    dev = torch.device("cuda", 0)
    n = 256
    for (h, c) in ((56, 64), (28, 128), (14, 256), (7, 512)):
        x = torch.randn((n, h, h, c), device=dev).bfloat16()
        dy = torch.randn((n, h, h, c), device=dev).bfloat16()
        out = torch.empty((c, 3, 3, c), dtype=torch.float32, device=dev)
        res = {}
        for mode in (1, 0):
            C.lib().dle_conv3x3_wgrad_mode(mode)
            res[mode] = timeit(lambda: F.conv2d_wgrad(dy, x, (3, 3), 1, 1, out=out))
        C.lib().dle_conv3x3_wgrad_mode(-1)
        fl = 2.0 * n * h * h * c * c * 9
        print("wgrad 256x%dx%dx%d k%d: halo %.1f us (%.0f TFLOP/s)   gemm %.1f us (%.0f TFLOP/s)" %
              (h, h, c, c, res[1], fl / res[1] / 1e6, res[0], fl / res[0] / 1e6), flush=True)


if __name__ == "__main__":
    main()
