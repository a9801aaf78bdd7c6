"""dot-interaction backward at the DLRM batch (65536 x 27 x 128, fp16): us per launch for a few grid sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deeplearningexamples_amd import functional as F


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


dev = torch.device("cuda", 0)
b, r, c = 65536, 27, 128
x = torch.randn((b, r, c), device=dev).half()
ow = (r * (r - 1) // 2 + c + 7) // 8 * 8
up = torch.randn((b, ow), device=dev).half()
t = timeit(lambda: F.dot_interact_bwd(x, up))
print("dot_interact_bwd %dx%dx%d: %.1f us (%.2f TB/s)" % (b, r, c, t, (2 * x.numel() * 2 + up.numel() * 2) / t / 1e6))
