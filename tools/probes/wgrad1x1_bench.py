"""1x1 weight gradients of the batch-256 ResNet-50 step: streaming kernel (csrc/wgrad1x1.hip) vs the split-K tile GEMM."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deeplearningexamples_amd import functional as F


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
    for (ko, c, m) in ((256, 64, 802816), (64, 256, 802816), (64, 64, 802816), (128, 256, 802816), (512, 128, 200704), (128, 512, 200704)):
        dy = torch.randn((m, ko), device=dev).bfloat16()
        x = torch.randn((m, c), device=dev).bfloat16()
        out = torch.empty((ko, c), dtype=torch.float32, device=dev)
        ts = timeit(lambda: F.wgrad1x1(dy, x, out))
        tg = timeit(lambda: F.gemm(dy, x, ko, c, m, False, False, out=out, splitk=F.pick_splitk(ko, c, m, target_blocks=1024)))
        by = m * (ko + c) * 2.0
        print("wgrad %dx%dx%d: stream %.1f us (%.2f TB/s)   split-K gemm %.1f us (%.2f TB/s)" % (ko, c, m, ts, by / ts / 1e6, tg, by / tg / 1e6), flush=True)


if __name__ == "__main__":
    main()
