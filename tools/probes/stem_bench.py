"""Stem kernels (csrc/stem.hip) against the generic implicit-GEMM path at the ResNet-50 batch-256 shape: us per launch."""
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
    n = int(os.environ.get("N", "256"))
    for dt in (torch.bfloat16, torch.float16):
        img = torch.randn((n, 3, 224, 224), device=dev)
        wm = (torch.randn((64, 3, 7, 7), device=dev) * 0.1).contiguous(memory_format=torch.channels_last)
        x4, x8 = F.nchw_to_nhwc(img, dt, 4), F.nchw_to_nhwc(img, dt, 8)
        w2 = F.stem_pack_weight(wm, dt)
        w8 = torch.zeros((64, 7, 7, 8), dtype=dt, device=dev)
        w8[..., :3] = wm.permute(0, 2, 3, 1).to(dt)
        dy = torch.randn((n, 112, 112, 64), device=dev).to(dt)
        g4 = torch.empty(64 * 147, dtype=torch.float32, device=dev)
        g8 = torch.empty((64, 7, 7, 8), dtype=torch.float32, device=dev)
        t = {"nhwc4": timeit(lambda: F.nchw_to_nhwc(img, dt, 4)), "nhwc8": timeit(lambda: F.nchw_to_nhwc(img, dt, 8)),
             "stem fwd+stats": timeit(lambda: F.stem_conv_fwd_bnstats(x4, w2)),
             "generic fwd+stats": timeit(lambda: F.conv2d_fwd_bnstats(x8, w8, 2, 3)),
             "stem wgrad": timeit(lambda: F.stem_conv_wgrad(dy, x4, g4)),
             "generic wgrad": timeit(lambda: F.conv2d_wgrad(dy, x8, (7, 7), 2, 3, out=g8))}
        byt = (x4.numel() + dy.numel()) * 2
        print(dt, " ".join("%s %.1f us" % kv for kv in t.items()),
              "| stem fwd %.2f TB/s, wgrad %.2f TB/s" % (byt / t["stem fwd+stats"] / 1e6, byt / t["stem wgrad"] / 1e6), flush=True)


if __name__ == "__main__":
    main()
