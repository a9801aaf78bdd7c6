"""fwd GEMM time vs K at fixed M,N: intercept = prologue + epilogue per tile round, slope = per-K-tile cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeplearningexamples_amd import functional as F
from tools.microbench import timeit
m, n = int(sys.argv[1]), int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "fwd"
dev = torch.device("cuda", 0)
for k in (64, 128, 256, 512, 1024, 2048, 4096):
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    b = torch.randn(n, k, device=dev).to(torch.bfloat16)
    if mode == "fwd":
        t = timeit(lambda: F.gemm(a, b, m, n, k, True, True), iters=20, warmup=3)
    elif mode == "f32":
        t = timeit(lambda: F.gemm(a, b, m, n, k, True, True, out_dtype=torch.float32), iters=20, warmup=3)
    print(mode, m, n, k, "us %.1f" % (t * 1e6), "TF %.0f" % (2.0 * m * n * k / t / 1e12), flush=True)
