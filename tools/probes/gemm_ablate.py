import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    import torch
    from deeplearningexamples_amd import functional as F
    dev = torch.device("cuda", 0)
    def timeit(fn, iters=10, warmup=3):
        for _ in range(warmup): fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(iters): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / iters * 1e3
    for m, n, k in [(802816, 256, 64), (802816, 256, 256), (65536, 1024, 1024)]:
        a = torch.randn(m, k, device=dev).bfloat16(); b = torch.randn(n, k, device=dev).bfloat16()
        out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        print("   %dx%dx%d: %.1f us" % (m, n, k, timeit(lambda: F.gemm(a, b, m, n, k, True, True, out=out))), flush=True)
else:
    for v2 in ("1", "0"):
        for skip in ("0", "1", "2", "4", "3", "6", "7"):
            env = dict(os.environ, DLE_GEMM_V2=v2, DLE_GEMM_SKIP=skip)
            print("V2=%s skip=%s (1 no stores, 2 no DMA, 4 no MFMA)" % (v2, skip), flush=True)
            subprocess.run([sys.executable, __file__, "child"], env=env)
            if v2 == "0": break
