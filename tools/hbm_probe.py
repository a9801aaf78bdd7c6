import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.microbench import timeit
dev = torch.device("cuda", 0)
for mb in (32, 134, 512, 2048):
    n = mb * 1000 * 1000 // 2
    a = torch.empty(n, dtype=torch.bfloat16, device=dev)
    b = torch.randn(n, device=dev).to(torch.bfloat16)
    t = timeit(lambda: a.fill_(1.0), iters=20)
    t2 = timeit(lambda: a.copy_(b), iters=20)
    t3 = timeit(lambda: b.sum(), iters=20)
    print("MB %d fill %.1f us %.2f TB/s | copy %.1f us %.2f TB/s (r+w) | sum %.1f us %.2f TB/s" % (mb, t * 1e6, n * 2 / t / 1e12, t2 * 1e6, n * 4 / t2 / 1e12, t3 * 1e6, n * 2 / t3 / 1e12))
