"""Run one GEMM flavour a few times (for rocprofv3 --pmc passes):  gemm_one.py fwd|dgrad|wgrad M N K [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeplearningexamples_amd import functional as F

mode, m, n, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device("cuda", 0)
dt = torch.bfloat16
a = torch.randn(m, k, device=dev).to(dt)
b = torch.randn(n, k, device=dev).to(dt)
gy = torch.randn(m, n, device=dev).to(dt)
for _ in range(iters):
    if mode == "fwd":
        F.gemm(a, b, m, n, k, True, True)
    elif mode == "dgrad":
        F.linear_dgrad(gy, b)
    else:
        F.linear_wgrad(gy, a)
torch.cuda.synchronize()
