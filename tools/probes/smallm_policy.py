"""Which tile should gemm_smallm.hip pick?  Times dle_gemm on the few-row shapes of the Tacotron2 step with the tile pinned to
64 x 16 (4 wavefronts) and 64 x 32 (8 wavefronts) -- DLE_GEMM_SMALLM_TN, read per call -- and with the built-in policy.
Weights rotate through enough copies (>= 192 MB) that a launch finds them in the Infinity Cache / HBM, not in L2, as inside a
training step.  Usage (GPU box): python tools/probes/smallm_policy.py > gpurun_out/smallm_policy.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("DLE_GEMM_SMALLM_TN", "0")          # must exist before the library reads it the first time
from deeplearningexamples_amd import _cabi as C             # noqa: E402
from deeplearningexamples_amd import functional as F        # noqa: E402

SHAPES = [(128, 1536, 4096, False), (128, 2560, 4096, False), (128, 4096, 2560, False), (128, 4096, 1536, True),
          (128, 4096, 1024, False), (128, 1024, 4096, False), (128, 128, 1024, False), (128, 1024, 128, False),
          (128, 1024, 256, True), (128, 256, 1024, False), (128, 512, 1024, False), (128, 2048, 512, False), (64, 4096, 1536, False),
          (256, 4096, 1536, False), (128, 88, 1536, False)]


def bench(m, n, k, add, tn, nst=0, dtype=torch.float16, iters=200):
    dev = torch.device("cuda", 0)
    copies = max(2, min(64, int(192e6 // (n * k * 2)) + 1))
    ws = [torch.randn(n, k, device=dev, dtype=dtype) * 0.05 for _ in range(copies)]
    x = torch.randn(m, k, device=dev, dtype=dtype)
    src = torch.randn(m, n, device=dev, dtype=dtype) if add else None
    out = torch.empty(m, n, device=dev, dtype=dtype)
    os.environ["DLE_GEMM_SMALLM_TN"] = str(tn)
    os.environ["DLE_GEMM_SMALLM_NST"] = str(nst)
    run = lambda i: F.gemm(x, ws[i % copies], m, n, k, True, True, out=out, act=C.ACT_ADD if add else C.ACT_NONE, mask_src=src)
    for i in range(20):
        run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()                                # launch-overhead free: the loop is captured once
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for i in range(iters):
                run(i)
    g.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    ref = (x.float() @ ws[(iters - 1) % copies].float().t() + (src.float() if add else 0)).to(dtype)
    err = float((out.float() - ref.float()).abs().max() / ref.float().abs().max())
    return e0.elapsed_time(e1) * 1000.0 / (3 * iters), err


def main():
    variants = [(0, 0), (16, 4), (16, 7), (32, 3), (32, 4), (32, 6)]
    print("us per launch, weights from beyond L2; columns: tile columns / ring stages (0/0 = the built-in policy)")
    print("%-24s " % "M x N x K" + " ".join("%8s" % ("%d/%d" % v) for v in variants))
    for m, n, k, add in SHAPES:
        t = [bench(m, n, k, add, tn, nst) for tn, nst in variants]
        assert all(e < 2e-2 for _, e in t), t
        print("%-24s " % ("%dx%dx%d%s" % (m, n, k, "+src" if add else "")) + " ".join("%8.2f" % x for x, _ in t))


if __name__ == "__main__":
    main()
