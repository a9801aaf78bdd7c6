"""Per-kernel microbenchmarks on one MI355X (HIP-event timed): achieved GB/s / TFLOP/s vs roofline.

    python tools/microbench.py [dot] [emb] [gemm] [mt]     -> JSON lines on stdout
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeplearningexamples_amd import functional as F, _cabi as C, multi_tensor as mt  # noqa: E402

HBM_PEAK, MFMA_PEAK = 8000.0, 2500.0   # GB/s, TFLOP/s (MI355X_MICROARCH.md)
dev = torch.device("cuda", 0)


def timeit(fn, iters=20, warmup=5):
    for _ in range(warmup):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def report(name, sec, bytes_=None, flops=None, **kw):
    d = {"kernel": name, "us": round(sec * 1e6, 2)}
    if bytes_ is not None:
        d["GBps"] = round(bytes_ / sec / 1e9, 1)
        d["hbm_frac"] = round(bytes_ / sec / 1e9 / HBM_PEAK, 3)
    if flops is not None:
        d["TFLOPs"] = round(flops / sec / 1e12, 1)
        d["mfma_frac"] = round(flops / sec / 1e12 / MFMA_PEAK, 3)
    d.update(kw)
    print(json.dumps(d), flush=True)


def bench_dot():
    for b in (8192, 65536):
        x = torch.rand(b, 27, 128, device=dev).half()
        ug = torch.rand(b, 480, device=dev).half()
        t = timeit(lambda: F.dot_interact_fwd(x))
        report("dot_fwd_f16", t, bytes_=b * (27 * 128 + 480) * 2, batch=b)
        t = timeit(lambda: F.dot_interact_bwd(x, ug))
        report("dot_bwd_f16", t, bytes_=b * (2 * 27 * 128 + 480 + 128) * 2, batch=b)


def bench_emb():
    rows, d = 20_000_000, 128
    w = torch.randn(rows, d, device=dev)
    for b in (8192, 65536):
        idx = torch.randint(0, rows, (b, 26), device=dev)
        g = torch.randn(b, 26, d, device=dev).half()
        for od, nm, eb in ((torch.float16, "f16", 2), (torch.float32, "f32", 4)):
            t = timeit(lambda: F.emb_gather_fwd(w, idx, out_dtype=od))
            report("emb_gather_" + nm, t, bytes_=b * 26 * (d * 4 + d * eb + 8), batch=b)
        t = timeit(lambda: F.emb_sparse_sgd_(w, idx, g, 1e-6))
        report("emb_sparse_sgd_f16grad", t, bytes_=b * 26 * (d * 2 + 2 * d * 4 + 8), batch=b)


def bench_gemm():
    shapes = [(8192, 1024, 1024), (8192, 1024, 480), (65536, 1024, 1024), (32768, 4096, 1024),
              (32768, 1024, 4096), (4096, 4096, 4096), (8192, 8192, 8192)]
    for dtype, nm in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
        for m, n, k in shapes:
            a = torch.randn(m, k, device=dev).to(dtype)
            b = torch.randn(n, k, device=dev).to(dtype)
            t = timeit(lambda: F.gemm(a, b, m, n, k, True, True), iters=10, warmup=3)
            report("gemm_fwd_" + nm, t, flops=2.0 * m * n * k, mnk=[m, n, k])
            if dtype == torch.bfloat16:
                continue
            gy = torch.randn(m, n, device=dev).to(dtype)
            t = timeit(lambda: F.linear_dgrad(gy, b), iters=10, warmup=3)
            report("gemm_dgrad_" + nm, t, flops=2.0 * m * n * k, mnk=[m, n, k])
            t = timeit(lambda: F.linear_wgrad(gy, a), iters=10, warmup=3)
            report("gemm_wgrad_" + nm, t, flops=2.0 * m * n * k, mnk=[m, n, k])
            # library GEMM through torch for orientation only (never the product path)
            t = timeit(lambda: torch.nn.functional.linear(a, b), iters=10, warmup=3)
            report("hipblaslt_ref_fwd_" + nm, t, flops=2.0 * m * n * k, mnk=[m, n, k])


def bench_gemmbert():
    """The BERT-Large / DLRM linear-layer shapes, forward / data-grad / weight-grad, bf16."""
    shapes = [(16384, 4096, 1024), (16384, 1024, 4096), (16384, 3072, 1024), (16384, 1024, 1024),
              (65536, 1024, 1024), (65536, 512, 1024), (8192, 8192, 8192)]
    dtype = torch.bfloat16
    for m, n, k in shapes:
        a = torch.randn(m, k, device=dev).to(dtype)
        b = torch.randn(n, k, device=dev).to(dtype)
        gy = torch.randn(m, n, device=dev).to(dtype)
        t = timeit(lambda: F.gemm(a, b, m, n, k, True, True), iters=10, warmup=3)
        report("gemm_fwd_bf16", t, flops=2.0 * m * n * k, mnk=[m, n, k])
        t = timeit(lambda: F.linear_dgrad(gy, b), iters=10, warmup=3)
        report("gemm_dgrad_bf16", t, flops=2.0 * m * n * k, mnk=[m, n, k])
        t = timeit(lambda: F.linear_wgrad(gy, a), iters=10, warmup=3)
        report("gemm_wgrad_bf16", t, flops=2.0 * m * n * k, mnk=[m, n, k])
        t = timeit(lambda: torch.nn.functional.linear(a, b), iters=10, warmup=3)
        report("hipblaslt_ref_fwd_bf16", t, flops=2.0 * m * n * k, mnk=[m, n, k])


def bench_mt():
    n = 336_000_000 // 4
    sizes = [n // 64] * 64
    g = [torch.randn(s, device=dev) for s in sizes]
    p = [torch.randn(s, device=dev) for s in sizes]
    m = [torch.zeros(s, device=dev) for s in sizes]
    v = [torch.zeros(s, device=dev) for s in sizes]
    noop = torch.zeros(1, dtype=torch.int32, device=dev)
    one = torch.ones(1, device=dev)
    step = torch.ones(1, dtype=torch.int32, device=dev)
    tg, t4, t2 = mt.TensorTable([g]), mt.TensorTable([g, p, m, v]), mt.TensorTable([g, p])
    tot = sum(sizes)
    t = timeit(lambda: mt.l2norm(tg, noop, True))
    report("mt_l2norm_f32", t, bytes_=tot * 4)
    t = timeit(lambda: mt.lamb_stage1(t4, noop, 0.9, 0.999, 0.1, step, True, 1e-6, 1, 0.01, one, one, one))
    report("mt_lamb_stage1_f32", t, bytes_=tot * 4 * 7)
    _, pn = mt.l2norm(mt.TensorTable([p]), noop, True)
    t = timeit(lambda: mt.lamb_stage2(t2, noop, pn, pn, one, 0.01, False))
    report("mt_lamb_stage2_f32", t, bytes_=tot * 4 * 3)
    t3 = mt.TensorTable([g, p, m])
    t = timeit(lambda: mt.sgd(t3, 0.1, 0.9, 0.0, 1e-4, False))
    report("mt_sgd_momentum_f32", t, bytes_=tot * 4 * 5)


if __name__ == "__main__":
    which = sys.argv[1:] or ["dot", "emb", "gemm", "mt"]
    for w in which:
        globals()["bench_" + w]()
