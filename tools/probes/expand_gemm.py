"""The streaming "expand" GEMM (csrc/gemm_expand.hip) against the tile kernels of gemm_dma.hip on the channel-widening 1x1
convolution shapes of the ResNet-50 step (batch 256): same dle_gemm call, DLE_GEMM_EXPAND = 0 / 1 (read per call).
    python tools/probes/expand_gemm.py > gpurun_out/expand_gemm.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["DLE_GEMM_EXPAND"] = "1"
os.environ["DLE_EXPAND_STATS_KMIN"] = "32"
from deeplearningexamples_amd import _cabi as C             # noqa: E402
from deeplearningexamples_amd import functional as F        # noqa: E402

SHAPES = [  # (M, N, K, epilogue, weights k-contiguous)
    (802816, 256, 64, "masked", False), (200704, 512, 128, "masked", False), (50176, 1024, 256, "masked", False),
    (802816, 256, 128, "add", False), (200704, 512, 256, "add", False),
    (802816, 256, 64, "none", True), (200704, 512, 128, "none", True), (50176, 1024, 256, "none", True), (802816, 128, 64, "none", True),
    (65536, 512, 256, "none", True), (65536, 1024, 256, "add", True), (10000, 1024, 256, "none", True), (70001, 256, 64, "masked", False)]


def run(m, n, k, epi, b_kc, dtype, expand, iters=20):
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(m + n + k)
    a = (torch.randn(m, k, generator=g) * 0.5).to(dtype).to(dev)
    w = (torch.randn((n, k) if b_kc else (k, n), generator=g) * 0.1).to(dtype).to(dev)
    gd = torch.Generator(device=dev).manual_seed(m + 3 * n + 7 * k)
    src = torch.randn(m, n, device=dev, dtype=dtype, generator=gd) if epi != "none" else None
    bits = torch.randint(0, 256, (m * n // 8,), device=dev, dtype=torch.uint8, generator=gd) if epi == "masked" else None
    out = torch.empty(m, n, device=dev, dtype=dtype)
    act = {"none": C.ACT_NONE, "add": C.ACT_ADD, "masked": C.ACT_ADD_MASKED}[epi]
    os.environ["DLE_GEMM_EXPAND"] = "1" if expand else "0"
    call = lambda: F.gemm(a, w, m, n, k, True, b_kc, out=out, act=act, mask_src=src, aux=bits)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000.0 / iters
    nbytes = 2 * (m * k + n * k + m * n * (2 if src is not None else 1)) + (m * n // 8 if bits is not None else 0)
    # fp32 reference on a slice of rows
    rows = slice(0, 4096)
    ref = a[rows].float() @ (w.float().t() if b_kc else w.float())
    if epi == "add":
        ref = ref + src[rows].float()
    elif epi == "masked":
        keep = ((bits[:4096 * n // 8].to(torch.int32).unsqueeze(1) >> torch.arange(8, device=dev, dtype=torch.int32)) & 1).reshape(4096, n)
        ref = ref + src[rows].float() * keep
    err = float((out[rows].float() - ref).abs().max() / ref.abs().max())
    tail = out[-3:].float().clone()
    return us, nbytes / us / 1e6, err, out.clone(), tail


def conv_stats(nb, hw, c, ko, dtype, expand, iters=20):
    """1x1 convolution + BatchNorm statistics (dle_conv2d_fwd_colstats + dle_bn_stats_from_partials) with / without the expand path."""
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(nb + hw + c + ko)
    x = (torch.randn(nb, hw, hw, c, generator=g) * 0.5).to(dtype).to(dev)
    w = (torch.randn(ko, 1, 1, c, generator=g) * 0.1).to(dtype).to(dev)
    os.environ["DLE_GEMM_EXPAND"] = "1" if expand else "0"
    os.environ["DLE_EXPAND_STATS_KMIN"] = "32"
    call = lambda: F.conv2d_fwd_bnstats(x, w, 1, 0)
    for _ in range(3):
        y, mean, rstd = call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        y, mean, rstd = call()
    e1.record()
    torch.cuda.synchronize()
    yf = y.float().view(-1, ko)
    ref_mean, ref_var = yf.mean(0), yf.var(0, unbiased=False)
    em = float((mean - ref_mean).abs().max() / ref_mean.abs().max())
    er = float((rstd - torch.rsqrt(ref_var + 1e-5)).abs().max() / torch.rsqrt(ref_var + 1e-5).abs().max())
    return e0.elapsed_time(e1) * 1000.0 / iters, y.clone(), em, er


def main():
    print("1x1 convolution + BatchNorm statistics (two launches): us per call, tile kernel vs expand kernel; statistics vs torch on the stored output")
    for dtype in (torch.bfloat16, torch.float16):
        for nb, hw, c, ko in ((256, 56, 64, 256), (256, 28, 128, 512), (256, 14, 256, 1024), (256, 56, 128, 256), (256, 56, 64, 128)):
            t0, y0, em0, er0 = conv_stats(nb, hw, c, ko, dtype, False)
            t1, y1, em1, er1 = conv_stats(nb, hw, c, ko, dtype, True)
            same = bool(torch.equal(y0, y1))
            print("%-30s %9.1f %9.1f   x %.2f   output identical: %s   mean / rstd error: tile %.1e %.1e expand %.1e %.1e" % (
                "%dx%dx%dx%d k%d %s" % (nb, hw, hw, c, ko, str(dtype)[6:]), t0, t1, t0 / t1, same, em0, er0, em1, er1))
            assert em1 < 1e-3 and er1 < 1e-3
            del y0, y1
            torch.cuda.empty_cache()
    gemm_table()


def gemm_table():
    print("%-34s %10s %8s %10s %8s %8s   max |new - old| / max |old|" % ("M x N x K epilogue", "tile us", "TB/s", "expand us", "TB/s", "x"))
    for dtype in (torch.bfloat16, torch.float16):
        for m, n, k, epi, b_kc in SHAPES:
            t0, bw0, e0, o0, _ = run(m, n, k, epi, b_kc, dtype, False)
            t1, bw1, e1, o1, _ = run(m, n, k, epi, b_kc, dtype, True)
            diff = float((o1.float() - o0.float()).abs().max() / o0.float().abs().max())
            tol = 8e-3 if dtype == torch.bfloat16 else 1e-3
            flag = "" if (e1 < 2 * tol + 1e-3 and diff < 2 * tol) else "   <-- MISMATCH (err vs fp32: old %.2e new %.2e)" % (e0, e1)
            print("%-34s %10.1f %8.2f %10.1f %8.2f %8.2f   %.2e%s" % ("%dx%dx%d %s %s %s" % (m, n, k, epi, "kc" if b_kc else "nc", str(dtype)[6:]),
                                                                     t0, bw0, t1, bw1, t0 / t1, diff, flag))
            del o0, o1
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
