"""Split-K choice for the weight-gradient GEMMs (C[Mo, No] fp32 = G[K, Mo]^T X[K, No], both operands m-contiguous) of the DLRM
MLPs and the ResNet-50 1x1 convolutions: time every split from 1 to 256 next to functional.pick_splitk's choice.
Run once as is and once with DLE_GEMM_BIG=0 (pins the 128x128 tile) to separate the tile choice from the split choice:
    python tools/probes/splitk_policy.py > gpurun_out/splitk_policy_auto.txt
    DLE_GEMM_BIG=0 python tools/probes/splitk_policy.py > gpurun_out/splitk_policy_small_tile.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deeplearningexamples_amd import functional as F        # noqa: E402

SHAPES = [  # DLRM (batch 65536)
    (256, 512, 65536), (512, 1024, 65536), (1024, 480, 65536), (1024, 1024, 65536), (128, 256, 65536), (512, 16, 65536),
    # ResNet-50 (batch 256)
    (256, 64, 802816), (64, 256, 802816), (128, 256, 802816), (512, 128, 200704), (128, 512, 200704), (1024, 256, 50176),
    (256, 1024, 50176), (2048, 512, 12544), (512, 2048, 12544), (64, 64, 802816)]


def bench(mo, no, k, s, dtype=torch.bfloat16, iters=20):
    dev = torch.device("cuda", 0)
    gs = [torch.randn(k, mo, device=dev, dtype=dtype) * 0.1 for _ in range(2)]
    xs = [torch.randn(k, no, device=dev, dtype=dtype) * 0.1 for _ in range(2)]
    out = torch.empty(mo, no, device=dev, dtype=torch.float32)
    run = lambda i: F.gemm(gs[i & 1], xs[i & 1], mo, no, k, False, False, out=out, splitk=s)
    for i in range(4):
        run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    ref = gs[(iters - 1) & 1][:65536].float().t() @ xs[(iters - 1) & 1][:65536].float() if k <= 65536 else None
    if ref is not None:
        err = float((out - ref).abs().max() / ref.abs().max())
        assert err < 2e-2, (mo, no, k, s, err)
    return e0.elapsed_time(e1) * 1000.0 / iters


def main():
    print("tile pin: DLE_GEMM_BIG=%s" % os.environ.get("DLE_GEMM_BIG", "(auto)"))
    splits = [1, 2, 4, 8, 16, 32, 64, 128, 256]
    print("%-22s %8s | %s" % ("Mo x No x K", "policy", " ".join("%7d" % s for s in splits)))
    for mo, no, k in SHAPES:
        pol = F.pick_splitk(mo, no, k)
        pol_rn = F.pick_splitk(mo, no, k, target_blocks=1024)
        t_pol = bench(mo, no, k, pol)
        row = []
        for s in splits:
            row.append(bench(mo, no, k, s) if s <= (k + 63) // 64 else float("nan"))
        best = min(range(len(splits)), key=lambda i: row[i] if row[i] == row[i] else 1e9)
        print("%-22s %4d:%6.1f | %s   best %d (%.1f us)  [target 1024 -> %d]" % (
            "%dx%dx%d" % (mo, no, k), pol, t_pol, " ".join("%7.1f" % t for t in row), splits[best], row[best], pol_rn))


if __name__ == "__main__":
    main()
