"""K sweep of the ping-pong GEMM (gemm8) vs the tile kernel: time = a + b * ktiles per layout (us per K tile = slope).
    python tools/gemm8_ksweep.py [M N]"""
import json, os, sys
os.environ.setdefault("DLE_GEMM_8PH_MIN_ITEMS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeplearningexamples_amd import functional as F, _cabi as C
dev = torch.device("cuda", 0)
lib = C.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096

def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

for layout in ("nt", "nn", "tn"):
    rows = []
    for K in (512, 1024, 2048, 4096, 8192):
        if layout == "nt":
            a = torch.randn(M, K, device=dev).bfloat16(); b = torch.randn(N, K, device=dev).bfloat16(); kc = (True, True)
        elif layout == "nn":
            a = torch.randn(M, K, device=dev).bfloat16(); b = torch.randn(K, N, device=dev).bfloat16(); kc = (True, False)
        else:
            a = torch.randn(K, M, device=dev).bfloat16(); b = torch.randn(K, N, device=dev).bfloat16(); kc = (False, False)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        r = {"layout": layout, "mnk": [M, N, K]}
        for mode, key in ((0, "old"), (1, "new")):
            lib.dle_gemm8_mode(mode)
            t = timeit(lambda: F.gemm(a, b, M, N, K, kc[0], kc[1], out=out))
            r["us_" + key] = round(t, 1); r["tf_" + key] = round(2.0 * M * N * K / t / 1e6, 1)
        rows.append(r)
        print(json.dumps(r), flush=True)
    for key in ("old", "new"):
        (k0, t0), (k1, t1) = [(x["mnk"][2] // 64, x["us_" + key]) for x in (rows[1], rows[-1])]
        b_ = (t1 - t0) / (k1 - k0)
        tiles_per_cu = ((M + 255) // 256) * ((N + 255) // 256) / 256.0
        print(json.dumps({"layout": layout, "kernel": key, "us_per_ktile_per_round": round(b_ / max(tiles_per_cu, 1), 4),
                          "intercept_us": round(t0 - b_ * k0, 2)}), flush=True)
lib.dle_gemm8_mode(1)
