"""Split-K sweep of the weight-gradient layout (tn) on the ping-pong kernel (gemm8): us per call incl. the slab reduce.
    python tools/gemm8_splitk_sweep.py"""
import json, os, sys
os.environ.setdefault("DLE_GEMM_8PH_MIN_ITEMS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeplearningexamples_amd import functional as F, _cabi as C
dev = torch.device("cuda", 0)
lib = C.lib()

def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

for (m, n, k, dt) in ((256, 512, 65536, torch.float16), (512, 1024, 65536, torch.float16), (1024, 1024, 65536, torch.float16),
                      (1024, 1024, 32768, torch.bfloat16), (3072, 1024, 32768, torch.bfloat16), (4096, 1024, 32768, torch.bfloat16),
                      (1024, 256, 50176, torch.bfloat16), (2048, 512, 12544, torch.bfloat16)):
    a = torch.randn(k, m, device=dev).to(dt); b = torch.randn(k, n, device=dev).to(dt)
    out = torch.empty(m, n, dtype=torch.float32, device=dev)
    row = {"mnk": [m, n, k], "default_splitk": F.pick_splitk(m, n, k, 1024)}
    for sk in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        if sk > k // 128: continue
        for mode, key in ((0, "old"), (1, "new")):
            lib.dle_gemm8_mode(mode)
            try:
                t = timeit(lambda: F.gemm(a, b, m, n, k, False, False, out=out, splitk=sk))
            except Exception as ex:
                t = None
            row["%s_sk%d" % (key, sk)] = round(t, 1) if t else None
    print(json.dumps(row), flush=True)
lib.dle_gemm8_mode(1)
