"""Time dle_gemm on a list of shapes (forward layout), HIP events.
    python tools/probes/gemm_shapes.py [MxNxK ...]      (DLE_GEMM_BIG=0/1, DLE_GEMM_GM=g select the tile / walk order)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deeplearningexamples_amd import functional as F
dev = torch.device("cuda", 0)
def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
from deeplearningexamples_amd import _cabi as C
# MxNxK[:flags]  flags: d = B stored [K][N] (data-gradient layout), s = + 16-bit addend (ACT_ADD), m = + addend under a 1-bit mask
arg_specs = [(a.split(":") + [""])[:2] for a in sys.argv[1:]]
arg_shapes = [tuple(int(v) for v in a.split("x")) + (f,) for a, f in arg_specs]
shapes = arg_shapes or [(802816, 256, 64), (802816, 256, 128), (802816, 256, 256), (802816, 64, 256), (802816, 128, 64), (200704, 512, 128),
          (50176, 1024, 256), (16384, 4096, 1024), (16384, 1024, 4096), (65536, 1024, 1024)]
for spec in shapes:
    m, n, k = spec[:3]
    fl = spec[3] if len(spec) > 3 else ""
    a = torch.randn(m, k, device=dev).bfloat16()
    b = (torch.randn(k, n, device=dev) if "d" in fl else torch.randn(n, k, device=dev)).bfloat16()
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    src = torch.randn(m, n, device=dev).bfloat16() if ("s" in fl or "m" in fl) else None
    bits = torch.randint(0, 255, (m * n // 8,), device=dev, dtype=torch.uint8) if "m" in fl else None
    act = C.ACT_ADD_MASKED if "m" in fl else C.ACT_ADD if "s" in fl else C.ACT_NONE
    t = timeit(lambda: F.gemm(a, b, m, n, k, True, "d" not in fl, out=out, act=act, mask_src=src, aux=bits))
    byts = (m * k + n * k + m * n) * 2 + (m * n * 2 if src is not None else 0) + (m * n // 8 if bits is not None else 0)
    print("%-3s %8dx%5dx%5d  %8.1f us  %7.1f TF  %6.2f TB/s" % (fl or "fwd", m, n, k, t * 1e6, 2.0 * m * n * k / t / 1e12, byts / t / 1e12), flush=True)
if arg_shapes:
    sys.exit(0)
x = torch.empty(802816 * 256, device=dev, dtype=torch.bfloat16)
y = torch.empty_like(x)
t = timeit(lambda: y.copy_(x))
print("torch copy 411MB->411MB %.1f us  %.2f TB/s" % (t * 1e6, 2 * x.numel() * 2 / t / 1e12))
t = timeit(lambda: y.zero_())
print("torch fill 411MB %.1f us  %.2f TB/s" % (t * 1e6, x.numel() * 2 / t / 1e12))
