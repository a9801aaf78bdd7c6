import os, sys
os.environ["DLE_GEMM_8PH_MIN_ITEMS"] = "1"
sys.path.insert(0, "/root/repo")
import torch
from deeplearningexamples_amd import functional as F, _cabi as C
lib = C.lib()
dev = torch.device("cuda", 0)
m, n, k = 512, 512, 1024
a = torch.randn(k, m, device=dev).bfloat16(); b = torch.randn(k, n, device=dev).bfloat16()
outs = []
for mode in (0, 1):
    lib.dle_gemm8_mode(mode)
    o = torch.empty(m, n, dtype=torch.float32, device=dev)
    F.gemm(a, b, m, n, k, False, False, out=o, splitk=2)
    torch.cuda.synchronize(); outs.append(o)
bad = (outs[0] != outs[1])
print("bad frac", bad.float().mean().item())
blk = bad[:32, :32].int()
print("bad rows of first block:", blk.sum(1).tolist())
print("bad cols of first block:", blk.sum(0).tolist())
ref = a.float().t() @ b.float()
print("err old", (outs[0]-ref).abs().max().item(), "new", (outs[1]-ref).abs().max().item())
# where does new row r come from?
o1 = outs[1][:32, :32]; o0 = outs[0][:64, :64]
for r in range(16, 32):
    src = [(rr) for rr in range(64) if torch.equal(o0[rr, :4], o1[r, :4])]
    print(r, src)
