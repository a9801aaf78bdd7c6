"""How much of the sparse-update time is the duplicate chains of the 1-2 k-row tables?  Times dle_emb_sgd_dedup_ws on the criteo_f15
cardinalities (batch 65536, dim 128, fp16 gradient inside a [B, 27, 128] tensor) and on the same list with those tables inflated
to 50 k rows (their lookups stay, their chains go).   python tools/probes/emb_chain_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from deeplearningexamples_amd import functional as F
import bench

dev = torch.device("cuda", 0)
B, D = 65536, 128


def run(sizes, label):
    T = len(sizes)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    g = torch.Generator().manual_seed(1)
    idx = torch.stack([torch.randint(0, int(s), (B,), generator=g) for s in sizes], 1)
    rows = (idx + torch.from_numpy(off[:-1])).to(dev)
    w = torch.zeros(int(off[-1]), D, device=dev)
    grad = (torch.randn(B, T + 1, D, generator=g) * 0.01).half().to(dev)
    ws = F.EmbUpdateWorkspace(off, D, dev)
    lr = torch.tensor(0.1, device=dev)

    def step():
        F.emb_sgd_dedup_(w, rows, grad[:, 1:, :], ws, lr, grad_batch_stride=(T + 1) * D)
    for _ in range(3):
        step()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(20):
        step()
    e.record(); torch.cuda.synchronize()
    print("%-60s %7.1f us" % (label, s.elapsed_time(e) / 20 * 1e3), flush=True)


crit = list(bench.CRITEO_F15)
run(crit, "criteo_f15")
run([50000 if 128 < s <= 4096 else s for s in crit], "968 / 1382 / 2209-row tables inflated to 50 k rows")
run([50000 if 128 < s <= 8192 else s for s in crit], "... and the 7105 / 7339-row tables")
run([s for s in crit if s > 128], "criteo_f15 without the 8 tiny tables")
