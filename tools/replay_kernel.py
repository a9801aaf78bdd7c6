"""Replay one recorded C-ABI launch of a bench workload many times (target of rocprofv3 --pmc passes).

    python tools/replay_kernel.py --workload rn50 [--key "dle_gemm[802816x256x64]"] [--iters 50]
Without --key the heaviest launch of the dominant kernel family (bench.py's roofline.heaviest_shape) is replayed.  Prints the key."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from deeplearningexamples_amd import _cabi

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="rn50")
ap.add_argument("--key", default=None)
ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
args = argparse.Namespace(batch=None, dtype=None, max_table_size=None)
torch.cuda.set_device(0)
wl = bench.WORKLOADS[a.workload](args, 0, 1, torch.device("cuda", 0))
for _ in range(2):
    wl.step()
timer = _cabi.KernelTimer()
_cabi.set_timer(timer)
wl.step()
torch.cuda.synchronize()
_cabi.set_timer(None)
key = a.key
if key is None:
    spp = {"rn50": 256, "bert": 256, "dlrm": 65536}[a.workload]
    roof, _ = bench.roofline_from(timer, 1, a.workload, spp, 1.0)
    key = roof["heaviest_shape"]
name, tag = (key[:key.index("[")], key[key.index("[") + 1:-1]) if "[" in key else (key, None)
fn, cargs, st = timer.last[(name, tag)]
with torch.cuda.stream(st):
    for _ in range(a.iters):
        fn(*cargs)
torch.cuda.synchronize()
print(json.dumps({"replayed": key, "iters": a.iters}))
