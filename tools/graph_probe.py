"""Probe: eager launch vs HIP-graph replay of one train step (host launch overhead)."""
import argparse, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="dlrm")
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
args = argparse.Namespace(batch=None, dtype=None, max_table_size=None)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
wl = bench.WORKLOADS[a.workload](args, 0, 1, dev)
for _ in range(5):
    wl.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    wl.step()
torch.cuda.synchronize()
print("eager  ms/step %.3f" % ((time.perf_counter() - t0) / a.steps * 1e3))
# host-only cost: time to ENQUEUE the steps (no sync)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    wl.step()
enq = time.perf_counter() - t0
torch.cuda.synchronize()
print("enqueue ms/step %.3f" % (enq / a.steps * 1e3))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        wl.step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        wl.step()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        g.replay()
    torch.cuda.synchronize()
    print("graph  ms/step %.3f  loss %s" % ((time.perf_counter() - t0) / a.steps * 1e3, float(wl.loss)))
except Exception as e:
    print("graph capture failed:", repr(e)[:2000])
