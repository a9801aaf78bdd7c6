"""Run three train steps of one bench workload (target of rocprofv3 --pmc passes: tools/pmc_r04_kernels.sh).
    python tools/replay_step.py rn50|bert|dlrm"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "rn50"
args = argparse.Namespace(batch=None, dtype=None, max_table_size=None)
torch.cuda.set_device(0)
wl = bench.WORKLOADS[name](args, 0, 1, torch.device("cuda", 0))
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
print("done", name)
