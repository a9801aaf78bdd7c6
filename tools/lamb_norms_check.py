"""Debug probe: LAMB's per-tensor norms from inside stage 1 (dle_mt_lamb_stage1_norms) against the two l2norm sweeps, on the real
BERT-Large tensors of a bench step.  python tools/lamb_norms_check.py"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from deeplearningexamples_amd import multi_tensor as mt  # noqa: E402

real = mt.lamb_stage1_norms
worst = {"pn": 0.0, "un": 0.0, "calls": 0, "pn_ne": 0, "un_ne": 0}


def probe(table, noop, *args):
    _, pn_ref = mt.l2norm(mt.TensorTable([table._keep[1]]), noop, per_tensor=True)
    pn, un = real(table, noop, *args)
    _, un_ref = mt.l2norm(mt.TensorTable([table._keep[0]]), noop, per_tensor=True)
    worst["calls"] += 1
    worst["pn"] = max(worst["pn"], float(((pn - pn_ref).abs() / pn_ref.clamp_min(1e-30)).max()))
    worst["un"] = max(worst["un"], float(((un - un_ref).abs() / un_ref.clamp_min(1e-30)).max()))
    worst["pn_ne"] += int((pn != pn_ref).sum())
    worst["un_ne"] += int((un != un_ref).sum())
    return pn, un


mt.lamb_stage1_norms = probe
a = argparse.Namespace(batch=None, dtype=None)
w = bench.WORKLOADS["bert"](a, 0, 1, torch.device("cuda", 0)) if hasattr(bench, "WORKLOADS") else None
for _ in range(3):
    w.step()
torch.cuda.synchronize()
print(worst)
