"""Run the RN50 forward twice on identical inputs and report the first unit whose output differs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import resnet_oracle as RO
from deeplearningexamples_amd.convnets.resnet import ResNet50
from deeplearningexamples_amd.convnets.engine import ResNetTrainer
from deeplearningexamples_amd import functional as F
dev = torch.device("cuda", 0)
state = RO.seeded_state(5)
model = ResNet50(device=dev)
model.load_state_dict({k: v.clone() for k, v in state.items()}, strict=False)
tr = ResNetTrainer(model, lr=0.0, compute_dtype=torch.float16, static_loss_scale=128.0)
x, y = RO.seeded_batch(77, 8, 64)
x = x.to(dev)
outs = []
for rep in range(3):
    rec = []
    xin = F.nchw_to_nhwc(x, torch.float16, 8)
    rec.append(("nhwc", xin.clone()))
    a0 = tr.stem.forward(xin); rec.append(("stem.t", tr.stem.saved[1].clone())); rec.append(("stem.y", a0.clone()))
    m0, am = F.maxpool_fwd(a0); rec.append(("maxpool", m0.clone()))
    h = m0
    for bi, (u1, u2, u3, ud) in enumerate(tr.blocks):
        res = ud.forward(h) if ud is not None else h
        if ud is not None: rec.append(("b%d.down.t" % bi, ud.saved[1].clone())); rec.append(("b%d.down.y" % bi, res.clone()))
        o1 = u1.forward(h); rec.append(("b%d.c1.t" % bi, u1.saved[1].clone())); rec.append(("b%d.c1.y" % bi, o1.clone()))
        o2 = u2.forward(o1); rec.append(("b%d.c2.t" % bi, u2.saved[1].clone())); rec.append(("b%d.c2.y" % bi, o2.clone()))
        h = u3.forward(o2, residual=res); rec.append(("b%d.c3.t" % bi, u3.saved[1].clone())); rec.append(("b%d.c3.y" % bi, h.clone()))
    outs.append(rec)
torch.cuda.synchronize()
for rep in (1, 2):
    for (n0, a), (n1, b) in zip(outs[0], outs[rep]):
        if not torch.equal(a, b):
            d = (a.float() - b.float()).abs()
            print("run", rep, "first difference at", n0, "max abs", float(d.max()), "count", int((d > 0).sum()), "of", a.numel())
            break
    else:
        print("run", rep, "identical")
