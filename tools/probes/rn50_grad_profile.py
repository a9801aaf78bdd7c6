"""Per-parameter relative L2 gradient error: HIP fp16 vs fp32 oracle, and 16-bit-storage oracle vs fp32 oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import resnet_oracle as RO
from deeplearningexamples_amd.convnets.resnet import ResNet50
from deeplearningexamples_amd.convnets.engine import ResNetTrainer
from deeplearningexamples_amd import functional as F
dev = torch.device("cuda", 0)
dt = torch.float16
b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
state = RO.seeded_state(5)
model = ResNet50(device=dev)
model.load_state_dict({k: v.clone() for k, v in state.items()}, strict=False)
tr = ResNetTrainer(model, lr=0.0, compute_dtype=dt, static_loss_scale=128.0)
x, y = RO.seeded_batch(77, b, 64)
o32 = RO.ResNet50Oracle(state, lr=0.0); o32.step(x, y)
o16 = RO.ResNet50Oracle(state, lr=0.0, storage_dtype=dt); o16.step(x, y)
logits = tr.forward(x.to(dev))
loss, dl = F.softmax_xent(logits, y.to(dev), smoothing=0.1, grad_dtype=dt, grad_scale=tr.scaler.scale)
tr.backward(dl)
torch.cuda.synchronize()
for n, p in model.named_parameters():
    ph = lambda t: (t.permute(0, 2, 3, 1) if t.dim() == 4 else t).reshape(-1).double()
    g = tr.gview[n].view(-1).cpu().double() / 128.0
    r32, r16 = ph(o32.p[n].grad), ph(o16.p[n].grad)
    e_hip = float((g - r32).norm() / r32.norm()); e_emu = float((r16 - r32).norm() / r32.norm())
    e_he = float((g - r16).norm() / r16.norm())
    if "conv" in n or n.startswith("fc") or "downsample.0" in n:
        print("%-34s hip-vs-fp32 %.4f   emu16-vs-fp32 %.4f   hip-vs-emu16 %.4f" % (n, e_hip, e_emu, e_he))
