"""First-step loss of the HIP RN50 forward vs the fp32 CPU oracle for several batches/seeds (precision noise?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import resnet_oracle as RO
from deeplearningexamples_amd.convnets.resnet import ResNet50
from deeplearningexamples_amd.convnets.engine import ResNetTrainer
from deeplearningexamples_amd import functional as F
dev = torch.device("cuda", 0)
state = RO.seeded_state(5)
for dt in (torch.float16, torch.bfloat16):
    model = ResNet50(device=dev)
    model.load_state_dict({k: v.clone() for k, v in state.items()}, strict=False)
    tr = ResNetTrainer(model, lr=0.0, compute_dtype=dt, static_loss_scale=128.0)
    orc = RO.ResNet50Oracle(state, lr=0.0)
    for seed, b in [(77, 8), (78, 8), (79, 8), (105, 32), (106, 32), (80, 16)]:
        x, y = RO.seeded_batch(seed, b, 64)
        with torch.no_grad():
            lo = float(orc.loss(orc.forward(x), y))
        logits = tr.forward(x.to(dev))
        loss, _ = F.softmax_xent(logits, y.to(dev), smoothing=0.1)
        lref = orc.forward(x).detach()
        err = float((logits.cpu() - lref).abs().max() / lref.abs().max())
        print(dt, "seed", seed, "batch", b, "oracle %.5f hip %.5f rel %.2e  max logit err / max logit %.2e" % (lo, loss.item(), abs(loss.item() - lo) / lo, err), flush=True)
