"""16-bit STORAGE floor of the first-step ResNet-50 parameter gradients for candidate parity fixtures (CPU only): relative L2
distance between the fp32 oracle's gradient and the same oracle with 16-bit storage emulation, per parameter (worst / median).
    python tools/probes/rn50_grad_floor.py "batch,size,bn3_gamma_scale" ...
The GPU test (tests/test_gpu_rn50_step.py) uses the fixture whose bf16 floor is < 5 %."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import resnet_oracle as RO  # noqa: E402

for spec in sys.argv[1:]:
    batch, size, damp = spec.split(",")
    batch, size, damp = int(batch), int(size), float(damp)
    state = RO.seeded_state(RO.RN50_STEP_CONFIG["seed"])
    if damp != 1.0:
        state = {k: (v * damp if k.endswith("bn3.weight") else v) for k, v in state.items()}
    x, y = RO.seeded_batch(77, batch, size)
    orc = RO.ResNet50Oracle(state, lr=0.0)
    orc.step(x, y)
    for dt in (torch.float16, torch.bfloat16):
        emu = RO.ResNet50Oracle(state, lr=0.0, storage_dtype=dt)
        emu.step(x, y)
        errs = [float((emu.p[n].grad - orc.p[n].grad).norm() / (orc.p[n].grad.norm() + 1e-12)) for n in orc.p]
        print("batch %d size %d bn3 x%.2f %s: gradient storage floor worst %.3f median %.3f" %
              (batch, size, damp, str(dt).split(".")[1], max(errs), float(np.median(errs))), flush=True)
