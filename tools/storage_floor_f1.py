"""16-bit STORAGE floors of the WaveGlow / Tacotron2 steps (the numbers the parity bars of tests/test_gpu_waveglow.py and
tests/test_gpu_tacotron2.py add to north_star's 1e-3): the product engines run on the CPU over the fp64-accumulating test doubles
of the C-ABI calls (tests/_waveglow_doubles.py, tests/_tacotron2_doubles.py) with fp16 / bf16 storage, against the fp32 oracles.

    python tools/storage_floor_f1.py [--full]  > profiles/old/r02_f1_storage_floors.txt      (CPU only; --full adds the 268 M network)
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def waveglow(cfg_name, cfg, seed, batch, segment, scale):
    from oracle import waveglow_oracle as WO
    from tests import _waveglow_doubles as D
    from deeplearningexamples_amd.waveglow.engine import WaveGlowTrainer
    from deeplearningexamples_amd.waveglow.model import WaveGlow
    mp = pytest.MonkeyPatch()
    D.install(mp)
    state = WO.seeded_state(cfg, seed)
    mel, audio = WO.seeded_inputs(dict(cfg=cfg, seed=seed, batch=batch, segment=segment))
    p = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    lo = WO.waveglow_loss(p, cfg, mel, audio, 1.0)
    lo.backward()
    for dt in (torch.float16, torch.bfloat16):
        model = WaveGlow(**cfg)
        model.load_reference_state(state)
        tr = WaveGlowTrainer(model, compute_dtype=dt, amp=True, init_loss_scale=scale)
        loss = tr.forward(mel, audio)
        tr.backward()
        errs = {k: float((tr.g[k] / scale - p[k].grad).norm() / p[k].grad.norm()) for k in p}
        worst = max(errs.items(), key=lambda kv: kv[1])
        print("waveglow %-8s %-9s loss rel %.2e   gradient rel L2: worst %.2e (%s)  median %.2e" %
              (cfg_name, str(dt).split(".")[1], abs(float(loss) - float(lo)) / abs(float(lo)), worst[1], worst[0],
               float(np.median(list(errs.values())))), flush=True)
    mp.undo()


def tacotron2(seeds, case=None, label="small", scale=1024.0):
    from oracle import tacotron2_oracle as TO
    from tests import _tacotron2_doubles as D
    from tests.test_tacotron2_host import _Replay
    from deeplearningexamples_amd.tacotron2.engine import Tacotron2Trainer
    from deeplearningexamples_amd.tacotron2.model import Tacotron2
    mp = pytest.MonkeyPatch()
    D.install(mp)
    c = case or TO.TACOTRON2_CASE
    cfg = c["cfg"]
    text, tl, mel, gate, ml = TO.seeded_batch(c)
    for dt in (torch.float16, torch.bfloat16):
        for seed in seeds:
            D.Masks.reset(seed)
            state = TO.seeded_state(cfg, c["seed"])
            model = Tacotron2(**cfg)
            model.load_reference_state(state)
            tr = Tacotron2Trainer(model, compute_dtype=dt, amp=True, init_loss_scale=scale)
            loss = tr.forward(text, tl, mel, gate)
            tr.backward()
            replay = _Replay(D.Masks.log, mel.shape[2], text.shape[0], cfg["attention_rnn_dim"], cfg["decoder_rnn_dim"])
            p = {k: v.clone().requires_grad_(True) for k, v in state.items()}
            lo, _ = TO.tacotron2_loss(p, cfg, text, tl, mel, gate, replay)
            lo.backward()
            errs = {k: float((tr.g[k] / scale - p[k].grad).norm() / p[k].grad.norm()) for k in p if float(p[k].grad.norm()) > 1e-5}
            worst = max(errs.items(), key=lambda kv: kv[1])
            print("tacotron2 %-7s %-9s masks %d  loss rel %.2e   gradient rel L2: worst %.2e (%s)  median %.2e" %
                  (label, str(dt).split(".")[1], seed, abs(float(loss) - float(lo.detach())) / abs(float(lo.detach())), worst[1], worst[0],
                   float(np.median(list(errs.values())))), flush=True)
    mp.undo()


if __name__ == "__main__":
    import warnings
    warnings.filterwarnings("ignore")
    from oracle import waveglow_oracle as WO
    from deeplearningexamples_amd.waveglow.model import DEFAULT_CONFIG
    if "--t2-default" in sys.argv:       # the reference's widths, the case tests/test_gpu_tacotron2.py checks on the GPU
        from oracle import tacotron2_oracle as TO
        tacotron2([1, 2, 3], TO.TACOTRON2_DEFAULT_CASE, "default", 65536.0)
        sys.exit(0)
    waveglow("small", WO.WAVEGLOW_SMALL, 7, 2, 2048, 65536.0)
    if "--full" in sys.argv:
        waveglow("default", DEFAULT_CONFIG, 11, 2, 2048, 4096.0)
    tacotron2([1, 2, 3, 4, 5, 6])
