"""Entry point mirroring SpeechSynthesis/Tacotron2/train.py for `-m Tacotron2` (second half):

    python -m deeplearningexamples_amd.tacotron2.train -m Tacotron2 -o out/ --amp -lr 1e-3 --epochs 2 -bs 128 \
        --weight-decay 1e-6 --grad-clip-thresh 1.0 --log-file nvlog.json --anneal-steps 500 1000 1500 --anneal-factor 0.3

Same flags (train.py:45-160, tacotron2/arg_parser.py:40-107), loop (train.py:444-500), DLLogger records (train_items_per_sec = mel
frames / s) and checkpoint files (`checkpoint_Tacotron2_<epoch>.pt`, train.py:185-255) as the reference; the shared pieces live
in waveglow/train.py.  Data: synthetic batches in TextMelCollate's layout (tacotron2/data_function.py:100-151: text sorted by
length, zero-padded mels, gate target 1 from the last frame on); text cleaning / the STFT front end are host-side and out of scope.
"""
import argparse
import os
import time

import numpy as np
import torch

from ..utils import dllogger as DLLogger
from ..utils.dist import init_from_env
from ..waveglow import train as WT
from .engine import Tacotron2Trainer
from .model import DEFAULT_CONFIG, Tacotron2, param_shapes


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Tacotron2 training on MI355X (train.py CLI of the reference, -m Tacotron2)")
    p.add_argument("-o", "--output", type=str, required=True)
    p.add_argument("-m", "--model-name", type=str, default="Tacotron2", choices=["Tacotron2"])
    p.add_argument("--log-file", type=str, default="nvlog.json")
    p.add_argument("--anneal-steps", nargs="*")
    p.add_argument("--anneal-factor", type=float, choices=[0.1, 0.3], default=0.1)
    p.add_argument("--seed", default=None, type=int)
    p.add_argument("--epochs", type=int, required=True)
    p.add_argument("--epochs-per-checkpoint", type=int, default=50)
    p.add_argument("--checkpoint-path", type=str, default="")
    p.add_argument("--resume-from-last", action="store_true")
    p.add_argument("--amp", action="store_true")
    p.add_argument("--cudnn-enabled", action="store_true", help="accepted for CLI compatibility; there is no cuDNN here")
    p.add_argument("--cudnn-benchmark", action="store_true", help="accepted for CLI compatibility")
    p.add_argument("-lr", "--learning-rate", type=float, required=True)
    p.add_argument("--weight-decay", default=1e-6, type=float)
    p.add_argument("--grad-clip-thresh", default=1.0, type=float)
    p.add_argument("-bs", "--batch-size", type=int, required=True)
    p.add_argument("--bench-class", type=str, default="")
    for k, v in DEFAULT_CONFIG.items():                                # tacotron2/arg_parser.py: --n-mel-channels, --prenet-dim, ...
        p.add_argument("--" + k.replace("_", "-"), default=v, type=type(v))
    p.add_argument("--iters-per-epoch", default=50, type=int, help="synthetic data: iterations that make up one epoch")
    p.add_argument("--compute-dtype", default="fp16", choices=["fp16", "bf16"])
    args, _ = p.parse_known_args(argv)
    return args


def get_model_config(args):
    """models.get_model_config('Tacotron2', args) (models.py:97-130), the keys this port implements."""
    return {k: getattr(args, k) for k in DEFAULT_CONFIG}


def parameter_order(cfg):
    """model.parameters() order of the reference's Tacotron2 = the order of model.param_shapes."""
    return [n for n, _ in param_shapes(cfg)]


class SyntheticTextMel:
    """Device-resident batches in TextMelCollate's layout; a fixed pool, cycled."""

    def __init__(self, batch, n_symbols, n_mel, device, seed, pool=4):
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.items, self.num_items = [], []
        for _ in range(pool):
            tl = torch.sort(torch.randint(60, 161, (batch,), generator=g), descending=True).values
            ml = (tl.float() * (5.0 + 0.8 * torch.rand(batch, generator=g))).long()
            text = torch.zeros(batch, int(tl.max()), dtype=torch.long)
            mel = torch.zeros(batch, n_mel, int(ml.max()))
            gate = torch.zeros(batch, int(ml.max()))
            for i in range(batch):
                text[i, :tl[i]] = torch.randint(1, n_symbols, (int(tl[i]),), generator=g)
                mel[i, :, :ml[i]] = torch.randn(n_mel, int(ml[i]), generator=g) * 1.5 - 4.0
                gate[i, ml[i] - 1:] = 1
            self.items.append(tuple(t.to(device) for t in (text, tl, mel, gate)))
            self.num_items.append(int(ml.sum()))

    def __getitem__(self, i):
        return self.items[i % len(self.items)], self.num_items[i % len(self.items)]


def main(argv=None):
    args = parse_args(argv)
    rank, world, local = init_from_env("nccl")
    dev = torch.device("cuda", local)
    if args.seed is not None:
        torch.manual_seed(args.seed + local)
        np.random.seed(args.seed + local)
    os.makedirs(args.output, exist_ok=True)
    DLLogger.init(backends=[DLLogger.JSONStreamBackend(DLLogger.Verbosity.DEFAULT, os.path.join(args.output, args.log_file)),
                            DLLogger.StdOutBackend(DLLogger.Verbosity.VERBOSE)] if rank == 0 else [])
    for k, v in vars(args).items():
        DLLogger.log(step="PARAMETER", data={k: v})
    DLLogger.log(step="PARAMETER", data={"model_name": "Tacotron2_PyT"})
    config = get_model_config(args)
    model = Tacotron2(device=dev, **config)
    trainer = Tacotron2Trainer(model, lr=args.learning_rate, weight_decay=args.weight_decay, grad_clip_thresh=args.grad_clip_thresh,
                               compute_dtype=torch.float16 if args.compute_dtype == "fp16" else torch.bfloat16, amp=args.amp,
                               world_size=world, seed=(args.seed or 1234), rank=rank)
    names = parameter_order(config)
    start_epoch = 0
    if args.resume_from_last:
        args.checkpoint_path = WT.get_last_checkpoint_filename(args.output, args.model_name)
    if args.checkpoint_path:
        config, start_epoch = WT.load_checkpoint(trainer, args.checkpoint_path, local, names)
    data = SyntheticTextMel(args.batch_size, config["n_symbols"], config["n_mel_channels"], dev, (args.seed or 0) + 1000 * rank)
    iteration = start_epoch * args.iters_per_epoch
    torch.cuda.synchronize()
    run_start = time.perf_counter()
    loss_v, ips_epoch = float("nan"), 0.0
    for epoch in range(start_epoch, args.epochs):
        ips_sum = 0.0
        t_epoch = time.perf_counter()
        for i in range(args.iters_per_epoch):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            trainer.set_lr(WT.adjust_learning_rate(epoch, args.learning_rate, args.anneal_steps, args.anneal_factor))
            batch, num_items = data[iteration]
            loss = trainer.train_step(*batch)
            if world > 1:
                from ..utils.comm import allreduce_mean_
                loss = allreduce_mean_(loss.clone())
            loss_v = float(loss.item())
            if np.isnan(loss_v):
                raise Exception("loss is NaN")
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ips = num_items * world / dt
            ips_sum += ips
            DLLogger.log(step=(epoch, i), data={"train_loss": loss_v, "train_items_per_sec": ips, "train_iter_time": dt})
            iteration += 1
        ips_epoch = ips_sum / max(args.iters_per_epoch, 1)
        DLLogger.log(step=(epoch,), data={"train_items_per_sec": ips_epoch, "train_loss": loss_v,
                                          "train_epoch_time": time.perf_counter() - t_epoch})
        if epoch % args.epochs_per_checkpoint == 0 and args.bench_class in ("", "train"):
            WT.save_checkpoint(trainer, epoch, config, args.output, args.model_name, local, world, names)
        if rank == 0:
            DLLogger.flush()
    torch.cuda.synchronize()
    DLLogger.log(step=tuple(), data={"run_time": time.perf_counter() - run_start, "train_loss": loss_v,
                                     "train_items_per_sec": ips_epoch})
    if rank == 0:
        DLLogger.flush()
    return loss_v


if __name__ == "__main__":
    main()
