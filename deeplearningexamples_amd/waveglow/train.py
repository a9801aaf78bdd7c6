"""Entry point mirroring SpeechSynthesis/Tacotron2/train.py for `-m WaveGlow`:

    python -m deeplearningexamples_amd.waveglow.train -m WaveGlow -o out/ --amp -lr 1e-4 --epochs 2 -bs 10 \
        --segment-length 8000 --weight-decay 0 --grad-clip-thresh 65504.0 --log-file nvlog.json
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m deeplearningexamples_amd.waveglow.train ...

Same flags (train.py:45-160, waveglow/arg_parser.py:30-64), the same loop (train.py:444-500: adjust_learning_rate, forward,
criterion, scaled backward, unscale + clip_grad_norm_, Adam step, scaler.update), DLLogger records (train_loss,
train_items_per_sec = audio samples / s) and checkpoint files (`checkpoint_WaveGlow_<epoch>.pt` + the `_last` symlink,
train.py:185-255: epoch, RNG states, config, state_dict, torch.optim.Adam state, GradScaler state) that the reference's own
load_checkpoint reads.  Data: synthetic LJSpeech-shaped segments resident on the device (there is no dataset in this
environment; MelAudioLoader's STFT front end, waveglow/data_function.py:33-77, is host-side and out of the hot path).
"""
import argparse
import os
import time

import numpy as np
import torch

from ..utils import dllogger as DLLogger
from ..utils.dist import init_from_env
from .engine import WaveGlowTrainer
from .model import WaveGlow


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="WaveGlow training on MI355X (train.py CLI of the reference, -m WaveGlow)")
    p.add_argument("-o", "--output", type=str, required=True)
    p.add_argument("-d", "--dataset-path", type=str, default="./")
    p.add_argument("-m", "--model-name", type=str, default="WaveGlow", choices=["WaveGlow"])
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
    # waveglow/arg_parser.py
    p.add_argument("--n-mel-channels", default=80, type=int)
    p.add_argument("--flows", default=12, type=int)
    p.add_argument("--groups", default=8, type=int)
    p.add_argument("--early-every", default=4, type=int)
    p.add_argument("--early-size", default=2, type=int)
    p.add_argument("--sigma", default=1.0, type=float)
    p.add_argument("--segment-length", default=4000, type=int)
    p.add_argument("--wn-kernel-size", default=3, type=int)
    p.add_argument("--wn-channels", default=512, type=int)
    p.add_argument("--wn-layers", default=8, type=int)
    # this port
    p.add_argument("--iters-per-epoch", default=100, type=int, help="synthetic data: iterations that make up one epoch")
    p.add_argument("--compute-dtype", default="fp16", choices=["fp16", "bf16"], help="16-bit storage type of the AMP path")
    args, _ = p.parse_known_args(argv)
    return args


def get_model_config(args):
    """models.get_model_config('WaveGlow', args) (models.py:131-146)."""
    return dict(n_mel_channels=args.n_mel_channels, n_flows=args.flows, n_group=args.groups, n_early_every=args.early_every,
                n_early_size=args.early_size,
                WN_config=dict(n_layers=args.wn_layers, kernel_size=args.wn_kernel_size, n_channels=args.wn_channels))


def adjust_learning_rate(epoch, learning_rate, anneal_steps, anneal_factor):
    """train.py:324-342."""
    p = 0
    if anneal_steps is not None:
        for a_step in anneal_steps:
            if epoch >= int(a_step):
                p += 1
    if anneal_factor == 0.3:
        return learning_rate * ((0.1 ** (p // 2)) * (1.0 if p % 2 == 0 else 0.3))
    return learning_rate * (anneal_factor ** p)


def reference_parameter_order(cfg):
    """model.parameters() order of the reference's WaveGlow (module registration order: upsample, WN[k] = in_layers,
    res_skip_layers, cond_layers, start, end -- waveglow/model.py:95-136 -- then convinv[k]; a weight_norm'd conv holds bias,
    weight_g, weight_v).  torch.optim.Adam.state_dict() indexes its state by position in this order."""
    nl = cfg["WN_config"]["n_layers"]
    names = ["upsample.weight", "upsample.bias"]
    for k in range(cfg["n_flows"]):
        pre = "WN.%d." % k
        for group in ("in_layers", "res_skip_layers", "cond_layers"):
            for i in range(nl):
                names += [pre + "%s.%d.%s" % (group, i, s) for s in ("bias", "weight_g", "weight_v")]
        names += [pre + "start.bias", pre + "start.weight_g", pre + "start.weight_v", pre + "end.weight", pre + "end.bias"]
    names += ["convinv.%d.conv.weight" % k for k in range(cfg["n_flows"])]
    return names


def optimizer_state_dict(trainer, names=None):
    """The state_dict torch.optim.Adam(model.parameters(), lr, weight_decay) would hold after the trainer's steps.  names: the
    reference model's parameters() order (default: WaveGlow's)."""
    names = names or reference_parameter_order(trainer.cfg)
    step = int(trainer.step_t.item())
    state = {}
    if step > 0:
        for i, n in enumerate(names):
            state[i] = {"step": torch.tensor(float(step)), "exp_avg": trainer.m[n].detach().cpu().clone(),
                        "exp_avg_sq": trainer.v[n].detach().cpu().clone()}
    group = {"lr": trainer.lr, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": trainer.wd, "amsgrad": False,
             "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
             "decoupled_weight_decay": False, "params": list(range(len(names)))}
    return {"state": state, "param_groups": [group]}


def load_optimizer_state_dict(trainer, sd, names=None):
    names = names or reference_parameter_order(trainer.cfg)
    steps = set()
    with torch.no_grad():
        for i, n in enumerate(names):
            st = sd["state"].get(i)
            if st is None:
                continue
            trainer.m[n].copy_(st["exp_avg"].to(trainer.dev))
            trainer.v[n].copy_(st["exp_avg_sq"].to(trainer.dev))
            steps.add(int(float(st["step"])))
    if len(steps) > 1:
        raise ValueError("the parameters of this checkpoint have taken different numbers of Adam steps: %s" % sorted(steps))
    trainer.step_t.fill_(steps.pop() if steps else 0)
    trainer.set_lr(float(sd["param_groups"][0]["lr"]))


def scaler_state_dict(trainer):
    """torch.cuda.amp.GradScaler.state_dict() (train.py:211)."""
    sc = trainer.scaler
    if not sc.enabled:
        return {}
    return {"scale": float(sc.scale.item()), "growth_factor": sc.growth_factor, "backoff_factor": sc.backoff_factor,
            "growth_interval": sc.growth_interval, "_growth_tracker": int(sc.growth_tracker.item())}


def load_scaler_state_dict(trainer, sd):
    sc = trainer.scaler
    if not sc.enabled or not sd:
        return
    sc.scale.fill_(float(sd["scale"]))
    sc.inv_scale.fill_(1.0 / float(sd["scale"]))
    sc.growth_tracker.fill_(int(sd["_growth_tracker"]))
    sc.growth_factor, sc.backoff_factor, sc.growth_interval = sd["growth_factor"], sd["backoff_factor"], sd["growth_interval"]


def save_checkpoint(trainer, epoch, config, output_dir, model_name, local_rank, world_size, names=None):
    """train.py:185-226 (rank 0 writes; every rank's RNG state is recorded)."""
    rng = torch.random.get_rng_state()
    cuda_rng = torch.cuda.get_rng_state(local_rank) if torch.cuda.is_available() else torch.zeros(1, dtype=torch.uint8)
    if world_size > 1:
        import torch.distributed as dist
        rngs, cudas = [None] * world_size, [None] * world_size
        dist.all_gather_object(rngs, rng)
        dist.all_gather_object(cudas, cuda_rng)
    else:
        rngs, cudas = [rng], [cuda_rng]
    if local_rank != 0:
        return None
    ckpt = {"epoch": epoch, "cuda_rng_state_all": torch.stack(cudas), "random_rng_states_all": torch.stack(rngs),
            "config": config, "state_dict": {k: v.detach().cpu().clone() for k, v in trainer.model.state_dict().items()},
            "optimizer": optimizer_state_dict(trainer, names), "scaler": scaler_state_dict(trainer)}
    if hasattr(trainer, "_rng_base"):
        # not a reference key (its loader ignores it): the position of the trainer's counter-based dropout stream (Tacotron2),
        # which torch's restored RNG state does not drive -- a resumed run continues the mask sequence
        ckpt["dle_rng_calls"] = int(trainer._rng_base.item())
    name = "checkpoint_{}_{}.pt".format(model_name, epoch)
    path = os.path.join(output_dir, name)
    torch.save(ckpt, path)
    link = os.path.join(output_dir, "checkpoint_{}_last.pt".format(model_name))
    if os.path.lexists(link):
        os.remove(link)
    os.symlink(name, link)
    return path


def get_last_checkpoint_filename(output_dir, model_name):
    link = os.path.join(output_dir, "checkpoint_{}_last.pt".format(model_name))
    return os.path.join(output_dir, os.readlink(link)) if os.path.exists(link) else ""


def load_checkpoint(trainer, filepath, local_rank, names=None):
    """train.py:239-255 -> (config, first epoch to run)."""
    ckpt = torch.load(filepath, map_location="cpu", weights_only=False)
    if torch.cuda.is_available():
        dev_id = local_rank % torch.cuda.device_count()
        torch.cuda.set_rng_state(ckpt["cuda_rng_state_all"][dev_id % len(ckpt["cuda_rng_state_all"])])
    if "random_rng_states_all" in ckpt:
        torch.random.set_rng_state(ckpt["random_rng_states_all"][local_rank % len(ckpt["random_rng_states_all"])])
    elif "random_rng_state" in ckpt:
        torch.random.set_rng_state(ckpt["random_rng_state"])
    else:
        raise Exception("Model checkpoint must have either 'random_rng_state' or 'random_rng_states_all' key.")
    # the reference wraps the model in DistributedDataParallel when distributed: its multi-GPU files carry "module." keys
    trainer.model.load_reference_state({(k[7:] if k.startswith("module.") else k): v for k, v in ckpt["state_dict"].items()})
    if hasattr(trainer, "_rng_base"):
        trainer._rng_base.fill_(int(ckpt.get("dle_rng_calls", 0)))
    load_optimizer_state_dict(trainer, ckpt["optimizer"], names)
    load_scaler_state_dict(trainer, ckpt["scaler"])
    return ckpt["config"], ckpt["epoch"] + 1


class SyntheticMelAudio:
    """Device-resident (mel, audio) batches of MelAudioLoader's shapes: audio segments in [-1, 1], log-mel-like frames, one frame
    per 256 samples (+ the frame the centred STFT adds).  A fixed pool of batches, cycled."""

    def __init__(self, batch, segment, n_mel, device, seed, pool=4):
        g = torch.Generator(device="cpu").manual_seed(seed)
        frames = segment // 256 + 1
        self.items = [((torch.randn(batch, n_mel, frames, generator=g) * 2.0 - 5.0).to(device),
                       (torch.randn(batch, segment, generator=g) * 0.2).clamp_(-1, 1).to(device)) for _ in range(pool)]
        self.num_items = batch * segment

    def __getitem__(self, i):
        return self.items[i % len(self.items)]


def main(argv=None):
    args = parse_args(argv)
    rank, world, local = init_from_env("nccl")
    dev = torch.device("cuda", local)
    if args.seed is not None:
        torch.manual_seed(args.seed + local)
        np.random.seed(args.seed + local)
    os.makedirs(args.output, exist_ok=True)
    if rank == 0:
        DLLogger.init(backends=[DLLogger.JSONStreamBackend(DLLogger.Verbosity.DEFAULT, os.path.join(args.output, args.log_file)),
                                DLLogger.StdOutBackend(DLLogger.Verbosity.VERBOSE)])
    else:
        DLLogger.init(backends=[])
    for k, v in vars(args).items():
        DLLogger.log(step="PARAMETER", data={k: v})
    DLLogger.log(step="PARAMETER", data={"model_name": "WaveGlow_PyT"})
    config = get_model_config(args)
    model = WaveGlow(**config, device=dev)
    trainer = WaveGlowTrainer(model, lr=args.learning_rate, weight_decay=args.weight_decay, grad_clip_thresh=args.grad_clip_thresh,
                              sigma=args.sigma, compute_dtype=torch.float16 if args.compute_dtype == "fp16" else torch.bfloat16,
                              amp=args.amp, world_size=world)
    start_epoch = 0
    if args.resume_from_last:
        args.checkpoint_path = get_last_checkpoint_filename(args.output, args.model_name)
    if args.checkpoint_path:
        config, start_epoch = load_checkpoint(trainer, args.checkpoint_path, local)
    data = SyntheticMelAudio(args.batch_size, args.segment_length, args.n_mel_channels, dev, (args.seed or 0) + 1000 * rank)
    iteration = start_epoch * args.iters_per_epoch
    torch.cuda.synchronize()
    run_start = time.perf_counter()
    loss_v, ips_epoch = float("nan"), 0.0
    for epoch in range(start_epoch, args.epochs):
        torch.cuda.synchronize()
        t_epoch = time.perf_counter()
        ips_sum = 0.0
        for i in range(args.iters_per_epoch):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            trainer.set_lr(adjust_learning_rate(epoch, args.learning_rate, args.anneal_steps, args.anneal_factor))
            mel, audio = data[iteration]
            loss = trainer.train_step(mel, audio)
            if world > 1:
                from ..utils.comm import allreduce_mean_
                loss = allreduce_mean_(loss.clone())
            loss_v = float(loss.item())                      # the reference reads the loss every iteration too (train.py:476-481)
            # torch.logdet of a matrix with a negative determinant is NaN in the reference (model.py:74); the kernels return
            # log|det| and the sign, so the same condition is raised here instead of training on
            if np.isnan(loss_v) or float(trainer.signs.min().item()) < 0:
                raise Exception("loss is NaN")
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ips = data.num_items * world / dt
            ips_sum += ips
            DLLogger.log(step=(epoch, i), data={"train_loss": loss_v, "train_items_per_sec": ips, "train_iter_time": dt})
            iteration += 1
        torch.cuda.synchronize()
        ips_epoch = ips_sum / max(args.iters_per_epoch, 1)
        DLLogger.log(step=(epoch,), data={"train_items_per_sec": ips_epoch, "train_loss": loss_v,
                                          "train_epoch_time": time.perf_counter() - t_epoch})
        if epoch % args.epochs_per_checkpoint == 0 and args.bench_class in ("", "train"):
            save_checkpoint(trainer, epoch, config, args.output, args.model_name, local, world)
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
