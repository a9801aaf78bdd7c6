"""Entry point mirroring SpeechSynthesis/Tacotron2/train.py for `-m WaveGlow`:

    python -m deeplearningexamples_amd.waveglow.train -m WaveGlow -o out/ --amp -lr 1e-4 --epochs 2 -bs 10 \
        --segment-length 8000 --weight-decay 0 --grad-clip-thresh 65504.0 --log-file nvlog.json
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m deeplearningexamples_amd.waveglow.train ...

Same flags (train.py:45-160, waveglow/arg_parser.py:30-64), the same loop (train.py:444-500: adjust_learning_rate, forward,
criterion, scaled backward, unscale + clip_grad_norm_, Adam step, scaler.update), DLLogger records (train_loss,
train_items_per_sec = audio samples / s) and checkpoint files (`checkpoint_WaveGlow_<epoch>.pt` + the `_last` symlink,
train.py:185-255: epoch, RNG states, config, state_dict, torch.optim.Adam state, GradScaler state) that the reference's own
load_checkpoint reads.  Data: the filelists of --training-files / --validation-files under -d through MelAudioLoader
(waveglow/data_function.py) and a DataLoader set up as train.py:420-436 does, or -- with --synthetic-data, the benchmark mode of
this port: there is no dataset in this environment -- LJSpeech-shaped segments resident on the device.  After every epoch the
validation pass of train.py:273-318 runs (forward + criterion, nothing updated) and logs val_loss / val_items_per_sec.
The pieces both speech trainers share (the common flags, checkpoint files, data plumbing, the epoch loop) live here;
tacotron2/train.py adds its model flags and trainer.
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


class ParseFromConfigFile(argparse.Action):
    """--config-file (tacotron2_common/utils.py:36-49): {group: {flag: value}} JSON, every entry set on the namespace."""

    def __init__(self, option_strings, type, dest, help=None, required=False):
        super().__init__(option_strings=option_strings, type=type, dest=dest, help=help, required=required)

    def __call__(self, parser, namespace, values, option_string=None):
        import json
        with open(values, "r") as f:
            data = json.load(f)
        for group in data:
            for k, v in data[group].items():
                setattr(namespace, k.replace("-", "_"), v)


def common_parser(description, model_name):
    """The flags of train.py:45-160 (every one parses; the cuDNN / dist-url family is accepted and unused: there is no cuDNN, and
    ranks come from the launcher's environment)."""
    p = argparse.ArgumentParser(description=description, allow_abbrev=False)
    p.add_argument("-o", "--output", type=str, required=True, help="Directory to save checkpoints")
    p.add_argument("-d", "--dataset-path", type=str, default="./", help="Path to dataset")
    p.add_argument("-m", "--model-name", type=str, default=model_name, choices=[model_name])
    p.add_argument("--log-file", type=str, default="nvlog.json")
    p.add_argument("--anneal-steps", nargs="*", help="Epochs after which decrease learning rate")
    p.add_argument("--anneal-factor", type=float, choices=[0.1, 0.3], default=0.1)
    p.add_argument("--config-file", action=ParseFromConfigFile, type=str, help="Path to configuration file")
    p.add_argument("--seed", default=None, type=int)
    t = p.add_argument_group("training setup")
    t.add_argument("--epochs", type=int, required=True)
    t.add_argument("--epochs-per-checkpoint", type=int, default=50)
    t.add_argument("--checkpoint-path", type=str, default="")
    t.add_argument("--resume-from-last", action="store_true")
    t.add_argument("--dynamic-loss-scaling", type=bool, default=True)
    t.add_argument("--amp", action="store_true")
    t.add_argument("--cudnn-enabled", action="store_true", help="accepted for CLI compatibility; there is no cuDNN here")
    t.add_argument("--cudnn-benchmark", action="store_true", help="accepted for CLI compatibility")
    t.add_argument("--disable-uniform-initialize-bn-weight", action="store_true",
                   help="keep BatchNorm weights at 1 instead of U[0, 1) (models.py:53-62)")
    o = p.add_argument_group("optimization setup")
    o.add_argument("--use-saved-learning-rate", default=False, type=bool)
    o.add_argument("-lr", "--learning-rate", type=float, required=True)
    o.add_argument("--weight-decay", default=1e-6, type=float)
    o.add_argument("--grad-clip-thresh", default=1.0, type=float)
    o.add_argument("-bs", "--batch-size", type=int, required=True)
    o.add_argument("--grad-clip", default=5.0, type=float, help="parsed and unused, as in the reference")
    d = p.add_argument_group("dataset parameters")
    d.add_argument("--load-mel-from-disk", action="store_true")
    d.add_argument("--training-files", default="filelists/ljs_audio_text_train_filelist.txt", type=str)
    d.add_argument("--validation-files", default="filelists/ljs_audio_text_val_filelist.txt", type=str)
    d.add_argument("--text-cleaners", nargs="*", default=["english_cleaners"], type=str)
    a = p.add_argument_group("audio parameters")
    a.add_argument("--max-wav-value", default=32768.0, type=float)
    a.add_argument("--sampling-rate", default=22050, type=int)
    a.add_argument("--filter-length", default=1024, type=int)
    a.add_argument("--hop-length", default=256, type=int)
    a.add_argument("--win-length", default=1024, type=int)
    a.add_argument("--mel-fmin", default=0.0, type=float)
    a.add_argument("--mel-fmax", default=8000.0, type=float)
    g = p.add_argument_group("distributed setup")
    g.add_argument("--rank", default=0, type=int, help="unused: RANK comes from the launcher")
    g.add_argument("--world-size", default=1, type=int, help="unused: WORLD_SIZE comes from the launcher")
    g.add_argument("--dist-url", type=str, default="tcp://localhost:23456", help="unused: MASTER_ADDR / MASTER_PORT")
    g.add_argument("--group-name", type=str, default="group_name")
    g.add_argument("--dist-backend", default="nccl", type=str, choices=["nccl"])
    p.add_argument("--bench-class", type=str, default="")
    x = p.add_argument_group("this port")
    x.add_argument("--synthetic-data", action="store_true",
                   help="device-resident synthetic batches of the dataset's shapes instead of the filelists (benchmarks)")
    x.add_argument("--iters-per-epoch", default=100, type=int, help="synthetic data: iterations that make up one epoch")
    x.add_argument("--compute-dtype", default="fp16", choices=["fp16", "bf16"], help="16-bit storage type of the AMP path")
    return p


def parse_args(argv=None):
    p = common_parser("WaveGlow training on MI355X (train.py CLI of the reference, -m WaveGlow)", "WaveGlow")
    w = p.add_argument_group("WaveGlow parameters")                    # waveglow/arg_parser.py:30-64
    w.add_argument("--n-mel-channels", default=80, type=int)
    w.add_argument("--flows", default=12, type=int)
    w.add_argument("--groups", default=8, type=int)
    w.add_argument("--early-every", default=4, type=int)
    w.add_argument("--early-size", default=2, type=int)
    w.add_argument("--sigma", default=1.0, type=float)
    w.add_argument("--segment-length", default=4000, type=int)
    w.add_argument("--wn-kernel-size", default=3, type=int)
    w.add_argument("--wn-channels", default=512, type=int)
    w.add_argument("--wn-layers", default=8, type=int)
    # parse_known_args, as the reference does (Tacotron2/train.py:349,382): a command line shared by both models, or a launcher's
    # --local_rank, must not abort the run; what is ignored is said on stderr
    args, unknown = p.parse_known_args(argv)
    if unknown:
        import sys
        print("warning: ignored command-line arguments (not used by this model): %s" % " ".join(unknown), file=sys.stderr)
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
        return self.items[i % len(self.items)], self.num_items


class SyntheticEpochs:
    """An epoch = `iters` batches of a cycled pool (`pool[i]` -> (trainer arguments, items in the batch))."""

    def __init__(self, pool, iters):
        self.pool, self.iters, self.count = pool, iters, 0

    def __len__(self):
        return self.iters

    def epoch(self, epoch):
        for _ in range(self.iters):
            yield self.pool[self.count]
            self.count += 1


class LoaderEpochs:
    """The DataLoader of train.py:420-436 / :277-282 (one worker, drop_last, DistributedSampler(seed) + set_epoch when
    distributed, shuffle otherwise) + the model's batch_to_gpu; to_trainer: (x, y) -> the trainer's positional arguments."""

    def __init__(self, dataset, batch_size, collate_fn, batch_to_gpu, to_trainer, device, world, rank, seed, train, drop_last):
        from torch.utils.data import DataLoader
        from torch.utils.data.distributed import DistributedSampler
        self.sampler = None
        if world > 1:
            self.sampler = DistributedSampler(dataset, num_replicas=world, rank=rank, seed=(seed or 0)) if train else \
                DistributedSampler(dataset, num_replicas=world, rank=rank)
        self.loader = DataLoader(dataset, num_workers=1, shuffle=(train and self.sampler is None), sampler=self.sampler,
                                 batch_size=batch_size, pin_memory=False, drop_last=drop_last, collate_fn=collate_fn)
        self.batch_to_gpu, self.to_trainer, self.device, self.train = batch_to_gpu, to_trainer, device, train

    def __len__(self):
        return len(self.loader)

    def epoch(self, epoch):
        if self.sampler is not None and self.train:
            self.sampler.set_epoch(epoch)
        for batch in self.loader:
            x, y, num_items = self.batch_to_gpu(batch, self.device)
            yield self.to_trainer(x, y), int(num_items.item())


def filelist(args, which):
    """--training-files / --validation-files as given, or relative to the dataset directory."""
    path = getattr(args, which)
    if not os.path.exists(path) and os.path.exists(os.path.join(args.dataset_path, path)):
        path = os.path.join(args.dataset_path, path)
    if not os.path.exists(path):
        raise SystemExit("%s: no such filelist (pass --synthetic-data to train on synthetic batches)" % path)
    return path


def init_run(args, model_tag):
    """Ranks, seeds (train.py:361-363), the output directory, DLLogger (train.py:365-378).  -> (rank, world, local, device)."""
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
    DLLogger.log(step="PARAMETER", data={"model_name": model_tag})
    return rank, world, local, dev


def validate(trainer, eval_fn, val_data, epoch, iteration, world):
    """train.py:273-318: the criterion over the validation set with the model in eval mode; nothing is updated."""
    from ..utils.comm import allreduce_mean_
    val_loss, ips_sum, n = 0.0, 0.0, 0
    for i, (batch, num_items) in enumerate(val_data.epoch(epoch)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = eval_fn(*batch)
        if world > 1:
            loss = allreduce_mean_(loss.clone())
        val_loss += float(loss.item())
        torch.cuda.synchronize()
        ips = num_items * world / (time.perf_counter() - t0)
        DLLogger.log(step=(epoch, iteration, i), data={"val_items_per_sec": ips})
        ips_sum += ips
        n += 1
    val_loss, ips = val_loss / max(n, 1), ips_sum / max(n, 1)
    DLLogger.log(step=(epoch,), data={"val_loss": val_loss})
    DLLogger.log(step=(epoch, iteration), data={"val_items_per_sec": ips})
    return val_loss, ips


def train_loop(args, trainer, config, names, train_data, val_data, eval_fn, start_epoch, rank, world, local, check=None):
    """The epoch loop of train.py:444-560: per iteration adjust_learning_rate, one trainer step (forward, criterion, scaled
    backward, unscale + clip_grad_norm_, Adam, scaler.update), the loss read back (as the reference does) and logged with
    items / s; per epoch the validation pass and the checkpoint."""
    from ..utils.comm import allreduce_mean_
    iteration = start_epoch * len(train_data)
    torch.cuda.synchronize()
    run_start = time.perf_counter()
    loss_v, ips_epoch, val_loss, val_ips = float("nan"), 0.0, 0.0, 0.0
    for epoch in range(start_epoch, args.epochs):
        torch.cuda.synchronize()
        t_epoch = time.perf_counter()
        ips_sum, n = 0.0, 0
        for i, (batch, num_items) in enumerate(train_data.epoch(epoch)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            DLLogger.log(step=(epoch, i), data={"glob_iter/iters_per_epoch": "%d/%d" % (iteration, len(train_data))})
            trainer.set_lr(adjust_learning_rate(epoch, args.learning_rate, args.anneal_steps, args.anneal_factor))
            loss = trainer.train_step(*batch)
            if world > 1:
                loss = allreduce_mean_(loss.clone())
            loss_v = float(loss.item())                      # the reference reads the loss every iteration too (train.py:476-481)
            if np.isnan(loss_v) or (check is not None and check()):
                raise Exception("loss is NaN")
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ips = num_items * world / dt
            ips_sum += ips
            n += 1
            DLLogger.log(step=(epoch, i), data={"train_loss": loss_v, "train_items_per_sec": ips, "train_iter_time": dt})
            iteration += 1
        torch.cuda.synchronize()
        ips_epoch = ips_sum / max(n, 1)
        DLLogger.log(step=(epoch,), data={"train_items_per_sec": ips_epoch, "train_loss": loss_v,
                                          "train_epoch_time": time.perf_counter() - t_epoch})
        val_loss, val_ips = validate(trainer, eval_fn, val_data, epoch, iteration, world)
        if epoch % args.epochs_per_checkpoint == 0 and args.bench_class in ("", "train"):
            save_checkpoint(trainer, epoch, config, args.output, args.model_name, local, world, names)
        if rank == 0:
            DLLogger.flush()
    torch.cuda.synchronize()
    DLLogger.log(step=tuple(), data={"run_time": time.perf_counter() - run_start, "val_loss": val_loss, "train_loss": loss_v,
                                     "train_items_per_sec": ips_epoch, "val_items_per_sec": val_ips})
    if rank == 0:
        DLLogger.flush()
    return loss_v


def main(argv=None):
    args = parse_args(argv)
    rank, world, local, dev = init_run(args, "WaveGlow_PyT")
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
        if args.use_saved_learning_rate:
            args.learning_rate = trainer.lr
    if args.synthetic_data:
        seed = (args.seed or 0) + 1000 * rank
        train_data = SyntheticEpochs(SyntheticMelAudio(args.batch_size, args.segment_length, args.n_mel_channels, dev, seed),
                                     args.iters_per_epoch)
        val_data = SyntheticEpochs(SyntheticMelAudio(args.batch_size, args.segment_length, args.n_mel_channels, dev, seed + 7, pool=2), 2)
    else:
        from torch.utils.data.dataloader import default_collate
        from .data_function import MelAudioLoader, batch_to_gpu
        mk = lambda which, train: LoaderEpochs(MelAudioLoader(args.dataset_path, filelist(args, which), args), args.batch_size,
                                               default_collate, batch_to_gpu, lambda x, y: x, dev, world, rank, args.seed, train,
                                               drop_last=train or args.bench_class == "perf-train")
        train_data, val_data = mk("training_files", True), mk("validation_files", False)

    # torch.logdet of a matrix with a negative determinant is NaN in the reference (model.py:74); the kernels return log|det| and
    # the sign, so the same condition is raised here instead of training on
    return train_loop(args, trainer, config, None, train_data, val_data, trainer.eval_loss, start_epoch, rank, world, local,
                      check=lambda: float(trainer.signs.min().item()) < 0)


if __name__ == "__main__":
    main()
