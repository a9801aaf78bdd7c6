"""Training entry point for ResNet-50 on MI355X with the reference's CLI (subset that drives the train step).

Mirrors Classification/ConvNets/main.py:89-356 (flags), :359-608 (prepare_for_training) and
image_classification/training.py:205-254,314-432 (train / train_loop) and logger.py's metric names
(train.loss, train.compute_ips, train.total_ips, train.lr).  Launch one process per GPU:
    python -m torch.distributed.run --nproc-per-node 8 -m deeplearningexamples_amd.convnets.main \
        --arch resnet50 --data-backend synthetic --batch-size 256 --amp --epochs 1 --prof 100
"""
import argparse
import time

import torch

from ..utils import dllogger
from ..utils.dist import init_from_env, is_main_process
from .engine import ResNetTrainer, lr_cosine_policy, lr_linear_policy, lr_step_policy
from .resnet import ResNet50


def add_parser_arguments(parser):
    p = parser
    p.add_argument("--arch", "-a", default="resnet50", choices=["resnet50"])
    p.add_argument("--data-backend", default="synthetic", choices=["synthetic"],
                   help="only the synthetic loader is on the hot path (dataloaders.py:520-577)")
    p.add_argument("--epochs", default=90, type=int)
    p.add_argument("--batch-size", "-b", default=256, type=int, help="mini-batch size per GPU")
    p.add_argument("--optimizer-batch-size", default=-1, type=int)
    p.add_argument("--lr", "--learning-rate", default=0.1, type=float, dest="lr")
    p.add_argument("--lr-schedule", default="step", choices=["step", "linear", "cosine"])
    p.add_argument("--end-lr", default=0.0, type=float)
    p.add_argument("--warmup", default=0, type=int)
    p.add_argument("--label-smoothing", default=0.0, type=float)
    p.add_argument("--momentum", default=0.9, type=float)
    p.add_argument("--weight-decay", "--wd", default=1e-4, type=float, dest="weight_decay")
    p.add_argument("--bn-weight-decay", action="store_true")
    p.add_argument("--nesterov", action="store_true")
    p.add_argument("--amp", action="store_true", help="16-bit compute with fp32 master weights")
    p.add_argument("--amp-dtype", default="bf16", choices=["bf16", "fp16"])
    p.add_argument("--static-loss-scale", type=float, default=1.0)
    p.add_argument("--prof", type=int, default=-1, help="stop after this many iterations (training.py:246-248)")
    p.add_argument("--image-size", default=224, type=int)
    p.add_argument("--num_classes", "--num-classes", default=1000, type=int, dest="num_classes")
    p.add_argument("--seed", default=None, type=int)
    p.add_argument("--print-freq", "-p", default=10, type=int)
    p.add_argument("--workspace", default="./")
    p.add_argument("--raport-file", default="experiment_raport.json")
    p.add_argument("--steps-per-epoch", default=5004, type=int, help="ImageNet at global batch 256: 1281167 // 256")
    return p


def get_lr_policy(args):
    if args.lr_schedule == "step":
        return lr_step_policy(args.lr, [30, 60, 80], 0.1, args.warmup)
    if args.lr_schedule == "cosine":
        return lr_cosine_policy(args.lr, args.warmup, args.epochs, end_lr=args.end_lr)
    return lr_linear_policy(args.lr, args.warmup, args.epochs)


def synthetic_loader(batch_size, image_size, num_classes, device, steps, seed):
    """SynteticDataLoader: ONE fixed randn batch (+ randint targets) yielded `steps` times."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn((batch_size, 3, image_size, image_size), generator=g).to(device)
    y = torch.randint(0, num_classes, (batch_size,), generator=g).to(device)
    for _ in range(steps):
        yield x, y


def train_loop(trainer, args, lr_policy, device, rank, world):
    it_total, t_start = 0, time.time()
    for epoch in range(args.epochs):
        steps = args.steps_per_epoch if args.prof <= 0 else min(args.steps_per_epoch, args.prof)
        t_prev = time.time()
        for i, (x, y) in enumerate(synthetic_loader(args.batch_size, args.image_size, args.num_classes, device, steps,
                                                    (args.seed or 0) + rank)):
            lr = float(lr_policy(i, epoch))
            trainer.set_lr(lr)
            loss = trainer.train_step(x, y)
            it_total += 1
            if i % args.print_freq == 0:
                torch.cuda.synchronize()
                now = time.time()
                ips = world * args.batch_size * (1 if i == 0 else args.print_freq) / max(now - t_prev, 1e-9)
                t_prev = now
                if is_main_process():
                    dllogger.log(step=(epoch, i), data={"train.loss": float(loss.item()), "train.lr": lr,
                                                        "train.compute_ips": ips, "train.total_ips": ips})
        if 0 < args.prof <= it_total:
            break
    torch.cuda.synchronize()
    return it_total, time.time() - t_start


def main(argv=None):
    args = add_parser_arguments(argparse.ArgumentParser(description="ResNet-50 training on MI355X")).parse_args(argv)
    rank, world, local = init_from_env()
    device = torch.device("cuda", local)
    if args.seed is not None:
        torch.manual_seed(args.seed)        # (replicas are synchronised by the trainer's broadcast from rank 0 either way)
    if is_main_process():
        dllogger.init([dllogger.JSONStreamBackend(dllogger.Verbosity.VERBOSE, args.workspace.rstrip("/") + "/" + args.raport_file),
                       dllogger.StdOutBackend(dllogger.Verbosity.DEFAULT)])
        dllogger.log(step="PARAMETER", data=vars(args))
    if not args.amp:
        raise SystemExit("this path computes in 16 bits with fp32 master weights: pass --amp (the reference's fp32 / TF32 "
                         "recipes are not built)")
    # main.py:405-416: the optimizer steps on optimizer-batch-size samples = batch_size_multiplier micro-batches per rank
    bsm = 1
    if args.optimizer_batch_size >= 0:
        tbs = world * args.batch_size
        if args.optimizer_batch_size % tbs != 0:
            raise SystemExit("--optimizer-batch-size %d is not a multiple of world x batch-size = %d" % (args.optimizer_batch_size, tbs))
        bsm = args.optimizer_batch_size // tbs
    model = ResNet50(num_classes=args.num_classes, device=device)
    dtype = torch.bfloat16 if args.amp_dtype == "bf16" else torch.float16
    trainer = ResNetTrainer(model, lr=args.lr, momentum=args.momentum, weight_decay=args.weight_decay,
                            nesterov=args.nesterov, label_smoothing=args.label_smoothing, compute_dtype=dtype,
                            static_loss_scale=args.static_loss_scale, world_size=world,
                            bn_weight_decay=args.bn_weight_decay, grad_acc_steps=bsm)
    iters, secs = train_loop(trainer, args, get_lr_policy(args), device, rank, world)
    if is_main_process():
        dllogger.log(step=tuple(), data={"train.total_ips": world * args.batch_size * iters / secs, "iterations": iters})
        dllogger.flush()
    return trainer


if __name__ == "__main__":
    main()
