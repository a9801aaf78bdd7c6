"""Training / evaluation entry point for ResNet-50 on MI355X with the reference's command line.

Mirrors Classification/ConvNets/main.py:89-356 (flags), :359-608 (prepare_for_training: resume, loss choice, loaders, optimizer,
LR policy), :611-650 (main) and image_classification/training.py:205-254 (train), :257-311 (validate), :314-432 (train_loop:
per-epoch validation, best_prec1, checkpoint_{epoch:04}.pth.tar through utils.Checkpointer, early stopping), logger.py's metric
names (train.loss, train.compute_ips, train.total_ips, train.lr, val.top1, val.top5, val.loss).  Launch one process per GPU:
    python -m torch.distributed.run --nproc-per-node 8 -m deeplearningexamples_amd.convnets.main \
        --arch resnet50 --data-backend synthetic --batch-size 256 --amp --epochs 1 --prof 100 /data/imagenet
Flags of the reference that select machinery outside this path (DALI, TorchScript, EMA, RMSprop, other architectures) are
parsed and rejected with a message instead of an argparse error.
"""
import argparse
import os
import time

import numpy as np
import torch

from ..utils import checkpoint as ckpt
from ..utils import dllogger
from ..utils.dist import init_from_env, is_main_process
from .dataloaders import MixUpWrapper, get_pytorch_train_loader, get_pytorch_val_loader, get_synthetic_loader
from .engine import ResNetTrainer, lr_cosine_policy, lr_linear_policy, lr_step_policy
from .resnet import ResNet50


def add_parser_arguments(parser):
    p = parser
    p.add_argument("data", metavar="DIR", nargs="?", default=None, help="path to dataset (unused by --data-backend synthetic)")
    p.add_argument("--data-backend", default="synthetic", choices=["pytorch", "synthetic", "dali-gpu", "dali-cpu"],
                   help="synthetic (dataloaders.py:520-577) or pytorch (ImageFolder + PrefetchedWrapper, :354-517); DALI is not built")
    p.add_argument("--interpolation", default="bilinear")
    p.add_argument("--arch", "-a", default="resnet50", choices=["resnet50"])
    p.add_argument("-j", "--workers", default=5, type=int)
    p.add_argument("--prefetch", default=2, type=int)
    p.add_argument("--epochs", default=90, type=int)
    p.add_argument("--run-epochs", default=-1, type=int, help="run only N epochs, used for checkpointing runs")
    p.add_argument("--early-stopping-patience", default=-1, type=int)
    p.add_argument("--image-size", default=224, type=int)
    p.add_argument("--batch-size", "-b", default=256, type=int, help="mini-batch size per GPU")
    p.add_argument("--optimizer-batch-size", default=-1, type=int)
    p.add_argument("--lr", "--learning-rate", default=0.1, type=float, dest="lr")
    p.add_argument("--lr-schedule", default="step", choices=["step", "linear", "cosine"])
    p.add_argument("--end-lr", default=0.0, type=float)
    p.add_argument("--warmup", default=0, type=int)
    p.add_argument("--label-smoothing", default=0.0, type=float)
    p.add_argument("--mixup", default=0.0, type=float, metavar="ALPHA")
    p.add_argument("--optimizer", default="sgd", choices=("sgd", "rmsprop"))
    p.add_argument("--momentum", default=0.9, type=float)
    p.add_argument("--weight-decay", "--wd", default=1e-4, type=float, dest="weight_decay")
    p.add_argument("--bn-weight-decay", action="store_true")
    p.add_argument("--rmsprop-alpha", default=0.9, type=float)
    p.add_argument("--rmsprop-eps", default=1e-3, type=float)
    p.add_argument("--nesterov", action="store_true")
    p.add_argument("--print-freq", "-p", default=10, type=int)
    p.add_argument("--resume", default=None, type=str, metavar="PATH", help="path to latest checkpoint")
    p.add_argument("--static-loss-scale", type=float, default=1.0)
    p.add_argument("--prof", type=int, default=-1, help="run only N iterations per epoch (training.py:246-248)")
    p.add_argument("--amp", action="store_true", help="16-bit compute with fp32 master weights")
    p.add_argument("--amp-dtype", default="bf16", choices=["bf16", "fp16"])
    p.add_argument("--seed", default=None, type=int)
    p.add_argument("--gather-checkpoints", default=0, type=int,
                   help="keep the N last numbered checkpoints (0: all of them stay, the reference's slicing)")
    p.add_argument("--raport-file", default="experiment_raport.json")
    p.add_argument("--evaluate", action="store_true", help="evaluate checkpoint / model")
    p.add_argument("--training-only", action="store_true", help="do not evaluate")
    p.add_argument("--no-checkpoints", action="store_false", dest="save_checkpoints")
    p.add_argument("--jit", default="no", choices=["no", "script"])
    p.add_argument("--checkpoint-filename", default="checkpoint.pth.tar")
    p.add_argument("--workspace", default="./")
    p.add_argument("--memory-format", default="nchw", choices=["nchw", "nhwc"],
                   help="layout of the LOADER's batches; the kernels compute in NHWC either way")
    p.add_argument("--use-ema", default=None, type=float)
    p.add_argument("--augmentation", default=None, choices=[None, "autoaugment"])
    p.add_argument("--gpu-affinity", default="none")
    p.add_argument("--topk", default=5, type=int)
    # model arguments (models/resnet.py:222-256, models/model.py:153-172)
    p.add_argument("--num_classes", "--num-classes", default=1000, type=int, dest="num_classes")
    p.add_argument("--last_bn_0_init", default=False, type=lambda s: str(s).lower() in ("1", "true", "yes"))
    p.add_argument("--conv_init", default="fan_in", choices=["fan_in", "fan_out"])
    p.add_argument("--pretrained-from-file", default=None, type=str, metavar="PATH")
    p.add_argument("--steps-per-epoch", default=5004, type=int,
                   help="synthetic loader only: iterations per epoch (ImageNet at global batch 256: 1281167 // 256)")
    return p


def _reject_unbuilt(args):
    if args.data_backend.startswith("dali"):
        raise SystemExit("--data-backend %s: DALI is not part of this path; use pytorch or synthetic" % args.data_backend)
    if args.optimizer != "sgd":
        raise SystemExit("--optimizer rmsprop (the EfficientNet recipe) is not built; ResNet-50 trains with sgd")
    if args.use_ema is not None:
        raise SystemExit("--use-ema (the EfficientNet recipe) is not built")
    if args.augmentation is not None:
        raise SystemExit("--augmentation autoaugment needs PIL image ops on the host: not built")
    if not args.amp:
        raise SystemExit("this path computes in 16 bits with fp32 master weights: pass --amp (the reference's fp32 / TF32 "
                         "recipes are not built)")


def get_lr_policy(args):
    if args.lr_schedule == "step":
        return lr_step_policy(args.lr, [30, 60, 80], 0.1, args.warmup)
    if args.lr_schedule == "cosine":
        return lr_cosine_policy(args.lr, args.warmup, args.epochs, end_lr=args.end_lr)
    return lr_linear_policy(args.lr, args.warmup, args.epochs)


def accuracy(output, target, topk=(1,)):
    """utils.py:101-114: precision@k in percent."""
    maxk = max(topk)
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.view(1, -1).expand(maxk, -1))
    return [correct[:k].float().sum() * (100.0 / target.size(0)) for k in topk]


def train(trainer, loader, lr_fn, args, epoch, world):
    """training.py:205-254."""
    n, t_prev = 0, time.time()
    for i, (x, y) in enumerate(loader):
        lr = float(lr_fn(i))
        trainer.set_lr(lr)
        loss = trainer.train_step(x, y)
        n += 1
        if i % args.print_freq == 0:
            loss_v = float(trainer.reduced_loss(loss).item())           # utils.reduce_tensor; the only host sync of the loop
            now = time.time()
            ips = world * args.batch_size * (1 if i == 0 else args.print_freq) / max(now - t_prev, 1e-9)
            t_prev = now
            if is_main_process():
                dllogger.log(step=(epoch, i), data={"train.loss": loss_v, "train.lr": lr, "train.compute_ips": ips,
                                                    "train.total_ips": ips})
        if 0 < args.prof <= i + 1:
            break
    return n


def validate(trainer, loader, args, epoch, world):
    """training.py:257-311: evaluation-mode forward, loss + top-1 / top-k, averaged over the batches (and ranks)."""
    s1 = sk = sl = cnt = 0.0
    for i, (x, y) in enumerate(loader):
        loss, out = trainer.eval_step(x, y)
        p1, pk = accuracy(out, y, (1, args.topk))
        vals = torch.stack([p1, pk, loss.reshape(())])
        if world > 1:
            from ..utils.comm import allreduce_mean_
            vals = allreduce_mean_(vals)
        bs = x.shape[0]
        v = vals.tolist()
        s1, sk, sl, cnt = s1 + v[0] * bs, sk + v[1] * bs, sl + v[2] * bs, cnt + bs
        if 0 < args.prof <= i + 1:
            break
    cnt = max(cnt, 1.0)
    res = {"val.top1": s1 / cnt, "val.top%d" % args.topk: sk / cnt, "val.loss": sl / cnt}
    if is_main_process():
        dllogger.log(step=(epoch,), data=res)
    return res["val.top1"]


def train_loop(trainer, args, lr_policy, train_loader, train_len, val_loader, start_epoch, best_prec1, world):
    """training.py:314-432."""
    checkpointer = ckpt.Checkpointer(args.checkpoint_filename, args.workspace, args.gather_checkpoints)
    end_epoch = min(start_epoch + args.run_epochs, args.epochs) if args.run_epochs != -1 else args.epochs
    save = args.save_checkpoints and not args.evaluate
    since_best, iters, t0 = 0, 0, time.time()
    print("RUNNING EPOCHS FROM %d TO %d" % (start_epoch, end_epoch))
    for epoch in range(start_epoch, end_epoch):
        if not args.evaluate:
            iters += train(trainer, train_loader, lambda i: lr_policy(i, epoch), args, epoch, world)
        prec1 = -1
        if not args.training_only:
            prec1 = validate(trainer, val_loader, args, epoch, world)
            is_best = prec1 > best_prec1
            best_prec1 = max(prec1, best_prec1)
        else:
            is_best, best_prec1 = False, 0
        if save and is_main_process():
            checkpointer.save_checkpoint(ckpt.rn50_trainer_state(trainer, epoch + 1, best_prec1), is_best,
                                         filename="checkpoint_%04d.pth.tar" % epoch)
        if args.early_stopping_patience > 0:
            since_best = 0 if is_best else since_best + 1
            if since_best >= args.early_stopping_patience:
                break
    torch.cuda.synchronize()
    return iters, time.time() - t0, best_prec1


def main(argv=None):
    args = add_parser_arguments(argparse.ArgumentParser(description="ResNet-50 training on MI355X")).parse_args(argv)
    _reject_unbuilt(args)
    rank, world, local = init_from_env()
    device = torch.device("cuda", local)
    if args.seed is not None:
        torch.manual_seed(args.seed + local)     # main.py:381-384 (replicas are synchronised by the trainer's broadcast from rank 0)
        np.random.seed(args.seed + local)
    os.makedirs(args.workspace, exist_ok=True)
    if is_main_process():
        dllogger.init([dllogger.JSONStreamBackend(dllogger.Verbosity.VERBOSE, os.path.join(args.workspace, args.raport_file)),
                       dllogger.StdOutBackend(dllogger.Verbosity.DEFAULT)])
        dllogger.log(step="PARAMETER", data=vars(args))
    # main.py:405-416: the optimizer steps on optimizer-batch-size samples = batch_size_multiplier micro-batches per rank
    bsm = 1
    if args.optimizer_batch_size >= 0:
        tbs = world * args.batch_size
        if args.optimizer_batch_size % tbs != 0:
            raise SystemExit("--optimizer-batch-size %d is not a multiple of world x batch-size = %d" % (args.optimizer_batch_size, tbs))
        bsm = args.optimizer_batch_size // tbs
    model = ResNet50(num_classes=args.num_classes, last_bn_0_init=args.last_bn_0_init, device=device)
    if args.conv_init == "fan_out":
        for m in model.modules():
            if isinstance(m, torch.nn.Conv2d):
                torch.nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
    if args.pretrained_from_file:
        sd = torch.load(args.pretrained_from_file, map_location=device)
        model.load_state_dict({(k[7:] if k.startswith("module.") else k): v for k, v in sd.get("state_dict", sd).items()})
    dtype = torch.bfloat16 if args.amp_dtype == "bf16" else torch.float16
    trainer = ResNetTrainer(model, lr=args.lr, momentum=args.momentum, weight_decay=args.weight_decay,
                            nesterov=args.nesterov, label_smoothing=args.label_smoothing, compute_dtype=dtype,
                            static_loss_scale=args.static_loss_scale, world_size=world,
                            bn_weight_decay=args.bn_weight_decay, grad_acc_steps=bsm)
    start_epoch, best_prec1 = 0, 0.0
    if args.resume is not None:                  # main.py:419-452
        if os.path.isfile(args.resume):
            print("=> loading checkpoint '%s'" % args.resume)
            start_epoch, best_prec1 = ckpt.rn50_trainer_load(trainer, torch.load(args.resume, map_location=device, weights_only=False))
            print("=> loaded checkpoint '%s' (epoch %d)" % (args.resume, start_epoch))
            if start_epoch >= args.epochs:
                print("Launched training for %d, checkpoint already run %d" % (args.epochs, start_epoch))
                raise SystemExit(1)
        else:
            print("=> no checkpoint found at '%s'" % args.resume)
    elif args.pretrained_from_file:
        trainer.refresh_working_copies()
    if args.data_backend == "synthetic":
        get_train, get_val = get_synthetic_loader, get_synthetic_loader
    else:
        get_train, get_val = get_pytorch_train_loader, get_pytorch_val_loader
    kw = dict(workers=args.workers, memory_format=args.memory_format, prefetch_factor=args.prefetch, device=device, rank=rank,
              world=world, seed=args.seed, steps_per_epoch=args.steps_per_epoch, interpolation=args.interpolation)
    train_loader, train_len = get_train(args.data, args.image_size, args.batch_size, args.num_classes, start_epoch=start_epoch, **kw)
    if args.mixup != 0.0:
        train_loader = MixUpWrapper(args.mixup, train_loader)
    val_loader, _ = get_val(args.data, args.image_size, args.batch_size, args.num_classes, **kw)
    iters, secs, best = train_loop(trainer, args, get_lr_policy(args), train_loader, train_len, val_loader, start_epoch,
                                   best_prec1, world)
    if is_main_process():
        data = {"iterations": iters, "best_prec1": best}
        if iters:
            data["train.total_ips"] = world * args.batch_size * iters / secs
        dllogger.log(step=tuple(), data=data)
        dllogger.flush()
    print("Experiment ended")
    return trainer


if __name__ == "__main__":
    main()
