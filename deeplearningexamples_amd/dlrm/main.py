"""DLRM training entry point on MI355X with the reference's flags (subset that drives the train step).

Mirrors Recommendation/DLRM/dlrm/scripts/main.py:43-143 (flags; argparse here, absl in the reference), :387-611
(setup: device mapping, per-rank model, LR compensation) and :621-717 (loop, average_train_throughput).
    python -m torch.distributed.run --nproc-per-node 8 -m deeplearningexamples_amd.dlrm.main \
        --dataset_type synthetic_gpu --amp --batch_size 65536 --max_steps 200
"""
import argparse
import os
import time

import torch

from ..utils import dllogger
from ..utils.graph import GraphedStep
from ..utils.dist import init_from_env, is_main_process
from . import placement as P
from .data import FeatureSpec, ParametricDataset, prefetcher
from .engine import DlrmTrainer
from .model import DistributedDlrm
from .utils import LearningRateScheduler, StepTimer

CRITEO_F15 = [7912889, 33823, 582469, 245828, 11, 2209, 10667, 104, 4, 968, 15, 8165896, 17139, 2675940, 7156453,
              302516, 12022, 97, 35, 7339, 20046, 4, 7105, 1382, 63, 5554114]


def _int_list(s):
    return [int(x) for x in s.split(",")] if isinstance(s, str) else list(s)


def parse_flags(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--mode", default="train", choices=["train"])
    p.add_argument("--seed", type=int, default=12345)
    p.add_argument("--batch_size", type=int, default=65536)
    p.add_argument("--lr", type=float, default=24)
    p.add_argument("--epochs", type=int, default=1)
    p.add_argument("--max_steps", type=int, default=None)
    p.add_argument("--warmup_factor", type=int, default=0)
    p.add_argument("--warmup_steps", type=int, default=8000)
    p.add_argument("--decay_steps", type=int, default=24000)
    p.add_argument("--decay_start_step", type=int, default=48000)
    p.add_argument("--decay_power", type=int, default=2)
    p.add_argument("--decay_end_lr", type=float, default=0)
    p.add_argument("--embedding_type", default="joint_fused",
                   choices=["joint", "custom_cuda", "multi_table", "joint_sparse", "joint_fused"])
    p.add_argument("--embedding_dim", type=int, default=128)
    p.add_argument("--top_mlp_sizes", type=_int_list, default=[1024, 1024, 512, 256, 1])
    p.add_argument("--bottom_mlp_sizes", type=_int_list, default=[512, 256, 128])
    p.add_argument("--interaction_op", default="cuda_dot", choices=["cuda_dot", "dot"])
    p.add_argument("--dataset_type", default="synthetic_gpu", choices=["synthetic_gpu", "parametric"],
                   help="parametric: the split-binary files a feature_spec.yaml describes (dlrm/data/datasets.py:64-223)")
    p.add_argument("--dataset", default=None, help="directory holding feature_spec.yaml and the train/ test/ files")
    p.add_argument("--feature_spec", default="feature_spec.yaml")
    p.add_argument("--synthetic_dataset_num_entries", type=int, default=int(2 ** 15 * 1024))
    p.add_argument("--synthetic_dataset_table_sizes", type=_int_list, default=CRITEO_F15)
    p.add_argument("--synthetic_dataset_numerical_features", type=int, default=13)
    p.add_argument("--max_table_size", type=int, default=None)
    p.add_argument("--hash_indices", action="store_true")
    p.add_argument("--log_path", default="./log.json")
    p.add_argument("--print_freq", type=int, default=200)
    p.add_argument("--benchmark_warmup_steps", type=int, default=0)
    p.add_argument("--amp", action="store_true")
    p.add_argument("--cuda_graphs", action="store_true", help="capture the train step in a HIP graph (main.py:120,194-274)")
    p.add_argument("--optimized_mlp", action="store_true", default=True)
    p.add_argument("--bottom_features_ordered", action="store_true")
    p.add_argument("--freeze_mlps", action="store_true")
    p.add_argument("--freeze_embeddings", action="store_true")
    return p.parse_args(argv)


def main(argv=None):
    flags = parse_flags(argv)
    rank, world, local = init_from_env()
    device = torch.device("cuda", local)
    torch.manual_seed(flags.seed)
    sizes = list(flags.synthetic_dataset_table_sizes)
    if flags.dataset_type == "parametric":
        sizes = FeatureSpec.from_yaml(os.path.join(flags.dataset, flags.feature_spec)).get_categorical_sizes()
    if flags.max_table_size:
        sizes = [min(s, flags.max_table_size) for s in sizes]
    mapping = P.get_device_mapping(sizes, num_gpus=world)
    batch_sizes = P.get_gpu_batch_sizes(flags.batch_size, num_gpus=world) if world > 1 else (flags.batch_size,)
    mine = mapping["embedding"][rank]
    if is_main_process():
        dllogger.init([dllogger.JSONStreamBackend(dllogger.Verbosity.VERBOSE, flags.log_path),
                       dllogger.StdOutBackend(dllogger.Verbosity.DEFAULT)])
        dllogger.log(step="PARAMETER", data=vars(flags))
    model = DistributedDlrm(
        num_numerical_features=flags.synthetic_dataset_numerical_features,
        categorical_feature_sizes=[sizes[t] for t in mine],
        bottom_mlp_sizes=flags.bottom_mlp_sizes if rank == mapping["bottom_mlp"] else None,
        top_mlp_sizes=flags.top_mlp_sizes, vectors_per_gpu=mapping["vectors_per_gpu"],
        embedding_device_mapping=mapping["embedding"], world_num_categorical_features=len(sizes),
        embedding_dim=flags.embedding_dim, hash_indices=flags.hash_indices, fp16=flags.amp, device=device,
        world_size=world, bottom_features_ordered=flags.bottom_features_ordered)
    trainer = DlrmTrainer(model, lr=flags.lr, batch_sizes_per_gpu=batch_sizes, vectors_per_gpu=mapping["vectors_per_gpu"],
                          rank=rank, world_size=world, amp=flags.amp, freeze_mlps=flags.freeze_mlps,
                          freeze_embeddings=flags.freeze_embeddings)
    sched = LearningRateScheduler(flags.warmup_steps, flags.warmup_factor, flags.decay_steps, flags.decay_start_step,
                                  flags.decay_power, flags.decay_end_lr / flags.lr)
    loader = None
    if flags.dataset_type == "parametric":
        # every rank reads the numerical features only if it owns the bottom MLP and the categorical files of ITS tables
        spec = FeatureSpec.from_yaml(os.path.join(flags.dataset, flags.feature_spec))
        names = spec.get_categorical_feature_names()
        loader = ParametricDataset(spec, "train", batch_size=flags.batch_size, numerical_features_enabled=rank == mapping["bottom_mlp"],
                                   categorical_features_to_read=[names[t] for t in mine], drop_last_batch=True)
        num = cat = click = None
    else:
        g = torch.Generator(device="cpu").manual_seed(flags.seed)                 # same global batch on every rank
        num = torch.rand((flags.batch_size, flags.synthetic_dataset_numerical_features), generator=g)
        cat = torch.cat([torch.randint(0, s, (flags.batch_size, 1), generator=g) for s in sizes], dim=1)
        click = torch.randint(0, 2, (flags.batch_size,), generator=g).float().to(device)
        num = num.to(device) if rank == mapping["bottom_mlp"] else None
        cat = cat[:, mine].contiguous().to(device) if mine else None
    steps_per_epoch = len(loader) if loader is not None else max(flags.synthetic_dataset_num_entries // flags.batch_size - 1, 1)
    # CudaGraphWrapper (main.py:610-611): eager warm-up steps, one capture, then copy-in + replay per step
    step_fn = GraphedStep(trainer.train_step, enabled=flags.cuda_graphs and world == 1)
    timer, times, moving_loss = StepTimer(), [], torch.zeros(1, device=device)
    step = 0
    for epoch in range(flags.epochs):
        batches = prefetcher(iter(loader), device) if loader is not None else None
        for i in range(steps_per_epoch):
            if batches is not None:
                num, cat, click = next(batches)
                num = num.float() if num is not None else None
            timer.click(synchronize=True)
            if flags.max_steps and step > flags.max_steps:
                break
            trainer.set_lr_factor(sched.step())
            moving_loss += step_fn(num, cat, click)
            step += 1
            if timer.measured is not None and step > flags.benchmark_warmup_steps:
                times.append(timer.measured)
            if step % flags.print_freq == 0 and is_main_process():
                dllogger.log(step=(epoch, i), data={"loss": float(moving_loss.item()) / flags.print_freq,
                                                    "step_time": timer.measured, "lr": trainer.base_lr * trainer.lr_factor})
                moving_loss.zero_()
    torch.cuda.synchronize()
    if is_main_process():
        avg = flags.batch_size / (sum(times) / max(len(times), 1)) if times else 0.0
        dllogger.log(step=tuple(), data={"average_train_throughput": avg, "training_loss": float(moving_loss.item())})
        dllogger.flush()
    return trainer


if __name__ == "__main__":
    main()
