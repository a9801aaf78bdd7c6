"""DLRM training / evaluation entry point on MI355X with the reference's flags.

Mirrors Recommendation/DLRM/dlrm/scripts/main.py:43-143 (flags; argparse here, absl in the reference), :387-611
(setup: device mapping, per-rank model, LR compensation, --load_checkpoint_path, --mode test), :621-717 (loop: --test_freq /
--test_after validation passes with AUC, --auc_threshold stop, --save_checkpoint_path at the end, average_train_throughput)
and :733-835 (dist_evaluate).  Checkpoints are the reference's directory format (utils/checkpoint.py).
    python -m torch.distributed.run --nproc-per-node 8 -m deeplearningexamples_amd.dlrm.main \
        --dataset_type synthetic_gpu --amp --batch_size 65536 --max_steps 200
"""
import argparse
import os
import sys
import time

import torch

from ..utils import checkpoint as ckpt
from ..utils import dllogger
from ..utils.graph import GraphedStep
from ..utils.dist import init_from_env, is_main_process
from . import placement as P
from .data import FeatureSpec, ParametricDataset, prefetcher
from .engine import DlrmTrainer
from .model import DistributedDlrm
from .utils import LearningRateScheduler, StepTimer, SyntheticDataset, evaluate

CRITEO_F15 = [7912889, 33823, 582469, 245828, 11, 2209, 10667, 104, 4, 968, 15, 8165896, 17139, 2675940, 7156453,
              302516, 12022, 97, 35, 7339, 20046, 4, 7105, 1382, 63, 5554114]


def _int_list(s):
    return [int(x) for x in s.split(",")] if isinstance(s, str) else list(s)


def _absl_bool(v):
    """absl.flags boolean values (flags.DEFINE_boolean; scripts/main.py:81-143): true / t / 1 / false / f / 0, any case."""
    t = str(v).strip().lower()
    if t in ("true", "t", "1"):
        return True
    if t in ("false", "f", "0"):
        return False
    raise argparse.ArgumentTypeError("Non-boolean argument to boolean flag: %r" % (v,))


def _add_bool(p, name, default, help=None, short_name=None):
    """A boolean flag with absl's three spellings: --name, --name=True|False (the reference's own test scripts write
    --amp=True --cuda_graphs=True --optimized_mlp=False, tests/test_all_configs.sh:17-27) and --noname."""
    names = ["--" + name] + (["--" + short_name] if short_name else [])
    p.add_argument(*names, dest=name, nargs="?", const=True, default=default, type=_absl_bool, help=help)
    p.add_argument(*["--no" + n[2:] for n in names], dest=name, action="store_false", help=argparse.SUPPRESS)


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--mode", default="train", choices=["train", "test", "inference_benchmark"])
    p.add_argument("--seed", type=int, default=12345)
    p.add_argument("--batch_size", type=int, default=65536)
    p.add_argument("--test_batch_size", type=int, default=65536)
    p.add_argument("--lr", type=float, default=24)
    p.add_argument("--epochs", type=int, default=1)
    p.add_argument("--max_steps", type=int, default=None)
    p.add_argument("--warmup_factor", type=int, default=0)
    p.add_argument("--warmup_steps", type=int, default=8000)
    p.add_argument("--decay_steps", type=int, default=24000)
    p.add_argument("--decay_start_step", type=int, default=48000)
    p.add_argument("--decay_power", type=int, default=2)
    p.add_argument("--decay_end_lr", type=float, default=0)
    p.add_argument("--embedding_type", default="custom_cuda",
                   choices=["joint", "custom_cuda", "multi_table", "joint_sparse", "joint_fused"])
    p.add_argument("--embedding_dim", type=int, default=128)
    p.add_argument("--top_mlp_sizes", type=_int_list, default=[1024, 1024, 512, 256, 1])
    p.add_argument("--bottom_mlp_sizes", type=_int_list, default=[512, 256, 128])
    p.add_argument("--interaction_op", default="cuda_dot", choices=["cuda_dot", "dot", "cat"])
    p.add_argument("--dataset_type", default="parametric", choices=["synthetic_gpu", "parametric"],
                   help="parametric: the split-binary files a feature_spec.yaml describes (dlrm/data/datasets.py:64-223)")
    p.add_argument("--dataset", default=None, help="directory holding feature_spec.yaml and the train/ test/ files")
    p.add_argument("--feature_spec", default="feature_spec.yaml")
    p.add_argument("--synthetic_dataset_num_entries", type=int, default=int(2 ** 15 * 1024))
    p.add_argument("--synthetic_dataset_table_sizes", type=_int_list, default=26 * [10 ** 5])
    p.add_argument("--synthetic_dataset_numerical_features", type=int, default=13)
    p.add_argument("--max_table_size", type=int, default=None)
    _add_bool(p, "hash_indices", False)
    _add_bool(p, "shuffle_batch_order", False, short_name="shuffle")
    _add_bool(p, "synthetic_dataset_use_feature_spec", False)
    p.add_argument("--load_checkpoint_path", default=None, help="directory written by --save_checkpoint_path (or by the reference)")
    p.add_argument("--save_checkpoint_path", default=None)
    p.add_argument("--test_freq", type=int, default=None, help="validation pass every N steps (default: once per epoch)")
    p.add_argument("--test_after", type=float, default=0, help="no validation before this many epochs")
    p.add_argument("--auc_threshold", type=float, default=None, help="stop as soon as the validation AUC reaches this value")
    p.add_argument("--auc_device", default="GPU", choices=["GPU", "CPU"])
    p.add_argument("--base_device", default="cuda", choices=["cuda"])
    p.add_argument("--backend", default="nccl")
    p.add_argument("--inference_benchmark_batch_sizes", type=_int_list, default=[1, 64, 4096])
    p.add_argument("--inference_benchmark_steps", type=int, default=200)
    _add_bool(p, "Adam_embedding_optimizer", False)
    _add_bool(p, "Adam_MLP_optimizer", False)
    p.add_argument("--log_path", default="./log.json")
    p.add_argument("--print_freq", type=int, default=200)
    p.add_argument("--benchmark_warmup_steps", type=int, default=0)
    _add_bool(p, "amp", False)
    _add_bool(p, "cuda_graphs", False, help="capture the train step in a HIP graph (main.py:120,194-274)")
    _add_bool(p, "optimized_mlp", True)
    _add_bool(p, "bottom_features_ordered", False)
    _add_bool(p, "freeze_mlps", False)
    _add_bool(p, "freeze_embeddings", False)
    p.add_argument("--embedding_sharding", default="table", choices=["table", "row"],
                   help="table: whole tables per rank, the reference's get_device_mapping (default); row: every table cut into "
                        "row ranges over the ranks, ids routed by value and exchanged before the vectors (dlrm/row_sharded.py)")
    return p


def parse_flags(argv=None):
    f = build_parser().parse_args(argv)
    if f.mode == "inference_benchmark":
        raise SystemExit("--mode inference_benchmark: inference is outside this path (the train step and its validation pass)")
    if f.Adam_embedding_optimizer or f.Adam_MLP_optimizer:
        raise SystemExit("--Adam_*_optimizer: the path implements the reference's default SGD recipe")
    if f.interaction_op == "cat":
        raise SystemExit("--interaction_op cat: the path implements the dot interaction (cuda_dot / dot), the reference's default")
    if (f.dataset_type == "parametric" or f.synthetic_dataset_use_feature_spec) and f.dataset is None:
        raise SystemExit("--dataset: a directory with feature_spec.yaml is required for --dataset_type parametric (the default); "
                         "--dataset_type synthetic_gpu needs none")
    return f


def main(argv=None):
    flags = parse_flags(argv)
    rank, world, local = init_from_env()
    device = torch.device("cuda", local)
    torch.manual_seed(flags.seed)
    # load_feature_spec (scripts/main.py:183-190): table cardinalities AND the number of numerical features come from the spec
    # unless the data is synthetic without --synthetic_dataset_use_feature_spec
    if flags.dataset_type == "synthetic_gpu" and not flags.synthetic_dataset_use_feature_spec:
        sizes = list(flags.synthetic_dataset_table_sizes)
    else:
        fspec = FeatureSpec.from_yaml(os.path.join(flags.dataset, flags.feature_spec))
        sizes = fspec.get_categorical_sizes()
        flags.synthetic_dataset_numerical_features = fspec.get_number_of_numerical_features()
    if flags.max_table_size:
        sizes = [min(s, flags.max_table_size) for s in sizes]
    if flags.embedding_sharding == "row":
        return main_row_sharded(flags, sizes, rank, world, device)
    mapping = P.get_device_mapping(sizes, num_gpus=world)
    batch_sizes = P.get_gpu_batch_sizes(flags.batch_size, num_gpus=world) if world > 1 else (flags.batch_size,)
    mine = mapping["embedding"][rank]
    if is_main_process():
        dllogger.init([dllogger.JSONStreamBackend(dllogger.Verbosity.VERBOSE, flags.log_path),
                       dllogger.StdOutBackend(dllogger.Verbosity.DEFAULT)])
        dllogger.log(step="PARAMETER", data=vars(flags))
        if not flags.amp:
            # the reference's --amp=False is an fp32 run; this path has 16-bit MFMA kernels only: fp32 master weights and
            # optimizer as in the reference, bf16 activations / products (fp32 accumulation), no loss scaler
            print("--amp=False: products and activations in bf16 with fp32 accumulation and fp32 master weights "
                  "(there is no fp32 compute path); --amp=True is the reference's fp16 + GradScaler recipe")
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
    writer = ckpt.make_distributed_checkpoint_writer(mapping, rank, is_main_process(), dict(vars(flags)))
    if flags.load_checkpoint_path:               # main.py:493-495
        ckpt.make_distributed_checkpoint_loader(mapping, rank, device=device).load_checkpoint(model, flags.load_checkpoint_path)
    loader = test_loader = None
    test_bs = flags.test_batch_size // world * world
    test_batches = P.get_gpu_batch_sizes(test_bs, num_gpus=world) if world > 1 else (test_bs,)
    has_bottom = rank == mapping["bottom_mlp"]
    if flags.dataset_type == "parametric":
        # every rank reads the numerical features only if it owns the bottom MLP and the categorical files of ITS tables
        spec = FeatureSpec.from_yaml(os.path.join(flags.dataset, flags.feature_spec))
        names = spec.get_categorical_feature_names()
        kw = dict(numerical_features_enabled=has_bottom, categorical_features_to_read=[names[t] for t in mine])
        loader = ParametricDataset(spec, "train", batch_size=flags.batch_size, drop_last_batch=True, **kw)
        try:
            test_loader = ParametricDataset(spec, "test", batch_size=test_bs, drop_last_batch=False, **kw)
        except (KeyError, OSError, ValueError):
            test_loader = None                   # a feature spec without a test mapping: training only
        num = cat = click = None
    else:
        g = torch.Generator(device="cpu").manual_seed(flags.seed)                 # same global batch on every rank
        num = torch.rand((flags.batch_size, flags.synthetic_dataset_numerical_features), generator=g)
        cat = torch.cat([torch.randint(0, s, (flags.batch_size, 1), generator=g) for s in sizes], dim=1)
        click = torch.randint(0, 2, (flags.batch_size,), generator=g).float().to(device)
        num = num.to(device) if has_bottom else None
        cat = cat[:, mine].contiguous().to(device) if mine else None
        # data/factories.py: the synthetic test set is another SyntheticDataset (one fixed batch, num_entries / batch of them)
        gt = torch.Generator(device="cpu").manual_seed(flags.seed + 1)
        tds = SyntheticDataset(min(flags.synthetic_dataset_num_entries, 4 * test_bs), device="cpu", batch_size=test_bs,
                               numerical_features=flags.synthetic_dataset_numerical_features,
                               categorical_feature_sizes=sizes, generator=gt)
        tnum = tds._num_tensor.to(device) if has_bottom else None
        tcat = tds._cat_tensor[:, mine].contiguous().to(device) if mine else None
        tclick = tds._label_tensor.to(device)
        test_loader = [(tnum, tcat, tclick)] * len(tds)

    def run_test():
        """dist_evaluate (main.py:733-835): forward only over the test set, logits of every rank gathered, AUC + BCE loss."""
        if test_loader is None:
            return None, None
        if isinstance(test_loader, list):
            batches = test_loader
        else:
            def batches_iter():
                for tn, tc, tk in prefetcher(iter(test_loader), device):
                    yield (tn.float() if tn is not None else None), tc, tk
            batches = batches_iter()
        plan_batches = test_batches

        def gen():
            for tn, tc, tk in batches:
                n = tk.shape[0]
                if n != test_bs:                 # last batch: padded to the static test batch, outputs cut back (main.py:782-797)
                    pad = test_bs - n
                    if tn is not None:
                        tn = torch.cat([tn, torch.zeros((pad, tn.shape[1]), dtype=tn.dtype, device=tn.device)])
                    if tc is not None:
                        tc = torch.cat([tc, torch.zeros((pad, tc.shape[1]), dtype=tc.dtype, device=tc.device)])
                yield tn, tc, tk, n
        return trainer.evaluate(gen(), plan_batches)
    if flags.mode == "test":                     # main.py:506-513
        auc, vloss = run_test()
        if is_main_process():
            dllogger.log(step=tuple(), data={"best_auc": auc, "best_validation_loss": vloss})
            dllogger.flush()
        return trainer
    steps_per_epoch = len(loader) if loader is not None else max(flags.synthetic_dataset_num_entries // flags.batch_size - 1, 1)
    # CudaGraphWrapper (main.py:610-611): eager warm-up steps, one capture, then copy-in + replay per step
    step_fn = GraphedStep(trainer.train_step, enabled=flags.cuda_graphs and world == 1)
    timer, times, moving_loss = StepTimer(), [], torch.zeros(1, device=device)
    step = 0
    test_freq = flags.test_freq if flags.test_freq is not None else steps_per_epoch - 1
    best_auc, best_loss, best_epoch, t_start, hit = 0.0, 1e6, 0.0, time.time(), False
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
            gstep = steps_per_epoch * epoch + i                      # main.py:676-699
            if test_freq > 0 and gstep % test_freq == 0 and gstep > 0 and gstep / steps_per_epoch >= flags.test_after:
                auc, vloss = run_test()
                if auc is not None:
                    if is_main_process():
                        print("Epoch %d step %d. auc %.6f" % (epoch, i, auc))
                        dllogger.log(step=(epoch, i), data={"auc": auc, "validation_loss": vloss})
                    if auc > best_auc:
                        best_auc, best_epoch = auc, epoch + (i + 1) / steps_per_epoch
                    best_loss = min(best_loss, vloss)
                    if flags.auc_threshold and auc >= flags.auc_threshold:
                        print("Hit target accuracy AUC %s at epoch %.2f in %ds. " %
                              (flags.auc_threshold, gstep / steps_per_epoch, int(time.time() - t_start)))
                        hit = True
                        break
        if hit:
            break
    torch.cuda.synchronize()
    if flags.save_checkpoint_path:               # main.py:706-707
        writer.save_checkpoint(model, flags.save_checkpoint_path, epoch, step)
    if is_main_process():
        avg = flags.batch_size / (sum(times) / max(len(times), 1)) if times else 0.0
        dllogger.log(step=tuple(), data={"best_auc": best_auc, "best_validation_loss": best_loss, "best_epoch": best_epoch,
                                         "average_train_throughput": avg, "training_loss": float(moving_loss.item())})
        dllogger.flush()
    return trainer


def main_row_sharded(flags, sizes, rank, world, device):
    """--embedding_sharding row (BASELINE.json configs[3] as worded; dlrm/row_sharded.py): synthetic data, training loop only --
    checkpoints and the validation pass follow the reference's table-wise layout and stay with the default placement."""
    from .row_sharded import RowShardedDlrmTrainer, build_row_sharded_model
    if flags.dataset_type != "synthetic_gpu" or flags.load_checkpoint_path or flags.save_checkpoint_path or flags.mode != "train":
        raise SystemExit("--embedding_sharding row: synthetic training only (checkpoints / validation use the table-wise placement)")
    batch_sizes = P.get_gpu_batch_sizes(flags.batch_size, num_gpus=world) if world > 1 else (flags.batch_size,)
    if is_main_process():
        dllogger.init([dllogger.JSONStreamBackend(dllogger.Verbosity.VERBOSE, flags.log_path),
                       dllogger.StdOutBackend(dllogger.Verbosity.DEFAULT)])
        dllogger.log(step="PARAMETER", data=vars(flags))
    model, plan = build_row_sharded_model(flags.synthetic_dataset_numerical_features, sizes, flags.bottom_mlp_sizes, flags.top_mlp_sizes,
                                          rank, world, embedding_dim=flags.embedding_dim, device=device,
                                          compute_dtype=torch.float16 if flags.amp else torch.bfloat16)
    trainer = RowShardedDlrmTrainer(model, plan, flags.lr, batch_sizes, rank=rank, world_size=world, amp=flags.amp,
                                    freeze_mlps=flags.freeze_mlps, freeze_embeddings=flags.freeze_embeddings)
    sched = LearningRateScheduler(flags.warmup_steps, flags.warmup_factor, flags.decay_steps, flags.decay_start_step,
                                  flags.decay_power, flags.decay_end_lr / flags.lr)
    g = torch.Generator(device="cpu").manual_seed(flags.seed)                     # the same global batch on every rank
    num = torch.rand((flags.batch_size, flags.synthetic_dataset_numerical_features), generator=g).to(device)
    cat = torch.cat([torch.randint(0, s, (flags.batch_size, 1), generator=g) for s in sizes], dim=1).to(device)
    click = torch.randint(0, 2, (flags.batch_size,), generator=g).float().to(device)
    steps = flags.max_steps or max(flags.synthetic_dataset_num_entries // flags.batch_size - 1, 1)
    timer, times, moving_loss = StepTimer(), [], torch.zeros(1, device=device)
    for step in range(1, steps + 1):
        timer.click(synchronize=True)
        trainer.set_lr_factor(sched.step())
        moving_loss += trainer.train_step(num, cat, click)
        if timer.measured is not None and step > flags.benchmark_warmup_steps:
            times.append(timer.measured)
        if step % flags.print_freq == 0 and is_main_process():
            dllogger.log(step=(0, step), data={"loss": float(moving_loss.item()) / flags.print_freq, "step_time": timer.measured})
            moving_loss.zero_()
    torch.cuda.synchronize()
    if is_main_process():
        avg = flags.batch_size / (sum(times) / max(len(times), 1)) if times else 0.0
        dllogger.log(step=tuple(), data={"average_train_throughput": avg, "training_loss": float(moving_loss.item()),
                                         "embedding_sharding": "row"})
        dllogger.flush()
    return trainer


if __name__ == "__main__":
    main()
