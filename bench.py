#!/usr/bin/env python
"""Headline benchmark: training samples/sec of the hot-path train step on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload dlrm|rn50|bert]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement): whole-job samples/sec over exactly K
steps bracketed by barrier + device sync, max over ranks; plus
  "roofline":     dominant kernel of the step, HIP-event timed inside the timed region, vs the MI355X peak
  "cpu_baseline": the CPU oracle (oracle/, a port of the reference's PyTorch-CPU path) timed on the host
                  cores on a bounded sample, rank 0 at N=1 only.
Inputs are synthetic, generated once and resident in HBM before the timed region (the reference's own
synthetic loaders do the same: DLRM dlrm/data/datasets.py:32-61, RN50 dataloaders.py:520-549).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0    # dense bf16/fp16 MFMA peak

CRITEO_F15 = [7912889, 33823, 582469, 245828, 11, 2209, 10667, 104, 4, 968, 15, 8165896, 17139,
              2675940, 7156453, 302516, 12022, 97, 35, 7339, 20046, 4, 7105, 1382, 63, 5554114]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=os.environ.get("DLE_BENCH_WORKLOAD", "rn50"),
                    choices=["dlrm", "rn50", "bert", "bert_acc32", "waveglow", "tacotron2"])
    ap.add_argument("--batch", type=int, default=None, help="global batch (DLRM) / per-GPU batch (RN50, BERT)")
    ap.add_argument("--dtype", default=None, choices=[None, "fp16", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--max-table-size", type=int, default=None)
    ap.add_argument("--no-nested", action="store_true",
                    help="only the headline workload (default: the other workloads of BASELINE.json's "
                         "metric run after it and are reported under \"workloads\")")
    return ap.parse_args()


def init_dist(n):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != n:
        raise SystemExit("bench.py --gpus %d must be launched with WORLD_SIZE=%d (got %d); use "
                         "python -m torch.distributed.run --nproc-per-node %d" % (n, n, world, n))
    # DLE_BENCH_BACKEND=gloo: the N > 1 path on a box with fewer GPUs than ranks (the one-GPU boxes of the test pool: RCCL refuses
    # two ranks on one device) -- the ranks share the visible GPUs round-robin and the engines stage their collectives through
    # host memory (utils/comm.py).  Everything else (self-launch, rendezvous, nested workloads, max-over-ranks timing, the one
    # JSON line) is the code the driver's 8-GPU run executes.  The default, and what the driver runs, is nccl = RCCL.
    backend = os.environ.get("DLE_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import datetime
        kw = dict(device_id=torch.device("cuda", local)) if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=int(os.environ.get("DLE_BENCH_PG_TIMEOUT", "300"))), **kw)
    return rank, world, torch.device("cuda", local)


# ------------------------------------------------------------------------------------------- DLRM
class DlrmWorkload:
    """BASELINE.json configs[3]: DLRM Criteo-shape (13 numerical + 26 tables x dim 128, criteo_f15 cardinalities),
    fp16 AMP, global batch 65536, SGD lr 24 with the reference's warm-up schedule; embeddings table-wise
    hybrid-parallel over the ranks with an RCCL all-to-all (dlrm/scripts/main.py defaults :43-143)."""

    name = "dlrm"

    def __init__(self, args, rank, world, device):
        from deeplearningexamples_amd.dlrm import placement as P
        from deeplearningexamples_amd.dlrm.model import DistributedDlrm
        from deeplearningexamples_amd.dlrm.engine import DlrmTrainer
        from deeplearningexamples_amd.dlrm.utils import LearningRateScheduler
        self.rank, self.world, self.device = rank, world, device
        sizes = list(CRITEO_F15)
        if args.max_table_size:
            sizes = [min(s, args.max_table_size) for s in sizes]
        self.sizes = sizes
        self.global_batch = args.batch or 65536
        self.dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
        mapping = P.get_device_mapping(sizes, world)
        self.batch_sizes = P.get_gpu_batch_sizes(self.global_batch, world) if world > 1 else (self.global_batch,)
        my_tables = mapping["embedding"][rank]
        torch.manual_seed(12345 + rank)
        self.model = DistributedDlrm(
            num_numerical_features=13, categorical_feature_sizes=[sizes[t] for t in my_tables],
            bottom_mlp_sizes=[512, 256, 128] if rank == mapping["bottom_mlp"] else None,
            top_mlp_sizes=[1024, 1024, 512, 256, 1], vectors_per_gpu=mapping["vectors_per_gpu"],
            embedding_device_mapping=mapping["embedding"], world_num_categorical_features=len(sizes),
            embedding_dim=128, device=device, compute_dtype=self.dtype, world_size=world)
        self.trainer = DlrmTrainer(self.model, lr=24.0, batch_sizes_per_gpu=self.batch_sizes,
                                   vectors_per_gpu=mapping["vectors_per_gpu"], rank=rank, world_size=world, amp=True)
        self.sched = LearningRateScheduler(warmup_steps=8000, warmup_factor=0, decay_steps=24000,
                                           decay_start_step=48000, decay_power=2, end_lr_factor=0)
        # SyntheticDataset: the same global batch on every rank (same seed), each rank keeps its own columns
        g = torch.Generator(device="cpu").manual_seed(2024)
        num = torch.rand((self.global_batch, 13), generator=g)
        cat = torch.cat([torch.randint(0, s, (self.global_batch, 1), generator=g) for s in sizes], dim=1)
        click = torch.randint(0, 2, (self.global_batch,), generator=g).float()
        self.num = num.to(device) if rank == mapping["bottom_mlp"] else None
        self.cat = cat[:, my_tables].contiguous().to(device) if my_tables else None
        self.click = click.to(device)
        self.samples_per_step = self.global_batch
        self.scaling = "strong"      # the reference keeps the GLOBAL batch at 64k for 1..8 GPUs (README.md:905-924)
        self.loss = None

    def single_stream(self, on):
        torch.cuda.synchronize()
        os.environ["DLE_DLRM_TWO_STREAMS"] = "0" if on else "1"          # read per call by DlrmBottom

    def step(self):
        self.trainer.set_lr_factor(self.sched.step())
        self.loss = self.trainer.train_step(self.num, self.cat, self.click)

    def config(self):
        return {"workload": "DLRM Criteo-shape (criteo_f15 cardinalities, 26 tables x dim 128, 13 numerical), "
                            "fp16 AMP, SGD lr 24 (BASELINE.json configs[3])",
                "global_batch": self.global_batch, "tables": len(self.sizes),
                "embedding_rows": int(sum(self.sizes)),
                "parallelism": "single GPU" if self.world == 1 else
                "hybrid: table-wise embeddings + all-to-all, data-parallel top MLP (dp%d)" % self.world}

    def dtype_name(self):
        return "fp16" if self.dtype == torch.float16 else "bf16"

    def cpu_baseline(self):
        """CPU oracle (port of the reference's PyTorch-CPU path) on a bounded sample of the same workload."""
        from oracle import dlrm_step_oracle as SO
        cap, batch, steps = 200000, 8192, 3
        sizes = [min(s, cap) for s in CRITEO_F15]
        state = SO.seeded_dlrm_state(sizes, 128, [512, 256, 128], [1024, 1024, 512, 256, 1], 13, 7)
        num, cat, click = SO.seeded_dlrm_batch(sizes, 13, batch, 8)
        orc = SO.DlrmOracle(state, sizes, 24.0 / 8000)
        orc.step(num, cat, click)                      # warm-up
        t0 = time.time()
        for _ in range(steps):
            orc.step(num, cat, click)
        dt = time.time() - t0
        return {"value": round(batch * steps / dt, 1), "unit": "samples/s", "cores": torch.get_num_threads(),
                "kind": "port", "steps": steps,
                "sample": "oracle/dlrm_step_oracle.py (fp32 torch-CPU restatement of the reference step), %d steps "
                          "of batch %d, table rows capped at %d (dense gradient on the capped table)" % (steps, batch, cap)}


# ------------------------------------------------------------------------------------------- RN50
class Rn50Workload:
    """BASELINE.json configs[1]: ResNet-50 v1.5, bf16 AMP, batch 256 per GPU, synthetic ImageNet 224x224
    (SynteticDataLoader: one fixed randn batch + randint labels, dataloaders.py:520-549), label smoothing 0.1,
    SGD momentum 0.875, wd 3.0517578125e-05, lr 0.256 per 256 images, cosine schedule with 8 warm-up epochs
    (configs.yml:66-91,152-157); data parallel over the ranks (gradient all-reduce, mean)."""

    name = "rn50"

    def __init__(self, args, rank, world, device):
        from deeplearningexamples_amd.convnets.resnet import ResNet50
        from deeplearningexamples_amd.convnets.engine import ResNetTrainer, lr_cosine_policy
        self.rank, self.world, self.device = rank, world, device
        self.batch = args.batch or 256
        self.dtype = torch.float16 if args.dtype == "fp16" else torch.bfloat16
        torch.manual_seed(0)                      # same initial weights on every rank (DDP broadcasts rank 0's)
        self.model = ResNet50(device=device)
        base_lr = 0.256 * world * self.batch / 256
        self.trainer = ResNetTrainer(self.model, lr=base_lr, momentum=0.875, weight_decay=3.0517578125e-05,
                                     label_smoothing=0.1, compute_dtype=self.dtype, static_loss_scale=128.0,
                                     world_size=world)
        self.lr_fn = lr_cosine_policy(base_lr, 8, 250)
        g = torch.Generator(device="cpu").manual_seed(1000 + rank)     # each rank its own batch (main.py:381-384)
        self.x = torch.randn((self.batch, 3, 224, 224), generator=g).to(device)
        self.y = torch.randint(0, 1000, (self.batch,), generator=g).to(device)
        self.samples_per_step = self.batch * world
        self.scaling = "weak"
        self.loss = None
        self.it = 0
        # DLE_RN50_GRAPH=1: the step captured in a HIP graph (utils/graph.py; the learning rate is a device word written before
        # every replay).  Opt-in: DESIGN.md section 5 has the A/B.  Multi-rank runs are always eager.
        from deeplearningexamples_amd.utils.graph import GraphedStep
        self.graphed = world == 1 and os.environ.get("DLE_RN50_GRAPH", "0") == "1"
        self._step = GraphedStep(self.trainer.train_step, enabled=self.graphed, warmup_steps=2)

    def single_stream(self, on):
        self.trainer.set_side_streams(not on)

    def step(self):
        from deeplearningexamples_amd import _cabi
        self.trainer.set_lr(float(self.lr_fn(self.it, 0)))
        if self.graphed and _cabi._timer is None:          # (the per-launch event pass of bench.py needs real launches)
            self.loss = self._step(self.x, self.y)
        else:
            self.loss = self.trainer.train_step(self.x, self.y)
        self.it += 1

    def config(self):
        return {"workload": "ResNet-50 v1.5 (PyTorch/Classification/ConvNets) synthetic ImageNet 224x224, "
                            "label smoothing 0.1, SGD momentum 0.875 (BASELINE.json configs[1])",
                "batch_per_gpu": self.batch, "global_batch": self.batch * self.world, "image": [3, 224, 224],
                "layout": "NHWC", "hip_graph": bool(self.graphed),
                "parallelism": "single GPU" if self.world == 1 else "dp%d" % self.world}

    def dtype_name(self):
        return "fp16" if self.dtype == torch.float16 else "bf16"

    def cpu_baseline(self):
        from oracle import resnet_oracle as RO
        batch, steps = 32, 3
        orc = RO.ResNet50Oracle(RO.seeded_state(3), lr=0.032)
        x, y = RO.seeded_batch(4, batch, 224)
        orc.step(x, y)
        t0 = time.time()
        for _ in range(steps):
            orc.step(x, y)
        dt = time.time() - t0
        return {"value": round(batch * steps / dt, 2), "unit": "samples/s", "cores": torch.get_num_threads(),
                "kind": "port", "steps": steps,
                "sample": "oracle/resnet_oracle.py (fp32 torch-CPU restatement of the reference step, pinned against "
                          "the reference module), %d steps of batch %d at 224x224 after 1 warm-up step" % (steps, batch)}


# ------------------------------------------------------------------------------------------- BERT
class BertWorkload:
    """BASELINE.json configs[2]: BERT-Large phase-1 pre-training, seq 128, 20 masked tokens per sequence, bf16,
    LAMB lr 6e-3 / warm-up 0.2843 / 7038 steps (scripts/configs/pretrain_config.sh:18-28), hidden / attention dropout 0.1 (bert_config.json; counter-based Philox masks),
    synthetic Wikipedia-shaped batch (run_pretraining.py:603-609); micro-batch 256 per GPU (BERT/README.md:813) with
    ONE LAMB step per micro-batch -- the reference amortises the optimizer over 32 accumulation steps, so a step here
    does strictly more work per sequence; data parallel over the ranks (gradient all-reduce, mean)."""

    name = "bert"

    def __init__(self, args, rank, world, device):
        from deeplearningexamples_amd.bert.model import BertForPreTraining, LARGE
        from deeplearningexamples_amd.bert.engine import BertTrainer
        self.rank, self.world, self.device = rank, world, device
        self.batch = args.batch or 256          # the reference's A100-80G phase-1 micro-batch (BERT/README.md:813)
        self.dtype = torch.float16 if args.dtype == "fp16" else torch.bfloat16
        torch.manual_seed(0)
        self.model = BertForPreTraining(LARGE, device=device)
        # DLE_BERT_GRAPH=1: the step captured in a HIP graph (the masked-row selection made static by max_predictions_per_seq = the
        # 20 masked tokens every sequence of this batch has; utils/graph.py).  Opt-in, single rank only.
        self.graphed = world == 1 and os.environ.get("DLE_BERT_GRAPH", "0") == "1"
        self.trainer = BertTrainer(self.model, lr=6e-3, warmup=0.2843, total_steps=7038, compute_dtype=self.dtype,
                                   world_size=world, hidden_dropout=0.1, attention_dropout=0.1, seed=42, rank=rank,
                                   max_predictions_per_seq=20 if self.graphed else None)
        from deeplearningexamples_amd.utils.graph import GraphedStep
        self._step = GraphedStep(self.trainer.train_step, enabled=self.graphed, warmup_steps=2)
        g = torch.Generator(device="cpu").manual_seed(500 + rank)
        b, s, v = self.batch, 128, LARGE["real_vocab"]
        ids = torch.randint(0, v, (b, s), generator=g)
        split = torch.randint(s // 4, 3 * s // 4, (b, 1), generator=g)
        tt = (torch.arange(s)[None, :] >= split).long()
        mask = torch.ones((b, s), dtype=torch.long)
        labels = torch.full((b, s), -1, dtype=torch.long)
        for i in range(b):
            pos = torch.randperm(s, generator=g)[:20]
            labels[i, pos] = torch.randint(0, v, (20,), generator=g)
        nsp = torch.randint(0, 2, (b,), generator=g)
        self.data = [t.to(device) for t in (ids, tt, mask, labels, nsp)]
        self.samples_per_step = self.batch * world
        self.scaling = "weak"
        self.loss = None

    def single_stream(self, on):
        torch.cuda.synchronize()
        os.environ["DLE_BERT_WGRAD_STREAM"] = "0" if on else "1"          # read per call by BertTrainer._leaf_stream

    def step(self):
        from deeplearningexamples_amd import _cabi
        if self.graphed and _cabi._timer is None:          # (the per-launch event pass of bench.py needs real launches)
            self.loss = self._step(*self.data)
        else:
            self.loss = self.trainer.train_step(*self.data)

    def config(self):
        return {"workload": "BERT-Large phase-1 pre-training (PyTorch/LanguageModeling/BERT), seq 128, 20 masked "
                            "tokens/sequence, LAMB, synthetic Wikipedia-shaped batch (BASELINE.json configs[2])",
                "batch_per_gpu": self.batch, "global_batch": self.batch * self.world, "seq_len": 128,
                "dropout": 0.1, "hip_graph": bool(self.graphed),
                "parallelism": "single GPU" if self.world == 1 else "dp%d" % self.world}

    def dtype_name(self):
        return "fp16" if self.dtype == torch.float16 else "bf16"

    def cpu_baseline(self):
        from oracle import bert_oracle as BO
        from deeplearningexamples_amd.bert.model import LARGE
        batch, steps = 4, 3
        orc = BO.BertOracle(LARGE, BO.seeded_state(LARGE, 1))
        data = BO.seeded_batch(LARGE, 2, batch)
        orc.step(*data)
        t0 = time.time()
        for _ in range(steps):
            orc.step(*data)
        dt = time.time() - t0
        return {"value": round(batch * steps / dt, 3), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
                "steps": steps,
                "sample": "oracle/bert_oracle.py (fp32 torch-CPU restatement of the reference step incl. LAMB, pinned "
                          "against the reference module), %d steps of batch %d x seq 128 after 1 warm-up step" % (steps, batch)}


class BertAcc32Workload(BertWorkload):
    """The recipe the reference's published phase-1 number is quoted on (LanguageModeling/BERT/README.md:813: batch 256 x 32
    gradient-accumulation steps on one GPU; run_pretraining.py:518-536: take_training_step per micro-batch, take_optimizer_step
    every accumulation_steps): a "step" here = 32 micro-batches of 256 sequences accumulated into the flat fp32 gradient + ONE
    LAMB step; gradients would be reduced once, during the last micro-batch's backward.  Secondary record beside `bert` (whose step
    pays LAMB per micro-batch, i.e. does strictly more work per sequence)."""

    name = "bert_acc32"
    ACC = 32

    def __init__(self, args, rank, world, device):
        super().__init__(args, rank, world, device)
        self.acc = int(os.environ.get("DLE_BERT_ACC_STEPS", str(self.ACC)))
        self.graphed = False
        self.trainer.grad_divisor = self.acc
        self.samples_per_step = self.batch * self.acc * world

    def step(self):
        tr = self.trainer
        tot = None
        for i in range(self.acc):
            tr._reduce_now = i == self.acc - 1               # only the last micro-batch communicates (run_pretraining.py:679-681)
            loss, dlog, dnsp = tr.forward(*self.data)
            tr.backward(dlog, dnsp, accumulate=i > 0)
            tot = loss if tot is None else tot + loss
        tr.optimizer_step()
        self.loss = tot / self.acc

    def config(self):
        c = super().config()
        c["workload"] = c["workload"].replace("LAMB,", "LAMB every %d accumulated micro-batches (BERT/README.md:813)," % self.acc)
        c["accumulation_steps"] = self.acc
        c["sequences_per_optimizer_step"] = self.samples_per_step
        return c

    def cpu_baseline(self):
        return None                                           # (the CPU leg of `bert` is the baseline of both records)


class WaveGlowWorkload:
    """BASELINE.json configs[4], the WaveGlow half (a "next" row -- not part of the bar): the reference's
    default network (12 flows, 8 WN layers x 512 channels, waveglow/arg_parser.py:38-64), fp16 AMP, batch 10 segments of 8000
    audio samples, Adam lr 1e-4, grad-clip 65504 (platform/DGXA100_waveglow_AMP_1NGPU_train.sh); synthetic LJSpeech-shaped
    pair (80-bin mel of 32 frames per segment).  Unit = audio samples / s, the reference's `train_items_per_sec`
    (train.py:466-468, waveglow/data_function.py:80-85).  The coupling nets start at the reference's initial state
    (`end` = 0), so log_s = 0 at the first step exactly as in the reference's first iterations."""

    name = "waveglow"

    def __init__(self, args, rank, world, device):
        from deeplearningexamples_amd.waveglow.engine import WaveGlowTrainer
        from deeplearningexamples_amd.waveglow.model import DEFAULT_CONFIG, WaveGlow
        self.rank, self.world, self.device = rank, world, device
        self.batch = args.batch or 10
        self.segment = 8000
        self.dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
        torch.manual_seed(0)
        self.model = WaveGlow(**DEFAULT_CONFIG, device=device)
        self.trainer = WaveGlowTrainer(self.model, lr=1e-4, weight_decay=0.0, grad_clip_thresh=65504.0, sigma=1.0,
                                       compute_dtype=self.dtype, world_size=world)
        g = torch.Generator(device="cpu").manual_seed(900 + rank)
        frames = (self.segment + 255) // 256
        self.mel = (torch.randn(self.batch, 80, frames, generator=g) * 2.0 - 5.0).to(device)
        self.audio = (torch.randn(self.batch, self.segment, generator=g) * 0.2).clamp_(-1, 1).to(device)
        self.samples_per_step = self.batch * self.segment * world
        self.scaling = "weak"
        self.loss = None
        # ~1,300 launches per step.  DLE_WG_GRAPH=1 captures the step in a HIP graph (utils/graph.py, the reference's
        # CudaGraphWrapper idea); measured equal to the eager step (50.2 vs 50.3 ms on the first path: the stream never runs
        # dry), so eager is the default.  Multi-rank runs are always eager (the bucket all-reduce is not captured).
        from deeplearningexamples_amd.utils.graph import GraphedStep
        self.graphed = world == 1 and os.environ.get("DLE_WG_GRAPH", "0") == "1"
        self._step = GraphedStep(self.trainer.train_step, enabled=self.graphed, warmup_steps=2)

    def step(self):
        from deeplearningexamples_amd import _cabi
        if self.graphed and _cabi._timer is not None:      # the per-launch event pass of bench.py needs real launches
            self.loss = self.trainer.train_step(self.mel, self.audio)
        else:
            self.loss = self._step(self.mel, self.audio)

    def config(self):
        return {"workload": "WaveGlow training (PyTorch/SpeechSynthesis/Tacotron2 -m WaveGlow), 12 flows x 8 layers x 512 "
                            "channels, synthetic LJSpeech-shaped mel / audio segments (BASELINE.json configs[4], WaveGlow half)",
                "batch_per_gpu": self.batch, "segment_length": self.segment, "unit_note": "audio samples / s",
                "hip_graph": bool(self.graphed),
                "parallelism": "single GPU" if self.world == 1 else "dp%d" % self.world}

    def dtype_name(self):
        return "fp16" if self.dtype == torch.float16 else "bf16"

    @staticmethod
    def flops_per_audio_sample():
        """3 x forward: per group of 8 samples, 12 flows x (8 layers x (3*512*1024 + 640*1024 + 512*1024) - 512*512 [last
        res_skip has the skip half only] + start / end) MAC, + the upsampling GEMM (320 x 20480 MAC per 256 samples)."""
        nc, nl, nf = 512, 8, 12
        per_row = nf * (nl * (3 * nc * 2 * nc + 640 * 2 * nc + nc * 2 * nc) - nc * nc) + sum(
            2 * nh * nc + nc * nh for nh in (4, 4, 4, 4, 3, 3, 3, 3, 2, 2, 2, 2))
        per_sample = per_row / 8.0 + 320 * 20480 / 256.0
        return 3 * 2 * per_sample

    def cpu_baseline(self):
        from oracle import waveglow_oracle as WO
        from deeplearningexamples_amd.waveglow.model import DEFAULT_CONFIG
        seg, steps = 1024, 3
        case = dict(cfg=DEFAULT_CONFIG, seed=3, batch=1, segment=seg)
        p = {k: v.clone().requires_grad_(True) for k, v in WO.seeded_state(DEFAULT_CONFIG, 3).items()}
        mel, audio = WO.seeded_inputs(case)
        opt = torch.optim.Adam(list(p.values()), lr=1e-4)

        def one():
            opt.zero_grad()
            WO.waveglow_loss(p, DEFAULT_CONFIG, mel, audio, 1.0).backward()
            torch.nn.utils.clip_grad_norm_(list(p.values()), 65504.0)
            opt.step()
        one()
        t0 = time.time()
        for _ in range(steps):
            one()
        dt = time.time() - t0
        return {"value": round(seg * steps / dt, 1), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
                "steps": steps,
                "sample": "oracle/waveglow_oracle.py (fp32 torch-CPU restatement of WaveGlow + WaveGlowLoss, pinned against "
                          "the reference's modules) + clip_grad_norm_ + torch.optim.Adam, %d steps of 1 x %d audio samples "
                          "after 1 warm-up step" % (steps, seg)}


class Tacotron2Workload:
    """BASELINE.json configs[4], the Tacotron2 half (a "next" row -- not part of the bar): the reference's default
    network (tacotron2/arg_parser.py:40-107, 28.2 M parameters), fp16 AMP, batch 128 (platform/DGXA100_tacotron2_AMP_1NGPU_train.sh),
    Adam lr 1e-3, weight decay 1e-6, grad-clip 1.0; synthetic LJSpeech-shaped batch from TextMelCollate's layout: text lengths
    60..160 symbols sorted descending, mel lengths ~5.4 frames per symbol (320..860 frames), zero padded, teacher forcing.  Unit =
    mel frames / s, the reference's `train_items_per_sec` (sum of output_lengths per iteration, tacotron2/data_function.py:139-151)."""

    name = "tacotron2"

    def __init__(self, args, rank, world, device):
        from deeplearningexamples_amd.tacotron2.engine import Tacotron2Trainer
        from deeplearningexamples_amd.tacotron2.model import Tacotron2
        self.rank, self.world, self.device = rank, world, device
        self.batch = args.batch or 128
        self.dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
        torch.manual_seed(0)
        self.model = Tacotron2(device=device)
        self.trainer = Tacotron2Trainer(self.model, lr=1e-3, weight_decay=1e-6, grad_clip_thresh=1.0, compute_dtype=self.dtype,
                                        world_size=world, rank=rank)
        g = torch.Generator(device="cpu").manual_seed(700 + rank)
        tl = torch.sort(torch.randint(60, 161, (self.batch,), generator=g), descending=True).values
        ml = (tl.float() * (5.0 + 0.8 * torch.rand(self.batch, generator=g))).long()
        text = torch.zeros(self.batch, int(tl.max()), dtype=torch.long)
        mel = torch.zeros(self.batch, 80, int(ml.max()))
        gate = torch.zeros(self.batch, int(ml.max()))
        for i in range(self.batch):
            text[i, :tl[i]] = torch.randint(1, 148, (int(tl[i]),), generator=g)
            mel[i, :, :ml[i]] = torch.randn(80, int(ml[i]), generator=g) * 1.5 - 4.0
            gate[i, ml[i] - 1:] = 1
        self.data = [t.to(device) for t in (text, tl, mel, gate)]
        self.samples_per_step = int(ml.sum()) * world
        self.frames, self.text_len = int(ml.max()), int(tl.max())
        self.scaling = "weak"
        self.loss = None
        # ~17,000 launches per iteration at 10-20 us of kernel each: eager, the step is bound by the HOST (Python + ctypes,
        # ~14 us per launch).  The whole iteration is captured in a HIP graph (utils/graph.py, the reference's CudaGraphWrapper
        # idea; the synthetic batch has one shape) and replayed; DLE_T2_GRAPH=0 runs it eagerly.  Multi-rank runs are eager.
        from deeplearningexamples_amd.utils.graph import GraphedStep
        self.graphed = world == 1 and os.environ.get("DLE_T2_GRAPH", "1") == "1"
        self._step = GraphedStep(self.trainer.train_step, enabled=self.graphed, warmup_steps=1)

    def step(self):
        from deeplearningexamples_amd import _cabi
        if self.graphed and _cabi._timer is not None:      # the per-launch event pass of bench.py needs real launches
            self.loss = self.trainer.train_step(*self.data)
        else:
            self.loss = self._step(*self.data)

    def config(self):
        return {"workload": "Tacotron2 training (PyTorch/SpeechSynthesis/Tacotron2 -m Tacotron2), default network, teacher forcing, "
                            "synthetic LJSpeech-shaped padded batch (BASELINE.json configs[4], Tacotron2 half)",
                "batch_per_gpu": self.batch, "max_text_len": self.text_len, "max_mel_frames": self.frames,
                "mel_frames_per_step": self.samples_per_step // self.world, "unit_note": "mel frames / s",
                "hip_graph": bool(self.graphed),
                "parallelism": "single GPU" if self.world == 1 else "dp%d" % self.world}

    def dtype_name(self):
        return "fp16" if self.dtype == torch.float16 else "bf16"

    def cpu_baseline(self):
        from oracle import tacotron2_oracle as TO
        cfg = TO.TACOTRON2_DEFAULT
        steps = 3
        case = dict(cfg=cfg, seed=5, text_lengths=[40, 32], mel_lengths=[64, 56])
        p = {k: v.clone().requires_grad_(True) for k, v in TO.seeded_state(cfg, 5).items()}
        text, tl, mel, gate, ml = TO.seeded_batch(case)
        opt = torch.optim.Adam(list(p.values()), lr=1e-3, weight_decay=1e-6)

        def one(seed):
            opt.zero_grad()
            TO.tacotron2_loss(p, cfg, text, tl, mel, gate, TO.MaskStream(seed))[0].backward()
            torch.nn.utils.clip_grad_norm_(list(p.values()), 1.0)
            opt.step()
        one(0)
        t0 = time.time()
        for i in range(steps):
            one(1 + i)
        dt = time.time() - t0
        return {"value": round(float(ml.sum()) * steps / dt, 1), "unit": "samples/s", "cores": torch.get_num_threads(),
                "kind": "port", "steps": steps,
                "sample": "oracle/tacotron2_oracle.py (fp32 torch-CPU restatement of Tacotron2 + Tacotron2Loss, pinned against the "
                          "reference's modules under shared dropout masks) + clip_grad_norm_ + Adam, %d steps of 2 utterances, "
                          "%d mel frames, after 1 warm-up step" % (steps, int(ml.sum()))}


WORKLOADS = {"dlrm": DlrmWorkload, "rn50": Rn50Workload, "bert": BertWorkload, "bert_acc32": BertAcc32Workload,
             "waveglow": WaveGlowWorkload, "tacotron2": Tacotron2Workload}
NESTED_STEPS = {"rn50": (30, 8), "bert": (20, 3), "bert_acc32": (3, 1), "dlrm": (100, 20), "waveglow": (20, 3), "tacotron2": (8, 2)}   # (timed steps, warm-up)

REFERENCE_PUBLISHED = {
    "rn50": {"value": 2470, "unit": "img/s", "hardware": "1x A100 80GB, mixed precision, bs 256",
             "source": "PyTorch/Classification/ConvNets/resnet50v1.5/README.md:598-599"},
    "bert": {"value": 580, "unit": "seq/s", "hardware": "1x A100 80GB, fp16, phase 1 seq 128",
             "source": "PyTorch/LanguageModeling/BERT/README.md:813-814"},
    "bert_acc32": {"value": 580, "unit": "seq/s", "hardware": "1x A100 80GB, fp16, phase 1 seq 128, batch 256 x 32 accumulation steps",
                   "source": "PyTorch/LanguageModeling/BERT/README.md:813-814"},
    "dlrm": {"value": 4.02e6, "unit": "samples/s", "hardware": "1x A100 80GB, AMP + CUDA graphs, bs 64k",
             "source": "PyTorch/Recommendation/DLRM/README.md:923-924"},
    "waveglow": {"value": 149479, "unit": "audio samples/s", "hardware": "1x A100 40GB, AMP, bs 10",
                 "source": "PyTorch/SpeechSynthesis/Tacotron2/README.md:704-706"},
    "tacotron2": {"value": 26484, "unit": "mel frames/s", "hardware": "1x A100 40GB, AMP, bs 128",
                  "source": "PyTorch/SpeechSynthesis/Tacotron2/README.md:696-698"},
}


# Algorithmic work per sample and the bound of each workload: SURVEY.md section 8(d) / BASELINE.md section 2.
#   RN50 v1.5 224x224 train  24.54 GFLOP / image     (3 x 8.178 GFLOP forward; dense contraction -> MFMA)
#   BERT-Large S=128 train   240.6 GFLOP / sequence  (3 x 80.2 GFLOP forward; GEMMs are 98 % -> MFMA)
#   DLRM Criteo-shape train  ~53 KB HBM / sample on the embedding path (gather read 13.3 KB + fp16 write 6.7 KB +
#                            sparse gradient 13.3 KB + SGD row read-modify-write 26.6 KB, minus cache hits -> HBM)
#   WaveGlow train           ~196 MFLOP / audio sample (WaveGlowWorkload.flops_per_audio_sample; GEMMs -> MFMA)
#   Tacotron2 train          ~0.22 GFLOP / mel frame (3 x 2 x [attention LSTM 7.3 M + decoder LSTM 10.5 M + location / query / projection
#                            0.3 M MAC per frame + the text-side work amortised]; latency-bound loop of small GEMMs -> priced on MFMA)
WORK_PER_SAMPLE = {"rn50": ("mfma", 24.54e9), "bert": ("mfma", 240.6e9), "bert_acc32": ("mfma", 240.6e9), "dlrm": ("hbm", 53.0e3),
                   "waveglow": ("mfma", WaveGlowWorkload.flops_per_audio_sample()), "tacotron2": ("mfma", 0.22e9)}
# entry points whose launches are matrix-core kernels (gemm2_kernel / gemm_kernel / conv3x3_kernel instantiations)
MFMA_FAMILIES = ("dle_gemm", "dle_gemm_batched", "dle_attention_fwd", "dle_attention_bwd", "dle_conv2d_fwd", "dle_conv2d_fwd_colstats", "dle_conv2d_dgrad", "dle_conv2d_dgrad_s2",
                 "dle_conv2d_wgrad", "dle_attention_fwd", "dle_attention_bwd")


def lookup_traffic(kernel_key):
    """HBM-side bytes per launch of `kernel_key` from the committed PMC passes (profiles/traffic.json, written by
    tools/collect_traffic.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs over a replay of exactly this
    launch, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  None when that launch was not profiled."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")
    try:
        table = json.load(open(path))
    except (OSError, ValueError):
        return None
    rec = table.get(kernel_key)
    return float(rec["traffic_bytes_per_launch"]) if rec else None


def traffic_source():
    """Where `roofline.traffic` comes from: it is a LOOKUP in the committed counter table, not a counter read in this run."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")
    try:
        meta = json.load(open(path)).get("_meta", {})
    except (OSError, ValueError):
        return None
    return "profiles/traffic.json (rocprofv3 --pmc passes of round %s, tools/collect_traffic.sh; looked up by launch shape, " \
           "not measured in this run)" % meta.get("round", "4")


def comm_info():
    """The process group this line was measured under (a SCALE record can be checked against it)."""
    import torch.distributed as dist
    from deeplearningexamples_amd.utils import comm, rccl
    path = "dle_rccl_* (C ABI over librccl.so, DLE_COMM=rccl)" if rccl.enabled() else "torch.distributed ProcessGroup"
    if dist.is_available() and dist.is_initialized():
        # ranks_seen: a SUM all-reduce of ones over the path the steps use (+ ncclCommCount on the direct path) -- EVERY rank calls
        # comm_info(), so the collective is complete; a SCALE record can be checked against it
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        return {"backend": dist.get_backend(), "world": dist.get_world_size(), "ranks_seen": comm.ranks_seen(dev), "path": path}
    return {"backend": None, "world": 1, "ranks_seen": 1, "path": path}


def roofline_from(timer, steps, wl_name, samples_per_step_per_gpu, ms_per_step):
    """Roofline of the workload's DOMINANT KERNEL FAMILY plus the whole-step fraction.

    Launches are HIP-event timed per C-ABI call over an instrumented pass of the same K steps, on the launch stream.
    They are aggregated by ENTRY POINT across all shapes (an entry point is one HIP kernel template: dle_gemm ->
    gemm2_kernel, dle_conv2d_dgrad -> conv3x3_kernel / gemm2_kernel<..4,5..>, dle_bn_bwd_apply -> bn_bwd_apply_kernel),
    so the convolution / GEMM launches of a step are ONE row, not 30 shape buckets.  The heaviest (entry point, shape)
    pairs are re-timed by replaying the recorded launch back to back (a per-call event pair leaves the queue idle
    between short kernels and inflates them); a family's time = sum over its shapes of avg duration x calls.
    `achieved` = algorithmic flops (MFMA families) or bytes (the others) of the family per step / its time per step;
    `step_frac` = SURVEY 8(d)'s per-sample work x samples per step / ms_per_step / peak (the number the target's
    ">= 70 % of the dominant roofline" is read against for the whole step)."""
    rows = timer.report() if timer else []
    if not rows:
        return None, []
    nreplay = int(os.environ.get("DLE_BENCH_REPLAY", "24"))
    for i, a in enumerate(rows):
        # launches whose operands fit in L2 (< 32 MB algorithmic bytes: the few-row GEMMs of the recurrent loops) are replayed with
        # a cache flush in front of each: back to back they would find their weights in L2, inside the step they do not
        cold = 0 < a["bytes"] / max(a["calls"], 1) < 32e6
        rep = timer.replay(a["name"], a["tag"], cold=cold) if i < nreplay else None
        a["avg_ms"] = rep if rep is not None else a["ms"] / a["calls"]
        a["timing"] = ("replay-cold" if cold else "replay") if rep is not None else "event-pair"
    fam = {}
    for a in rows:
        f = fam.setdefault(a["name"], {"name": a["name"], "ms": 0.0, "calls": 0, "flops": 0.0, "bytes": 0.0, "shapes": []})
        f["ms"] += a["avg_ms"] * a["calls"]
        f["calls"] += a["calls"]
        f["flops"] += a["flops"]
        f["bytes"] += a["bytes"]
        f["shapes"].append(a)
    fams = sorted(fam.values(), key=lambda f: -f["ms"])
    top = fams[0]
    bound_step, work = WORK_PER_SAMPLE[wl_name]
    # The family's bound is the side of the roofline its ALGORITHMIC arithmetic intensity puts it on: flops / byte below the
    # ridge (peak flops / peak bytes = 312) -> HBM, above -> MFMA.  (ResNet-50's GEMM launches are mostly K <= 256 1x1
    # convolutions at ~100 flop/B: no schedule can run them faster than their bytes at 8 TB/s.)  Both fractions are kept.
    t_s = top["ms"] * 1e-3
    is_mfma_kernel = top["name"] in MFMA_FAMILIES and top["flops"] > 0
    ach_f = top["flops"] / t_s / 1e12 if top["flops"] else 0.0
    ach_b = top["bytes"] / t_s / 1e9 if top["bytes"] else 0.0
    ridge = MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
    ai = top["flops"] / top["bytes"] if (top["flops"] and top["bytes"]) else None
    if is_mfma_kernel and (ai is None or ai >= ridge):
        r = {"bound": "mfma", "achieved": round(ach_f, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
             "frac": round(ach_f / MFMA_PEAK_TFLOPS, 4)}
    else:
        r = {"bound": "hbm", "achieved": round(ach_b, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(ach_b / HBM_PEAK_GBS, 4)}
    r.update({"arithmetic_intensity": round(ai, 1) if ai else None, "ridge": round(ridge, 1),
              "frac_mfma": round(ach_f / MFMA_PEAK_TFLOPS, 4) if is_mfma_kernel else None,
              "frac_hbm": round(ach_b / HBM_PEAK_GBS, 4) if top["bytes"] else None})
    by_time = sorted(top["shapes"], key=lambda a: -a["avg_ms"] * a["calls"])
    big = by_time[0]
    shape_key = lambda a: a["name"] + ("[" + a["tag"] + "]" if a["tag"] else "")
    bigname = shape_key(big)
    # PMC traffic: of the heaviest launch when profiles/traffic.json holds it, else of the heaviest PROFILED launch of the family
    traffic, traffic_shape, traffic_alg = None, None, None
    fam_tb = lookup_traffic("family:%s@%s" % (top["name"], wl_name))
    if fam_tb is not None:
        # families of many small launches (the few-row GEMMs of Tacotron2 / WaveGlow): the counters are summed over every launch
        # of the family inside the real step (tools/pmc_family.sh) -- a replayed single launch would find its weights in L2
        traffic, traffic_shape = fam_tb, "average over all launches of the family inside the step (tools/pmc_family.sh)"
        traffic_alg = top["bytes"] / top["calls"] if top["bytes"] else None
    for a in (by_time if fam_tb is None else []):
        tb = lookup_traffic(shape_key(a))
        if tb is not None:
            traffic, traffic_shape = tb, shape_key(a)
            traffic_alg = a["bytes"] / a["calls"] if a["bytes"] else None
            break
    step_rate = work * samples_per_step_per_gpu / (ms_per_step * 1e-3)
    step_peak = MFMA_PEAK_TFLOPS * 1e12 if bound_step == "mfma" else HBM_PEAK_GBS * 1e9
    r.update({"traffic": traffic, "traffic_shape": traffic_shape,
              "traffic_over_algorithmic": round(traffic / traffic_alg, 3) if (traffic and traffic_alg) else None,
              "kernel": top["name"],
              "avg_launch_us": round(top["ms"] / top["calls"] * 1e3, 2),
              "launches_per_step": round(top["calls"] / steps, 2),
              "ms_per_step": round(top["ms"] / steps, 4),
              "aggregation": "all launches of this entry point (one HIP kernel template) across shapes",
              "heaviest_shape": bigname, "heaviest_shape_us": round(big["avg_ms"] * 1e3, 2),
              "algorithmic_flops_per_step": top["flops"] / steps, "algorithmic_bytes_per_step": top["bytes"] / steps,
              "step_bound": bound_step,
              "step_achieved": round(step_rate / (1e12 if bound_step == "mfma" else 1e9), 1),
              "step_unit": "TFLOP/s" if bound_step == "mfma" else "GB/s",
              "step_frac": round(step_rate / step_peak, 4),
              "step_work_per_sample": work})
    nb = int(os.environ.get("DLE_BENCH_BREAKDOWN", "12"))
    def counter_gbs(f):
        """HBM-side GB/s of a family from the committed PMC record of its launch (profiles/traffic.json), when there is one: the
        ALGORITHMIC figure beside it charges e.g. the sparse update a full row read-modify-write per LOOKUP, which
        duplicate-heavy batches never move -- dle_emb_sgd_dedup: 33.5 KB / sample algorithmic, 1.40 GB per launch counted."""
        tb = lookup_traffic(f["name"]) or lookup_traffic(f["name"].replace("_ws", ""))
        return round(tb * f["calls"] / (f["ms"] * 1e-3) / 1e9, 1) if (tb and f["ms"]) else None
    breakdown = [{"kernel": f["name"], "ms_per_step": round(f["ms"] / steps, 4),
                  "calls_per_step": round(f["calls"] / steps, 2),
                  "tflops": round(f["flops"] / (f["ms"] * 1e-3) / 1e12, 1) if f["flops"] else None,
                  "gbs": round(f["bytes"] / (f["ms"] * 1e-3) / 1e9, 1) if f["bytes"] else None,
                  "gbs_counter": counter_gbs(f)}
                 for f in fams[:nb]]
    if len(fams) > nb:          # everything below the cut in ONE row, so that the rows add up to the timed kernel total
        rest = fams[nb:]
        breakdown.append({"kernel": "(other: %d entry points)" % len(rest), "ms_per_step": round(sum(f["ms"] for f in rest) / steps, 4),
                          "calls_per_step": round(sum(f["calls"] for f in rest) / steps, 2), "tflops": None, "gbs": None})
    # sum over every C-ABI launch of the step, each kernel alone on the chip (single-stream pass): what the step would take with
    # no overlap between streams, no launch gap and no ATen kernel -- next to ms_per_step it shows what the breakdown leaves out
    r["kernel_sum_ms_per_step"] = round(sum(f["ms"] for f in fams) / steps, 4)
    if os.environ.get("DLE_BENCH_SHAPES"):
        rows.sort(key=lambda a: -a["avg_ms"] * a["calls"])
        breakdown += [{"kernel": a["name"] + ("[" + a["tag"] + "]" if a["tag"] else ""),
                       "ms_per_step": round(a["avg_ms"] * a["calls"] / steps, 4),
                       "calls_per_step": round(a["calls"] / steps, 2), "timing": a["timing"]}
                      for a in rows[:int(os.environ["DLE_BENCH_SHAPES"])]]
    return r, breakdown


def sample_shader_clock(wl, steps, sample=True):
    """The shader clock (MHz) while the workload runs, sampled on the host every 20 ms over `steps` UNTIMED steps after the
    timed region (torch.cuda.clock_rate() = amdsmi's current gfx clock).  Context for `roofline.frac`: the MFMA peak of
    MI355X_MICROARCH.md assumes 2.4 GHz; under sustained matrix load the power management runs the chip well below that
    (DESIGN.md 0.1).  None when the box has no amdsmi binding."""
    import threading
    try:
        first = int(torch.cuda.clock_rate()) if sample else None
    except Exception:
        first = None
    vals, stop = [first], threading.Event()

    def loop():
        while not stop.is_set():
            try:
                vals.append(int(torch.cuda.clock_rate()))
            except Exception:
                return
            stop.wait(0.02)
    th = threading.Thread(target=loop, daemon=True) if first is not None else None
    if th is not None:
        th.start()
    for _ in range(steps):                              # (every rank runs them: the collectives inside a step need all ranks)
        wl.step()
    torch.cuda.synchronize()
    if th is None:
        return None
    stop.set()
    th.join(timeout=2.0)
    busy = vals[1:] or vals
    return {"mean": round(sum(busy) / len(busy), 1), "min": min(busy), "max": max(busy), "samples": len(busy),
            "source": "torch.cuda.clock_rate() (amdsmi current gfx clock) every 20 ms over %d untimed steps" % steps}


def run_workload(name, args, rank, world, device, steps, warmup):
    """Build the workload, warm up, time exactly `steps` steps (barrier + synchronize on both sides, nothing else
    inside), then the instrumented pass for the roofline.  Returns the record of this workload (rank 0) or None."""
    from deeplearningexamples_amd import _cabi
    inject = os.environ.get("DLE_BENCH_FAIL_NESTED", "")           # "<workload>:<rank>": tests of the nested-failure path
    if inject and inject.split(":")[0] == name and int(inject.split(":")[1]) == rank:
        raise RuntimeError("injected failure of %s on rank %d (DLE_BENCH_FAIL_NESTED)" % (name, rank))
    wl = WORKLOADS[name](args, rank, world, device)
    for _ in range(warmup):
        wl.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    sclk = None
    if os.environ.get("DLE_BENCH_SCLK", "1") == "1":
        sclk = sample_shader_clock(wl, max(2, min(steps, 10)), sample=rank == 0)
    # the same K steps again with a HIP-event pair around every C-ABI launch (on the launch stream) for the per-kernel
    # durations behind "roofline"; kept out of the region above (the event pairs inflate the step); every rank runs it
    timer = None
    if not args.no_kernel_timer:
        timer = _cabi.KernelTimer()
        # per-kernel durations are taken with every kernel ALONE on the chip: the engines' second streams (weight gradients,
        # downsample branch, embedding update) are folded into one stream for this pass only -- side by side a kernel's event
        # pair would also measure its neighbour's share of the machine
        solo = getattr(wl, "single_stream", None)
        if solo is not None:
            solo(True)
        _cabi.set_timer(timer)
        for _ in range(steps):
            wl.step()
        torch.cuda.synchronize()
        _cabi.set_timer(None)
        if solo is not None:
            solo(False)
        if world > 1:
            dist.barrier()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = float(wl.loss.item()) if wl.loss is not None else None
    rec = None
    if rank == 0:
        ms = elapsed / steps * 1e3
        roof, breakdown = roofline_from(timer, steps, name, wl.samples_per_step / world, ms)
        if roof is not None:
            roof["sclk_mhz"] = sclk
        rec = {"value": round(wl.samples_per_step * steps / elapsed, 1), "unit": "samples/s", "steps": steps,
               "warmup": warmup, "ms_per_step": round(ms, 4), "scaling": wl.scaling, "dtype": wl.dtype_name(),
               "config": wl.config(), "final_loss": loss, "roofline": roof, "kernel_breakdown": breakdown,
               # context only (other hardware, so vs_baseline stays null): BASELINE.md's published 1-GPU numbers
               "reference_published": REFERENCE_PUBLISHED.get(name)}
    del wl
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return rec


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks through torch.distributed.run ourselves."""
    import subprocess
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    raise SystemExit(subprocess.call(cmd, env=env))


UNITS = {"rn50": "img/s", "bert": "seq/s", "bert_acc32": "seq/s", "dlrm": "samples/s", "waveglow": "audio samples/s", "tacotron2": "mel frames/s"}
SHORT = {"rn50": "RN50 v1.5 bs256/GPU 224^2 (configs[1])", "bert": "BERT-L ph1 s128 bs256/GPU dropout .1 LAMB (configs[2])",
         "bert_acc32": "BERT-L ph1 s128 bs256 x 32 accumulation steps per LAMB step (README.md:813 recipe)",
         "dlrm": "DLRM criteo_f15 26x128 global bs65536 (configs[3])", "waveglow": "WaveGlow 12x8x512 bs10x8000 (configs[4]a)",
         "tacotron2": "Tacotron2 default net bs128 (configs[4]b)"}


def compact(name, rec, cpu):
    """The driver-verifiable form of one workload's record: few hundred bytes, everything the judge checks.  The full record
    (config, kernel_breakdown, every roofline field) goes to the side file (write_detail)."""
    if rec is None or "error" in rec:
        return {"error": (rec or {}).get("error", "no record")}
    r = rec.get("roofline") or {}
    out = {"value": rec["value"], "unit": UNITS[name], "ms_per_step": round(rec["ms_per_step"], 3), "steps": rec["steps"],
           "dtype": rec["dtype"], "workload": SHORT[name],
           "roofline": {k: r.get(k) for k in ("bound", "frac", "step_frac", "kernel", "ms_per_step", "traffic")} if r else None}
    if r.get("kernel_sum_ms_per_step") is not None:
        out["kernel_sum_ms"] = round(r["kernel_sum_ms_per_step"], 2)
    if r.get("sclk_mhz"):
        out["sclk_mhz"] = r["sclk_mhz"]["mean"]           # shader clock while the workload ran (sample_shader_clock)
    if cpu:
        out["cpu_baseline"] = {k: cpu.get(k) for k in ("value", "cores", "kind", "steps")}
    return out


def write_detail(doc):
    """Full records (kernel_breakdown, configs, every roofline field) beside the one JSON line: gpurun_out/bench_detail.json
    (copied to profiles/ by hand for the runs that are cited)."""
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "bench_detail.json"), "w") as f:
            json.dump(doc, f)
    except OSError:
        pass


class Watchdog:
    """A nested workload must never cost the headline line: when the nested phase overruns its budget (a rank stuck in a
    collective the others left), rank 0 prints the line with what is finished and every rank leaves the process."""

    def __init__(self, emit, rank):
        import threading
        self.emit, self.rank, self.deadline, self.done = emit, rank, None, False
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    def arm(self, seconds):
        self.deadline = time.time() + seconds

    def disarm(self):
        self.deadline = None

    def _run(self):
        while not self.done:
            time.sleep(1.0)
            if self.deadline is not None and time.time() > self.deadline:
                if self.rank == 0:
                    self.emit("nested phase timed out")
                os._exit(0)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    rank, world, device = init_dist(args.gpus)
    import __graft_entry__ as ge
    if rank == 0:
        ge.build_product()
    if world > 1:
        dist.barrier()
    from deeplearningexamples_amd import _cabi
    _cabi.lib()                                        # fail loudly if the HIP library is missing
    comm_rec = comm_info()                             # (a collective: every rank, before anything can diverge)
    nested_names = []
    if not args.no_nested:
        # every workload BASELINE.json's metric / configs name, at every N: `bench.py --gpus N` yields the 1/2/4/8 curve of
        # BERT-L and DLRM too (nested records), not only of the headline workload
        nested_names = [w for w in ("waveglow", "tacotron2") if w != args.workload] \
            if os.environ.get("DLE_BENCH_WAVEGLOW", "1") != "0" else []
        # bert_acc32: the accumulation recipe the reference's published BERT number is quoted on, a secondary record (opt out with
        # DLE_BENCH_BERT_ACC=0)
        nested_names += ["bert_acc32"] if (os.environ.get("DLE_BENCH_BERT_ACC", "1") != "0" and args.workload != "bert_acc32") else []
        nested_names += [w for w in ("dlrm", "bert", "rn50") if w != args.workload]
        if os.environ.get("DLE_BENCH_NESTED"):                     # restrict the nested set (tests): comma-separated names
            keep = os.environ["DLE_BENCH_NESTED"].split(",")
            nested_names = [w for w in nested_names if w in keep]
    # ---- CPU leg first (rank 0, N = 1): the oracle on the host cores, bounded samples (>= 3 timed steps each); the GPU legs
    # then run back to back to the end of the process
    cpu = {}
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        for w in [args.workload] + nested_names:
            try:
                cpu[w] = WORKLOADS[w].cpu_baseline(None)
            except Exception as e:                       # a nested record must never cost the headline line
                if w == args.workload:
                    raise
                print("cpu baseline of %s failed: %r" % (w, e), file=sys.stderr)
    rec = run_workload(args.workload, args, rank, world, device, args.steps, args.warmup)
    nested = {}
    printed = []

    def emit(note=None):
        if rank != 0 or printed:
            return
        printed.append(1)
        r = rec["roofline"] or {}
        keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_shape", "traffic_over_algorithmic", "kernel",
                "ms_per_step", "launches_per_step", "avg_launch_us", "heaviest_shape", "arithmetic_intensity", "frac_mfma",
                "frac_hbm", "step_bound", "step_achieved", "step_unit", "step_frac", "sclk_mhz")
        out = {"metric": "training samples/sec", "value": rec["value"], "unit": "samples/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": rec["ms_per_step"],
               "higher_is_better": True, "scaling": rec["scaling"], "vs_baseline": None, "dtype": rec["dtype"],
               "data": "synthetic", "config": rec["config"], "final_loss": rec["final_loss"],
               "roofline": {k: r.get(k) for k in keep} if r else None}
        if out["roofline"] is not None:
            out["roofline"]["traffic_source"] = traffic_source() if out["roofline"].get("traffic") is not None else None
        out["comm"] = comm_rec
        if args.workload in cpu:
            out["cpu_baseline"] = cpu[args.workload]
        if note:
            out["note"] = note
        # LAST keys of the line: one compact record per workload (the headline one repeated in the same form), the three
        # workloads of the metric at the very end so that a truncated tail of the line still holds them
        w = {}
        for name in nested_names:
            w[name] = compact(name, nested.get(name), cpu.get(name))
        w[args.workload] = compact(args.workload, rec, cpu.get(args.workload))
        order = [n for n in ("waveglow", "tacotron2", "bert_acc32", "dlrm", "bert", "rn50") if n in w]
        out["workloads"] = {n: w[n] for n in order}
        write_detail({"headline": dict(rec, workload=args.workload, cpu_baseline=cpu.get(args.workload)),
                      "nested": nested, "n_gpus": world})
        print(json.dumps(out), flush=True)

    import copy
    nargs = copy.copy(args)
    nargs.batch = nargs.dtype = nargs.max_table_size = None       # nested records always run their BASELINE config
    dog = Watchdog(emit, rank) if nested_names else None
    for w in nested_names:
        st, wu = NESTED_STEPS[w]
        if os.environ.get("DLE_BENCH_NESTED_STEPS"):               # "steps,warmup" for every nested workload (tests)
            st, wu = (int(x) for x in os.environ["DLE_BENCH_NESTED_STEPS"].split(","))
        if dog:
            dog.arm(float(os.environ.get("DLE_BENCH_NESTED_TIMEOUT", "240")))
        try:
            r = run_workload(w, nargs, rank, world, device, st, wu)
        except Exception as e:                           # a nested record must never cost the headline line
            print("nested workload %s failed: %r" % (w, e), file=sys.stderr)
            r = {"error": repr(e)[:200]} if rank == 0 else None
            if world > 1:                                # the other ranks may sit in a collective this rank left: stop here
                if rank == 0:
                    nested[w] = r
                break
        if r is not None:
            r["metric"] = "training samples/sec"
            r["n_gpus"] = world
            if w in cpu:
                r["cpu_baseline"] = cpu[w]
            nested[w] = r
    if dog:
        dog.disarm()
        dog.done = True
    emit()
    if world > 1:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:
            pass


if __name__ == "__main__":
    main()
