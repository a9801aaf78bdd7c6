"""BERT pre-training entry point on MI355X with the reference's command line.

Mirrors LanguageModeling/BERT/run_pretraining.py:140-321 (flags), :323-375 (setup_training), :377-486
(prepare_model_and_optimizer: --resume_from_checkpoint / --init_checkpoint / --resume_step / --phase2 / --resume_phase2),
:489-515 (checkpoint_step: ckpt_{step}.pt every --num_steps_per_checkpoint optimizer steps and at the end, the three most
recent files kept), :658-736 (loop with gradient accumulation) and its dllogger keys (average_loss, learning_rate,
training_sequences_per_second, e2e_train_time, final_loss, raw_train_time).  Checkpoint files are the reference's
(utils/checkpoint.py: they load into the reference's BertForPreTraining / FusedLAMBAMP / GradScaler and back).
The lddl loader (un-vendored) is replaced by a synthetic loader with the same 5-key int64 batch
(run_pretraining.py:603-609).
    python -m torch.distributed.run --nproc-per-node 8 -m deeplearningexamples_amd.bert.run_pretraining \
        --train_batch_size 256 --gradient_accumulation_steps 2 --max_steps 20 --bf16
"""
import argparse
import os
import time

import torch

from ..utils import checkpoint as ckpt
from ..utils import dllogger
from ..utils.graph import GraphedStep
from ..utils.dist import init_from_env, is_main_process
from .engine import BertTrainer
from .model import LARGE, BertForPreTraining, config_from_json


def parse_arguments(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--input_dir", default=None, type=str, help="lddl parquet shards (external loader; synthetic batches here)")
    p.add_argument("--config_file", default=None, type=str, help="BERT model config json (default: BERT-Large)")
    p.add_argument("--bert_model", default="bert-large-uncased", type=str)
    p.add_argument("--output_dir", default=None, type=str, help="where ckpt_{step}.pt files are written / resumed from")
    p.add_argument("--vocab_file", default=None, type=str)
    p.add_argument("--init_checkpoint", default=None, type=str, help="start from this checkpoint's weights (step and LR reset)")
    p.add_argument("--max_seq_length", default=512, type=int)
    p.add_argument("--max_predictions_per_seq", default=80, type=int)
    p.add_argument("--num_train_epochs", default=3.0, type=float)
    p.add_argument("--local_rank", type=int, default=int(os.getenv("LOCAL_RANK", -1)))
    p.add_argument("--amp", action="store_true", help="accepted: 16-bit compute is always on (--fp16 / --bf16 pick the type)")
    p.add_argument("--loss_scale", type=float, default=0.0)
    p.add_argument("--checkpoint_activations", action="store_true", help="accepted; activations fit in 288 GB and are kept")
    p.add_argument("--resume_from_checkpoint", action="store_true")
    p.add_argument("--resume_step", type=int, default=-1)
    p.add_argument("--num_steps_per_checkpoint", type=int, default=100)
    p.add_argument("--phase2", action="store_true", help="sequence length 512 phase: step / LR restart from a phase-1 checkpoint")
    p.add_argument("--resume_phase2", action="store_true")
    p.add_argument("--phase1_end_step", type=int, default=7038)
    p.add_argument("--do_train", action="store_true")
    p.add_argument("--use_env", action="store_true")
    p.add_argument("--profile", action="store_true")
    p.add_argument("--profile-start", default=0, type=int)
    p.add_argument("--num_workers", type=int, default=4)
    p.add_argument("--no_dense_sequence_output", action="store_true")
    p.add_argument("--disable_jit_fusions", action="store_true", help="accepted: there is no TorchScript in this path")
    p.add_argument("--train_batch_size", default=32, type=int, help="per-GPU batch of one optimizer step")
    p.add_argument("--learning_rate", default=5e-5, type=float)
    p.add_argument("--max_steps", default=1000, type=float)
    p.add_argument("--warmup_proportion", default=0.01, type=float)
    p.add_argument("--gradient_accumulation_steps", type=int, default=1)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--fp16", action="store_true")
    p.add_argument("--bf16", action="store_true")
    p.add_argument("--init_loss_scale", type=int, default=2 ** 20)
    p.add_argument("--log_freq", type=float, default=1.0)
    p.add_argument("--steps_this_run", type=int, default=-1)
    p.add_argument("--json-summary", type=str, default="results/dllogger.json")
    p.add_argument("--disable_progress_bar", action="store_true")
    p.add_argument("--skip_checkpoint", action="store_true")
    p.add_argument("--allreduce_post_accumulation", action="store_true",
                   help="accepted: gradients are always reduced once per optimizer step, on the last micro-batch")
    p.add_argument("--allreduce_post_accumulation_fp16", action="store_true",
                   help="16-bit wire format for the gradient buckets (run_pretraining.py:261,416-417): the compute dtype")
    p.add_argument("--cuda_graphs", action="store_true",
                   help="capture the micro-step and the optimizer step in HIP graphs (run_pretraining.py:310,602-640)")
    args = p.parse_args(argv)
    args._defaults = {k: p.get_default(k) for k in IGNORED_FLAGS}
    if args.steps_this_run < 0:
        args.steps_this_run = int(args.max_steps)
    if args.train_batch_size % args.gradient_accumulation_steps:
        raise ValueError("train_batch_size must be divisible by gradient_accumulation_steps")
    return args


IGNORED_FLAGS = {
    # flag -> what happens instead (printed once at start-up by rank 0: a flag that parses must not silently do nothing)
    "input_dir": "the lddl parquet loader (run_pretraining.py:557-570) is an external package that is not part of this path: "
                 "training runs on SYNTHETIC batches of the same 5-key int64 schema, the shards under this directory are NOT read",
    "checkpoint_activations": "activations are kept (they fit in 288 GB); no recomputation happens",
    "amp": "16-bit compute is always on; --fp16 / --bf16 choose the type (default bf16)",
    "disable_jit_fusions": "there is no TorchScript in this path; the fused kernels are always used",
    "allreduce_post_accumulation": "gradients are always reduced once per optimizer step, on the last micro-batch",
    "vocab_file": "no text is tokenised in this path (synthetic batches)",
    "profile": "use rocprofv3 around the command instead",
    "no_dense_sequence_output": "the MLM head always runs on the masked rows only (dense sequence output)",
}


def warn_ignored_flags(args, defaults, log=print):
    """One line per flag that was GIVEN (differs from its default) and has no effect here.  Returns the list of flag names."""
    hit = [k for k in IGNORED_FLAGS if getattr(args, k, None) != defaults.get(k)]
    for k in hit:
        log("WARNING: --%s is accepted for command-line compatibility and IGNORED: %s" % (k, IGNORED_FLAGS[k]))
    return hit


def synthetic_batches(cfg, micro_batch, seq, max_pred, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    v = cfg["real_vocab"]
    ids = torch.randint(0, v, (micro_batch, seq), generator=g)
    split = torch.randint(seq // 4, 3 * seq // 4, (micro_batch, 1), generator=g)
    tt = (torch.arange(seq)[None, :] >= split).long()
    mask = torch.ones((micro_batch, seq), dtype=torch.long)
    labels = torch.full((micro_batch, seq), -1, dtype=torch.long)
    for i in range(micro_batch):
        pos = torch.randperm(seq, generator=g)[:max_pred]
        labels[i, pos] = torch.randint(0, v, (max_pred,), generator=g)
    nsp = torch.randint(0, 2, (micro_batch,), generator=g)
    batch = {"input_ids": ids, "token_type_ids": tt, "attention_mask": mask, "labels": labels, "next_sentence_labels": nsp}
    batch = {k: t.to(device) for k, t in batch.items()}
    while True:
        yield batch


def main(argv=None):
    args = parse_arguments(argv)
    rank, world, local = init_from_env()
    device = torch.device("cuda", local)
    torch.manual_seed(args.seed)
    cfg = dict(config_from_json(args.config_file) if args.config_file else LARGE, seq=args.max_seq_length)
    if is_main_process():
        if os.path.dirname(args.json_summary):
            os.makedirs(os.path.dirname(args.json_summary), exist_ok=True)
        dllogger.init([dllogger.JSONStreamBackend(dllogger.Verbosity.VERBOSE, args.json_summary),
                       dllogger.StdOutBackend(dllogger.Verbosity.DEFAULT)])
        dllogger.log(step="PARAMETER", data={"Config": [str({k: v for k, v in vars(args).items() if k != "_defaults"})]})
        import sys
        warn_ignored_flags(args, args._defaults, log=lambda m: print(m, file=sys.stderr, flush=True))
    model = BertForPreTraining(cfg, device=device)
    dtype = torch.float16 if args.fp16 else torch.bfloat16
    trainer = BertTrainer(model, lr=args.learning_rate, warmup=args.warmup_proportion, total_steps=int(args.max_steps),
                          compute_dtype=dtype, init_loss_scale=float(args.init_loss_scale), world_size=world,
                          seed=args.seed, rank=rank, allreduce_dtype=dtype if args.allreduce_post_accumulation_fp16 else None,
                          max_predictions_per_seq=args.max_predictions_per_seq if args.cuda_graphs else None)
    # ---- resume (run_pretraining.py:388-452)
    global_step = 0
    if args.resume_from_checkpoint or args.init_checkpoint:
        if args.init_checkpoint:
            path = args.init_checkpoint
        else:
            if args.resume_step == -1:
                names = [f for f in os.listdir(args.output_dir) if f.startswith("ckpt_") and f.endswith(".pt")]
                if not names:
                    raise SystemExit("--resume_from_checkpoint: no ckpt_<step>.pt in %s" % args.output_dir)
                args.resume_step = max(int(f[:-3].split("_")[1]) for f in names)
            global_step = args.resume_step
            path = os.path.join(args.output_dir, "ckpt_%d.pt" % global_step)
        state = torch.load(path, map_location=device, weights_only=False)
        if (args.phase2 and not args.resume_phase2) or args.init_checkpoint:
            # phase 2 from a phase-1 checkpoint / fine start from given weights: step count and learning rate restart, the
            # saved loss scale is not taken over (run_pretraining.py:437-445)
            for group in state["optimizer"]["param_groups"]:
                group["step"] = torch.zeros_like(torch.as_tensor(group["step"]))
                group["lr"] = torch.as_tensor(float(args.learning_rate))
            state = dict(state, grad_scaler={})
        ckpt.bert_trainer_load(trainer, state)
        if args.phase2 and not args.init_checkpoint:
            global_step -= args.phase1_end_step
        if args.init_checkpoint:
            args.resume_step, global_step = 0, 0
        if is_main_process():
            print("resume step from ", args.resume_step)
    recent = []

    def checkpoint_step(step):
        """run_pretraining.py:489-515: ckpt_{step}.pt on the main process, the three most recent files stay."""
        torch.cuda.synchronize()
        if not is_main_process() or args.skip_checkpoint or not args.output_dir:
            return
        dllogger.log(step="PARAMETER", data={"checkpoint_step": step})
        os.makedirs(args.output_dir, exist_ok=True)
        shown = step if (args.resume_step < 0 or not args.phase2) else step + args.phase1_end_step
        out = os.path.join(args.output_dir, "ckpt_%d.pt" % shown)
        torch.save(ckpt.bert_trainer_state(trainer, epoch=0), out)
        if out in recent:
            recent.remove(out)
        recent.append(out)
        if len(recent) > 3:
            os.remove(recent.pop(0))

    acc = args.gradient_accumulation_steps
    micro = args.train_batch_size // acc
    it = synthetic_batches(cfg, micro, args.max_seq_length, args.max_predictions_per_seq, device, args.seed + rank)
    avg_loss = torch.zeros(1, device=device)
    t0 = time.time()
    # run_pretraining.py:602-640 keeps two graphs: micro-step + optimizer step, and the micro-step alone (accumulation).
    # Here the optimizer step is its own captured callable, so one micro-step graph per accumulate flag serves both.
    use_graphs = args.cuda_graphs and world == 1

    def micro_first(*b):
        loss, dlog, dnsp = trainer.forward(*b)
        trainer.backward(dlog, dnsp, accumulate=False)
        return loss

    def micro_acc(*b):
        loss, dlog, dnsp = trainer.forward(*b)
        trainer.backward(dlog, dnsp, accumulate=True)
        return loss

    def opt_step():
        trainer.optimizer_step()
        return trainer.lr_t
    side = torch.cuda.Stream() if use_graphs else None       # ONE side stream: the three captured callables depend on each other
    g_first, g_acc, g_opt = (GraphedStep(f, enabled=use_graphs, stream=side) for f in (micro_first, micro_acc, opt_step))
    trainer.grad_divisor = acc
    loss = avg_loss
    steps_run = 0
    for step in range(global_step, args.steps_this_run):
        for micro_step in range(acc):
            b = next(it)
            trainer._reduce_now = micro_step == acc - 1                   # no_sync on all but the last micro-step
            loss = (g_first if micro_step == 0 else g_acc)(b["input_ids"], b["token_type_ids"], b["attention_mask"],
                                                           b["labels"], b["next_sentence_labels"])
            avg_loss += loss / acc
        g_opt()
        steps_run += 1
        done = step + 1
        if done % max(int(args.log_freq), 1) == 0:
            if is_main_process():
                dllogger.log(step=(0, done), data={"average_loss": float(avg_loss.item()) / max(int(args.log_freq), 1),
                                                   "learning_rate": trainer.current_lr()})
            avg_loss.zero_()
            if done % args.num_steps_per_checkpoint == 0 and done < args.steps_this_run:
                checkpoint_step(done)
    if steps_run:
        checkpoint_step(args.steps_this_run)
    torch.cuda.synchronize()
    secs = time.time() - t0
    if is_main_process():
        seqs = args.train_batch_size * world * steps_run
        dllogger.log(step=tuple(), data={"e2e_train_time": secs, "training_sequences_per_second": seqs / secs,
                                         "final_loss": float(loss.item()), "raw_train_time": secs})
        dllogger.flush()
    return trainer


if __name__ == "__main__":
    main()
