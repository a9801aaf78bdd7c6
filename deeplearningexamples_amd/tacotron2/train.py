"""Entry point mirroring SpeechSynthesis/Tacotron2/train.py for `-m Tacotron2` (second half):

    python -m deeplearningexamples_amd.tacotron2.train -m Tacotron2 -o out/ --amp -lr 1e-3 --epochs 2 -bs 128 \
        --weight-decay 1e-6 --grad-clip-thresh 1.0 --log-file nvlog.json --anneal-steps 500 1000 1500 --anneal-factor 0.3

Same flags (train.py:45-160, tacotron2/arg_parser.py:40-107), loop (train.py:444-500), DLLogger records (train_items_per_sec = mel
frames / s) and checkpoint files (`checkpoint_Tacotron2_<epoch>.pt`, train.py:185-255) as the reference; the shared pieces live
in waveglow/train.py (common flags, epoch loop with the per-epoch validation pass, checkpoints).  Data: the filelists of
--training-files / --validation-files under -d through TextMelLoader + TextMelCollate (tacotron2/data_function.py; wavs through
the STFT front end of tacotron2/audio.py or, with --load-mel-from-disk, saved mels), or -- with --synthetic-data, this port's
benchmark mode -- device-resident batches in the collate's layout.  --mask-padding as model.py:648-655.
"""
import torch

from ..waveglow import train as WT
from .engine import Tacotron2Trainer
from .model import DEFAULT_CONFIG, Tacotron2, param_shapes


def parse_args(argv=None):
    p = WT.common_parser("Tacotron2 training on MI355X (train.py CLI of the reference, -m Tacotron2)", "Tacotron2")
    m = p.add_argument_group("Tacotron2 parameters")                 # tacotron2/arg_parser.py:40-107
    m.add_argument("--mask-padding", default=False, type=bool, help="Use mask padding")
    for k, v in DEFAULT_CONFIG.items():                                # --n-mel-channels, --prenet-dim, --p-attention-dropout, ...
        m.add_argument("--" + k.replace("_", "-"), default=v, type=type(v))
    m.add_argument("--max-decoder-steps", default=2000, type=int, help="inference only: kept in the checkpoint's config")
    m.add_argument("--gate-threshold", default=0.5, type=float, help="inference only: kept in the checkpoint's config")
    m.add_argument("--decoder-no-early-stopping", action="store_true", help="inference only: kept in the checkpoint's config")
    p.set_defaults(iters_per_epoch=50)
    # parse_known_args, as the reference does (Tacotron2/train.py:349,382): a command line shared by both models, or a launcher's
    # --local_rank, must not abort the run; what is ignored is said on stderr
    args, unknown = p.parse_known_args(argv)
    if unknown:
        import sys
        print("warning: ignored command-line arguments (not used by this model): %s" % " ".join(unknown), file=sys.stderr)
    return args


def get_model_config(args):
    """models.get_model_config('Tacotron2', args) (models.py:97-130): the dict the checkpoint's `config` key holds, with the
    reference's keys (its own load path rebuilds the model from it)."""
    cfg = dict(mask_padding=args.mask_padding)
    cfg.update({k: getattr(args, k) for k in DEFAULT_CONFIG})
    cfg.update(max_decoder_steps=args.max_decoder_steps, gate_threshold=args.gate_threshold,
               decoder_no_early_stopping=args.decoder_no_early_stopping)
    return cfg


def parameter_order(cfg):
    """model.parameters() order of the reference's Tacotron2 = the order of model.param_shapes."""
    return [n for n, _ in param_shapes(cfg)]


class SyntheticTextMel:
    """Device-resident batches in TextMelCollate's layout; a fixed pool, cycled.  pool[i] -> ((text, text_lengths, mel, gate,
    output_lengths), mel frames in the batch)."""

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
            self.items.append(tuple(t.to(device) for t in (text, tl, mel, gate, ml)))
            self.num_items.append(int(ml.sum()))

    def __getitem__(self, i):
        return self.items[i % len(self.items)], self.num_items[i % len(self.items)]


def main(argv=None):
    args = parse_args(argv)
    rank, world, local, dev = WT.init_run(args, "Tacotron2_PyT")
    config = get_model_config(args)
    model = Tacotron2(device=dev, uniform_initialize_bn_weight=not args.disable_uniform_initialize_bn_weight, **config)
    trainer = Tacotron2Trainer(model, lr=args.learning_rate, weight_decay=args.weight_decay, grad_clip_thresh=args.grad_clip_thresh,
                               compute_dtype=torch.float16 if args.compute_dtype == "fp16" else torch.bfloat16, amp=args.amp,
                               world_size=world, seed=(args.seed or 1234), rank=rank, mask_padding=args.mask_padding)
    names = parameter_order(config)
    start_epoch = 0
    if args.resume_from_last:
        args.checkpoint_path = WT.get_last_checkpoint_filename(args.output, args.model_name)
    if args.checkpoint_path:
        config, start_epoch = WT.load_checkpoint(trainer, args.checkpoint_path, local, names)
        if args.use_saved_learning_rate:
            args.learning_rate = trainer.lr
    if args.synthetic_data:
        seed = (args.seed or 0) + 1000 * rank
        nsym, nmel = config["n_symbols"], config["n_mel_channels"]
        train_data = WT.SyntheticEpochs(SyntheticTextMel(args.batch_size, nsym, nmel, dev, seed), args.iters_per_epoch)
        val_data = WT.SyntheticEpochs(SyntheticTextMel(args.batch_size, nsym, nmel, dev, seed + 7, pool=2), 2)
    else:
        from .data_function import TextMelCollate, TextMelLoader, batch_to_gpu
        collate = TextMelCollate(args.n_frames_per_step)
        to_trainer = lambda x, y: (x[0], x[1], x[2], y[1], x[4])            # (text, text_lengths, mel, gate target, output_lengths)
        mk = lambda which, train: WT.LoaderEpochs(TextMelLoader(args.dataset_path, WT.filelist(args, which), args), args.batch_size,
                                                  collate, batch_to_gpu, to_trainer, dev, world, rank, args.seed, train,
                                                  drop_last=train or args.bench_class == "perf-train")
        train_data, val_data = mk("training_files", True), mk("validation_files", False)
    return WT.train_loop(args, trainer, config, names, train_data, val_data, trainer.eval_loss, start_epoch, rank, world, local)


if __name__ == "__main__":
    main()
