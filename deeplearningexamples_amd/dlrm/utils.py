"""Synthetic data, LR schedule and step timing for the DLRM path.

Mirrors (Recommendation/DLRM/):
    dlrm/data/datasets.py:32-61      SyntheticDataset  (one fixed random batch, resident on the device)
    dlrm/scripts/utils.py:224-286    LearningRateScheduler (linear warm-up, polynomial decay)
    dlrm/scripts/utils.py:207-221    StepTimer
"""
import math
import time

import torch


class SyntheticDataset:
    """rand(B, num) fp32, randint(0, card_t) int64 per table, randint(0, 2) float labels -- generated once."""

    def __init__(self, num_entries, device="cuda", batch_size=32768, numerical_features=None,
                 categorical_feature_sizes=None, generator=None):
        n_cat = len(categorical_feature_sizes) if categorical_feature_sizes is not None else 0
        n_num = numerical_features or 0
        self._batches_per_epoch = math.ceil(num_entries / batch_size)
        kw = dict(device=device, generator=generator)
        self._num_tensor = torch.rand((batch_size, n_num), dtype=torch.float32, **kw) if n_num > 0 else None
        self._label_tensor = torch.randint(0, 2, (batch_size,), **kw).to(torch.float32)
        self._cat_tensor = torch.cat(
            [torch.randint(0, int(card), (batch_size, 1), dtype=torch.long, **kw)
             for card in categorical_feature_sizes], dim=1) if n_cat > 0 else None

    def __len__(self):
        return self._batches_per_epoch

    def __getitem__(self, idx):
        if idx >= self._batches_per_epoch:
            raise IndexError()
        return self._num_tensor, self._cat_tensor, self._label_tensor


class LearningRateScheduler:
    """lr factor: linear warm-up over `warmup_steps`, flat, then ((end - step)/decay_steps)**power."""

    def __init__(self, warmup_steps, warmup_factor, decay_steps, decay_start_step, decay_power=2, end_lr_factor=0):
        if decay_start_step < warmup_steps:
            raise ValueError("Learning rate warmup must finish before decay starts")
        self.current_step = 0
        self.warmup_steps, self.warmup_factor = warmup_steps, warmup_factor
        self.decay_steps, self.decay_start_step = decay_steps, decay_start_step
        self.decay_power, self.end_lr_factor = decay_power, end_lr_factor
        self.decay_end_step = decay_start_step + decay_steps

    def factor(self):
        s = self.current_step
        if s <= self.warmup_steps:
            unit = 1 / (self.warmup_steps * (2 ** self.warmup_factor)) if self.warmup_steps else 0.0
            return 1 - (self.warmup_steps - s) * unit
        if self.decay_start_step < s <= self.decay_end_step:
            return max(((self.decay_end_step - s) / self.decay_steps) ** self.decay_power, self.end_lr_factor)
        if s > self.decay_end_step:
            return self.end_lr_factor
        return 1

    def step(self):
        self.current_step += 1
        return self.factor()


class StepTimer:
    def __init__(self):
        self._previous = self._new = self.measured = None

    def click(self, synchronize=False):
        self._previous = self._new
        if synchronize:
            torch.cuda.synchronize()
        self._new = time.time()
        if self._previous is not None:
            self.measured = self._new - self._previous


def roc_auc_score(y_true: torch.Tensor, y_score: torch.Tensor) -> float:
    """Area under the ROC curve, the validation metric of the reference (dlrm/scripts/utils.py:289-320, sklearn's
    definition): thresholds at the distinct score values, trapezoid rule over (fpr, tpr).  Runs on the tensors' device
    (sort + prefix sum); float64 accumulation so that 10^8 validation samples keep their last digits."""
    y_true, y_score = y_true.reshape(-1), y_score.reshape(-1)
    if y_true.shape != y_score.shape:
        raise TypeError("Shape of y_true and y_score must match. Got %s and %s." % (tuple(y_true.shape), tuple(y_score.shape)))
    order = torch.argsort(y_score, descending=True)
    s, t = y_score[order], y_true[order].to(torch.float64)
    n = t.numel()
    last = torch.ones(n, dtype=torch.bool, device=s.device)          # last element of every run of equal scores
    last[:-1] = s[1:] != s[:-1]
    idx = torch.nonzero(last).reshape(-1)
    tps = torch.cumsum(t, 0)[idx]
    fps = (idx + 1).to(torch.float64) - tps
    zero = torch.zeros(1, dtype=torch.float64, device=s.device)
    tps, fps = torch.cat([zero, tps]), torch.cat([zero, fps])
    if float(tps[-1]) == 0.0 or float(fps[-1]) == 0.0:
        return float("nan")                                          # one class only: the curve is undefined
    return float(torch.trapz(tps / tps[-1], fps / fps[-1]).item())


@torch.no_grad()
def evaluate(model, batches, world_size=1, rank=0, batch_sizes_per_gpu=None, process_group=None, exchange=None):
    """Validation pass (dlrm/scripts/main.py:733-835 dist_evaluate): forward only, every rank's logits gathered, AUC and
    BCE loss over the whole validation set.  batches: iterable of (numerical or None, categorical or None, click).
    exchange: the trainer's bottom -> top all-to-all (DlrmTrainer._bottom_to_top) when world_size > 1.
    Returns (auc, loss)."""
    import torch.distributed as dist
    y_true, y_score = [], []
    for num, cat, click in batches:
        x = model.bottom_model(num, cat)
        if world_size > 1:
            x = exchange(x)
        out = model.top_model(x).reshape(-1).float()
        if world_size > 1:
            parts = [torch.empty(b, dtype=out.dtype, device=out.device) for b in batch_sizes_per_gpu]
            dist.all_gather(parts, out, group=process_group)
            out = torch.cat(parts)
        y_true.append(click.reshape(-1).float())
        y_score.append(out)
    y_true, y_score = torch.cat(y_true), torch.cat(y_score)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(y_score, y_true)
    return roc_auc_score(y_true, y_score), float(loss.item())
