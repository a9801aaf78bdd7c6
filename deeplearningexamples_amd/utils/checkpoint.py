"""Checkpoint files of the three trainers in the REFERENCE'S formats: a file written here loads
into the reference's own classes and the other way round.

Formats (paths relative to /root/reference/PyTorch/):
 * ResNet-50    Classification/ConvNets/image_classification/training.py:194-202,409-421, utils.py:26-80
     checkpoint_{epoch:04}.pth.tar (+ copies checkpoint.pth.tar / model_best.pth.tar) = torch.save of
     {"epoch", "best_prec1", "state_dict": model.state_dict(), "optimizer": torch.optim.SGD.state_dict()} with the
     parameter groups of get_sgd_optimizer (optimizers.py:34-56): [names containing "bn": weight_decay 0 | the rest].
 * BERT         LanguageModeling/BERT/run_pretraining.py:489-515, lamb_amp_opt/fused_lamb/fused_lamb.py:13-41,131-260
     ckpt_{step}.pt = {"model": state_dict, "optimizer": FusedLAMBAMP.state_dict(), "grad_scaler": GradScaler
     .state_dict(), "epoch"}; optimizer groups [decay | no_decay] of run_pretraining.py:341-349 in named_parameters()
     order, per-parameter state {exp_avg, exp_avg_sq}, per-group TENSOR entries "lr" (fp32 scalar) and "step" (int32 [1]).
 * DLRM         Recommendation/DLRM/dlrm/utils/checkpointing/model.py:37-134, distributed.py:23-123
     a directory: bottom_model.embeddings.{i}.bin (raw fp32 bytes of table i) + embeddings.{i}.meta.pt {"shape"},
     bottom_model.mlp.pt / top_model.mlp.pt {"weights": [...], "biases": [...]} fp32, top_model.out.pt (nn.Linear
     state_dict), metadata.pt {"data": {...}, "config": {...}}; every rank writes the tables it owns.

The functions that build / consume the dictionaries work on plain tensors (no kernel calls): the CPU suite checks them
against the reference's classes imported in this container; the trainer-facing wrappers below add the device plumbing
(momentum / moment buffers out of the flat optimizer state, 16-bit working copies refreshed after a load).
"""
import os
import shutil
from collections import OrderedDict
from typing import Any, Dict, Optional, Sequence

import numpy as np
import torch


def _strip_module(state):
    """Keys of a DistributedDataParallel-wrapped model carry a "module." prefix (the reference saves them as they are)."""
    return OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in state.items())


# ---------------------------------------------------------------------------------------------- ResNet-50
def sgd_param_groups(named_parameters, weight_decay, bn_weight_decay=False):
    """[(names, weight_decay)] in the order get_sgd_optimizer builds its groups (optimizers.py:37-51)."""
    names = [n for n, _ in named_parameters]
    if bn_weight_decay:
        return [(names, weight_decay)]
    return [([n for n in names if "bn" in n], 0), ([n for n in names if "bn" not in n], weight_decay)]


def rn50_optimizer_state(named_parameters, momentum_buffers: Optional[Dict[str, torch.Tensor]], lr, momentum,
                         weight_decay, nesterov=False, bn_weight_decay=False):
    """torch.optim.SGD.state_dict() of the reference's optimizer.  momentum_buffers: name -> tensor shaped like the
    parameter, or None / {} before the first step (torch creates the buffers lazily)."""
    named = list(named_parameters)
    groups, state, idx = [], {}, 0
    skeleton = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=float(lr), momentum=momentum,
                               weight_decay=weight_decay, nesterov=nesterov).state_dict()["param_groups"][0]
    for names, wd in sgd_param_groups(named, weight_decay, bn_weight_decay):
        g = dict(skeleton)
        g["weight_decay"] = wd
        g["params"] = list(range(idx, idx + len(names)))
        for n in names:
            if momentum_buffers:
                state[idx] = {"momentum_buffer": momentum_buffers[n]}
            idx += 1
        groups.append(g)
    return {"state": state, "param_groups": groups}


def rn50_momentum_from_optimizer_state(named_parameters, opt_state, bn_weight_decay=False):
    """name -> momentum buffer (or {} when the checkpoint was taken before the first step)."""
    named = list(named_parameters)
    order = [n for names, _ in sgd_param_groups(named, 0.0, bn_weight_decay) for n in names]
    saved = [i for g in opt_state["param_groups"] for i in g["params"]]
    if len(saved) != len(order):
        raise ValueError("optimizer state holds %d parameters, the model %d" % (len(saved), len(order)))
    out = {}
    for n, i in zip(order, saved):
        st = opt_state["state"].get(i)
        if st and st.get("momentum_buffer") is not None:
            out[n] = st["momentum_buffer"]
    return out


class Checkpointer:
    """File handling of image_classification/utils.py:26-80: numbered files, a copy under `last_filename`, a copy of
    the best one, only the newest `keep_last_n` numbered files kept."""

    def __init__(self, last_filename, checkpoint_dir="./", keep_last_n=0):
        self.last_filename, self.checkpoint_dir, self.keep_last_n = last_filename, checkpoint_dir, keep_last_n
        self.checkpoints = []

    def get_full_path(self, filename):
        return os.path.join(self.checkpoint_dir, filename)

    def cleanup(self):
        # the reference's slicing, as is (utils.py:40-46): keep_last_n = 0 (its default) -> checkpoints[:-0] is EMPTY, every
        # numbered file stays; -1 ("all checkpoints" in its help text) drops only the oldest one, exactly as there
        drop = self.checkpoints[:-self.keep_last_n]
        self.checkpoints = self.checkpoints[len(drop):]
        for f in drop:
            os.remove(self.get_full_path(f))

    def save_checkpoint(self, state, is_best, filename):
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_rank() != 0:
            raise RuntimeError("only rank 0 writes checkpoints")
        os.makedirs(self.checkpoint_dir, exist_ok=True)
        full = self.get_full_path(filename)
        torch.save(state, full)
        self.checkpoints.append(filename)
        shutil.copyfile(full, self.get_full_path(self.last_filename))
        if is_best:
            shutil.copyfile(full, self.get_full_path("model_best.pth.tar"))
        self.cleanup()


def rn50_trainer_state(trainer, epoch, best_prec1=0.0, lr=None):
    """The dictionary training.py:414-417 saves, from a ResNetTrainer."""
    trainer.sync_counters()
    model = trainer.model
    named = list(model.named_parameters())
    mom = None
    if not trainer.first_step and trainer.momentum != 0:
        mom = {}
        for n, p in named:
            m = trainer.mview[n]
            if p.dim() == 4:      # the flat state follows the channels_last memory order (KRSC)
                ko, ci, r, s = p.shape
                mom[n] = m.view(ko, r, s, ci).permute(0, 3, 1, 2).clone(memory_format=torch.preserve_format)
            else:
                mom[n] = m.view(p.shape).clone()
    return {"epoch": epoch, "best_prec1": best_prec1, "state_dict": model.state_dict(),
            "optimizer": rn50_optimizer_state(named, mom, float(trainer.lr.item()) if lr is None else lr, trainer.momentum,
                                              trainer.wd, trainer.nesterov, trainer.bn_weight_decay)}


def rn50_trainer_load(trainer, checkpoint):
    """Resume a ResNetTrainer from a reference-format dictionary.  Returns (start_epoch, best_prec1)."""
    model = trainer.model
    model.load_state_dict(_strip_module(checkpoint["state_dict"]))
    named = list(model.named_parameters())
    mom = rn50_momentum_from_optimizer_state(named, checkpoint["optimizer"], trainer.bn_weight_decay)
    trainer.flat_mom.zero_()
    for n, p in named:
        if n in mom:
            src = mom[n].to(device=trainer.dev, dtype=torch.float32)
            if p.dim() == 4:
                src = src.permute(0, 2, 3, 1)
            trainer.mview[n].copy_(src.reshape(-1))
    trainer.first_step = not mom
    trainer.steps_done = int(model.bn1.num_batches_tracked.item())
    trainer.refresh_working_copies()
    return checkpoint.get("epoch", 0), checkpoint.get("best_prec1", 0.0)


# ---------------------------------------------------------------------------------------------- BERT
BERT_NO_DECAY = ("bias", "gamma", "beta", "LayerNorm")      # run_pretraining.py:342


def lamb_param_groups(names: Sequence[str]):
    """[decay names, no-decay names] in named_parameters() order (run_pretraining.py:341-349)."""
    nd = [n for n in names if any(k in n for k in BERT_NO_DECAY)]
    return [[n for n in names if n not in set(nd)], nd]


def lamb_optimizer_state(names, exp_avg, exp_avg_sq, lr, step, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01,
                         max_grad_norm=1.0, with_moments=True):
    """FusedLAMBAMP.state_dict(): names in the reference model's named_parameters() order; lr a fp32 scalar tensor,
    step an int32 [1] tensor (both per group, as the optimizer keeps them on the device)."""
    groups, state, idx = [], {}, 0
    for gnames, wd in zip(lamb_param_groups(names), (weight_decay, 0.0)):
        g = {"weight_decay": wd, "lr": lr.detach().clone().reshape(()), "step": step.detach().clone().reshape(1).to(torch.int32),
             "bias_correction": True, "betas": tuple(betas), "eps": eps, "grad_averaging": True,
             "max_grad_norm": max_grad_norm, "params": list(range(idx, idx + len(gnames)))}
        for n in gnames:
            if with_moments:
                state[idx] = {"exp_avg": exp_avg[n], "exp_avg_sq": exp_avg_sq[n]}
            idx += 1
        groups.append(g)
    return {"state": state, "param_groups": groups}


def lamb_moments_from_state(names, opt_state):
    """-> (exp_avg by name, exp_avg_sq by name, step int, lr float) from a FusedLAMBAMP.state_dict()."""
    order = [n for g in lamb_param_groups(names) for n in g]
    saved = [i for g in opt_state["param_groups"] for i in g["params"]]
    if len(saved) != len(order):
        raise ValueError("optimizer state holds %d parameters, the model %d" % (len(saved), len(order)))
    m, v = {}, {}
    for n, i in zip(order, saved):
        st = opt_state["state"].get(i)
        if st:
            m[n], v[n] = st["exp_avg"], st["exp_avg_sq"]
    g0 = opt_state["param_groups"][0]
    step = int(torch.as_tensor(g0.get("step", 0)).reshape(-1)[0].item())
    return m, v, step, float(torch.as_tensor(g0["lr"]).item())


def grad_scaler_state(scale, growth_tracker, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
    """torch.cuda.amp.GradScaler.state_dict()."""
    return {"scale": float(scale), "growth_factor": growth_factor, "backoff_factor": backoff_factor,
            "growth_interval": growth_interval, "_growth_tracker": int(growth_tracker)}


def bert_model_state(model):
    """state_dict with the tied decoder weight under both of the reference's names (modeling.py:563-566)."""
    sd = model.state_dict()
    if "cls.predictions.decoder.weight" not in sd:
        sd["cls.predictions.decoder.weight"] = sd["bert.embeddings.word_embeddings.weight"]
    return sd


def bert_trainer_state(trainer, epoch=0):
    """The dictionary run_pretraining.py:501-504 saves, from a BertTrainer."""
    names = [n for n, _ in trainer.model.named_parameters()]
    sc = trainer.scaler
    return {"model": bert_model_state(trainer.model),
            "optimizer": lamb_optimizer_state(names, trainer.exp_avg, trainer.exp_avg_sq, trainer.lr_t, trainer.step_t,
                                              weight_decay=trainer.wd, max_grad_norm=float(trainer.max_norm_t.item()),
                                              with_moments=trainer.opt_steps > 0),
            "grad_scaler": grad_scaler_state(sc.scale.item(), sc.growth_tracker.item(), sc.growth_factor,
                                             sc.backoff_factor, sc.growth_interval) if sc.enabled else {},
            "epoch": epoch,
            # not a reference key (its loaders ignore it): the position of the counter-based dropout stream, so that a resumed
            # run continues the mask sequence instead of replaying it from 0
            "dle_rng_base": int(trainer._rng_base.item())}


def bert_trainer_load(trainer, checkpoint):
    model = trainer.model
    sd = _strip_module(checkpoint["model"])
    own = model.state_dict()
    model.load_state_dict(OrderedDict((k, v) for k, v in sd.items() if k in own), strict=True)
    names = [n for n, _ in model.named_parameters()]
    m, v, step, lr = lamb_moments_from_state(names, checkpoint["optimizer"])
    for n in names:
        if n in m:
            trainer.exp_avg[n].copy_(m[n].to(trainer.exp_avg[n].device))
            trainer.exp_avg_sq[n].copy_(v[n].to(trainer.exp_avg_sq[n].device))
        else:
            trainer.exp_avg[n].zero_()
            trainer.exp_avg_sq[n].zero_()
    trainer.step_t.fill_(step)
    trainer._rng_base.fill_(int(checkpoint.get("dle_rng_base", 0)))
    gs = checkpoint.get("grad_scaler") or {}
    if trainer.scaler.enabled and "scale" in gs:
        trainer.scaler.scale.fill_(gs["scale"])
        trainer.scaler.inv_scale.fill_(1.0 / gs["scale"])
        trainer.scaler.growth_tracker.fill_(gs.get("_growth_tracker", 0))
    trainer.refresh_working_copies()
    return checkpoint.get("epoch", 0)


# ---------------------------------------------------------------------------------------------- DLRM
_BOTTOM_MLP_FILE, _TOP_MLP_FILE, _TOP_OUT_FILE, _METADATA_FILE = "bottom_model.mlp.pt", "top_model.mlp.pt", "top_model.out.pt", "metadata.pt"


def _embedding_file(i):
    return "bottom_model.embeddings.%d.bin" % i


def _embedding_meta_file(i):
    return "embeddings.%d.meta.pt" % i


class DlrmCheckpointWriter:
    """Writes the parts of a DLRM model a rank owns (dlrm/utils/checkpointing/model.py:37-83)."""

    def __init__(self, embedding_indices: Sequence[int], config: Dict[str, Any]):
        self._embedding_indices, self._config = list(embedding_indices), config

    def save_embeddings(self, path, model):
        os.makedirs(path, exist_ok=True)
        for i, w in zip(self._embedding_indices, model.bottom_model.embeddings.weights):
            with open(os.path.join(path, _embedding_file(i)), "wb") as f:
                f.write(w.detach().cpu().numpy().astype(np.float32).tobytes())
            torch.save({"shape": torch.Size(w.shape)}, os.path.join(path, _embedding_meta_file(i)))

    @staticmethod
    def _mlp_state(mlp):
        return {"weights": [x.detach().to(torch.float32) for x in mlp.weights],
                "biases": [x.detach().to(torch.float32) for x in mlp.biases]}

    def save_bottom_mlp(self, path, model):
        os.makedirs(path, exist_ok=True)
        torch.save(self._mlp_state(model.bottom_model.mlp), os.path.join(path, _BOTTOM_MLP_FILE))

    def save_top_model(self, path, model):
        os.makedirs(path, exist_ok=True)
        top = model.top_model.module if hasattr(model.top_model, "module") else model.top_model
        torch.save(self._mlp_state(top.mlp), os.path.join(path, _TOP_MLP_FILE))
        torch.save(top.out.state_dict(), os.path.join(path, _TOP_OUT_FILE))

    def save_metadata(self, path, data):
        os.makedirs(path, exist_ok=True)
        torch.save({"data": data, "config": self._config}, os.path.join(path, _METADATA_FILE))


class DlrmCheckpointLoader:
    def __init__(self, embedding_indices: Sequence[int], device="cpu"):
        self._embedding_indices, self._device = list(embedding_indices), device

    def _load(self, path, name):
        data = torch.load(os.path.join(path, name), map_location=self._device)
        return {(k[7:] if k.startswith("module.") else k): v for k, v in data.items()}

    def load_embeddings(self, path, model):
        def tables():
            for i in self._embedding_indices:
                shape = torch.load(os.path.join(path, _embedding_meta_file(i)))["shape"]
                with open(os.path.join(path, _embedding_file(i)), "rb") as f:
                    yield torch.from_numpy(np.frombuffer(f.read(), dtype=np.float32).reshape(*shape).copy()).to(self._device)
        model.bottom_model.embeddings.load_weights(tables())

    def load_bottom_mlp(self, path, model):
        st = self._load(path, _BOTTOM_MLP_FILE)
        model.bottom_model.mlp.load_state(st["weights"], st["biases"])

    def load_top_model(self, path, model):
        top = model.top_model.module if hasattr(model.top_model, "module") else model.top_model
        st = self._load(path, _TOP_MLP_FILE)
        top.mlp.load_state(st["weights"], st["biases"])
        top.out.load_state_dict(self._load(path, _TOP_OUT_FILE))


class DistributedCheckpointWriter:
    """Every rank its tables, the bottom-MLP rank the bottom MLP, the main process the (data-parallel) top model and
    the metadata (dlrm/utils/checkpointing/distributed.py:23-65)."""

    def __init__(self, writer, device_mapping, rank, main_process):
        self._writer, self._device_mapping, self._main_process = writer, device_mapping, main_process
        self._has_bottom_mlp = rank == device_mapping["bottom_mlp"]
        self._distributed = len(device_mapping["embedding"]) > 1

    def save_checkpoint(self, model, checkpoint_path, epoch=None, step=None):
        self._writer.save_embeddings(checkpoint_path, model)
        if self._has_bottom_mlp:
            self._writer.save_bottom_mlp(checkpoint_path, model)
        if self._main_process:
            self._writer.save_top_model(checkpoint_path, model)
            self._writer.save_metadata(checkpoint_path, {"device_mapping": self._device_mapping, "epoch": epoch, "step": step})
        if self._distributed and torch.distributed.is_initialized():
            torch.distributed.barrier()


class DistributedCheckpointLoader:
    def __init__(self, loader, device_mapping, rank):
        self._loader = loader
        self._has_bottom_mlp = rank == device_mapping["bottom_mlp"]
        self.distributed = len(device_mapping["embedding"]) > 1

    def load_checkpoint(self, model, checkpoint_path):
        self._loader.load_top_model(checkpoint_path, model)
        if self._has_bottom_mlp:
            self._loader.load_bottom_mlp(checkpoint_path, model)
        self._loader.load_embeddings(checkpoint_path, model)
        if hasattr(model, "refresh_working_copies"):
            model.refresh_working_copies()
        if self.distributed and torch.distributed.is_initialized():
            torch.distributed.barrier()


def make_distributed_checkpoint_loader(device_mapping, rank, device="cpu"):
    return DistributedCheckpointLoader(DlrmCheckpointLoader(device_mapping["embedding"][rank], device), device_mapping, rank)


def make_distributed_checkpoint_writer(device_mapping, rank, is_main_process, config):
    return DistributedCheckpointWriter(DlrmCheckpointWriter(device_mapping["embedding"][rank], config), device_mapping,
                                       rank, is_main_process)
