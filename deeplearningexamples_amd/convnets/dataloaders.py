"""Input side of the ResNet-50 trainer.

Mirrors Classification/ConvNets/image_classification/dataloaders.py:
    :520-549  SynteticDataLoader    one fixed random batch (the benchmark's input)
    :340-351  fast_collate          PIL / array images -> one uint8 NCHW batch (no float conversion on the host)
    :354-409  PrefetchedWrapper     host -> device copy on a side stream one batch ahead, normalisation on the device
The normalisation itself does NOT run here: the trainer's first kernel takes the uint8 batch and writes the normalised
16-bit NHWC tensor in one pass (dle_u8_nchw_normalize_nhwc); the wrapper only moves bytes and overlaps the copy.
"""
import os

import numpy as np
import torch
import torch.utils.data
import torch.utils.data.distributed

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


class SyntheticDataLoader:
    """dataloaders.py:520-549: the same (randn images, randint targets) batch at every iteration."""

    def __init__(self, batch_size, num_classes, num_channels, height, width, device, length=10 ** 9, seed=None):
        g = torch.Generator(device="cpu")
        if seed is not None:
            g.manual_seed(seed)
        self.images = torch.randn((batch_size, num_channels, height, width), generator=g).to(device)
        self.target = torch.randint(0, num_classes, (batch_size,), generator=g).to(device)
        self.length = length

    def __len__(self):
        return self.length

    def __iter__(self):
        for _ in range(self.length):
            yield self.images, self.target


def fast_collate(batch):
    """[(HWC or CHW uint8 image, label)] -> (uint8 [N,3,H,W], int64 [N]); images of one size (dataloaders.py:340-351)."""
    imgs = [np.asarray(b[0], dtype=np.uint8) for b in batch]
    targets = torch.tensor([int(b[1]) for b in batch], dtype=torch.int64)
    out = torch.empty((len(imgs), 3, *(imgs[0].shape[:2] if imgs[0].shape[-1] == 3 else imgs[0].shape[1:])), dtype=torch.uint8)
    for i, a in enumerate(imgs):
        if a.ndim < 3:
            a = np.repeat(a[..., None], 3, axis=-1)
        out[i] = torch.from_numpy(a if a.shape[0] == 3 and a.shape[-1] != 3 else np.rollaxis(a, 2).copy())
    return out, targets


class PrefetchedWrapper:
    """One batch ahead on a side stream: pinned host batch -> device copy overlaps the previous step; yields
    (uint8 NCHW device tensor, int64 targets) -- the trainer normalises in its first kernel."""

    def __init__(self, dataloader, device, start_epoch=0):
        self.dataloader, self.device, self.epoch = dataloader, device, start_epoch

    def __len__(self):
        return len(self.dataloader)

    def __iter__(self):
        sampler = getattr(self.dataloader, "sampler", None)
        if isinstance(sampler, torch.utils.data.distributed.DistributedSampler):
            sampler.set_epoch(self.epoch)
        self.epoch += 1
        stream = torch.cuda.Stream(device=self.device)
        prev = None
        for images, target in self.dataloader:
            with torch.cuda.stream(stream):
                nxt = (images.pin_memory().to(self.device, non_blocking=True) if not images.is_cuda else images,
                       target.pin_memory().to(self.device, non_blocking=True) if not target.is_cuda else target)
            if prev is not None:
                yield prev
            torch.cuda.current_stream(self.device).wait_stream(stream)
            for t in nxt:
                t.record_stream(torch.cuda.current_stream(self.device))
            prev = nxt
        if prev is not None:
            yield prev


# ---------------------------------------------------------------------------------------------- loader factories (main.py:494-540)
def get_synthetic_loader(data_path, image_size, batch_size, num_classes, start_epoch=0, device="cuda", rank=0, seed=None,
                         steps_per_epoch=5004, memory_format="nchw", **_):
    """dataloaders.py:552-577.  The reference's loader is endless (its epochs end through --prof); here an epoch is
    `steps_per_epoch` iterations of the same batch.  -> (loader, its length)."""
    ld = SyntheticDataLoader(batch_size, num_classes, 3, image_size, image_size, device, length=steps_per_epoch,
                             seed=None if seed is None else seed + rank)
    if memory_format == "nhwc":
        ld.images = ld.images.contiguous(memory_format=torch.channels_last)
    return ld, steps_per_epoch


class NpyImageFolder(torch.utils.data.Dataset):
    """root/<class>/<image>.npy with uint8 HWC (or HW) arrays = ImageFolder over PRE-DECODED images: the layout the `pytorch`
    backend reads when torchvision / PIL are not installed (this image).  Train: random-resized-crop (scale 0.08-1, ratio 3/4-4/3,
    nearest sampling) + horizontal flip, the parameters of transforms.RandomResizedCrop; val: centre crop after a resize to
    size + 32."""

    def __init__(self, root, image_size, train, seed=0):
        self.size, self.train = image_size, train
        classes = sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d)))
        self.samples = [(os.path.join(root, c, f), i) for i, c in enumerate(classes)
                        for f in sorted(os.listdir(os.path.join(root, c))) if f.endswith(".npy")]
        if not self.samples:
            raise FileNotFoundError("no <class>/<image>.npy files under %s" % root)
        self.rng = np.random.default_rng(seed)

    def __len__(self):
        return len(self.samples)

    def _resample(self, a, top, left, h, w, size):
        ys = np.clip((top + (np.arange(size) + 0.5) * h / size).astype(np.int64), 0, a.shape[0] - 1)
        xs = np.clip((left + (np.arange(size) + 0.5) * w / size).astype(np.int64), 0, a.shape[1] - 1)
        return a[ys][:, xs]

    def __getitem__(self, i):
        path, label = self.samples[i]
        a = np.load(path)
        if a.ndim == 2:
            a = np.repeat(a[..., None], 3, axis=-1)
        hh, ww = a.shape[:2]
        if self.train:
            area = hh * ww * self.rng.uniform(0.08, 1.0)
            ratio = np.exp(self.rng.uniform(np.log(3 / 4), np.log(4 / 3)))
            w, h = min(ww, int(round(np.sqrt(area * ratio)))), min(hh, int(round(np.sqrt(area / ratio))))
            top, left = int(self.rng.integers(0, hh - h + 1)), int(self.rng.integers(0, ww - w + 1))
            out = self._resample(a, top, left, h, w, self.size)
            if self.rng.random() < 0.5:
                out = out[:, ::-1]
        else:
            s = self.size + 32
            scale = s / min(hh, ww)
            rh, rw = int(round(hh * scale)), int(round(ww * scale))
            full = a[np.clip(((np.arange(rh) + 0.5) / scale).astype(np.int64), 0, hh - 1)][:, np.clip(((np.arange(rw) + 0.5) / scale).astype(np.int64), 0, ww - 1)]
            t, l = (rh - self.size) // 2, (rw - self.size) // 2
            out = full[t:t + self.size, l:l + self.size]
        return np.ascontiguousarray(out), label


def _image_folder(root, image_size, train, interpolation, seed):
    try:                                                   # the reference's path: torchvision ImageFolder + PIL transforms
        from torchvision import datasets, transforms
        from torchvision.transforms import InterpolationMode
        mode = {"bicubic": InterpolationMode.BICUBIC, "bilinear": InterpolationMode.BILINEAR}[interpolation]
        tf = [transforms.RandomResizedCrop(image_size, interpolation=mode), transforms.RandomHorizontalFlip()] if train else \
            [transforms.Resize(image_size + 32, interpolation=mode), transforms.CenterCrop(image_size)]
        return datasets.ImageFolder(root, transforms.Compose(tf))
    except (ImportError, AttributeError, KeyError):
        return NpyImageFolder(root, image_size, train, seed)


def _pytorch_loader(data_path, split, image_size, batch_size, train, start_epoch, workers, prefetch_factor, device, rank, world,
                    seed, interpolation):
    if not data_path:
        raise SystemExit("--data-backend pytorch needs the dataset directory (positional argument DIR with train/ and val/)")
    ds = _image_folder(os.path.join(data_path, split), image_size, train, interpolation or "bilinear", (seed or 0) + rank)
    sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=train) if world > 1 else None
    kw = dict(persistent_workers=True, prefetch_factor=prefetch_factor) if workers > 0 else {}
    ld = torch.utils.data.DataLoader(ds, sampler=sampler, batch_size=batch_size, shuffle=(train and sampler is None),
                                     num_workers=workers, pin_memory=True, collate_fn=fast_collate, drop_last=train, **kw)
    return PrefetchedWrapper(ld, device, start_epoch), len(ld)


def get_pytorch_train_loader(data_path, image_size, batch_size, num_classes, start_epoch=0, workers=5, prefetch_factor=2,
                             device="cuda", rank=0, world=1, seed=None, interpolation="bilinear", **_):
    """dataloaders.py:410-461: <DIR>/train, shuffled (DistributedSampler when distributed), drop_last, uint8 batches prefetched
    to the device on a side stream; the trainer normalises in its first kernel."""
    return _pytorch_loader(data_path, "train", image_size, batch_size, True, start_epoch, workers, prefetch_factor, device, rank,
                           world, seed, interpolation)


def get_pytorch_val_loader(data_path, image_size, batch_size, num_classes, workers=5, prefetch_factor=2, device="cuda", rank=0,
                           world=1, seed=None, interpolation="bilinear", **_):
    """dataloaders.py:464-517: <DIR>/val, resize to size + 32, centre crop, no shuffling."""
    return _pytorch_loader(data_path, "val", image_size, batch_size, False, 0, workers, prefetch_factor, device, rank, world,
                           seed, interpolation)


class MixUpWrapper:
    """mixup.py:19-44: per batch c ~ Beta(alpha, alpha), a random permutation; inputs c x + (1 - c) x[perm] (fp32 on the device,
    the input pipeline's arithmetic -- uint8 batches are normalised first), targets handed to the trainer as
    (y, y[perm], c): the loss is linear in the mixed one-hot target."""

    def __init__(self, alpha, dataloader):
        self.alpha, self.dataloader = alpha, dataloader

    def __len__(self):
        return len(self.dataloader)

    def __iter__(self):
        for x, y in self.dataloader:
            c = float(np.random.beta(self.alpha, self.alpha))
            perm = torch.randperm(x.shape[0], device=x.device)
            if x.dtype == torch.uint8:
                mean = torch.tensor(IMAGENET_MEAN, device=x.device).view(1, 3, 1, 1) * 255.0
                std = torch.tensor(IMAGENET_STD, device=x.device).view(1, 3, 1, 1) * 255.0
                x = (x.float() - mean) / std
            yield c * x + (1.0 - c) * x[perm], (y, y[perm], c)
