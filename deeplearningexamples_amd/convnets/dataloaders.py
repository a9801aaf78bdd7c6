"""Input side of the ResNet-50 trainer.

Mirrors Classification/ConvNets/image_classification/dataloaders.py:
    :520-549  SynteticDataLoader    one fixed random batch (the benchmark's input)
    :340-351  fast_collate          PIL / array images -> one uint8 NCHW batch (no float conversion on the host)
    :354-409  PrefetchedWrapper     host -> device copy on a side stream one batch ahead, normalisation on the device
The normalisation itself does NOT run here: the trainer's first kernel takes the uint8 batch and writes the normalised
16-bit NHWC tensor in one pass (dle_u8_nchw_normalize_nhwc); the wrapper only moves bytes and overlaps the copy.
"""
import numpy as np
import torch

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
