"""pynvml surface used by Classification/ConvNets/image_classification/gpu_affinity.py (nvmlInit is called even for
--gpu-affinity none, gpu_affinity.py:393).  Device count comes from torch; CPU affinity = all cores."""
import os


class NVMLError(Exception):
    pass


def nvmlInit():
    return None


def nvmlShutdown():
    return None


def nvmlDeviceGetCount():
    import torch
    return torch.cuda.device_count()


class _Handle:
    def __init__(self, index):
        self.index = index


def nvmlDeviceGetHandleByIndex(index):
    return _Handle(index)


def nvmlDeviceGetName(handle):
    import torch
    return torch.cuda.get_device_name(handle.index)


def nvmlDeviceGetUUID(handle):
    return "GPU-%08d" % handle.index


def nvmlDeviceGetCpuAffinity(handle, length):
    """-> `length` 64-bit words, bit i set = logical CPU i allowed (all CPUs: no NUMA pinning information here)."""
    n = os.cpu_count() or 1
    words = []
    for w in range(length):
        lo, hi = 64 * w, min(64 * (w + 1), n)
        words.append(((1 << max(hi - lo, 0)) - 1) if hi > lo else 0)
    return words
