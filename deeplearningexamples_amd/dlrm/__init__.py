"""DLRM (PyTorch/Recommendation/DLRM) train-step path on MI355X."""
from .placement import get_device_mapping, get_gpu_batch_sizes, argsort, distribute_to_buckets  # noqa: F401
