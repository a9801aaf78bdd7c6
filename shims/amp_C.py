"""amp_C attributes FusedLAMBAMP.__init__ reads (fused_lamb.py:30-35)."""
from fused_lamb_CUDA import multi_tensor_l2norm, multi_tensor_lamb  # noqa: F401
