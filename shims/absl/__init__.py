"""Small absl-py stand-in (absl is not installed here): the flags / app / logging surface of
Recommendation/DLRM/dlrm/scripts/main.py:19,34-143,401-403,839 and dlrm/nn/embeddings.py:19."""
from . import app, flags, logging  # noqa: F401
