"""lddl (NVIDIA/lddl, un-vendored): only the loader factory run_pretraining.py:557-570 calls."""
