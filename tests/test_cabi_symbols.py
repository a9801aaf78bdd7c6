"""The C-ABI library loads and exports every symbol include/dle_mi355x.h declares (no compute, CPU ok)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "dle_mi355x.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dle_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    import __graft_entry__ as ge
    ge.build()
    from deeplearningexamples_amd import _cabi
    names = _declared()
    assert len(names) >= 15
    h = ctypes.CDLL(_cabi.LIB_PATH)
    for n in names:
        assert hasattr(h, n), "library does not export %s" % n
    assert sorted(_cabi.declared_symbols()) == names, "python binding and header disagree"
    lib = _cabi.lib()
    assert lib.dle_abi_version() == 1
    assert lib.dle_dot_interact_out_width(27, 128) == 480      # host-only entry point


def test_product_path_fails_loudly_without_gpu_tensor():
    import pytest
    import torch
    from deeplearningexamples_amd import functional as F
    with pytest.raises(RuntimeError):
        F.dot_interact_fwd(torch.zeros(2, 27, 128, dtype=torch.float16))
