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


def test_ctypes_signatures_match_header_types():
    """Every ctypes signature in _cabi._SIGS has the arity and the per-argument type class of its declaration in
    include/dle_mi355x.h (a mismatch is undefined behaviour at call time, not an exception)."""
    import ctypes as ct
    from deeplearningexamples_amd import _cabi
    src = open(os.path.join(ROOT, "include", "dle_mi355x.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    decls = re.findall(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(dle_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src)
    assert len(decls) >= 55

    def klass(ctype_text):
        t = ctype_text.strip()
        t = re.sub(r"\b[A-Za-z_][A-Za-z0-9_]*$", "", t).strip() if not t.endswith("*") and " " in t else t
        t = t.replace("const ", "").strip()
        if t.endswith("*") or t == "hipStream_t":
            return "ptr"
        return {"int": "int", "int64_t": "i64", "uint64_t": "u64", "float": "float", "void": "void"}[t]

    pyk = {ct.c_void_p: "ptr", ct.c_char_p: "ptr", ct.c_int: "int", ct.c_int64: "i64", ct.c_uint64: "u64",
           ct.c_float: "float"}
    checked = 0
    for ret, name, params in decls:
        restype, argtypes = _cabi._SIGS[name]
        plist = [p for p in (x.strip() for x in params.replace("\n", " ").split(",")) if p and p != "void"]
        got = [klass(p) for p in plist]
        want = [pyk[a] for a in argtypes]
        assert got == want, "%s: header %s vs ctypes %s" % (name, got, want)
        rk = klass(ret + " x")
        assert pyk.get(restype, "ptr") == rk, "%s: return type %s vs %s" % (name, rk, restype)
        checked += 1
    assert checked == len(_cabi._SIGS)


def test_product_never_imports_the_oracle_or_the_test_doubles():
    """oracle/ and tests/ are checkers: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline legs may touch them."""
    pkg = os.path.join(ROOT, "deeplearningexamples_amd")
    pat = re.compile(r"^\s*(from|import)\s+(oracle|tests)\b", re.M)
    bad = []
    for base in (pkg, os.path.join(ROOT, "shims")):
        for d, _, files in os.walk(base):
            for f in files:
                if f.endswith(".py") and pat.search(open(os.path.join(d, f)).read()):
                    bad.append(os.path.join(d, f))
    assert not bad, bad
    # bench.py: every oracle import sits inside a cpu_baseline() body
    src = open(os.path.join(ROOT, "bench.py")).read()
    for m in re.finditer(r"^(\s*)from oracle import", src, re.M):
        head = src[:m.start()]
        last_def = head.rfind("\n    def ")
        assert head[last_def:].lstrip().startswith("def cpu_baseline("), src[m.start():m.start() + 60]


def test_host_side_planning_entry_points():
    """The host-only sizing functions of the round-4 kernels (no GPU needed): scratch of the DLRM sparse update (one-hot partial
    blocks of the tiny tables + eight lists per row for the mid tables), of the fused head, of the streaming weight gradients."""
    import numpy as np
    from deeplearningexamples_amd import _cabi
    lib = _cabi.lib()
    crit = [7912889, 33823, 582469, 245828, 11, 2209, 10667, 104, 4, 968, 15, 8165896, 17139, 2675940, 7156453, 302516, 12022, 97,
            35, 7339, 20046, 4, 7105, 1382, 63, 5554114]
    off = np.concatenate([[0], np.cumsum(crit)]).astype(np.int64)
    p = off.ctypes.data_as(ctypes.c_void_p)
    batch, dim = 65536, 128
    # 8 tables of <= 128 rows -> 512 // 8 = 64 batch slices of one [128][128] fp32 block each
    assert lib.dle_emb_onehot_workspace_bytes(8, batch) == 8 * 64 * 128 * 128 * 4
    assert lib.dle_emb_onehot_workspace_bytes(0, batch) == 0
    total = lib.dle_emb_sgd_workspace_bytes(p, len(crit), dim, batch)
    mid_rows = 968 + 1382 + 2209                       # (7105 / 7339 rows: above the 4096-row envelope of the sub-lists)
    onehot = 8 * 64 * 128 * 128 * 4
    heads = (mid_rows * 8 * 4 + 255) // 256 * 256
    assert total == onehot + heads + mid_rows * 8 * dim * 4
    # a table list without tiny / mid tables needs no scratch; a small batch still gets >= 4 tiles per slice
    big = np.asarray([0, 100000, 300000], dtype=np.int64)
    assert lib.dle_emb_sgd_workspace_bytes(big.ctypes.data_as(ctypes.c_void_p), 2, dim, batch) == 0
    assert lib.dle_emb_onehot_workspace_bytes(8, 256) == 8 * 1 * 128 * 128 * 4
    # fused DLRM head: one partial row [dw | column sums | d bias, loss] per workgroup, at most 1024 workgroups
    assert lib.dle_head_bce_workspace_bytes(65536, 256) == 1024 * (2 * 256 + 2) * 4
    assert lib.dle_head_bce_workspace_bytes(100, 64) == 2 * (2 * 64 + 2) * 4
    assert lib.dle_wgrad1x1_workspace() == 64 << 20
    assert lib.dle_conv3x3_wgrad_workspace() == 256 * 64 * 9 * 64 * 4
