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
