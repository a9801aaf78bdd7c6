"""Build recipe for the gfx950 C-ABI library (explicit hipcc, in-tree, incremental).

    python -m deeplearningexamples_amd.build [--force]

Produces deeplearningexamples_amd/lib/libdle_mi355x.so from csrc/*.hip.  hipcc cross-compiles
for gfx950 without a GPU present, so this also runs in the CPU-only build container.
"""
import concurrent.futures as cf
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(HERE, "lib", "libdle_mi355x.so")
MANIFEST = os.path.join(HERE, "lib", "libdle_mi355x.manifest.json")   # content hashes of the sources the .so was built from
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-Wno-unused-result", "-DNDEBUG"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _sha(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for q in paths:
        h.update(open(q, "rb").read())
    return h.hexdigest()


def _compile(src, force):
    """One object per source.  Up to date = the CONTENT hash of (flags, the source, every header of csrc/ and include/)
    recorded next to the object matches -- file times are not preserved everywhere this tree travels."""
    obj = os.path.join(OBJ, src[:-4] + ".o")
    inc = os.path.join(os.path.dirname(HERE), "include")
    headers = sorted(os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h"))
    headers += sorted(os.path.join(inc, h) for h in os.listdir(inc) if h.endswith(".h")) if os.path.isdir(inc) else []
    want = _sha([os.path.join(CSRC, src)] + headers, " ".join(FLAGS))
    stamp = obj + ".sha256"
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return obj, False
    cmd = [hipcc()] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout[-4000:], r.stderr[-8000:]))
    open(stamp, "w").write(want)
    return obj, True


def _source_hashes():
    out = {"flags": " ".join(FLAGS)}
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")):
            out[f] = hashlib.sha256(open(os.path.join(CSRC, f), "rb").read()).hexdigest()
    return out


def build(force=False, verbose=True):
    # The .so travels to the GPU box without its object files (and file times are not preserved there): decide
    # "up to date" from CONTENT hashes recorded next to the library, not from mtimes of objects that are not there.
    hashes = _source_hashes()
    if not force and os.path.exists(LIB) and os.path.exists(MANIFEST):
        try:
            if json.load(open(MANIFEST)) == hashes:
                if verbose:
                    print("up to date:", LIB)
                return LIB
        except (OSError, ValueError):
            pass
    os.makedirs(OBJ, exist_ok=True)
    srcs = sources()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in res]
    if any(c for _, c in res) or not os.path.exists(LIB) or _stale(LIB, objs):
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + ["-ldl", "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-8000:])
        if verbose:
            print("built", LIB, "(%d objects, %d recompiled)" % (len(objs), sum(c for _, c in res)))
    elif verbose:
        print("up to date:", LIB)
    json.dump(hashes, open(MANIFEST, "w"), indent=0)          # only after a successful link of objects that match `hashes`
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
