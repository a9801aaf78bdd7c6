"""functional.pick_splitk (the K-slice count of a weight-gradient GEMM: host logic) against the measured sweep committed in
profiles/r05_gemm8_splitk_sweep.jsonl (tools/gemm8_splitk_sweep.py on one MI355X: microseconds per call, slab reduce
included, for 1 ... 256 slices of every weight-gradient shape of the workloads).  The policy must name a slice count whose
measured time is within 8 % of the best measured one (it interpolates between the swept powers of two: 5 slices for BERT's
QKV gradient), and it must stay inside the kernel's envelope (slices <= K tiles, one- and two-tile outputs capped at 64).
CPU only."""
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SWEEP = os.path.join(os.path.dirname(HERE), "profiles", "r05_gemm8_splitk_sweep.jsonl")


def _records():
    return [json.loads(l) for l in open(SWEEP) if l.strip()]


@pytest.mark.parametrize("rec", _records(), ids=lambda r: "x".join(map(str, r["mnk"])))
def test_policy_is_at_the_measured_optimum(rec, monkeypatch):
    monkeypatch.delenv("DLE_SPLITK_TARGET", raising=False)
    from deeplearningexamples_amd import functional as F
    m, n, k = rec["mnk"]
    s = F.pick_splitk(m, n, k)
    assert s == rec["default_splitk"], "the sweep was taken with another policy: re-run tools/gemm8_splitk_sweep.py"
    times = {int(key[len("new_sk"):]): v for key, v in rec.items() if key.startswith("new_sk")}
    best = min(times.values())
    if s in times:
        assert times[s] <= 1.08 * best
    else:                                   # between two swept counts: no worse than the better neighbour + 8 %
        lo = max(c for c in times if c < s)
        hi = min(c for c in times if c > s)
        assert min(times[lo], times[hi]) <= 1.08 * best


def test_envelope(monkeypatch):
    monkeypatch.delenv("DLE_SPLITK_TARGET", raising=False)
    from deeplearningexamples_amd import functional as F
    for (m, n, k) in [(256, 256, 64), (256, 256, 128), (256, 512, 1 << 20), (4096, 1024, 256), (1024, 1024, 512),
                      (64, 64, 802816), (8192, 8192, 8192), (30528, 1024, 5120)]:
        s = F.pick_splitk(m, n, k)
        assert 1 <= s <= (k + 63) // 64
        if m >= 256 and n >= 256 and ((m + 255) // 256) * ((n + 255) // 256) <= 2:
            assert s <= 64
    assert F.pick_splitk(8192, 8192, 8192) == 1            # enough tiles: no split
