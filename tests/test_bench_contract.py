"""The committed default bench line (profiles/r03_bench_default.json, produced by `python bench.py` on one MI355X) carries every
field of the contract: metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline /
dtype / data / config.workload, the `roofline` and `cpu_baseline` objects, and one compact record per workload as the LAST key.
CPU only (reads the committed file)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    return json.loads(open(os.path.join(ROOT, "profiles", "r03_bench_default.json")).read().strip().splitlines()[-1])


def test_headline_fields():
    d = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "workloads"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and d["dtype"] == "bf16" and "model" not in d["config"] and "ResNet-50" in d["config"]["workload"]
    assert abs(d["value"] - d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] in (8000.0, 2500.0)
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] and r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"] and c["steps"] >= 3
    assert list(d)[-1] == "workloads"                      # the compact records close the line (a truncated tail still shows them)


def test_every_workload_has_a_compact_record():
    w = _line()["workloads"]
    assert list(w) == ["waveglow", "tacotron2", "dlrm", "bert", "rn50"]          # the three workloads of the metric last
    for name, rec in w.items():
        for k in ("value", "unit", "ms_per_step", "steps", "dtype", "workload", "roofline", "cpu_baseline"):
            assert k in rec, (name, k)
        assert rec["value"] > 0 and rec["ms_per_step"] > 0 and rec["steps"] >= 3
        for k in ("bound", "frac", "step_frac", "kernel", "ms_per_step", "traffic"):
            assert k in rec["roofline"], (name, k)
        assert 0 < rec["roofline"]["frac"] < 1 and 0 < rec["roofline"]["step_frac"] < 1
        assert len(json.dumps(rec)) < 520, name            # compact: five records fit a 4 KB tail
    assert w["tacotron2"]["roofline"]["traffic"] is None   # (rocprofv3 segfaults in PMC mode on that workload: DESIGN.md 5)
    assert all(w[k]["roofline"]["traffic"] for k in ("rn50", "bert", "dlrm", "waveglow"))
