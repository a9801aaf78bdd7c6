"""bench.py's record-building code on a synthetic kernel timer (CPU only, no GPU): the family aggregation, the choice of the
roofline side by arithmetic intensity, achieved / peak / frac / step_frac arithmetic, the per-shape replay bookkeeping (launches of
one entry point over different tables / shapes are separate rows), the breakdown adding up to the timed kernel total, and the
compact per-workload record the driver's 4 KB tail must hold.  A code change in bench.py that breaks the contract fails here."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402


class FakeTimer:
    """report() rows as _cabi.KernelTimer produces them; replay() returns a fixed per-launch time for the rows it knows."""

    def __init__(self, rows, replays=None):
        self.rows, self.replays, self.asked = rows, replays or {}, []

    def report(self):
        return [dict(r) for r in self.rows]

    def replay(self, name, tag, iters=20, warmup=3, cold=False):
        self.asked.append((name, tag, cold))
        return self.replays.get((name, tag))


def _rows():
    # 10 steps; dle_gemm: two HBM-side shapes (K = 64), 40 launches; a BatchNorm pass; LAMB over two tables (different tags)
    return [
        {"name": "dle_gemm", "tag": "802816x256x64", "calls": 20, "ms": 4.0, "bytes": 20 * 514e6, "flops": 20 * 26.3e9},
        {"name": "dle_gemm", "tag": "256x64x802816", "calls": 20, "ms": 2.4, "bytes": 20 * 514e6, "flops": 20 * 26.3e9},
        {"name": "dle_bn_bwd_apply", "tag": "M802816xC256", "calls": 10, "ms": 2.3, "bytes": 10 * 1.28e9, "flops": 0.0},
        {"name": "dle_mt_lamb_stage1", "tag": "392t,335000000e", "calls": 10, "ms": 5.8, "bytes": 10 * 9.4e9, "flops": 0.0},
        {"name": "dle_mt_lamb_stage1", "tag": "200t,1000000e", "calls": 10, "ms": 0.08, "bytes": 10 * 28e6, "flops": 0.0},
    ]


def test_roofline_from_aggregates_families_and_prices_them():
    t = FakeTimer(_rows(), {("dle_gemm", "802816x256x64"): 0.15, ("dle_gemm", "256x64x802816"): 0.10})
    r, bd = bench.roofline_from(t, 10, "rn50", 256, 23.0)
    # family = entry point across shapes; replayed rows use the replay time, the others their event-pair average
    fam = {b["kernel"]: b for b in bd}
    assert abs(fam["dle_gemm"]["ms_per_step"] - (0.15 * 20 + 0.10 * 20) / 10) < 1e-6
    assert abs(fam["dle_mt_lamb_stage1"]["ms_per_step"] - (5.8 + 0.08) / 10) < 1e-6      # two tables, BOTH counted at their own time
    assert abs(fam["dle_bn_bwd_apply"]["ms_per_step"] - 0.23) < 1e-6
    # the dominant family here is LAMB (0.588 ms / step) -> priced in bytes
    assert r["kernel"] == "dle_mt_lamb_stage1" and r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == bench.HBM_PEAK_GBS
    ach = (10 * 9.4e9 + 10 * 28e6) / (5.88e-3) / 1e9
    assert abs(r["achieved"] - ach) <= 0.01 * ach and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # whole step: SURVEY 8(d)'s work per sample x samples / time / the peak of the bound 8(d) names
    bound, work = bench.WORK_PER_SAMPLE["rn50"]
    assert bound == "mfma" and abs(r["step_frac"] - work * 256 / 23.0e-3 / (bench.MFMA_PEAK_TFLOPS * 1e12)) < 1e-3
    # the rows add up: kernel_sum = every launch of the step
    total = sum(b["ms_per_step"] for b in bd)
    assert abs(total - r["kernel_sum_ms_per_step"]) <= 1e-3 and abs(total - (0.5 + 0.23 + 0.588)) < 1e-3


def test_roofline_side_follows_arithmetic_intensity():
    # a K = 1024 GEMM family (619 flop/B > the 312 ridge) is priced against the MFMA peak, a K = 64 one against HBM
    big = [{"name": "dle_gemm", "tag": "32768x4096x1024", "calls": 10, "ms": 2.4, "bytes": 10 * 343e6, "flops": 10 * 275e9}]
    r, _ = bench.roofline_from(FakeTimer(big), 10, "bert", 256, 80.0)
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == bench.MFMA_PEAK_TFLOPS
    assert abs(r["achieved"] - 275e9 / 0.24e-3 / 1e12) < 2 and abs(r["frac"] - r["achieved"] / 2500.0) < 1e-3
    small = [{"name": "dle_gemm", "tag": "802816x256x64", "calls": 10, "ms": 1.9, "bytes": 10 * 514e6, "flops": 10 * 26.3e9}]
    r, _ = bench.roofline_from(FakeTimer(small), 10, "rn50", 256, 23.0)
    assert r["bound"] == "hbm" and r["frac_mfma"] is not None and r["frac_hbm"] == r["frac"]
    assert r["arithmetic_intensity"] < r["ridge"] == round(2500e12 / 8000e9, 1)


def test_breakdown_folds_the_tail_into_one_row(monkeypatch):
    rows = [{"name": "k%02d" % i, "tag": None, "calls": 10, "ms": 10.0 - i * 0.5, "bytes": 1e9, "flops": 0.0} for i in range(16)]
    monkeypatch.setenv("DLE_BENCH_BREAKDOWN", "12")
    r, bd = bench.roofline_from(FakeTimer(rows), 10, "dlrm", 65536, 3.0)
    assert len(bd) == 13 and bd[-1]["kernel"].startswith("(other: 4")
    assert abs(sum(b["ms_per_step"] for b in bd) - sum(x["ms"] for x in rows) / 10) < 1e-3
    assert abs(r["kernel_sum_ms_per_step"] - sum(x["ms"] for x in rows) / 10) < 1e-3


def test_small_operand_launches_are_replayed_cold():
    rows = [{"name": "dle_gemm", "tag": "128x4096x1536", "calls": 100, "ms": 1.0, "bytes": 100 * 13e6, "flops": 100 * 1.6e9}]
    t = FakeTimer(rows, {("dle_gemm", "128x4096x1536"): 0.013})
    _, bd = bench.roofline_from(t, 10, "tacotron2", 1000, 120.0)
    assert t.asked == [("dle_gemm", "128x4096x1536", True)]                # < 32 MB of operands per launch -> cache flush in front


def test_compact_record_is_small_and_complete():
    t = FakeTimer(_rows())
    r, bd = bench.roofline_from(t, 10, "rn50", 256, 23.0)
    rec = {"value": 11130.4, "unit": "samples/s", "steps": 10, "warmup": 2, "ms_per_step": 23.0, "scaling": "weak", "dtype": "bf16",
           "config": {"workload": "x"}, "final_loss": 6.9, "roofline": r, "kernel_breakdown": bd}
    cpu = {"value": 4.7, "unit": "samples/s", "cores": 128, "kind": "port", "steps": 3, "sample": "a long description " * 10}
    c = bench.compact("rn50", rec, cpu)
    for k in ("value", "unit", "ms_per_step", "steps", "dtype", "workload", "roofline", "cpu_baseline"):
        assert k in c, k
    for k in ("bound", "frac", "step_frac", "kernel", "ms_per_step", "traffic"):
        assert k in c["roofline"], k
    assert c["unit"] == "img/s" and c["cpu_baseline"] == {"value": 4.7, "cores": 128, "kind": "port", "steps": 3}
    assert len(json.dumps(c)) < 560                                      # five of them fit the driver's 4 KB tail
    assert bench.compact("bert", {"error": "boom"}, None) == {"error": "boom"}
    assert bench.compact("bert", None, None) == {"error": "no record"}


def test_nested_steps_time_enough_steps():
    # nested workloads are timed for >= 20 steps (Tacotron2 >= 8)
    for name, (steps, warm) in bench.NESTED_STEPS.items():
        if name == "bert_acc32":           # a step = 32 micro-batches (~2.2 s): 3 timed steps = 96 forward / backward passes
            assert steps >= 3 and warm >= 1
            continue
        assert steps >= (8 if name == "tacotron2" else 10 if name == "waveglow" else 20), (name, steps)
        assert warm >= 2


def test_line_says_where_traffic_comes_from_and_which_process_group_ran():
    """`roofline.traffic` is a lookup in the committed counter table (the line says so, with the round it was collected in), and
    the line names the process group it was measured under (backend, world) so that a scaling record can be checked against it."""
    import bench
    src = bench.traffic_source()
    assert src is not None and "profiles/traffic.json" in src and "round" in src and "not measured in this run" in src
    info = bench.comm_info()                                           # no process group in this test
    assert (info["backend"], info["world"], info["ranks_seen"]) == (None, 1, 1) and "torch.distributed" in info["path"]
