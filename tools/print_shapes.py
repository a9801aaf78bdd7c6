"""Per-(entry point, shape) rows of a bench.py detail file produced with DLE_BENCH_SHAPES=<n>:
    python tools/print_shapes.py gpurun_out/bench_detail.json"""
import json
import sys

d = json.load(open(sys.argv[1]))["headline"]
print("%s  %.3f ms/step  kernel_sum %s" % (d["config"]["workload"][:50], d["ms_per_step"], (d.get("roofline") or {}).get("kernel_sum_ms_per_step")))
for b in d.get("kernel_breakdown", []):
    print("%-92s %8.4f ms  x%-6s %s" % (b["kernel"][:92], b["ms_per_step"], b["calls_per_step"], b.get("timing", "")))
