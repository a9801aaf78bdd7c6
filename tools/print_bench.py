"""Compact view of a bench.py JSON line: python tools/print_bench.py <file> [<detail file to diff against> ...]"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print("%s  value %.1f  %.3f ms/step  family %s %.3f ms  frac %s  step_frac %s  traffic/alg %s" % (
    d["config"]["workload"][:40], d["value"], d["ms_per_step"], r.get("kernel"), r.get("ms_per_step", 0), r.get("frac"),
    r.get("step_frac"), r.get("traffic_over_algorithmic")))
for k, v in (d.get("workloads") or {}).items():
    rr = v.get("roofline") or {}
    print("   %-10s %12.1f %-16s %8.3f ms  %s frac %s step_frac %s traffic %s" % (
        k, v["value"], v["unit"], v["ms_per_step"], rr.get("kernel"), rr.get("frac"), rr.get("step_frac"), rr.get("traffic")))
