#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_tmp; rm -rf $out
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVES -d $out -o x --output-format csv -- python $R/tools/gemm_one.py $1 $2 $3 $4 4 > /dev/null 2>&1
python - "$out" <<'PY'
import csv, sys, glob, collections
d = sys.argv[1]
kt = {}
for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])):
    kt[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for r in csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])):
    k = r["Kernel_Name"]
    if "gemm" not in k: continue
    name, ns = kt.get(r["Dispatch_Id"], ("?", 0))
    v = float(r["Counter_Value"])
    print("%-40s %-22s %14.0f  dur_us %.1f  -> %.3f GHz-equivalent" % (k[:40], r["Counter_Name"], v, ns / 1e3, v / max(ns, 1)))
PY
