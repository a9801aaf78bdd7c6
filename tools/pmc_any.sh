#!/bin/bash
# usage: pmc_any.sh "<counters>" <kernel-substring> -- cmd...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
ctrs=$1; sub=$2; shift 3
args=(); for a in "$@"; do if [ -f "$R/$a" ]; then args+=("$R/$a"); else args+=("$a"); fi; done
out=$R/gpurun_out/pmc_tmp; rm -rf $out
rocprofv3 --kernel-trace --pmc $ctrs -d $out -o x --output-format csv -- "${args[@]}" > $R/gpurun_out/pmc_any.log 2>&1
python - "$out" "$sub" <<'PY'
import csv, sys, glob, collections
d, sub = sys.argv[1], sys.argv[2]
kt = {}
for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])):
    kt[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
agg = collections.defaultdict(lambda: [0.0, 0, 0.0])
for r in csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])):
    k = r["Kernel_Name"]
    if sub not in k: continue
    a = agg[(k[:48], r["Counter_Name"])]
    a[0] += float(r["Counter_Value"]); a[1] += 1; a[2] += kt.get(r["Dispatch_Id"], 0)
for (k, c), (v, n, ns) in sorted(agg.items()):
    print("%-50s %-28s %16.0f  (n=%d, avg dur %.1f us)" % (k, c, v / n, n, ns / n / 1e3))
PY
