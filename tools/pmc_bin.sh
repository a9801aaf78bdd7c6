#!/bin/bash
# PMC passes over a standalone binary: pmc_bin.sh "<kernel substring>" "<counter group>;<counter group>;..." -- cmd args...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
sub=$1; groups=$2; shift 3
IFS=';' read -ra GR <<< "$groups"
for grp in "${GR[@]}"; do
  out=$R/gpurun_out/pmc_tmp; rm -rf $out
  rocprofv3 --kernel-trace --pmc $grp -d $out -o x --output-format csv -- "$@" > /dev/null 2>&1
  python - "$out" "$sub" <<'PY'
import csv, sys, glob, collections
d, sub = sys.argv[1], sys.argv[2]
kt = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kt[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
agg = collections.defaultdict(lambda: [0.0, 0, 0.0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub not in k: continue
        a = agg[(k[:40], r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1; a[2] += kt.get(r["Dispatch_Id"], 0)
for (k, c), (v, n, ns) in sorted(agg.items()):
    print("%-42s %-30s %16.0f  (n=%d, avg dur %.1f us)" % (k, c, v / n, n, ns / n / 1e3))
PY
done
