#!/bin/bash
# One rocprofv3 --pmc pass per counter GROUP (groups separated by ';') over one command; per-kernel averages for every kernel
# whose name contains one of the comma-separated substrings.  (pmc_any.sh runs the command once per kernel family; a ResNet-50
# step holds all of them, so one run per group is enough.)
# usage: pmc_multi.sh "<grp1 ctrs>;<grp2 ctrs>" "sub1,sub2,..." -- cmd...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
groups=$1; subs=$2; shift 3
args=(); for a in "$@"; do if [ -f "$R/$a" ]; then args+=("$R/$a"); else args+=("$a"); fi; done
IFS=';' read -ra GR <<< "$groups"
for grp in "${GR[@]}"; do
  out=$R/gpurun_out/pmc_tmp; rm -rf $out
  rocprofv3 --kernel-trace --pmc $grp -d $out -o x --output-format csv -- "${args[@]}" > $R/gpurun_out/pmc_multi.log 2>&1
  python - "$out" "$subs" <<'PY'
import csv, sys, glob, collections, re
d, subs = sys.argv[1], sys.argv[2].split(",")
kt = {}
for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])):
    kt[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Grid_Size_X", r.get("Grid_Size", "")))
agg = collections.defaultdict(lambda: [0.0, 0, 0.0])
for r in csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])):
    k = r["Kernel_Name"]
    if not any(s in k for s in subs): continue
    k = re.sub(r"\(.*", "", k)
    dur, grid = kt.get(r["Dispatch_Id"], (0, ""))
    a = agg[(k[:60], grid, r["Counter_Name"])]
    a[0] += float(r["Counter_Value"]); a[1] += 1; a[2] += dur
for (k, g, c), (v, n, ns) in sorted(agg.items()):
    print("%-62s grid %-9s %-28s %16.0f  (n=%d, avg dur %.1f us)" % (k, g, c, v / n, n, ns / n / 1e3))
PY
  rm -rf $out
done
