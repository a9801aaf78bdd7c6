#!/bin/bash
# HBM-side traffic of one entry point of a bench workload: two PMC passes (FETCH_SIZE, WRITE_SIZE; own runs, kernel
# trace only) over tools/replay_kernel.py.  The replayed launches are the most frequent (kernel, grid) pair of the run.
# usage: tools/pmc_traffic.sh <workload> [key]   -> one JSON line (bytes per launch; FETCH_SIZE doubled: gfx950 tallies
# 128-byte requests at 64 B for 16-byte-per-lane streams, MI355X_MICROARCH.md "HBM")
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
wl=$1; key=$2
for c in FETCH_SIZE WRITE_SIZE; do
  out=$R/gpurun_out/pmc_$c; rm -rf $out
  if [ -n "$key" ]; then rocprofv3 --kernel-trace --pmc $c -d $out -o x --output-format csv -- python $R/tools/replay_kernel.py --workload $wl --key "$key" > $R/gpurun_out/pmc_traffic_$c.log 2>&1
  else rocprofv3 --kernel-trace --pmc $c -d $out -o x --output-format csv -- python $R/tools/replay_kernel.py --workload $wl > $R/gpurun_out/pmc_traffic_$c.log 2>&1; fi
done
python - "$R" "$wl" <<'PY'
import csv, sys, glob, collections, json
R, wl = sys.argv[1], sys.argv[2]
res = {"workload": wl}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    d = R + "/gpurun_out/pmc_" + c
    log = open(R + "/gpurun_out/pmc_traffic_%s.log" % c).read()
    for line in log.splitlines():
        if line.startswith('{"replayed"'):
            res["kernel"] = json.loads(line)["replayed"]
    rows = [r for r in csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])) if r["Counter_Name"] == c]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    iters = 50
    for line in log.splitlines():
        if line.startswith('{"replayed"'):
            iters = json.loads(line)["iters"]
    names = [r["Kernel_Name"] for r in rows]
    # the replay is the tail of the run: `iters` repetitions of the k kernels one entry-point call launches
    k = 1
    while k <= 6 and not all(names[-1 - i] == names[-1 - i - k] for i in range(k * iters - k)):
        k += 1
    tail = rows[-k * iters:]
    res["hip_kernels"] = sorted(set(r["Kernel_Name"][:70] for r in tail)); res["kernels_per_launch"] = k
    res["launches_" + c] = iters
    res[c + "_KB_per_launch"] = sum(float(r["Counter_Value"]) for r in tail) / iters
res["fetch_bytes_per_launch"] = res["FETCH_SIZE_KB_per_launch"] * 1024 * 2     # gfx950 correction (x2)
res["write_bytes_per_launch"] = res["WRITE_SIZE_KB_per_launch"] * 1024          # uncalibrated on gfx950
res["traffic_bytes_per_launch"] = res["fetch_bytes_per_launch"] + res["write_bytes_per_launch"]
print(json.dumps(res))
PY
