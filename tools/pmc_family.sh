#!/bin/bash
# HBM-side traffic of a whole kernel FAMILY inside the real step (run through gpurun): two PMC passes (FETCH_SIZE, WRITE_SIZE; own
# runs, kernel trace only) over a short eager bench.py run; the counters of every dispatch whose HIP symbol matches the family's
# regular expression are summed and divided by the number of C-ABI launches of the family (split-K reduce kernels belong to
# their GEMM launch).  For families of MANY SMALL launches whose operands outlive L2 only inside the step (the few-row GEMMs of
# Tacotron2 / WaveGlow: a back-to-back replay of one launch finds its weights in L2 and reports no traffic at all).
# usage: tools/pmc_family.sh <workload> <entry point> <symbol regex> <regex of symbols that count as launches>  -> one JSON line
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
wl=$1; entry=$2; rx=$3; launch_rx=$4
for c in FETCH_SIZE WRITE_SIZE; do
  out=$R/gpurun_out/pmcf_$c; rm -rf $out
  (cd $R && DLE_T2_GRAPH=0 rocprofv3 --kernel-trace --pmc $c -d $out -o x --output-format csv -- python bench.py --workload $wl --no-nested --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer > $R/gpurun_out/pmcf_$c.log 2>&1)
done
python - "$R" "$wl" "$entry" "$rx" "$launch_rx" <<'PY'
import csv, sys, glob, json, re
R, wl, entry, rx, launch_rx = sys.argv[1:6]
rx, launch_rx = re.compile(rx), re.compile(launch_rx)
res = {"workload": wl, "kernel": "family:%s@%s" % (entry, wl), "symbols": rx.pattern, "steps_profiled": 3}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(R + "/gpurun_out/pmcf_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    tot, launches, syms = 0.0, 0, set()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c or not rx.search(r["Kernel_Name"]):
            continue
        # dle_gemm_batched launches the same HIP symbol with grid.z = batch: another entry point, not part of the dle_gemm family
        gz, wz = r.get("Grid_Size_Z"), r.get("Workgroup_Size_Z")
        if gz is not None and wz is not None and int(gz) > int(wz):
            continue
        tot += float(r["Counter_Value"])
        syms.add(r["Kernel_Name"][:60])
        launches += 1 if launch_rx.search(r["Kernel_Name"]) else 0
    res["launches_" + c] = launches
    res[c + "_KB_per_launch"] = tot / max(launches, 1)
    res["hip_kernels"] = sorted(syms)
res["fetch_bytes_per_launch"] = res["FETCH_SIZE_KB_per_launch"] * 1024 * 2     # gfx950 correction (x2), MI355X_MICROARCH.md "HBM"
res["write_bytes_per_launch"] = res["WRITE_SIZE_KB_per_launch"] * 1024
res["traffic_bytes_per_launch"] = res["fetch_bytes_per_launch"] + res["write_bytes_per_launch"]
print(json.dumps(res))
PY
rm -rf $R/gpurun_out/pmcf_FETCH_SIZE $R/gpurun_out/pmcf_WRITE_SIZE
