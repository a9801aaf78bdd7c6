#!/bin/bash
# PMC passes (own runs, kernel-trace only) over one GEMM flavour; prints per-kernel counter averages.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mode=$1; M=$2; N=$3; K=$4
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS"; do
  out=$R/gpurun_out/pmc_tmp; rm -rf $out
  rocprofv3 --kernel-trace --pmc $grp -d $out -o x --output-format csv -- python $R/tools/gemm_one.py $mode $M $N $K 4 > /dev/null 2>&1
  f=$(find $out -name '*counter_collection.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "gemm2" not in k and "splitk" not in k:
        continue
    a = agg[(k[:60], r["Counter_Name"])]
    a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(agg.items()):
    print("%-62s %-34s %16.0f (avg over %d)" % (k, c, v / n, n))
PY
done
