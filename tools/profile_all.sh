#!/bin/bash
# Per-workload rocprofv3 kernel trace of bench.py (run through gpurun): writes gpurun_out/<tag>_<w>_kernel_stats.txt and
# the bench line of the same run.   usage: tools/profile_all.sh <tag> [workloads...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
for w in "${@:-rn50 bert dlrm}"; do
  for ww in $w; do
    out=$R/gpurun_out/prof_${tag}_$ww
    rm -rf $out
    (cd $R && rocprofv3 --kernel-trace --stats -d $out -o x -- python bench.py --workload $ww --no-nested --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timer > $R/gpurun_out/${tag}_${ww}_bench_under_rocprof.json 2> $R/gpurun_out/prof_${tag}_$ww.err)
    db=$(find $out -name '*.db' | head -1)
    python $R/tools/rocpd_stats.py $db > $R/gpurun_out/${tag}_${ww}_kernel_stats.txt 2>&1
    rm -rf $out $R/gpurun_out/prof_${tag}_$ww.err     # raw traces stay on the box: gpurun_out/ is capped at 64 MiB
  done
done
