# round-4 evidence, part 2: rocprofv3 kernel tables of the five workloads (final tree)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_all.sh r04 rn50 bert dlrm waveglow tacotron2
for w in rn50 bert dlrm waveglow tacotron2; do head -9 gpurun_out/r04_${w}_kernel_stats.txt | cut -c1-150; cut -c1-200 gpurun_out/r04_${w}_bench_under_rocprof.json; done
# the single-stream table of ResNet-50 (every kernel alone on the chip: what the per-kernel numbers of DESIGN.md are read from)
cd /tmp && export TMPDIR=/tmp
DLE_RN50_WGRAD_STREAM=0 DLE_RN50_BRANCH_STREAM=0 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04s -o x -- python $GRAFT_REPO_ROOT/bench.py --workload rn50 --no-nested --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timer > $GRAFT_REPO_ROOT/gpurun_out/r04_rn50_bench_single_stream.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find gpurun_out/prof_r04s -name '*.db' | head -1) > gpurun_out/r04_rn50_kernel_stats_single_stream.txt 2>&1; rm -rf gpurun_out/prof_r04s
head -40 gpurun_out/r04_rn50_kernel_stats_single_stream.txt | cut -c1-150
