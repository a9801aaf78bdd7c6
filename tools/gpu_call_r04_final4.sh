# round-4 evidence, part 4: the single-stream rocprofv3 table of DLRM (every kernel alone on the chip)
cd /tmp && export TMPDIR=/tmp
DLE_DLRM_TWO_STREAMS=0 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04d -o x -- python $GRAFT_REPO_ROOT/bench.py --workload dlrm --no-nested --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer > $GRAFT_REPO_ROOT/gpurun_out/r04_dlrm_bench_single_stream.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find gpurun_out/prof_r04d -name '*.db' | head -1) > gpurun_out/r04_dlrm_kernel_stats_single_stream.txt 2>&1; rm -rf gpurun_out/prof_r04d
head -30 gpurun_out/r04_dlrm_kernel_stats_single_stream.txt | cut -c1-150; cut -c1-200 gpurun_out/r04_dlrm_bench_single_stream.json
