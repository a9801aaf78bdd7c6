cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_convnet_ops.py tests/test_gpu_rn50_step.py tests/test_gpu_baseline_shapes.py tests/test_gpu_conv3x3_wgrad.py tests/test_gpu_stem.py -x -q 2>&1 | tail -8 > gpurun_out/r04e_rn50_tests.log; cat gpurun_out/r04e_rn50_tests.log
DLE_BENCH_SHAPES=60 python bench.py --workload rn50 --no-nested --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r04e_rn50.json 2> gpurun_out/r04e_rn50.err; cut -c1-200 gpurun_out/r04e_rn50.json; cp gpurun_out/bench_detail.json gpurun_out/r04e_detail_rn50.json
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_w3 -o x -- python $GRAFT_REPO_ROOT/tools/probes/wgrad3x3_bench.py > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find gpurun_out/prof_w3 -name "*_results.db" | head -1) 2>&1 | head -14
