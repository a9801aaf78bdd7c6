cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_convnet_ops.py tests/test_gpu_conv3x3_wgrad.py tests/test_gpu_rn50_step.py -x -q 2>&1 | tail -6 > gpurun_out/r04f_rn50_tests.log; cat gpurun_out/r04f_rn50_tests.log
timeout 120 python tools/probes/wgrad3x3_bench.py > gpurun_out/r04f_w3_bench.txt 2>&1; cat gpurun_out/r04f_w3_bench.txt
timeout 600 python -m pytest tests/test_gpu_dlrm_ops.py tests/test_gpu_dlrm_step.py -x -q 2>&1 | tail -6 > gpurun_out/r04f_dlrm_tests.log; cat gpurun_out/r04f_dlrm_tests.log
DLE_BENCH_SHAPES=12 python bench.py --workload dlrm --no-nested --no-cpu-baseline --steps 100 --warmup 20 > gpurun_out/r04f_dlrm.json 2> gpurun_out/r04f_dlrm.err; cut -c1-200 gpurun_out/r04f_dlrm.json; cp gpurun_out/bench_detail.json gpurun_out/r04f_detail_dlrm.json
DLE_BENCH_SHAPES=60 python bench.py --workload rn50 --no-nested --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r04f_rn50.json 2> gpurun_out/r04f_rn50.err; cut -c1-200 gpurun_out/r04f_rn50.json; cp gpurun_out/bench_detail.json gpurun_out/r04f_detail_rn50.json
