cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./tools/probes/bin/mall_bw > gpurun_out/r04d_mall_bw.txt 2>&1; cat gpurun_out/r04d_mall_bw.txt
timeout 300 python -m pytest tests/test_gpu_conv3x3_wgrad.py -x -q 2>&1 | tail -5
timeout 120 python tools/probes/wgrad3x3_bench.py > gpurun_out/r04d_w3_bench.txt 2>&1; cat gpurun_out/r04d_w3_bench.txt
