cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_convnet_ops.py tests/test_gpu_rn50_step.py tests/test_gpu_baseline_shapes.py tests/test_gpu_conv_bnload.py tests/test_gpu_graph.py tests/test_gpu_checkpoint.py -x -q 2>&1 | tail -6
for m in 1 0 1 0; do DLE_RN50_FUSE_BN=$m python bench.py --workload rn50 --no-nested --no-cpu-baseline --no-kernel-timer --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse_bn=$m', d['ms_per_step'], d['value'])"; done
DLE_BENCH_SHAPES=70 python bench.py --workload rn50 --no-nested --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r04k_rn50.json 2> gpurun_out/r04k_rn50.err; cut -c1-200 gpurun_out/r04k_rn50.json; cp gpurun_out/bench_detail.json gpurun_out/r04k_detail_rn50.json
