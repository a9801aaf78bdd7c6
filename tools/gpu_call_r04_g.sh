cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_conv_bnload.py -x -q 2>&1 | tail -12 > gpurun_out/r04g_bnload_tests.log; cat gpurun_out/r04g_bnload_tests.log
timeout 600 python -m pytest tests/test_gpu_rn50_step.py tests/test_gpu_baseline_shapes.py -x -q 2>&1 | tail -6 > gpurun_out/r04g_rn50_tests.log; cat gpurun_out/r04g_rn50_tests.log
for m in 1 0 1 0; do DLE_RN50_FUSE_BN=$m python bench.py --workload rn50 --no-nested --no-cpu-baseline --no-kernel-timer --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse_bn=$m', d['ms_per_step'], d['value'])"; done
DLE_BENCH_SHAPES=70 python bench.py --workload rn50 --no-nested --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r04g_rn50.json 2> gpurun_out/r04g_rn50.err; cut -c1-200 gpurun_out/r04g_rn50.json; cp gpurun_out/bench_detail.json gpurun_out/r04g_detail_rn50.json
timeout 300 python -m pytest tests/test_gpu_bert_step.py -x -q -k "losses_match or first_step" 2>&1 | tail -4
DLE_BENCH_SHAPES=12 python bench.py --workload bert --no-nested --no-cpu-baseline --steps 8 --warmup 2 > gpurun_out/r04g_bert.json 2> gpurun_out/r04g_bert.err; cut -c1-200 gpurun_out/r04g_bert.json; cp gpurun_out/bench_detail.json gpurun_out/r04g_detail_bert.json
