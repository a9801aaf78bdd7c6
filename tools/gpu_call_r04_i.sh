cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_wgrad1x1.py tests/test_gpu_conv_bnload.py -x -q 2>&1 | tail -12 > gpurun_out/r04i_tests.log; cat gpurun_out/r04i_tests.log
timeout 200 python tools/probes/wgrad1x1_bench.py > gpurun_out/r04i_w1_bench.txt 2>&1; cat gpurun_out/r04i_w1_bench.txt
timeout 600 python -m pytest tests/test_gpu_rn50_step.py -x -q 2>&1 | tail -4
for m in 1 0 1 0; do DLE_WGRAD1X1=$m python bench.py --workload rn50 --no-nested --no-cpu-baseline --no-kernel-timer --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgrad1x1=$m', d['ms_per_step'], d['value'])"; done
