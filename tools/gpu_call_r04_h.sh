cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for m in 1 0 1 0; do DLE_RN50_FUSE_BN=$m python bench.py --workload rn50 --no-nested --no-cpu-baseline --no-kernel-timer --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse_bn=$m', d['ms_per_step'], d['value'])"; done
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r04h_gpu_tests.log 2>&1; tail -6 gpurun_out/r04h_gpu_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r04h_bench_default.json 2> gpurun_out/r04h_bench_default.err; cp gpurun_out/bench_detail.json gpurun_out/r04h_bench_default_detail.json
python tools/print_bench.py gpurun_out/r04h_bench_default.json 2>/dev/null | head -30 || cut -c1-600 gpurun_out/r04h_bench_default.json
