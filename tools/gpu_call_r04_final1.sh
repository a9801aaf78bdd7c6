# round-4 evidence, part 1: the full GPU suite and the default bench line from the final tree
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
git log -1 --format=%h 2>/dev/null
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r04_gpu_tests.log 2>&1; tail -4 gpurun_out/r04_gpu_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err; cp gpurun_out/bench_detail.json gpurun_out/r04_bench_default_detail.json
python tools/print_bench.py gpurun_out/r04_bench_default.json
