cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_stem.py -x -q 2>&1 | tail -15 > gpurun_out/r04a_stem_tests.log; cat gpurun_out/r04a_stem_tests.log
python tools/probes/stem_bench.py > gpurun_out/r04a_stem_bench.txt 2>&1; cat gpurun_out/r04a_stem_bench.txt
python -m pytest tests/test_gpu_rn50_step.py tests/test_gpu_convnet_ops.py -x -q 2>&1 | tail -8 > gpurun_out/r04a_rn50_tests.log; cat gpurun_out/r04a_rn50_tests.log
python -m pytest tests/test_gpu_bert_step.py -x -q -k "large_24 or losses_match" -s 2>&1 | tail -25 > gpurun_out/r04a_bert24.log; cat gpurun_out/r04a_bert24.log
DLE_BENCH_SHAPES=40 python bench.py --workload rn50 --no-nested --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r04a_rn50.json 2> gpurun_out/r04a_rn50.err; cut -c1-300 gpurun_out/r04a_rn50.json; cp gpurun_out/bench_detail.json gpurun_out/r04a_detail_rn50.json
timeout 900 python -m pytest tests/test_gpu_bench_multirank.py -x -q 2>&1 | tail -25 > gpurun_out/r04a_bench_multirank.log; cat gpurun_out/r04a_bench_multirank.log
