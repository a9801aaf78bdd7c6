cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_stem.py -x -q 2>&1 | tail -15 > gpurun_out/r04b_stem_tests.log; cat gpurun_out/r04b_stem_tests.log
python tools/probes/stem_bench.py > gpurun_out/r04b_stem_bench.txt 2>&1; cat gpurun_out/r04b_stem_bench.txt
DLE_STEM_FWD_WG_PER_CU=4 python tools/probes/stem_bench.py 2>&1 | tail -2
python -m pytest tests/test_gpu_rn50_step.py tests/test_gpu_convnet_ops.py tests/test_gpu_baseline_shapes.py -x -q 2>&1 | tail -8 > gpurun_out/r04b_rn50_tests.log; cat gpurun_out/r04b_rn50_tests.log
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_stem -o x -- python $GRAFT_REPO_ROOT/tools/probes/stem_bench.py > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find gpurun_out/prof_stem -name "*_results.db" | head -1) 2>&1 | head -14
