#!/bin/bash
# Round-6 evidence, run through gpurun from the repo root:  tools/r06_evidence.sh <section> ...
#   shapes   RN50 bench line with the per-(entry point, shape) rows (DLE_BENCH_SHAPES) -> gpurun_out/r06_rn50_shapes.json
#   pmcrn50  SQ / LDS / GRBM counters of the ResNet-50 convolution kernels inside the real step -> gpurun_out/r06_pmc_rn50.txt
#   bench    python bench.py (the driver's command) -> gpurun_out/r06_bench_default{,_detail}.json
#   prof     rocprofv3 --kernel-trace --stats tables (multi-stream and single-stream); W="rn50 bert dlrm"
#   traffic  tools/collect_traffic.sh $W
#   tests    tail of pytest -m gpu
R=${GRAFT_REPO_ROOT:-/root/repo}
W=${W:-rn50 bert dlrm}
cd $R; mkdir -p gpurun_out
for sec in "$@"; do case $sec in
shapes)
  for w in ${SW:-rn50}; do
    DLE_BENCH_SHAPES=${NSHAPES:-140} DLE_BENCH_REPLAY=${NREPLAY:-60} python bench.py --workload $w --no-nested --no-cpu-baseline > gpurun_out/r06_${w}_shapes${TAG}.json 2> gpurun_out/r06_${w}_shapes.err
    cp gpurun_out/bench_detail.json gpurun_out/r06_${w}_shapes${TAG}_detail.json
    python tools/print_shapes.py gpurun_out/r06_${w}_shapes${TAG}_detail.json > gpurun_out/r06_${w}_shapes${TAG}.txt; head -${NSHAPES:-140} gpurun_out/r06_${w}_shapes${TAG}.txt; done ;;
pmcrn50)
  G="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE"
  { echo "# tools/r06_evidence.sh pmcrn50: rocprofv3 --kernel-trace --pmc <group> (one run per group) over tools/replay_step.py rn50 (3 steps, batch 256, bf16)"
    echo "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES = 32 x MFMAs summed over SIMDs;"
    echo "# GRBM_GUI_ACTIVE is summed over the 8 XCDs"
    DLE_RN50_WGRAD_STREAM=0 DLE_RN50_BRANCH_STREAM=0 tools/pmc_multi.sh "$G" "conv3x3_kernel,conv3x3_wgrad_kernel,gemm_expand_kernel,conv_bnload_kernel,conv_bnbwd_kernel,conv3x3_bn" -- python tools/replay_step.py rn50; } > gpurun_out/r06_pmc_rn50${TAG}.txt 2>&1
  tail -60 gpurun_out/r06_pmc_rn50${TAG}.txt ;;
bench)
  python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
  cp gpurun_out/bench_detail.json gpurun_out/r06_bench_default_detail.json
  tail -c 3000 gpurun_out/r06_bench_default.json ;;
prof)
  tools/profile_all.sh r06 $W
  for w in $W; do mv gpurun_out/r06_${w}_kernel_stats.txt gpurun_out/r06_${w}_kernel_stats_multi_stream.txt; mv gpurun_out/r06_${w}_bench_under_rocprof.json gpurun_out/r06_${w}_bench_under_rocprof_multi_stream.json; done
  DLE_RN50_WGRAD_STREAM=0 DLE_RN50_BRANCH_STREAM=0 DLE_BERT_WGRAD_STREAM=0 DLE_DLRM_TWO_STREAMS=0 tools/profile_all.sh r06 $W
  for w in $W; do mv gpurun_out/r06_${w}_kernel_stats.txt gpurun_out/r06_${w}_kernel_stats_single_stream.txt; mv gpurun_out/r06_${w}_bench_under_rocprof.json gpurun_out/r06_${w}_bench_single_stream.json; done
  for w in $W; do head -24 gpurun_out/r06_${w}_kernel_stats_single_stream.txt; done ;;
traffic)
  tools/collect_traffic.sh $W > gpurun_out/r06_traffic_collect.log 2>&1; tail -5 gpurun_out/r06_traffic_collect.log ;;
tests)
  python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r06_gpu_tests_tail.log; tail -5 gpurun_out/r06_gpu_tests_tail.log ;;
esac; done
