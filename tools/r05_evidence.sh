#!/bin/bash
# Round-5 evidence, run through gpurun from the repo root:  tools/r05_evidence.sh <section> ...
#   bench    python bench.py (the driver's command) -> gpurun_out/r05_bench_default{,_detail}.json
#   prof     rocprofv3 --kernel-trace --stats tables of rn50 / bert / dlrm, multi-stream and single-stream
#   pmc      SQ / LDS / GRBM counters + FETCH_SIZE / WRITE_SIZE of the ping-pong GEMM (gemm8) (standalone harness, tools/kbench)
#   traffic  tools/collect_traffic.sh rn50 bert dlrm -> gpurun_out/traffic_new.json
#   gemm     tools/gemm8_check.py full, tools/gemm8_ksweep.py, tools/gemm8_splitk_sweep.py
#   tests    tail of pytest -m gpu
#   W="bert dlrm" limits the prof / traffic sections to those workloads
R=${GRAFT_REPO_ROOT:-/root/repo}
W=${W:-rn50 bert dlrm}
cd $R; mkdir -p gpurun_out
for sec in "$@"; do case $sec in
bench)
  python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
  cp gpurun_out/bench_detail.json gpurun_out/r05_bench_default_detail.json
  tail -c 3000 gpurun_out/r05_bench_default.json ;;
prof)
  tools/profile_all.sh r05 $W
  for w in $W; do mv gpurun_out/r05_${w}_kernel_stats.txt gpurun_out/r05_${w}_kernel_stats_multi_stream.txt; mv gpurun_out/r05_${w}_bench_under_rocprof.json gpurun_out/r05_${w}_bench_under_rocprof_multi_stream.json; done
  DLE_RN50_WGRAD_STREAM=0 DLE_RN50_BRANCH_STREAM=0 DLE_BERT_WGRAD_STREAM=0 DLE_DLRM_TWO_STREAMS=0 tools/profile_all.sh r05 $W
  for w in $W; do mv gpurun_out/r05_${w}_kernel_stats.txt gpurun_out/r05_${w}_kernel_stats_single_stream.txt; mv gpurun_out/r05_${w}_bench_under_rocprof.json gpurun_out/r05_${w}_bench_single_stream.json; done
  head -30 gpurun_out/r05_bert_kernel_stats_single_stream.txt ;;
pmc)
  B=$R/tools/kbench/bin/gemm8_bench_plain
  G="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE;FETCH_SIZE;WRITE_SIZE"
  { echo "# tools/r05_evidence.sh pmc: rocprofv3 --kernel-trace --pmc <group> (own runs) over tools/kbench/gemm8_bench (13 launches per run)"
    echo "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES = 32 x MFMAs summed over SIMDs;"
    echo "# GRBM_GUI_ACTIVE is summed over the 8 XCDs; FETCH_SIZE (KB) counts 128-byte requests at 64 B on gfx950: double it (MI355X_MICROARCH.md)"
    for c in "32768 4096 1024 nt plain" "32768 4096 1024 nt bias" "32768 4096 1024 nt gelu" "32768 4096 1024 nn mul" "32768 1024 4096 nn add" "4096 1024 32768 tn plain 4" "8192 8192 8192 nt plain"; do
      echo "== gemm8_bench $c"; tools/pmc_bin.sh gemm8 "$G" -- $B $c; done; } > gpurun_out/r05_pmc_gemm8.txt 2>&1
  tail -40 gpurun_out/r05_pmc_gemm8.txt ;;
traffic)
  tools/collect_traffic.sh $W > gpurun_out/r05_traffic_collect.log 2>&1; tail -5 gpurun_out/r05_traffic_collect.log ;;
gemm)
  python tools/gemm8_check.py full 2>/dev/null | grep -v amdgpu > gpurun_out/r05_gemm8_vs_tile_kernels.jsonl
  { python tools/gemm8_ksweep.py 4096 4096; python tools/gemm8_ksweep.py 32768 4096; } 2>/dev/null | grep -v amdgpu > gpurun_out/r05_gemm8_ksweep.jsonl
  python tools/gemm8_splitk_sweep.py 2>/dev/null | grep -v amdgpu > gpurun_out/r05_gemm8_splitk_sweep.jsonl
  tail -3 gpurun_out/r05_gemm8_vs_tile_kernels.jsonl ;;
tests)
  python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r05_gpu_tests_tail.log; tail -3 gpurun_out/r05_gpu_tests_tail.log ;;
esac; done
