# round-4 evidence, part 3: PMC traffic of the heaviest launches + matrix-pipe / LDS counters of the new kernels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/collect_traffic.sh rn50 bert dlrm > gpurun_out/r04_traffic_collect.log 2>&1; tail -5 gpurun_out/r04_traffic_collect.log
echo "\"dle_gemm[256x64x802816]\": $(bash tools/pmc_traffic.sh rn50 'dle_gemm[256x64x802816]')" > gpurun_out/r04_traffic_wgrad1x1.json; cat gpurun_out/r04_traffic_wgrad1x1.json
bash tools/pmc_r04_kernels.sh > gpurun_out/r04_pmc_new_kernels.txt 2>&1; cat gpurun_out/r04_pmc_new_kernels.txt | head -80
# rocprofv3's per-pass output directories are scratch: only the summaries travel back (gpurun merges <= 64 MiB)
find gpurun_out -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
du -sh gpurun_out
