cd $GRAFT_REPO_ROOT
python tools/probes/smallm_policy.py > gpurun_out/smallm_policy_depth.txt 2>&1; cat gpurun_out/smallm_policy_depth.txt
bash tools/profile_all.sh r03 tacotron2 waveglow rn50 bert dlrm
for w in tacotron2 waveglow rn50 bert dlrm; do head -14 gpurun_out/r03_${w}_kernel_stats.txt; cat gpurun_out/r03_${w}_bench_under_rocprof.json | cut -c1-300; done
echo "{" > gpurun_out/traffic_family.json
echo "\"family:dle_gemm@tacotron2\": $(bash tools/pmc_family.sh tacotron2 dle_gemm 'gemm_smallm_kernel|gemm2_kernel|splitk_reduce' 'gemm_smallm_kernel|gemm2_kernel')," >> gpurun_out/traffic_family.json
echo "\"family:dle_gemm@waveglow\": $(bash tools/pmc_family.sh waveglow dle_gemm 'gemm_smallm_kernel|gemm2_kernel|splitk_reduce' 'gemm_smallm_kernel|gemm2_kernel')" >> gpurun_out/traffic_family.json
echo "}" >> gpurun_out/traffic_family.json
cat gpurun_out/traffic_family.json
tail -3 gpurun_out/pmcf_FETCH_SIZE.log
