cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
DLE_BENCH_SHAPES=60 python bench.py --workload dlrm --no-nested --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r04t_dlrm.json 2> gpurun_out/r04t_dlrm.err
cp gpurun_out/bench_detail.json gpurun_out/r04t_detail_dlrm.json
tail -c 400 gpurun_out/r04t_dlrm.json
