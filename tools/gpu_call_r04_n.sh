# per-shape tables (DLE_BENCH_SHAPES) of the DLRM and BERT steps
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for wl in dlrm bert; do
  DLE_BENCH_SHAPES=80 python bench.py --workload $wl --no-nested --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r04n_$wl.json 2> gpurun_out/r04n_$wl.err
  cp gpurun_out/bench_detail.json gpurun_out/r04n_detail_$wl.json
  tail -c 300 gpurun_out/r04n_$wl.json
done
