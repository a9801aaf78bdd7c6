# round-4 baseline: per-shape breakdown of the three metric workloads (DLE_BENCH_SHAPES), one gpurun call
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in rn50 bert dlrm; do
  st=20; wu=5
  [ $w = bert ] && st=8 && wu=2
  [ $w = dlrm ] && st=100 && wu=20
  DLE_BENCH_SHAPES=120 DLE_BENCH_REPLAY=60 python bench.py --workload $w --no-nested --no-cpu-baseline --steps $st --warmup $wu > gpurun_out/r04_base_$w.json 2> gpurun_out/r04_base_$w.err
  cp gpurun_out/bench_detail.json gpurun_out/r04_base_detail_$w.json
  cut -c1-400 gpurun_out/r04_base_$w.json
done
