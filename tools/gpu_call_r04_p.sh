cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
DLE_BENCH_SHAPES=12 python bench.py --workload dlrm --no-nested --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r04p_dlrm.json 2> gpurun_out/r04p_dlrm.err
cp gpurun_out/bench_detail.json gpurun_out/r04p_detail_dlrm.json
python /dev/stdin <<'PY'
import json
d=json.load(open('gpurun_out/r04p_detail_dlrm.json')); h=d.get('headline',d)
for b in h['kernel_breakdown']:
    if 'timing' in b: print("%-60s %8.1f us x %4.1f %s"%(b['kernel'][:60], b['ms_per_step']/b['calls_per_step']*1e3, b['calls_per_step'], b['timing']))
print(h['ms_per_step'])
PY
