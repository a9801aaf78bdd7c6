# bn3 reduction in the producing GEMM's epilogue: tests + RN50 A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_gemm_bnred.py tests/test_gpu_rn50_step.py tests/test_gpu_conv_bnbwd.py -x -q 2>&1 | tail -6
for v in 1 0; do
  DLE_RN50_FUSE_BNRED=$v python bench.py --workload rn50 --no-nested --no-cpu-baseline --no-kernel-timer --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse_bnred=$v', d['ms_per_step'], d['value'], d.get('final_loss'))"
done
DLE_BENCH_SHAPES=60 python bench.py --workload rn50 --no-nested --no-cpu-baseline --steps 10 --warmup 3 > /dev/null 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_detail.json')); h=d.get('headline',d)
for b in h['kernel_breakdown']:
    if 'timing' in b and ('bnred' in b['kernel'] or 'bn_bwd_reduce' in b['kernel'] or '+src+aux' in b['kernel']): print(b)
PY
