# dgrad + bias-gradient fusion: tests + A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_gemm_colsum.py tests/test_gpu_dlrm_step.py tests/test_gpu_dlrm_head.py tests/test_gpu_convnet_ops.py -x -q 2>&1 | tail -8
for v in 1 0 1 0; do
  DLE_GEMM_COLSUM=$v python bench.py --workload dlrm --no-nested --no-cpu-baseline --no-kernel-timer --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gemm_colsum=$v', d['ms_per_step'], d['value'])"
done
