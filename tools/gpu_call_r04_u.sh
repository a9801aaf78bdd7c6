# bias gradient from the GEMM epilogue: tests (DLRM + BERT) + BERT A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_gemm_colsum.py tests/test_gpu_dlrm_step.py tests/test_gpu_bert_step.py -x -q 2>&1 | tail -6
for v in 1 0 1 0; do
  DLE_BERT_FUSE_BIAS_GRAD=$v python bench.py --workload bert --no-nested --no-cpu-baseline --no-kernel-timer --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bert fuse_bias_grad=$v', d['ms_per_step'], d['value'])"
done
