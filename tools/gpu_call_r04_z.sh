# bn2's reduction inside the fused conv3 backward kernel: tests + RN50 A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_conv_bnbwd.py tests/test_gpu_gemm_bnred.py tests/test_gpu_rn50_step.py -x -q 2>&1 | tail -6
for v in 1 0 1 0; do
  DLE_RN50_FUSE_BNRED2=$v python bench.py --workload rn50 --no-nested --no-cpu-baseline --no-kernel-timer --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse_bnred2=$v', d['ms_per_step'], d['value'], d.get('final_loss'))"
done
