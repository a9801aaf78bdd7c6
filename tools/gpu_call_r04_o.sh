# fused DLRM head: tests + in-step A/B + kernel time
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dlrm_head.py tests/test_gpu_dlrm_step.py -x -q 2>&1 | tail -5
for v in 1 0 1 0; do
  DLE_DLRM_FUSE_HEAD=$v python bench.py --workload dlrm --no-nested --no-cpu-baseline --no-kernel-timer --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse_head=$v', d['ms_per_step'], d['value'])"
done
bash tools/gpu_call_r04_p.sh 2>&1 | grep -i "head\|^[0-9]"
