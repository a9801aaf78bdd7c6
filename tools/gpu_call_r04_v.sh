# sub-lists for the mid tables: tests + probe + in-step A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dlrm_ops.py tests/test_gpu_dlrm_step.py tests/test_gpu_bert_step.py -x -q 2>&1 | tail -6
for v in 1 0; do echo "DLE_EMB_MID=$v"; DLE_EMB_MID=$v python tools/probes/emb_chain_probe.py 2>&1 | grep -v amdgpu.ids | head -2; done
for v in 1 0 1 0; do
  DLE_EMB_MID=$v python bench.py --workload dlrm --no-nested --no-cpu-baseline --no-kernel-timer --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('emb_mid=$v', d['ms_per_step'], d['value'])"
done
