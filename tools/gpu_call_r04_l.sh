cd $GRAFT_REPO_ROOT
for m in 1 0 1 0; do DLE_EMB_SPEC=$m python bench.py --workload dlrm --no-nested --no-cpu-baseline --no-kernel-timer --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('emb_spec=$m', d['ms_per_step'], d['value'])"; done
timeout 300 python -m pytest tests/test_gpu_dlrm_ops.py -x -q 2>&1 | tail -2
