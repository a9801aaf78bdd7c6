cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_dlrm_ops.py tests/test_gpu_dlrm_step.py tests/test_gpu_shims.py -x -q 2>&1 | tail -6
for w in 2 3 1; do DLE_DOT_BWD_WG_PER_CU=$w python tools/probes/dot_bwd_bench.py 2>&1 | tail -1; done
for i in 1 2; do python bench.py --workload dlrm --no-nested --no-cpu-baseline --no-kernel-timer --steps 200 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dlrm', d['ms_per_step'], d['value'])"; done
