# one-hot MFMA tiny tables: tests + in-step A/B + kernel times
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dlrm_ops.py tests/test_gpu_dlrm_step.py -x -q 2>&1 | tail -8
for v in 1 1; do
  DLE_EMB_ONEHOT=$v python bench.py --workload dlrm --no-nested --no-cpu-baseline --no-kernel-timer --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('onehot=$v', d['ms_per_step'], d['value'])"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_q -o x -- python $GRAFT_REPO_ROOT/bench.py --workload dlrm --no-nested --no-cpu-baseline --no-kernel-timer --steps 10 --warmup 3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(ls gpurun_out/prof_q/*/*_results.db gpurun_out/prof_q/*_results.db 2>/dev/null | head -1) --skip-first 0 2>/dev/null | grep -i "emb_\|head_" | cut -c1-150
rm -rf gpurun_out/prof_q
