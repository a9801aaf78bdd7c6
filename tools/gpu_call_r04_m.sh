# staggered start of the 256x256 GEMM: probe on the BERT / DLRM shapes, then in-step A/B; + per-shape table of the RN50 step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SH="32768x4096x1024 32768x1024x4096 32768x1024x1024 32768x3072x1024 32768x4096x1024:d 32768x1024x4096:d 32768x1024x1024:d 65536x1024x1024 65536x512x1024"
for st in off 2:0 4:0 8:0 4:1 4:3 16:0; do
  echo "== DLE_GEMM_STAGGER=$st"
  if [ $st = off ]; then python tools/probes/gemm_shapes.py $SH; else DLE_GEMM_STAGGER=$st python tools/probes/gemm_shapes.py $SH; fi
done 2>&1 | tee gpurun_out/r04m_stagger_probe.txt
for st in off 4:0 off 4:0 8:0; do
  if [ $st = off ]; then unset DLE_GEMM_STAGGER; else export DLE_GEMM_STAGGER=$st; fi
  python bench.py --workload bert --no-nested --no-cpu-baseline --no-kernel-timer --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bert stagger=$st', d['ms_per_step'], d['value'])"
done 2>&1 | tee gpurun_out/r04m_stagger_bert.txt
unset DLE_GEMM_STAGGER
DLE_BENCH_SHAPES=120 python bench.py --workload rn50 --no-nested --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r04m_rn50.json 2> gpurun_out/r04m_rn50.err
cp gpurun_out/bench_detail.json gpurun_out/r04m_detail_rn50.json
tail -c 600 gpurun_out/r04m_rn50.json
