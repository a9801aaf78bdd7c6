#!/bin/bash
# Run through gpurun.  For each workload: bench.py names the heaviest launch of the dominant kernel family, two PMC passes measure its HBM-side
# traffic (tools/pmc_traffic.sh), the record lands in gpurun_out/traffic_new.json (merge into profiles/traffic.json).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
echo "{" > gpurun_out/traffic_new.json
first=1
for w in "${@:-rn50}"; do
  key=$(python bench.py --workload $w --no-nested --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['roofline']['heaviest_shape'])")
  rec=$(tools/pmc_traffic.sh $w "$key")
  [ $first -eq 1 ] || echo "," >> gpurun_out/traffic_new.json
  first=0
  echo "\"$key\": $rec" >> gpurun_out/traffic_new.json
done
echo "}" >> gpurun_out/traffic_new.json
cat gpurun_out/traffic_new.json
