# A/B of environment switches: bash tools/gpu_ab.sh "VAR=1 VAR2=0" "VAR=0" ...   (WORKLOADS="mh12345 mh01" selects the maps)
for cfg in "$@"; do
  for w in ${WORKLOADS:-mh12345}; do
  for rep in ${REPS:-1 2}; do
    env $cfg timeout 300 python bench.py --workload $w --steps 8 --warmup 2 --no-e2e --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', '$w', round(d['value'],2), d['phase_ms_per_iteration'])"
  done
  done
done
