# A/B of environment switches on the headline workload: bash tools/gpu_ab.sh "VAR=1 VAR2=0" "VAR=0" ...
for cfg in "$@"; do
  for rep in 1 2; do
    env $cfg timeout 300 python bench.py --steps 8 --warmup 2 --no-e2e --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', round(d['value'],2), d['phase_ms_per_iteration'])"
  done
done
