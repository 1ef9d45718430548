mkdir -p gpurun_out
for rep in 1 2; do for tr in 0 1; do
  COVGPU_HOST_TR=$tr timeout 300 python bench.py --workload mh12345 --steps 5 --warmup 2 --no-e2e --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mh12345 host_tr=$tr', round(d['value'],2), d['phase_ms_per_iteration'], round(d['ms_per_step'],2))"
done; done
