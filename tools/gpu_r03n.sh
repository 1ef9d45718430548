mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep -E "passed|failed|FAILED|Error|error|assert" | tail -20
for leaf in 450 600 780; do for w in mh12345 mh01; do
  COVGPU_ND_LEAF=$leaf timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --no-e2e --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w leaf $leaf', round(d['value'],2), d['phase_ms_per_iteration'], d['config']['layout']['nd_serial_panels'], d['config']['layout']['nd_front_mib'])"
done; done
COVGPU_TRACE_PANELS=1 timeout 300 python bench.py --workload mh12345 --steps 1 --warmup 1 --no-e2e --no-cpu-baseline 2>&1 >/dev/null | grep "covgpu marks" | tail -1
