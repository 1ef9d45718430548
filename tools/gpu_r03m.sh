mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep -E "passed|failed|FAILED|Error|error|assert" | tail -20
COVGPU_PLAN_TIMING=1 python tools/up_time.py 2>&1 | tail -8
timeout 600 python bench.py --workload mh12345 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), d['e2e_call'])"
