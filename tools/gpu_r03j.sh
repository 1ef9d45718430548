mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep -E "passed|failed|FAILED|Error|error|assert|world" | tail -30 > gpurun_out/r03j_gpu_tests_tail.txt
cat gpurun_out/r03j_gpu_tests_tail.txt
for w in mh12345; do
  timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --no-e2e --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', round(d['value'],2), d['phase_ms_per_iteration'], round(d['ms_per_step'],2))"
  timeout 300 python bench.py --workload $w --force-shard --steps 5 --warmup 2 --no-e2e --no-cpu-baseline 2>gpurun_out/r03j_fs.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w forced-shard', round(d['value'],2), d['phase_ms_per_iteration'], round(d['ms_per_step'],2), d['config']['sharding'])"
  tail -3 gpurun_out/r03j_fs.err
done
