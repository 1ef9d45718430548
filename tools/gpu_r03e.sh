mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_full.py tests/test_gpu_parity.py -m gpu -q --timeout 900 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -8
for mask in 0 2 4; do for w in mh12345; do
  COVGPU_CU_MASK=$mask COVGPU_ND_LEAF=600 COVGPU_TRACE_PANELS=1 timeout 300 python bench.py --workload $w --steps 3 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/r03e_bench_${w}_mask$mask.json 2> gpurun_out/r03e_marks_${w}_mask$mask.txt
  grep "covgpu marks" gpurun_out/r03e_marks_${w}_mask$mask.txt | tail -1
  python -c "
import json; d=json.loads(open('gpurun_out/r03e_bench_${w}_mask$mask.json').read().strip().splitlines()[-1]); print('$w mask $mask', round(d['value'],2), d['phase_ms_per_iteration'])"
done; done
COVGPU_ND_LEAF=600 timeout 300 python bench.py --workload mh01 --steps 3 --warmup 1 --no-e2e --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mh01', round(d['value'],2), d['phase_ms_per_iteration'])"
