mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | tail -12 > gpurun_out/r03f_gpu_tests_tail.txt
cat gpurun_out/r03f_gpu_tests_tail.txt
for w in mh12345 mh01; do
  COVGPU_TRACE_PANELS=1 timeout 300 python bench.py --workload $w --steps 3 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/r03f_bench_$w.json 2> gpurun_out/r03f_marks_$w.txt
  grep "covgpu marks" gpurun_out/r03f_marks_$w.txt | tail -1
  python -c "
import json; d=json.loads(open('gpurun_out/r03f_bench_$w.json').read().strip().splitlines()[-1]); print('$w', round(d['value'],2), d['phase_ms_per_iteration'], d['ms_per_step'])"
done
