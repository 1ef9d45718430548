mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -8 > gpurun_out/r03d_gpu_tests_tail.txt
cat gpurun_out/r03d_gpu_tests_tail.txt
for w in mh01 mh12345; do
  COVGPU_ND_LEAF=600 COVGPU_TRACE_PANELS=1 timeout 300 python bench.py --workload $w --steps 1 --warmup 1 --no-e2e --no-cpu-baseline > /dev/null 2> gpurun_out/r03d_marks_$w.txt
  grep "covgpu marks" gpurun_out/r03d_marks_$w.txt | tail -3
done
