# round 3, third GPU pass: GPU suite, leaf-size sweep of the multifrontal solve
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -8 > gpurun_out/r03c_gpu_tests_tail.txt
tail -5 gpurun_out/r03c_gpu_tests_tail.txt
for leaf in 450 600 900 1400; do for w in mh01 mh12345; do
  COVGPU_ND_LEAF=$leaf timeout 300 python bench.py --workload $w --steps 3 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/r03c_bench_${w}_leaf$leaf.json 2> /dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/r03c_bench_${w}_leaf$leaf.json').read().strip().splitlines()[-1]); print('$w leaf $leaf', round(d['value'],2), d['phase_ms_per_iteration'], d['config']['layout'].get('nd_serial_panels'), d['config']['layout'].get('nd_front_mib'))"
done; done
