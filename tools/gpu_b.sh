mkdir -p gpurun_out
for m in 0 1 2; do
COVGPU_CU_MASK=$m timeout 300 python bench.py --steps 3 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/b_$m.json 2> gpurun_out/b_$m.err
python -c "
import json; d=json.loads(open('gpurun_out/b_$m.json').read().strip().splitlines()[-1]); print('mask $m', round(d['value'],2), d['phase_ms_per_iteration'], round(d['roofline']['achieved'],1), d['roofline'].get('launches'), round(d['roofline'].get('avg_launch_ms',0),3), d['final_cost'])"
done
