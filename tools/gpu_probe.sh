mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCOVGPU_PROBE tools/panel_probe.hip -o /tmp/panel_probe 2>/dev/null && timeout 120 /tmp/panel_probe > gpurun_out/panel_probe.txt 2>&1
grep -E "k_trsm_sub|max\|L" gpurun_out/panel_probe.txt
COVGPU_TRSM_WAVES=1 timeout 120 /tmp/panel_probe 2>&1 | grep -E "k_trsm_sub" | sed 's/^/one wave: /'
for v in 4 1; do
COVGPU_TRSM_WAVES=$v timeout 300 python bench.py --steps 15 --warmup 2 --no-e2e --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/b.err
python -c "
import json; d=json.loads(open('gpurun_out/b.json').read().strip().splitlines()[-1]); print('bench waves=$v', round(d['value'],2), d['phase_ms_per_iteration'], round(d['roofline']['achieved'],1), d['roofline'].get('launches'), round(d['roofline'].get('avg_launch_ms',0),3), d['final_cost'])"
done
