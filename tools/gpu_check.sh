# panel probe + tests + a bench line (dev visit)
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCOVGPU_PROBE tools/panel_probe.hip -o /tmp/panel_probe 2>/dev/null && timeout 120 /tmp/panel_probe 2>&1 | grep -E "k_potrf_panel|max\|L"
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8
timeout 300 python bench.py --steps 15 --warmup 2 --no-e2e --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/b.err
python -c "
import json; d=json.loads(open('gpurun_out/b.json').read().strip().splitlines()[-1]); print('bench', round(d['value'],2), d['phase_ms_per_iteration'], round(d['roofline']['achieved'],1), d['roofline'].get('launches'), round(d['roofline'].get('avg_launch_ms',0),3), d['final_cost'])"
