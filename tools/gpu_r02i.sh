mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCOVGPU_PROBE tools/panel_probe.hip -o /tmp/panel_probe 2>/dev/null && timeout 120 /tmp/panel_probe > gpurun_out/r02i_panel_probe.txt 2>&1
cat gpurun_out/r02i_panel_probe.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full.py -m gpu -q -x --timeout 600 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 > gpurun_out/r02i_tests.txt
cat gpurun_out/r02i_tests.txt
timeout 300 python bench.py --steps 3 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/r02i_bench.json 2> gpurun_out/r02i_bench.err
COVGPU_PANEL=0 timeout 300 python bench.py --steps 3 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/r02i_bench_old.json 2> /dev/null
for f in gpurun_out/r02i_bench.json gpurun_out/r02i_bench_old.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', round(d['value'],2), d['phase_ms_per_iteration'], round(d['roofline']['achieved'],1), d['roofline'].get('launches'), round(d['roofline'].get('avg_launch_ms',0),3), d['final_cost'])"; done
