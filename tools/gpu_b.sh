mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCOVGPU_PROBE tools/panel_probe.hip -o /tmp/panel_probe 2>/dev/null && timeout 120 /tmp/panel_probe 2>&1 | grep -E "trsm_sub|max"
