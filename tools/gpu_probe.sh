mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCOVGPU_PROBE tools/panel_probe.hip -o /tmp/panel_probe 2>/dev/null && timeout 120 /tmp/panel_probe > gpurun_out/panel_probe.txt 2>&1
cat gpurun_out/panel_probe.txt
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCOVGPU_PROBE -DPANEL_NO_DEFER tools/panel_probe.hip -o /tmp/panel_probe2 2>/dev/null && timeout 120 /tmp/panel_probe2 2>&1 | grep "k_potrf_panel" | sed 's/^/NO_DEFER: /'
