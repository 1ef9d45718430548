"""The two probes the round-4 work on the serial panel chain leans on (tools/panel_probe.hip: k_potrf_panel / k_trsm_sub4 / k_bwd_step_sub
against a host Cholesky with their per-step timelines; tools/valu_probe.hip: issue rates and latencies of the chain's instructions) include
the product's kernel source: they must keep compiling for gfx950 (hipcc cross-compiles without a GPU). Round 5: tools/event_probe.hip (what an
event packet between two dependent kernels costs: DESIGN.md 4.6)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("src,defs", [("tools/panel_probe.hip", ["-DCOVGPU_PROBE"]), ("tools/valu_probe.hip", []), ("tools/event_probe.hip", [])])
def test_probe_compiles_for_gfx950(tmp_path, src, defs):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = str(tmp_path / "probe.o")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", *defs, os.path.join(ROOT, src), "-o", out],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert os.path.getsize(out) > 0
