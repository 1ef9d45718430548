#!/bin/bash
# HBM-traffic and matrix-pipe counter passes for the dominant kernel of bench.py's default workload (run ON the GPU box:
#   gpurun -- 'bash tools/pmc_pass.sh r02'), as /opt/skills/guides/MI355X_MICROARCH.md prescribes: one counter per pass,
# --kernel-trace only (no sys/hip/hsa trace domains), FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B).
# Writes gpurun_out/<tag>_pmc_*.csv and gpurun_out/pmc_traffic_current.json, stamped with the sha256 of the HIP sources:
# bench.py reports `roofline.traffic` only from a file whose stamp matches the sources it runs (copy it to profiles/).
tag=${1:-pmc}
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
# counter collection serialises kernels: a polling gate kernel (device-flag stream ordering, round 6) would wait for a producer the profiler has not
# let start. The counters are per kernel and do not depend on how the streams are ordered: HIP events here.
export COVGPU_GATES=0
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_$ctr -o pmc -- python $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --sustain-s 0 --a12-leg 0 > $root/gpurun_out/${tag}_pmc_$ctr.log 2>&1
done
rm -rf /tmp/pmc_mfma
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d /tmp/pmc_mfma -o pmc -- python $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --sustain-s 0 --a12-leg 0 > $root/gpurun_out/${tag}_pmc_mfma.log 2>&1
cd $root
python tools/rocpd_pmc.py FETCH_SIZE=$(ls /tmp/pmc_FETCH_SIZE/*.db | head -1) WRITE_SIZE=$(ls /tmp/pmc_WRITE_SIZE/*.db | head -1) gpurun_out/${tag}_pmc_hbm_traffic.csv > /dev/null
python tools/rocpd_counters.py $(ls /tmp/pmc_mfma/*.db | head -1) gpurun_out/${tag}_pmc_mfma.csv > /dev/null
python - <<PY
import csv, hashlib, json, os, subprocess
root = "$root"
h = hashlib.sha256()
d = os.path.join(root, "covins_amd", "csrc")
for f in sorted(os.listdir(d)):
    if f.endswith((".hip", ".hpp")):
        h.update(open(os.path.join(d, f), "rb").read())
out = {"workload": "mh12345", "kernels_sha256": h.hexdigest(),
       "git_sha": subprocess.run(["git", "rev-parse", "HEAD"], cwd=root, capture_output=True, text=True).stdout.strip() or "gpu-box snapshot (no .git)"}
# the trailing update runs in two tile forms (full 128x128 tiles; 64x64 quadrants for short tile lists): per-launch figures over both
def is_syrk(name): return "k_gemm_abt<0" in name or "k_gemm_abt_q<0" in name
calls = fetch = write = 0.0
names = []
for row in csv.DictReader(open(os.path.join(root, "gpurun_out", "${tag}_pmc_hbm_traffic.csv"))):
    if is_syrk(row["Name"]):
        c = int(row["Calls"]); calls += c; names.append(row["Name"].split("(")[0].replace("void covgpu::", ""))
        fetch += c * float(row["FETCH_x2_MB_per_call"]) * 1048576.0; write += c * float(row["WRITE_MB_per_call"]) * 1048576.0
if calls:
    out.update(kernel=" + ".join(sorted(set(names))), calls=int(calls), fetch_bytes_x2=fetch / calls, write_bytes=write / calls)
# the serial panel chain (bench.py: roofline = the kernel with the largest share of GPU time)
for row in csv.DictReader(open(os.path.join(root, "gpurun_out", "${tag}_pmc_hbm_traffic.csv"))):
    if "k_potrf_panel" in row["Name"]:
        out["potrf_bytes_per_launch"] = (float(row["FETCH_x2_MB_per_call"]) + float(row["WRITE_MB_per_call"])) * 1048576.0
        out["potrf_calls"] = int(row["Calls"])
# counter traffic of the whole linearise + landmark-Schur pass per trust-region iteration (bench.py: roofline_build.traffic)
BUILD = ("k_lm_lin", "k_kf_reduce", "k_pair_blocks", "k_imu_build", "k_imu_gather", "k_edge_build", "k_edge_gather_kf", "k_edge_gather_pair",
         "k_finalize_diag", "k_zero_many", "k_nd_zero", "k_cost_finish")
its = 0; bbytes = 0.0; parts = {}
for row in csv.DictReader(open(os.path.join(root, "gpurun_out", "${tag}_pmc_hbm_traffic.csv"))):
    nm = row["Name"].split("(")[0].replace("void covgpu::", "").replace("covgpu::", "").split("<")[0]
    if nm in BUILD:
        c = int(row["Calls"]); b = c * (float(row["FETCH_x2_MB_per_call"]) + float(row["WRITE_MB_per_call"])) * 1048576.0
        bbytes += b; parts[nm] = parts.get(nm, 0.0) + b
        if nm == "k_lm_lin": its = c
ext = 0.0
for row in csv.DictReader(open(os.path.join(root, "gpurun_out", "${tag}_pmc_hbm_traffic.csv"))):
    if "k_nd_extend" in row["Name"]:
        ext += int(row["Calls"]) * (float(row["FETCH_x2_MB_per_call"]) + float(row["WRITE_MB_per_call"])) * 1048576.0
if its:
    out["extend_add_bytes_per_iteration"] = ext / its   # (round 6: the extend-add stores the border x border tiles whole instead of k_nd_zero clearing them)
    out["build_bytes_per_iteration"] = bbytes / its
    out["build_bytes_by_kernel"] = {k: round(v / its / 1e6, 1) for k, v in sorted(parts.items(), key=lambda kv: -kv[1])}   # MB
busy = wsum = 0.0
for row in csv.DictReader(open(os.path.join(root, "gpurun_out", "${tag}_pmc_mfma.csv"))):
    if is_syrk(row["Name"]) and row.get("MfmaBusy_pct"):
        w = float(row.get("Calls") or 1); busy += w * float(row["MfmaBusy_pct"]) / 100.0; wsum += w
if wsum:
    out["mfma_busy_frac"] = busy / wsum   # (weighted by launches)
for row in csv.DictReader(open(os.path.join(root, "gpurun_out", "${tag}_pmc_mfma.csv"))):
    if "k_potrf_panel" in row["Name"] and row.get("MfmaBusy_pct"):
        out["potrf_mfma_busy_frac"] = float(row["MfmaBusy_pct"]) / 100.0
json.dump(out, open(os.path.join(root, "gpurun_out", "pmc_traffic_current.json"), "w"), indent=1)
print(json.dumps(out))
PY
