# round-2 final measurement pass (one MI355X): GPU tests, bench lines of every workload, kernel stats, PMC passes, probes
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s --timeout 900 2>&1 | grep -E "passed|failed|error|world|a12x1000|difference" | tail -14 > gpurun_out/r02k_gpu_tests_tail.txt
for w in mh01 mh123; do python bench.py --workload $w --steps 3 --warmup 1 --no-e2e > gpurun_out/r02k_bench_$w.json 2> gpurun_out/r02k_bench_$w.err; done
python bench.py --strategy lm --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/r02k_bench_lm.json 2> /dev/null
python bench.py --workload a12x1000 --steps 1 --warmup 0 --no-e2e --no-cpu-baseline > gpurun_out/r02k_bench_a12x1000.json 2> /dev/null
timeout 420 python bench.py --workload a12 --steps 1 --warmup 0 --no-e2e --no-cpu-baseline > gpurun_out/r02k_bench_a12_20k_kf.json 2> /dev/null
python bench.py --force-shard --steps 3 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/r02k_bench_forced_shard_1rank.json 2> /dev/null
COVGPU_PANEL=0 COVGPU_SB_BACK=0 COVGPU_POSE_RHS_Y=1 python bench.py --steps 3 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/r02k_bench_round2a_chain.json 2> /dev/null
root=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $root/gpurun_out/r02k_ks.log 2>&1
cd $root; python tools/rocpd_stats.py $(ls /tmp/ks/*.db | head -1) gpurun_out/r02k_kernel_stats.csv > /dev/null 2>&1
python tools/rocpd_iter_timeline.py $(ls /tmp/ks/*.db | head -1) 14 > gpurun_out/r02k_iteration_timeline.csv 2>/dev/null
bash tools/pmc_pass.sh r02k > gpurun_out/r02k_pmc.log 2>&1
mkdir -p profiles; cp gpurun_out/pmc_traffic_current.json profiles/pmc_traffic_current.json
python bench.py --steps 10 --warmup 2 > gpurun_out/r02k_bench.json 2> gpurun_out/r02k_bench.err
for t in panel_probe chain_probe; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCOVGPU_PROBE tools/$t.hip -o /tmp/$t 2>/dev/null && timeout 120 /tmp/$t > gpurun_out/r02k_$t.txt 2>&1; done
for t in diag_probe lat_probe; do hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/$t.hip -o /tmp/$t 2>/dev/null && timeout 60 /tmp/$t > gpurun_out/r02k_$t.txt 2>&1; done
tail -4 gpurun_out/r02k_gpu_tests_tail.txt; tail -2 gpurun_out/r02k_pmc.log
for f in gpurun_out/r02k_bench*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', round(d['value'],2), d['phase_ms_per_iteration'], round(d['roofline']['achieved'],1), d['roofline']['traffic'], d['config']['layout']['device_mib'], d['ate_rmse_m']['final'], d.get('cpu_baseline',{}).get('value'), d.get('max_pose_diff_gpu_cpu_m'), d.get('e2e_call',{}).get('t_call_s'))"; done
