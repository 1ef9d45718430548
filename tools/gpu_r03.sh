# round-3 measurement pass (one MI355X): GPU tests, bench lines of every workload, kernel stats, iteration timeline, PMC passes
tag=${1:-r03b}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s --timeout 900 2>&1 | grep -E "passed|failed|error|world|a12x1000|difference|pgo " | tail -30 > gpurun_out/${tag}_gpu_tests_tail.txt
for w in mh01 mh123; do python bench.py --workload $w --steps 3 --warmup 1 --no-e2e > gpurun_out/${tag}_bench_$w.json 2> gpurun_out/${tag}_bench_$w.err; done
python bench.py --strategy lm --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/${tag}_bench_lm.json 2> /dev/null
python bench.py --workload a12x1000 --steps 2 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/${tag}_bench_a12x1000.json 2> /dev/null
timeout 600 python bench.py --workload a12 --steps 1 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/${tag}_bench_a12_20k_kf.json 2> gpurun_out/${tag}_bench_a12.err
COVGPU_GBA_DENSE=1 python bench.py --workload mh01 --steps 2 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/${tag}_bench_mh01_one_front.json 2> /dev/null
root=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $root/gpurun_out/${tag}_ks.log 2>&1
cd $root; python tools/rocpd_stats.py $(ls /tmp/ks/*.db | head -1) gpurun_out/${tag}_kernel_stats.csv > /dev/null 2>&1
python tools/rocpd_iter_timeline.py $(ls /tmp/ks/*.db | head -1) 14 > gpurun_out/${tag}_iteration_timeline.csv 2>/dev/null
bash tools/pmc_pass.sh ${tag} > gpurun_out/${tag}_pmc.log 2>&1
mkdir -p profiles; cp gpurun_out/pmc_traffic_current.json profiles/pmc_traffic_current.json
COVGPU_TRACE_PANELS=1 python bench.py --steps 1 --warmup 1 --no-e2e --no-cpu-baseline 2>&1 >/dev/null | grep "covgpu marks" | tail -2 > gpurun_out/${tag}_marks_unprofiled.txt
python bench.py --steps 10 --warmup 2 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 > gpurun_out/${tag}_smoke.txt; cat gpurun_out/${tag}_smoke.txt
for i in 1 2; do python bench.py --force-shard --steps 8 --warmup 2 --no-e2e --no-cpu-baseline 2>/dev/null > gpurun_out/${tag}_bench_forced_shard_rccl_1rank_$i.json; python bench.py --steps 8 --warmup 2 --no-e2e --no-cpu-baseline 2>/dev/null > gpurun_out/${tag}_bench_plain_beside_$i.json; done
python tools/cpp_flatten_time.py mh12345 7 2>/dev/null | tail -1 > gpurun_out/${tag}_cpp_flatten.txt; cat gpurun_out/${tag}_cpp_flatten.txt
tail -4 gpurun_out/${tag}_gpu_tests_tail.txt; tail -2 gpurun_out/${tag}_pmc.log
for f in gpurun_out/${tag}_bench*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', round(d['value'],2), d['phase_ms_per_iteration'], round(d['roofline']['achieved'],1), d['roofline']['traffic'], d['config']['layout']['device_mib'], d['config']['layout']['nd_fronts'], d['ate_rmse_m']['final'], d.get('cpu_baseline',{}).get('value'), d.get('max_pose_diff_gpu_cpu_m'), d.get('e2e_call',{}).get('t_call_s'))" 2>&1 | tail -1; done
