mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -15 > gpurun_out/r02f_tests.txt
for w in mh01 mh123; do python bench.py --workload $w --steps 3 --warmup 1 --no-e2e > gpurun_out/r02f_bench_$w.json 2> gpurun_out/r02f_bench_$w.err; done
python bench.py --strategy lm --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/r02f_bench_lm.json 2> /dev/null
python bench.py --workload a12x1000 --steps 1 --warmup 0 --no-e2e > gpurun_out/r02f_bench_a12x1000.json 2> /dev/null
bash tools/pmc_pass.sh r02f > gpurun_out/r02f_pmc.log 2>&1
mkdir -p profiles; cp gpurun_out/pmc_traffic_current.json profiles/pmc_traffic_current.json
python bench.py --steps 3 --warmup 1 > gpurun_out/r02f_bench.json 2> gpurun_out/r02f_bench.err
tail -5 gpurun_out/r02f_tests.txt; tail -3 gpurun_out/r02f_pmc.log; for f in gpurun_out/r02f_bench*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', round(d['value'],2), d['phase_ms_per_iteration'], round(d['roofline']['achieved'],1), d['roofline']['traffic'], d['config']['layout']['device_mib'], d['ate_rmse_m']['final'], d.get('cpu_baseline',{}).get('value'), d.get('max_pose_diff_gpu_cpu_m'))"; done
