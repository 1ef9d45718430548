mkdir -p gpurun_out
timeout 600 python bench.py --workload mh12345 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03l_bench_e2e.json 2> gpurun_out/r03l_bench_e2e.err
python -c "
import json; d=json.loads(open('gpurun_out/r03l_bench_e2e.json').read().strip().splitlines()[-1]); print(round(d['value'],2), d['e2e_call'], d['pgo_call'], d['upload_s_not_in_value'])"
timeout 600 python bench.py --workload a12x1000 --steps 1 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/r03l_bench_a12x1000.json 2> gpurun_out/r03l_bench_a12x1000.err
python -c "
import json; d=json.loads(open('gpurun_out/r03l_bench_a12x1000.json').read().strip().splitlines()[-1]); print('a12x1000', round(d['value'],2), d['phase_ms_per_iteration'], d['config']['layout'], d['ate_rmse_m'], d['roofline']['achieved'])"
tail -2 gpurun_out/r03l_bench_a12x1000.err
