mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -8 > gpurun_out/r02g_tests.txt
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/r02g_bench.json 2> gpurun_out/r02g_bench.err
python bench.py --workload a12x1000 --steps 1 --warmup 0 --no-e2e --no-cpu-baseline > gpurun_out/r02g_bench_a12x1000.json 2> /dev/null
tail -4 gpurun_out/r02g_tests.txt; for f in gpurun_out/r02g_bench.json gpurun_out/r02g_bench_a12x1000.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', round(d['value'],2), d['phase_ms_per_iteration'], round(d['roofline']['achieved'],1), d['roofline']['launches'], round(d['roofline']['avg_launch_ms'],3), d['final_cost'])"; done
