mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -8 > gpurun_out/r02h_tests.txt
python bench.py --steps 3 --warmup 1 --no-e2e > gpurun_out/r02h_bench.json 2> gpurun_out/r02h_bench.err
python bench.py --workload a12x1000 --steps 1 --warmup 0 --no-e2e --no-cpu-baseline > gpurun_out/r02h_bench_a12x1000.json 2> /dev/null
root=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $root/gpurun_out/r02h_ks.log 2>&1
cd $root; python tools/rocpd_stats.py $(ls /tmp/ks/*.db | head -1) gpurun_out/r02h_kernel_stats.csv > /dev/null 2>&1 || cp /tmp/ks/*kernel_stats.csv gpurun_out/r02h_kernel_stats.csv
tail -4 gpurun_out/r02h_tests.txt; for f in gpurun_out/r02h_bench.json gpurun_out/r02h_bench_a12x1000.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', round(d['value'],2), d['phase_ms_per_iteration'], round(d['roofline']['achieved'],1), d['roofline'].get('launches'), round(d['roofline'].get('avg_launch_ms',0),3), d['final_cost'])"; done
head -25 gpurun_out/r02h_kernel_stats.csv
