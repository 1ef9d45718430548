# quick GPU check between two source changes: sharding + full-size parity tests, headline bench lines, level marks
tag=${1:-q}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_shard.py tests/test_gpu_full.py tests/test_gpu_parity.py -m gpu -q -x --timeout 600 2>&1 | grep -E "passed|failed|rror|assert" | tail -8 > gpurun_out/${tag}_tests.txt
for w in mh12345 mh01 mh123; do timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --no-e2e --no-cpu-baseline > gpurun_out/${tag}_bench_$w.json 2> gpurun_out/${tag}_bench_$w.err; done
COVGPU_TRACE_PANELS=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-e2e --no-cpu-baseline 2>&1 >/dev/null | grep "covgpu marks" | tail -2 > gpurun_out/${tag}_marks.txt
cat gpurun_out/${tag}_tests.txt
for f in gpurun_out/${tag}_bench_*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', round(d['value'],2), d['phase_ms_per_iteration'], d['config']['layout']['nd_fronts'], d['ate_rmse_m']['final'])" 2>&1 | tail -1; done
cat gpurun_out/${tag}_marks.txt
timeout 300 python bench.py --strategy lm --steps 5 --warmup 2 --no-e2e --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lm', round(d['value'],2), d['phase_ms_per_iteration'])"
