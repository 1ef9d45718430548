mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_golden.py tests/test_gpu_full.py -m gpu -q --timeout 900 2>&1 | tail -30 > gpurun_out/r02d_tests.txt
COVGPU_POTRF=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/r02d_bench_potrf1.json 2> gpurun_out/r02d_bench1.err
COVGPU_POTRF=2 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/r02d_bench_potrf2.json 2> gpurun_out/r02d_bench2.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02d_prof -o r02d -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $GRAFT_REPO_ROOT/gpurun_out/r02d_prof.log 2>&1
cd $GRAFT_REPO_ROOT
tail -8 gpurun_out/r02d_tests.txt; for f in gpurun_out/r02d_bench_potrf1.json gpurun_out/r02d_bench_potrf2.json; do python -c "
import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['phase_ms_per_iteration'], d['final_cost'])"; done
