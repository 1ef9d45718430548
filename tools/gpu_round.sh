mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -40 > gpurun_out/r02a_tests.txt
python bench.py --steps 3 --warmup 1 > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err
COVGPU_GBA_DENSE=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/r02a_bench_dense.json 2> gpurun_out/r02a_bench_dense.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02a_prof -o r02a -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $GRAFT_REPO_ROOT/gpurun_out/r02a_prof.log 2>&1
cd $GRAFT_REPO_ROOT; ls gpurun_out/r02a_prof | head; tail -5 gpurun_out/r02a_tests.txt; cat gpurun_out/r02a_bench.json | head -c 3000
