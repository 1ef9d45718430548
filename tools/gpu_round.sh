# one GPU-box visit of the round: tests, bench lines, kernel trace. Usage: bash tools/gpu_round.sh <tag>
tag=${1:-r02x}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -60 > gpurun_out/${tag}_tests.txt
python bench.py --steps 3 --warmup 1 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python bench.py --steps 3 --warmup 1 --force-shard > gpurun_out/${tag}_bench_shard1.json 2> gpurun_out/${tag}_bench_shard1.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof -o ${tag} -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof.log 2>&1
cd $GRAFT_REPO_ROOT; tail -12 gpurun_out/${tag}_tests.txt; head -c 1500 gpurun_out/${tag}_bench.json; echo; tail -3 gpurun_out/${tag}_bench_shard1.err; head -c 600 gpurun_out/${tag}_bench_shard1.json
