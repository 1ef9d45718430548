# round 3, first GPU pass of the multifrontal solve: step parity on small maps, the GPU suite, first bench lines
mkdir -p gpurun_out
timeout 300 python tools/nd_check.py tiny,small > gpurun_out/r03a_nd_check.txt 2>&1
tail -12 gpurun_out/r03a_nd_check.txt
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -15 > gpurun_out/r03a_gpu_tests_tail.txt
tail -15 gpurun_out/r03a_gpu_tests_tail.txt
for w in mh01 mh12345; do timeout 300 python bench.py --workload $w --steps 3 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/r03a_bench_$w.json 2> gpurun_out/r03a_bench_$w.err; tail -c 1500 gpurun_out/r03a_bench_$w.json; tail -3 gpurun_out/r03a_bench_$w.err; done
