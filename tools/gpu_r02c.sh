mkdir -p gpurun_out
python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_shard.py -m gpu -q --timeout 900 -s 2>&1 | grep -v "^+++\|^-->" | tail -40 > gpurun_out/r02c_tests.txt
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/r02c_bench.json 2> gpurun_out/r02c_bench.err
python bench.py --workload a12 --steps 1 --warmup 0 --no-e2e --iterations 6 > gpurun_out/r02c_bench_a12.json 2> gpurun_out/r02c_bench_a12.err
tail -25 gpurun_out/r02c_tests.txt; head -c 700 gpurun_out/r02c_bench.json; echo; tail -3 gpurun_out/r02c_bench_a12.err; head -c 2500 gpurun_out/r02c_bench_a12.json
