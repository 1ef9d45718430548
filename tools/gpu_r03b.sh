# round 3, second GPU pass: whole GPU suite, iteration timelines of the multifrontal solve (5-agent map, single agent)
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -25 > gpurun_out/r03b_gpu_tests_tail.txt
tail -12 gpurun_out/r03b_gpu_tests_tail.txt
root=$(pwd); cd /tmp && export TMPDIR=/tmp
for w in mh12345 mh01; do
  rm -rf /tmp/ks_$w
  rocprofv3 --kernel-trace -d /tmp/ks_$w -o ks -- python $root/bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --no-e2e > $root/gpurun_out/r03b_tl_$w.log 2>&1
  python $root/tools/rocpd_iter_timeline.py $(ls /tmp/ks_$w/*.db | head -1) 14 > $root/gpurun_out/r03b_iter_timeline_$w.csv
  python $root/tools/rocpd_stats.py $(ls /tmp/ks_$w/*.db | head -1) $root/gpurun_out/r03b_kernel_stats_$w.csv > /dev/null 2>&1
  head -14 $root/gpurun_out/r03b_kernel_stats_$w.csv
done
