mkdir -p gpurun_out
root=$(pwd); cd /tmp && export TMPDIR=/tmp
for w in mh12345; do
  rm -rf /tmp/ks_$w
  rocprofv3 --kernel-trace -d /tmp/ks_$w -o ks -- python $root/bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --no-e2e > $root/gpurun_out/r03i_tl_$w.log 2>&1
  python $root/tools/rocpd_iter_timeline.py $(ls /tmp/ks_$w/*.db | head -1) 14 > $root/gpurun_out/r03i_iter_timeline_$w.csv
  python $root/tools/rocpd_stats.py $(ls /tmp/ks_$w/*.db | head -1) $root/gpurun_out/r03i_kernel_stats_$w.csv > /dev/null 2>&1
  head -12 $root/gpurun_out/r03i_kernel_stats_$w.csv
done
