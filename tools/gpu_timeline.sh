mkdir -p gpurun_out
root=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ks
rocprofv3 --kernel-trace -d /tmp/ks -o ks -- python $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-e2e > $root/gpurun_out/tl.log 2>&1
cd $root; python tools/rocpd_iter_timeline.py $(ls /tmp/ks/*.db | head -1) 14 > gpurun_out/iter_timeline.csv; wc -l gpurun_out/iter_timeline.csv
python tools/rocpd_stats.py $(ls /tmp/ks/*.db | head -1) gpurun_out/tl_kernel_stats.csv > /dev/null 2>&1; head -12 gpurun_out/tl_kernel_stats.csv
