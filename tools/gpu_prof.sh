# kernel timeline of one trust-region iteration (profiled: dispatch gaps are inflated, durations are not)
tag=${1:-p}
root=$(pwd); mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $root/gpurun_out/${tag}_ks.log 2>&1
cd $root; python tools/rocpd_stats.py $(ls /tmp/ks/*.db | head -1) gpurun_out/${tag}_kernel_stats.csv > /dev/null 2>&1
python tools/rocpd_iter_timeline.py $(ls /tmp/ks/*.db | head -1) 14 > gpurun_out/${tag}_iteration_timeline.csv 2>/dev/null
head -30 gpurun_out/${tag}_kernel_stats.csv
