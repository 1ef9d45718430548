# round-4 measurement pass (one MI355X): usage  bash tools/gpu_r04.sh <tag> [tests|bench|prof|all]
tag=${1:-r04a}; what=${2:-all}
mkdir -p gpurun_out
if [[ $what == tests || $what == all ]]; then
  python -m pytest tests -m gpu -q -s --timeout 1500 2>&1 | grep -E "passed|failed|error|Error|world|a12|difference|pgo |one iteration|scaled step" | tail -40 > gpurun_out/${tag}_gpu_tests_tail.txt
  tail -5 gpurun_out/${tag}_gpu_tests_tail.txt
fi
if [[ $what == bench || $what == all ]]; then
  python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
  COVGPU_TRACE_PANELS=1 python bench.py --steps 1 --warmup 1 --no-e2e --no-cpu-baseline --sustain-s 0 2>&1 >/dev/null | grep "covgpu marks" | tail -2 > gpurun_out/${tag}_marks_unprofiled.txt
fi
if [[ $what == prof || $what == all ]]; then
  root=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ks
  rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --sustain-s 0 > $root/gpurun_out/${tag}_ks.log 2>&1
  cd $root; python tools/rocpd_stats.py $(ls /tmp/ks/*.db | head -1) gpurun_out/${tag}_kernel_stats.csv > /dev/null 2>&1
  python tools/rocpd_iter_timeline.py $(ls /tmp/ks/*.db | head -1) 14 > gpurun_out/${tag}_iteration_timeline.csv 2>/dev/null
fi
for f in gpurun_out/${tag}_bench*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', round(d['value'],2), d['phase_ms_per_iteration'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline'].get('chain_ms_per_iteration'), d['roofline_iteration']['frac'], d.get('sustained',{}).get('iterations_per_s'), d.get('cpu_baseline',{}).get('value'), d.get('max_pose_diff_gpu_cpu_m'), d.get('e2e_call',{}).get('t_call_s'))" 2>&1 | tail -1; done
