# round-6 measurement pass (one MI355X): usage  bash tools/gpu_r06.sh <tag> [tests|bench|side|prof|pmc|all ...]
tag=${1:-r06a}; shift; what=" ${*:-all} "
has() { [[ $what == *" $1 "* || $what == *" all "* ]]; }
mkdir -p gpurun_out
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -q -s --timeout 1500 2>&1 | grep -E "passed|failed|error|Error|world|a12|difference|pgo |one iteration|scaled step|device second" | tail -60 > gpurun_out/${tag}_gpu_tests_tail.txt
  tail -3 gpurun_out/${tag}_gpu_tests_tail.txt
fi
if has bench; then
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
  COVGPU_TRACE_PANELS=1 python bench.py --steps 1 --warmup 1 --no-e2e --no-cpu-baseline --sustain-s 0 --a12-leg 0 2>&1 >/dev/null | grep "covgpu marks" | tail -2 > gpurun_out/${tag}_marks_unprofiled.txt
  COVGPU_TRACE_PANELS=2 python bench.py --steps 1 --warmup 1 --no-e2e --no-cpu-baseline --sustain-s 0 --a12-leg 0 2>&1 >/dev/null | grep "covgpu marks" | tail -1 > gpurun_out/${tag}_marks_kernels.txt
fi
if has marks; then   # un-profiled event marks only
  COVGPU_TRACE_PANELS=1 python bench.py --steps 1 --warmup 1 --no-e2e --no-cpu-baseline --sustain-s 0 --a12-leg 0 2>&1 >/dev/null | grep "covgpu marks" | tail -2 > gpurun_out/${tag}_marks_unprofiled.txt
  cat gpurun_out/${tag}_marks_unprofiled.txt
fi
if has gatelog; then   # un-profiled timeline of the streams' hand-overs (signal / gate kernels stamp wall_clock64)
  COVGPU_GATE_LOG=1 python bench.py --steps 1 --warmup 1 --no-e2e --no-cpu-baseline --sustain-s 0 --a12-leg 0 2>&1 >/dev/null | grep "covgpu gate log" | tail -1 > gpurun_out/${tag}_gate_log.txt
fi
if has soak; then      # four minutes of solves back to back: no gate may time out
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --a12-leg 0 --sustain-s 240 > gpurun_out/${tag}_soak.json 2> gpurun_out/${tag}_soak.err; grep -c "timed out" gpurun_out/${tag}_soak.err
fi
if has quick; then   # the metric's line only, short
  timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --sustain-s 0 --a12-leg 0 > gpurun_out/${tag}_bench_quick.json 2> gpurun_out/${tag}_bench_quick.err
fi
if has side; then
  for w in mh01 mh123; do python bench.py --workload $w --steps 5 --warmup 2 --no-e2e --sustain-s 0 > gpurun_out/${tag}_bench_$w.json 2> gpurun_out/${tag}_bench_$w.err; done
  python bench.py --strategy lm --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --sustain-s 0 --a12-leg 0 > gpurun_out/${tag}_bench_lm.json 2> /dev/null
  for i in 1; do python bench.py --force-shard --steps 8 --warmup 2 --no-e2e --no-cpu-baseline --sustain-s 0 2>/dev/null > gpurun_out/${tag}_bench_forced_shard_rccl_1rank_$i.json; done
  COVGPU_FLATTEN_TIMING=1 python tools/cpp_flatten_time.py mh12345 5 gba 2>&1 | grep -E "GBA call|C\+\+" | tail -12 > gpurun_out/${tag}_cpp_call.txt
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > gpurun_out/${tag}_smoke.txt; cat gpurun_out/${tag}_smoke.txt
  COVGPU_TRACE_PANELS=1 timeout 600 python bench.py --workload a12 --steps 1 --warmup 1 --no-e2e --no-cpu-baseline --sustain-s 0 2>&1 >/dev/null | grep "covgpu marks" | tail -1 > gpurun_out/${tag}_marks_a12.txt
fi
if has prof; then
  root=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ks
  rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks -- python $root/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --sustain-s 0 --a12-leg 0 > $root/gpurun_out/${tag}_ks.log 2>&1
  cd $root; python tools/rocpd_stats.py $(ls /tmp/ks/*.db | head -1) gpurun_out/${tag}_kernel_stats.csv > /dev/null 2>&1
  python tools/rocpd_iter_timeline.py $(ls /tmp/ks/*.db | head -1) 14 > gpurun_out/${tag}_iteration_timeline.csv 2>/dev/null
fi
if has pmc; then
  bash tools/pmc_pass.sh ${tag} > gpurun_out/${tag}_pmc.log 2>&1
  mkdir -p profiles; cp gpurun_out/pmc_traffic_current.json profiles/pmc_traffic_current.json
  tail -2 gpurun_out/${tag}_pmc.log
fi
for f in gpurun_out/${tag}_bench*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', round(d['value'],2), d['phase_ms_per_iteration'], round(d['roofline']['frac'],5), round(d['roofline_potrf']['avg_launch_ms']*1e3,1), round(d['roofline_iteration']['frac'],4), d['config']['layout']['device_mib'], d['ate_rmse_m']['final'], d.get('cpu_baseline',{}).get('value'), d.get('max_pose_diff_gpu_cpu_m'), d.get('e2e_call',{}).get('t_call_s'), d.get('e2e_call_cpp',{}).get('t_call_s'), 'a12', (d.get('a12_leg') or {}).get('value'))" 2>&1 | tail -1; done
