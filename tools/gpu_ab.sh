# A/B of environment switches on one MI355X: bash tools/gpu_ab.sh <tag> "<label>:<ENV=v,ENV=v>" ...   (label alone = default environment)
tag=$1; shift
mkdir -p gpurun_out
for spec in "$@"; do
  label=${spec%%:*}; envs=""; [[ $spec == *:* ]] && envs=$(echo ${spec#*:} | tr ',' ' ')
  env $envs python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-e2e --sustain-s 0 ${BENCH_ARGS} > gpurun_out/${tag}_ab_${label}.json 2> gpurun_out/${tag}_ab_${label}.err
  python -c "
import json,sys
try:
    d=json.loads(open('gpurun_out/${tag}_ab_${label}.json').read().strip().splitlines()[-1])
    print('${label}', round(d['value'],2), {k: round(v,4) for k,v in d['phase_ms_per_iteration'].items()}, 'chain', round(d['roofline_potrf']['chain_ms_per_iteration'],4), 'potrf_us', round(d['roofline_potrf']['avg_launch_ms']*1e3,1), 'final', d['final_cost'], 'ate', round(d['ate_rmse_m']['final'],6))
except Exception as e:
    print('${label}', 'FAILED', e); print(open('gpurun_out/${tag}_ab_${label}.err').read()[-1500:])
"
done
