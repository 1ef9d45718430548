mkdir -p gpurun_out
python -c "
import torch; torch.cuda.init()
import ctypes; h=ctypes.CDLL('libamdhip64.so'); lo=ctypes.c_int(); hi=ctypes.c_int(); h.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)); print('priority range lo', lo.value, 'hi', hi.value)"
for v in n h; do
COVGPU_MID_PRIO=$v timeout 300 python bench.py --steps 8 --warmup 2 --no-e2e --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/b.err
python -c "
import json; d=json.loads(open('gpurun_out/b.json').read().strip().splitlines()[-1]); print('bench mid=$v', round(d['value'],2), d['phase_ms_per_iteration'], round(d['roofline']['achieved'],1), d['roofline'].get('launches'), round(d['roofline'].get('avg_launch_ms',0),3), d['final_cost'])"
done
