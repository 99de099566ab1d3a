#!/bin/bash
# development aid: the sweep launch that also registers the primaries (front.inc: FUSED) against the two launches, on ONE box:
# the bench with the CPU check, the A/B of the stage times, the phase stamps.   gpurun -- 'bash tools/dev/fused_probe.sh [rounds]'
export NWAYHIP_DEV=1
R=${1:-3}
mkdir -p gpurun_out/f1
NWAYHIP_FUSED_FRONT=1 timeout 120 python bench.py --steps 50 --warmup 5 --two-pipelines 0 > gpurun_out/f1/bench_fused.json 2> gpurun_out/f1/bench_fused.err
tail -c 400 gpurun_out/f1/bench_fused.err
python -c "
import json
d=json.load(open('gpurun_out/f1/bench_fused.json'))
print('fused bench: us per step', d['ms_per_step'] * 1e3, 'check', d['check']['ok'], 'path', d.get('path'))
"
timeout 600 bash tools/dev/ab.sh $R tree tree+NWAYHIP_FUSED_FRONT=1 2>&1 | tail -4
NWAYHIP_FUSED_FRONT=1 timeout 200 python tools/dev/phase_times.py 2>&1 | grep -v "^k_tail2\|^    .*landed  \|amdgpu.ids" | head -60
