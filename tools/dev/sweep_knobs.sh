#!/bin/bash
# development aid: the bench's sweep time under the plan's environment knobs
for cfg in "20 1" "19 2" "19 1" "18 2" "17 2"; do
	set -- $cfg
	NWAYHIP_COARSE_LOG2=$1 NWAYHIP_SWEEP_BLOCKS_PER_CU=$2 timeout 200 python bench.py --steps 60 --warmup 6 --cpu-sample 0 2>/dev/null | tail -1 |
		python -c "import sys,json; d=json.loads(sys.stdin.read()); print('coarse 2^$1 blocks/CU $2: step %.1f us  sweep %.2f us  survivors %d' % (d['ms_per_step']*1e3, d['roofline']['launch_ms']*1e3, d['config']['survivors_per_step_rank0']))"
done
