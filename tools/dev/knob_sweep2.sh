#!/bin/bash
# development aid: tools/status_probe.py of the standing configurations under the plan's development knobs (one value at a time)
#   bash tools/dev/knob_sweep2.sh "c1x c2x" NWAYHIP_TAILD_PER_BLOCK "0 4 8 16 32"
export NWAYHIP_DEV=1
CFGS=$1; KNOB=$2; VALS=$3
for cfg in $CFGS; do
	for v in $VALS; do
		echo "$cfg $KNOB=$v: $(env $KNOB=$v timeout 200 python tools/status_probe.py $cfg 2>&1 | grep -E '^wall' | cut -c1-60) $(env $KNOB=$v timeout 200 python tools/status_probe.py $cfg 2>&1 | grep -E 'stages' | cut -c1-160)"
	done
done
