#!/bin/bash
# development aid: tools/status_probe.py <config> with several builds of the library on the same box, alternating
#   bash tools/dev/ab_probe.sh <config> <rounds> name1 name2 ...
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
CFG=$1; R=$2; shift 2
for r in $(seq 1 $R); do
	for n in "$@"; do
		lib=$ROOT/tools/dev/bin/lib_$n.so; [ "$n" = tree ] && lib=$ROOT/nway_amd/csrc/libnwayhip.so
		echo "$n: $(NWAYHIP_LIBRARY=$lib python tools/status_probe.py $CFG 2>/dev/null | grep 'wall\|stages' | tr '\n' ' ' | cut -c1-230)"
	done
done
