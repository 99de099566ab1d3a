#!/bin/bash
# development aid: tools/status_probe.py of several configurations with several builds of the library on ONE box, alternating
#   bash tools/dev/ab_probe2.sh <rounds> "<configs>" name1 name2 ...   (libraries tools/dev/bin/lib_<name>.so; "tree" = the in-tree build)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
R=$1; CFGS=$2; shift 2
for cfg in $CFGS; do
	for r in $(seq 1 $R); do
		for n in "$@"; do
			lib=$ROOT/tools/dev/bin/lib_$n.so; [ "$n" = tree ] && lib=$ROOT/nway_amd/csrc/libnwayhip.so
			echo "$cfg $n: $(NWAYHIP_LIBRARY=$lib timeout 200 python tools/status_probe.py $cfg 2>&1 | grep -E '^wall' | cut -c1-250)"
		done
	done
done
