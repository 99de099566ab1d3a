#!/bin/bash
# development aid: tools/status_probe.py of the dense / 3-way configurations with several builds of the library on the SAME box
#   bash tools/dev/ab_configs.sh "c3d c4s c4d c1x c2x" tree ocml
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
CONFIGS=$1; shift
mkdir -p gpurun_out/abc
for c in $CONFIGS; do
	for n in "$@"; do
		lib=$ROOT/tools/dev/bin/lib_$n.so; [ "$n" = tree ] && lib=$ROOT/nway_amd/csrc/libnwayhip.so
		NWAYHIP_LIBRARY=$lib timeout 200 python tools/status_probe.py $c 2> /dev/null | grep "wall\|stages" | sed "s/^/$c $n: /" | tee -a gpurun_out/abc/summary.txt
	done
done
