#!/bin/bash
# development aid: the bench with several builds of the library on the SAME box, alternating (boxes differ by a microsecond):
#   bash tools/dev/ab.sh <rounds> name1 name2 ...     (libraries tools/dev/bin/lib_<name>.so; "tree" = the in-tree build;
#   name+NWAYHIP_X=v: with that development switch of the library)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
R=$1; shift
mkdir -p gpurun_out/ab
for r in $(seq 1 $R); do
	for n in "$@"; do
		base=${n%%+*}; extra=""; [ "$base" != "$n" ] && extra="NWAYHIP_DEV=1 ${n#*+}"   # name+VAR=value: a development switch of the library
		lib=$ROOT/tools/dev/bin/lib_$base.so; [ "$base" = tree ] && lib=$ROOT/nway_amd/csrc/libnwayhip.so
		env $extra NWAYHIP_LIBRARY=$lib timeout 150 python bench.py --steps 100 --warmup 5 --cpu-sample 0 --two-pipelines 0 --profile-stages 2> /dev/null | tail -1 > gpurun_out/ab/$n.$r.json
	done
done
python - "$@" <<'PY'
import json, sys, glob, statistics
for n in sys.argv[1:]:
	runs = [json.load(open(f)) for f in sorted(glob.glob('gpurun_out/ab/%s.*.json' % n))]
	ms = [r['ms_per_step'] * 1e3 for r in runs]
	st = {}
	for r in runs:
		for k, v in r.get('stages_ms', {}).items():
			st.setdefault(k, []).append(v * 1e3)
	print('%-14s step us: median %.2f  (%s)   stages (median us): %s' % (n, statistics.median(ms), ' '.join('%.1f' % x for x in ms),
		' '.join('%s %.1f' % (k, statistics.median(v)) for k, v in st.items() if statistics.median(v) > 0)))
PY
