#!/bin/bash
# Kernel traces of the configurations beside the bench (C3-D, C4-S, C4-D, the stand-ins of BASELINE configs[0] / [1]):
# run ON THE GPU BOX (through gpurun), the summary goes to gpurun_out/rocprof_configs_<tag>.txt (copy to profiles/).
#   gpurun --timeout 1500 -- 'bash tools/profile_configs.sh r04'
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/configs_$TAG
mkdir -p $OUT
cd $ROOT
SUMMARY=$ROOT/gpurun_out/rocprof_configs_$TAG.txt
: > $SUMMARY
for cfg in c3d c4s c4d c1x c2x; do
	timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$cfg -- python tools/status_probe.py $cfg > $OUT/$cfg.log 2>&1
	python - $OUT/$cfg $cfg >> $SUMMARY <<'PY'
import csv, glob, sys
out, cfg = sys.argv[1], sys.argv[2]
print('# rocprofv3 --kernel-trace --stats -- python tools/status_probe.py %s' % cfg)
for line in open(out + '.log'):
	if line.startswith(('plan:', 'path:', 'status', 'wall')):
		print('#   ' + line.rstrip()[:220])
fs = sorted(glob.glob(out + '/*/*kernel_stats.csv'))
if not fs:
	print('no kernel_stats.csv')
else:
	print('%-40s %6s %12s %10s %10s' % ('kernel', 'calls', 'total_us', 'avg_us', 'max_us'))
	for r in csv.DictReader(open(fs[-1])):
		nm = r['Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
		if nm.startswith(('k_', 'k_sweep')):
			print('%-40s %6s %12.1f %10.2f %10.2f' % (nm[:40], r['Calls'], float(r['TotalDurationNs']) / 1e3, float(r['AverageNs']) / 1e3, float(r['MaxNs']) / 1e3))
print()
PY
done
cat $SUMMARY
