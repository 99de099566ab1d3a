# FETCH_SIZE / WRITE_SIZE / L2 counters of the dense 2-way pass C3-D and of C5 whole (development aid; GPU box):
#   bash tools/dev/pmc_dense.sh r03   ->  gpurun_out/pmc_dense_r03.txt  (copy to profiles/)
# Three rocprofv3 passes per configuration (the TCC counters do not fit one pass; --pmc is never combined with
# other traces than the kernel trace), every pass of the SAME command; mean per dispatch, kernels of the run only.
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_dense_$TAG
mkdir -p $OUT
cd $ROOT
SUMMARY=$ROOT/gpurun_out/pmc_dense_$TAG.txt
: > $SUMMARY
for cfg in "c3d" "c3s 500000 100000000"; do
	name=$(echo $cfg | tr ' ' '_')
	for pass in "f FETCH_SIZE" "w WRITE_SIZE" "h TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
		set -- $pass; d=$1; shift
		timeout 400 rocprofv3 --pmc $@ --kernel-trace --output-format csv -d $OUT/$name/$d -- python tools/status_probe.py $cfg > $OUT/$name.$d.log 2>&1
	done
	python - $OUT/$name "$cfg" >> $SUMMARY <<'PY'
import csv, glob, collections, sys
out, cfg = sys.argv[1], sys.argv[2]
print('# rocprofv3 --pmc <counters> --kernel-trace -- python tools/status_probe.py %s   (one pass per counter group; mean per dispatch)' % cfg)
print('# FETCH_SIZE / WRITE_SIZE in KiB as reported (gfx950: a wide coalesced 16 B/lane stream is reported at HALF its bytes -- MI355X_MICROARCH.md)')
dur = {}
for d in ('f', 'w', 'h'):
	fs = glob.glob('%s/%s/*/*counter_collection.csv' % (out, d))
	if not fs:
		print(d, 'no counter file')
		continue
	acc = collections.defaultdict(list)
	for r in csv.DictReader(open(fs[0])):
		k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
		acc[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
	for (k, c), v in sorted(acc.items()):
		if k.startswith('k_'):
			print('%-34s %-14s %16.1f   (%d dispatches)' % (k[:34], c, sum(v) / len(v), len(v)))
	ks = glob.glob('%s/%s/*/*kernel_trace.csv' % (out, d))
	if ks and d == 'f':
		t = collections.defaultdict(list)
		for r in csv.DictReader(open(ks[0])):
			k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
			t[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
		for k, v in sorted(t.items()):
			if k.startswith('k_'):
				print('%-34s %-14s %16.2f   (us under the counter pass, mean of %d)' % (k[:34], 'duration', sum(v) / len(v), len(v)))
print()
PY
	grep -h "^path\|^plan\|^c3\|^wall\|^stages" $OUT/$name.f.log >> $SUMMARY
	echo >> $SUMMARY
done
cat $SUMMARY
