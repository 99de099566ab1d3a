# kernel trace of the bench (development aid): bash tools/dev/trace.sh <tag>
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python bench.py --steps 20 --warmup 3 --cpu-sample 0 > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | python tools/bench_summary.py
python - <<PY
import csv, glob
f = glob.glob('$OUT/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    print('%-40s %5s %9.2f us' % (r['Name'].replace('(anonymous namespace)::','').split('(')[0][:40], r['Calls'], float(r['AverageNs'])/1e3))
PY
