# kernel trace of one status_probe configuration: bash tools/dev/trace_probe.sh c3d
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python tools/status_probe.py "$@" > $OUT/probe.log 2>&1
python - <<PY
import csv, glob
f = glob.glob('$OUT/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    print('%-44s %5s %9.2f us' % (r['Name'].replace('(anonymous namespace)::','').split('(')[0][:44], r['Calls'], float(r['AverageNs'])/1e3))
PY
