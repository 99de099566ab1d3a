# kernel trace of the c5 job (5e5 x 1e8) as Z zones in one launch set: bash tools/dev/zones_trace.sh <tag> <zones> [streams; 0 = one launch set]
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
cat > /tmp/zones_job.py <<PY
import sys, time
sys.path.insert(0, '$GRAFT_REPO_ROOT')
import torch, bench
from nway_amd import distributed, _hip
dev = torch.device('cuda', 0)
zpr, streams = int(sys.argv[1]), int(sys.argv[2])
tabs = list(bench.make_workload(500000, 100000000, 78))
eng = distributed.ZoneShardedMatch(tabs[0], tabs[1:], 5.0, 0.9, dev, zones_per_rank=zpr, streams=max(streams, 1), one_launch=streams == 0)
for _ in range(5):
	eng.step()
torch.cuda.synchronize(dev)
t0 = time.perf_counter()
probe = torch.zeros(64, device=dev) if len(sys.argv) > 3 and sys.argv[3] == 'probe' else None
for _ in range(20):
	eng.step()
	if probe is not None:
		probe.add_(1.0)   # (a 64-element kernel right behind the tail: its "duration" in the trace = what the tail leaves behind)
torch.cuda.synchronize(dev)
print('us per pass', (time.perf_counter() - t0) * 1e6 / 20, 'batched', eng.batched)
PY
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python /tmp/zones_job.py ${2:-8} ${3:-0} ${4:-} > $OUT/job.log 2>&1
tail -2 $OUT/job.log
python - <<PY
import csv, glob
f = glob.glob('$OUT/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    print('%-40s %5s %9.2f us' % (r['Name'].replace('(anonymous namespace)::','').split('(')[0][:40], r['Calls'], float(r['AverageNs'])/1e3))
PY
