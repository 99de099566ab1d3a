#!/bin/bash
# BASELINE configs[4] (5e5 x 1e8, 5") on ONE GPU as Z declination zones in one launch set: kernel trace + stats, then FETCH_SIZE and
# WRITE_SIZE in passes of their own (--pmc is never combined with other traces), summarised into gpurun_out/<tag>/zones.txt.
#   gpurun --timeout 1500 -- 'bash tools/profile_zones.sh zones_r06 8'        then: cp gpurun_out/zones_r06/zones.txt profiles/rocprof_zones_r06.txt
TAG=${1:-zones_r06}
Z=${2:-8}
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
cat > /tmp/zones_job.py <<PY
import sys, time
sys.path.insert(0, '$ROOT')
import torch, bench
from nway_amd import distributed, _hip
dev = torch.device('cuda', 0)
zpr, steps = int(sys.argv[1]), int(sys.argv[2])
tabs = list(bench.make_workload(500000, 100000000, 78))
eng = distributed.ZoneShardedMatch(tabs[0], tabs[1:], 5.0, 0.9, dev, zones_per_rank=zpr, local_only=True)
for _ in range(5):
	eng.step()
torch.cuda.synchronize(dev)
t0 = time.perf_counter()
for _ in range(steps):
	eng.step()
torch.cuda.synchronize(dev)
st = eng.read_status()
print('zones %d: %.1f us per pass over %d passes, one launch set %s, owner-computes registration %s, rows %d, flags %d' % (zpr, (time.perf_counter() - t0) * 1e6 / steps, steps,
	eng.batched, eng.owner_computes, int(st[_hip.ST_ROWS]), int(st[_hip.ST_FLAGS])))
PY
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python /tmp/zones_job.py $Z 20 > $OUT/stats.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python /tmp/zones_job.py $Z 4 > $OUT/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python /tmp/zones_job.py $Z 4 > $OUT/write.log 2>&1
python - <<PY > $OUT/zones.txt
import csv, glob
short = lambda n: n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
print('# BASELINE configs[4] (5e5 x 1e8 uniform sky, 5 arcsec) on ONE GPU as $Z declination zones in ONE launch set (tools/profile_zones.sh)')
print('# ' + open('$OUT/stats.log').read().strip().splitlines()[-1])
print('# rocprofv3 --kernel-trace --stats (the launches of the set: 25 passes; k_sweep / k_register_x / k_tail2 / k_clear: the zones\' settling runs, one plan at a time)')
print('%-34s %6s %10s %10s' % ('kernel', 'calls', 'avg_us', 'max_us'))
for f in glob.glob('$OUT/stats/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if short(r['Name']).startswith('k_'):
            print('%-34s %6s %10.2f %10.2f' % (short(r['Name'])[:34], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MaxNs']) / 1e3))
print()
print('# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), mean per dispatch, MB (KiB as reported x 1024); the sweep streams with 16 B per lane:')
print('# gfx950 reports half of such a read (MI355X_MICROARCH.md) -- its uncounted half, 8 B x the sources streamed, is NOT added here')
acc = {}
for which in ('fetch', 'write'):
    for f in glob.glob('$OUT/%s/*/*counter_collection.csv' % which):
        for r in csv.DictReader(open(f)):
            acc.setdefault((short(r['Kernel_Name']), r['Counter_Name']), []).append(float(r['Counter_Value']))
for (k, c), v in sorted(acc.items()):
    if k.startswith('k_') and 'zones' in k:
        print('%-34s %-12s %10.1f' % (k[:34], c, sum(v) / len(v) * 1024 / 1e6))
PY
cat $OUT/zones.txt
