"""development aid: time per pass of the c5 job (5e5 x 1e8) as Z zones in one launch set, for the library named by $NWAYHIP_LIBRARY
and a cell factor:   python tools/dev/zones_ab.py <zones> [cell factor] [n_primary n_secondary]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from nway_amd import distributed, _hip

dev = torch.device('cuda', 0)
zpr = int(sys.argv[1]) if len(sys.argv) > 1 else 8
factor = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
n0 = int(sys.argv[3]) if len(sys.argv) > 3 else 500000
n1 = int(sys.argv[4]) if len(sys.argv) > 4 else 100000000
streams = int(os.environ.get('ZONES_STREAMS', '1'))
tabs = list(bench.make_workload(n0, n1, 78))
tuning = dict(sphere_cell_factor=factor) if factor > 0 else None
eng = distributed.ZoneShardedMatch(tabs[0], tabs[1:], 5.0, 0.9, dev, zones_per_rank=zpr, local_only=True, tuning=tuning, streams=streams)
for _ in range(10):
	eng.step()
torch.cuda.synchronize(dev)
plan0 = [z['plan'] for z in eng.zones if z['plan'] is not None][0]
plan0.profile(0b100011, every=1)
best = None
for rep in range(3):
	t0 = time.perf_counter()
	for _ in range(20):
		eng.step()
	torch.cuda.synchronize(dev)
	us = (time.perf_counter() - t0) * 1e6 / 20
	best = us if best is None else min(best, us)
n, ms = plan0.profile_read()
st = eng.read_status()
stage = dict((_hip.STAGE_NAMES[i], ms[i] * 1e3 / n[i]) for i in range(len(_hip.STAGE_NAMES)) if n[i])
print('%s zones %d x %d streams, factor %g: %.1f us per pass (best of 3 x 20), batched %s, rows %d, flags %d, registrations %d survivors %d tests %d; stages (events, us): %s' % (
	os.path.basename(os.environ.get('NWAYHIP_LIBRARY', 'libnwayhip.so')), zpr, streams, factor, best, eng.batched, int(st[_hip.ST_ROWS]), int(st[_hip.ST_FLAGS]),
	int(st[_hip.ST_REGISTRATIONS]), int(st[_hip.ST_SURVIVORS]), int(st[_hip.ST_TESTS]), ', '.join('%s %.1f' % kv for kv in stage.items())))
