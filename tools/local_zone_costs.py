"""ONE GPU, several declination zones (ZoneShardedMatch(zones_per_rank=Z, streams=S)): time per pass of the fixed-size jobs
BASELINE names against the same job as one zone.  Streams 0 = the zones as ONE launch set (round 6: one registration, one sweep,
one tail launch for all zones; nwayhip_zones_*), S >= 1 = a pass per zone, one after the other or round robin on S streams.  Set-up (the one-time bucketing of the catalogues by zone) is outside the pass,
as the set-up exchanges of the multi-GPU modes are.

    python tools/local_zone_costs.py [c3s|c4s|c5] > profiles/local_zone_costs_r06.md      (on the GPU box)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from nway_amd import distributed, _hip

dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
jobs = dict(c3s=([100000, 10000000], 5.0, [(1, 1), (2, 1), (2, 0), (4, 0)]),
	c4s=([100000, 1000000, 1000000], 10.0, [(1, 1), (2, 2), (3, 3)]),
	c5=([500000, 100000000], 5.0, [(1, 1), (4, 1), (4, 0), (5, 0), (6, 0), (8, 2), (8, 1), (8, 0), (12, 0), (16, 0)]))
which = sys.argv[1:] or ['c3s', 'c4s', 'c5']
print('| job | zones x streams | us per pass | rows | paths of the zones | factor against one zone |')
print('|---|---|---|---|---|---|')
for name in which:
	sizes, radius, grid = jobs[name]
	tabs = list(bench.make_workload(sizes[0], sizes[1], 78)) if len(sizes) == 2 else bench.make_workload3(sizes[0], sizes[1], sizes[2], 78)
	base = None
	for zpr, streams in grid:
		eng = distributed.ZoneShardedMatch(tabs[0], tabs[1:], radius, 0.9, dev, zones_per_rank=zpr, streams=max(streams, 1), one_launch=streams == 0)
		for _ in range(30):
			eng.step()
		torch.cuda.synchronize(dev)
		steps = 60 if name != 'c5' else 20
		t0 = time.perf_counter()
		for _ in range(steps):
			eng.step()
		torch.cuda.synchronize(dev)
		us = (time.perf_counter() - t0) * 1e6 / steps
		st = eng.read_status()
		assert int(st[_hip.ST_FLAGS]) == 0
		rows = int(st[_hip.ST_ROWS])
		base = base or (us, rows)
		assert rows == base[1], (rows, base)
		paths = sorted(set('%s/%s' % (z['plan'].description['sweep'], z['plan'].description['tail']) for z in eng.zones if z['plan'] is not None))
		assert eng.batched == (streams == 0 and zpr > 1), (eng.batched, zpr, streams)
		print('| %s %s | %d x %s | %.1f | %d | %s | %.2f |' % (name, ' x '.join('%g' % n for n in sizes), zpr, streams if streams else 'one launch set', us, rows, ', '.join(paths), base[0] / us))
		sys.stdout.flush()
		eng.close()
		del eng
		torch.cuda.empty_cache()
