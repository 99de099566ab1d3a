import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests')); 
import numpy as np, torch
import test_hip_fuzz as fz
import nway_amd as nw
from nway_amd import distributed, _hip
seed = 4445
rng = np.random.default_rng(7000 + seed)
k = int(rng.integers(2, 6))
tabs, radius = (fz.flat_case if seed % 2 == 0 else fz.sphere_case)(rng, k)
if seed % 2 == 1 and k > 4: tabs = tabs[:4]
comp = float(rng.choice([1.0, 0.9, 0.5]))
print('k', len(tabs), 'sizes', [len(t['ra']) for t in tabs], 'radius', radius, 'comp', comp)
dev = torch.device('cuda', 0)
whole = distributed.ZoneShardedMatch(tabs[0], tabs[1:], radius, comp, dev, zones_per_rank=1, local_only=True)
whole.step(); w = whole.gather_table()
print('whole', whole.zones[0]['plan'].description)
for zpr in (2, 8):
	eng = distributed.ZoneShardedMatch(tabs[0], tabs[1:], radius, comp, dev, zones_per_rank=zpr, local_only=True)
	eng.step(); g = eng.gather_table()
	print(zpr, [z['plan'].description['tail'] + '/' + str(z['plan'].description['link_slots']) for z in eng.zones if z['plan'] is not None])
	for key in w:
		a, b = np.asarray(g[key]), np.asarray(w[key])
		same = (a == b) | ((a != a) & (b != b))
		if not same.all(): print('   ', key, (~same).sum(), 'differ, max rel', np.nanmax(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))
	grp = np.bincount(np.asarray(w[tabs[0]['name']]))
	print('   largest group', grp.max())
