"""development aid: the 2-way sparse tail against the general path on the bench workload: which rows differ, and their groups"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import bench
import nway_amd as nw
from test_full_size import hip_table

n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
n1 = int(sys.argv[2]) if len(sys.argv) > 2 else 10000000
prim, sec = bench.make_workload(n0, n1, 1)
t, _ = hip_table(nw, [prim, sec], 5.0, 0.9)
g, _ = hip_table(nw, [prim, sec], 5.0, 0.9, link_slots=-1)
print('rows', len(t['ncat']), len(g['ncat']), t['_desc'])
bad = np.zeros(len(t['ncat']), dtype=bool)
for key in t:
	if key.startswith('_'):
		continue
	a, b = np.asarray(t[key]), np.asarray(g[key])
	d = ~((a == b) | (np.isnan(a.astype(float)) & np.isnan(b.astype(float))))
	if d.any():
		print('%-28s %d rows differ' % (key, d.sum()))
		bad |= d
prims = np.unique(t['PRIM'][bad])
print('primaries with a differing row:', prims[:20], '(wave %s, lane %s)' % ((prims[:20] // 64), (prims[:20] % 64)))
groups = np.bincount(t['PRIM'].astype(np.int64), minlength=n0)
print('groups of three or more rows:', (groups >= 3).sum(), 'their primaries (first 30):', np.flatnonzero(groups >= 3)[:30])
for p in prims[:6]:
	rows = np.flatnonzero(t['PRIM'] == p)
	wave = p // 64
	print('--- primary %d (wave %d lane %d): group sizes in its wave %s' % (p, wave, p % 64, groups[wave * 64:(wave + 1) * 64]))
	for key in ('SEC', 'Separation_PRIM_SEC', 'dist_bayesfactor', 'dist_post', 'prob_has_match', 'prob_this_match', 'match_flag'):
		print('   %-22s sparse %s | general %s' % (key, t[key][rows], g[key][rows]))
