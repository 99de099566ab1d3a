"""development aid: one configuration of tools/dev/soak_mid.py (by seed) through the fused path, the general path, the C oracle and
the numpy oracle -- who disagrees with whom
    python tools/dev/k5_probe.py 103"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import nway_amd as nw
import nway_oracle as orc
import nway_oracle_c as orc_c
from goldenutil import cat
from test_full_size import hip_table

seed = int(sys.argv[1])
rng = np.random.default_rng(5000 + seed)
k = int(rng.integers(2, int(os.environ.get('SOAK_KMAX', '6')) + 1))
n0 = int(10 ** rng.uniform(3, 4.7))
radius = float(rng.choice([2.0, 5.0, 10.0, 20.0]))
lam = 10 ** rng.uniform(-2, 0.7 if k == 2 else (0.3 if k < 5 else -0.5))
whole_sky = seed % 3 == 0
if whole_sky:
	area = 41252.96
	pos = lambda n: (rng.uniform(0, 360, n), np.degrees(np.arcsin(rng.uniform(-1, 1, n))))
else:
	side = float(10 ** rng.uniform(-0.3, 1.0))
	area = side * side
	c_ra, c_dec = rng.uniform(20, 340), rng.uniform(-40, 40 - side)
	pos = lambda n: (c_ra + rng.uniform(0, side, n), c_dec + rng.uniform(0, side, n))
ns = int(min(1e6, max(100, lam * area / (np.pi * (radius / 3600.) ** 2))))
pra, pdec = pos(n0)
tabs = [cat('P', pra, pdec, rng.uniform(0.3, radius / 4, n0), area)]
for c in range(1, k):
	ra, dec = pos(ns)
	m = min(ns, int(n0 * rng.uniform(0.2, 0.9)))
	ra[:m] = pra[:m] + rng.normal(0, radius / 5, m) / 3600. / np.maximum(np.cos(np.radians(pdec[:m])), 1e-3)
	dec[:m] = np.clip(pdec[:m] + rng.normal(0, radius / 5, m) / 3600., -90, 90)
	order = rng.permutation(ns)
	tabs.append(cat('S%d' % c, ra[order] % 360 if whole_sky else ra[order], dec[order], float(rng.uniform(0.2, 1.0)) * np.ones(ns), area))
comp = float(rng.choice([1.0, 0.9, 0.6]))
names = [t['name'] for t in tabs]
print('seed', seed, 'k', k, 'n0', n0, 'ns', ns, 'radius', radius, 'lambda %.3f' % lam, 'sky' if whole_sky else 'patch')
fused, _ = hip_table(nw, tabs, radius, comp)
general, _ = hip_table(nw, tabs, radius, comp, link_slots=-1)
c_or = orc_c.nway_match(tabs, radius, comp)
tables = dict(fused=fused, general=general, c_oracle=c_or)
if n0 * k < 200000 or os.environ.get('K5_NUMPY'):
	tables['numpy_oracle'] = orc.nway_match(tabs, radius, comp)
print('paths:', fused['_desc']['tail'], general['_desc']['tail'], 'rows', {n: len(t['ncat']) for n, t in tables.items()})
keys = names + ['ncat', 'match_flag', 'Separation_max', 'dist_bayesfactor', 'prob_this_match'] + ['Separation_%s_%s' % (names[i], names[j]) for i in range(k) for j in range(i + 1, k)]
ref = 'c_oracle'
for n, t in tables.items():
	if n == ref or len(t['ncat']) != len(tables[ref]['ncat']):
		continue
	for key in keys:
		a, b = np.asarray(t[key], dtype=float), np.asarray(tables[ref][key], dtype=float)
		bad = ~(np.isclose(a, b, rtol=1e-6, atol=1e-9) | (np.isnan(a) & np.isnan(b)))
		if bad.any():
			i = int(np.flatnonzero(bad)[0])
			print('  %s vs %s: %s differs in %d rows; first row %d (primary %d): %r vs %r' % (n, ref, key, bad.sum(), i, int(t[names[0]][i]), a[i], b[i]))
			rows = np.flatnonzero(np.asarray(t[names[0]]) == t[names[0]][i])
			if key.startswith('Separation_S') or key == 'match_flag':
				for r in rows[:12]:
					print('     row', r, [int(t[nm][r]) for nm in names], 'flag', int(t['match_flag'][r]), int(tables[ref]['match_flag'][r]), 'p_i %.17g %.17g' % (t['prob_this_match'][r], tables[ref]['prob_this_match'][r]))
			if key == 'match_flag' or key.startswith('Separation'):
				seps = ['Separation_%s_%s' % (names[i2], names[j2]) for i2 in range(k) for j2 in range(i2 + 1, k)]
				for r in rows:
					d = [(sk, float(t[sk][r]), float(tables[ref][sk][r])) for sk in seps if not (np.isclose(t[sk][r], tables[ref][sk][r], rtol=1e-9) or (np.isnan(t[sk][r]) and np.isnan(tables[ref][sk][r])))]
					if d:
						print('     row', r, [int(t[nm][r]) for nm in names], d)
				allsep = {}
				for r in rows:
					for sk in seps:
						allsep.setdefault(round(float(tables[ref][sk][r]), 9), (sk, [int(tables[ref][nm][r]) for nm in names]))
				print('     oracle separations of this primary:', sorted((v, w[0]) for v, w in allsep.items() if v == v)[:40])
			break
