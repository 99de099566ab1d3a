"""development aid: mid-size random matches (1e3..5e4 primaries, up to 1e6 secondaries, chance
neighbours per primary from 0.01 to a few, patches and the whole sky, k = 2..4) against the C
oracle -- covers the fused sparse kernels, their fall-back and the general path at sizes where
workgroup regions, scans and look-back chains are long
    python tools/dev/soak_mid.py 0 40        (on the GPU box; up to SOAK_KMAX = 8 catalogues by default since round 4: the k >= 5 failure of round 3 hid behind a default of 4)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import test_hip_parity as tp
import nway_amd as nw
from goldenutil import cat

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad, t0, rows = [], time.time(), 0
for seed in range(lo, hi):
	rng = np.random.default_rng(5000 + seed)
	k = int(rng.integers(2, int(os.environ.get('SOAK_KMAX', '8')) + 1))
	n0 = int(10 ** rng.uniform(3, 4.7))
	radius = float(rng.choice([2.0, 5.0, 10.0, 20.0]))
	lam = 10 ** rng.uniform(-2, 0.7 if k == 2 else (0.3 if k < 5 else -0.5))     # chance neighbours per primary and catalogue
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
	if os.environ.get('SOAK_MISSING') and whole_sky:
		# sources without a coordinate: they match nothing and must not disturb the others (a NaN anywhere also means the all-sky scheme)
		for t in tabs:
			for col in ('ra', 'dec'):
				t[col] = np.array(t[col], dtype=float)
				t[col][rng.choice(len(t[col]), size=int(rng.integers(1, 6)), replace=False)] = np.nan
	names = [t['name'] for t in tabs]
	kw = dict(correction='cli') if (k > 2 and seed % 2 == 0) else {}
	if os.environ.get('SOAK_CORR') == 'api':
		kw = {}
	try:
		t = tp.oracle_vs_hip(nw, tabs, radius, float(rng.choice([1.0, 0.9, 0.6])), names, oracle=tp.orc_c, **kw)
		rows += len(t['ncat'])
		print('seed %d ok: k=%d n0=%d ns=%d lambda=%.3f %s rows=%d' % (seed, k, n0, ns, lam, 'sky' if whole_sky else 'patch', len(t['ncat'])))
	except AssertionError as e:
		bad.append(seed)
		print('seed %d FAILED (k=%d n0=%d ns=%d lambda=%.3f): %s' % (seed, k, n0, ns, lam, " | ".join(str(e).strip().splitlines()[:12])[:900]))
print('%d configurations, %d rows, %d failures %s in %.0f s' % (hi - lo, rows, len(bad), bad, time.time() - t0))
