"""development aid: random sparse 3-way matches through the tail with four lanes per primary (csrc/tail3q.inc)
against the C oracle and, bit for bit, against the one-lane walk it replaces: 1e3..5e4 primaries, 1e-3..3e-2 chance
neighbours per primary, a few per cent of the primaries with a second (sometimes a third: the QUAD_DEEP repeat)
candidate in a catalogue, whole sky and flat patches, with and without the script's correction
    python tools/dev/soak_quad.py 0 60        (on the GPU box)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import nway_amd as nw
import nway_oracle_c as orc_c
from nway_amd import _hip
from goldenutil import cat, soak_compare
from test_full_size import hip_table, compare

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad, t0, rows, tails = [], time.time(), 0, {}
for seed in range(lo, hi):
	rng = np.random.default_rng(9000 + seed)
	n0 = int(10 ** rng.uniform(3, 4.7))
	radius = float(rng.choice([2.0, 5.0, 10.0, 20.0]))
	lam = 10 ** rng.uniform(-3, -1.5)
	whole_sky = seed % 2 == 0
	if whole_sky:
		area = 41252.96
		pos = lambda n: (rng.uniform(0, 360, n), np.degrees(np.arcsin(rng.uniform(-1, 1, n))))
	else:
		side = float(10 ** rng.uniform(-0.3, 0.6))
		area = side * side
		c_ra, c_dec = rng.uniform(20, 340), rng.uniform(-40, 40 - side)
		pos = lambda n: (c_ra + rng.uniform(0, side, n), c_dec + rng.uniform(0, side, n))
	pra, pdec = pos(n0)
	tabs = [cat('P', pra, pdec, rng.uniform(0.3, radius / 4, n0), area)]
	crowd = 0 if seed % 5 == 4 else int(n0 * rng.uniform(0.0, 0.05))
	third = seed % 7 == 3
	for c in (1, 2):
		ns = int(min(1e6, max(2 * n0, lam * area / (np.pi * (radius / 3600.) ** 2))))
		ra, dec = pos(ns)
		m = int(n0 * rng.uniform(0.2, 0.9))
		ra[:m] = pra[:m] + rng.normal(0, radius / 5, m) / 3600. / np.maximum(np.cos(np.radians(pdec[:m])), 1e-3)
		dec[:m] = np.clip(pdec[:m] + rng.normal(0, radius / 5, m) / 3600., -90, 90)
		who = rng.choice(n0, size=crowd, replace=False)  # a second candidate for these
		ra[m:m + crowd] = pra[who] + rng.normal(0, radius / 4, crowd) / 3600. / np.maximum(np.cos(np.radians(pdec[who])), 1e-3)
		dec[m:m + crowd] = np.clip(pdec[who] + rng.normal(0, radius / 4, crowd) / 3600., -90, 90)
		if third and c == 2:
			ra[m + crowd:m + crowd + 2] = pra[5]
			dec[m + crowd:m + crowd + 2] = np.clip(pdec[5] + np.array([1, -1]) * radius / 10 / 3600., -90, 90)
		order = rng.permutation(ns)
		tabs.append(cat('S%d' % c, ra[order] % 360 if whole_sky else ra[order], dec[order], float(rng.uniform(0.2, 1.0)) * np.ones(ns), area))
	names = [t['name'] for t in tabs]
	corr = seed % 3 == 0
	opt = dict(correction=_hip.CORRECTION_CLI) if corr else {}
	comp = float(rng.choice([1.0, 0.9, 0.6]))
	try:
		q, _ = hip_table(nw, tabs, radius, comp, tuning=dict(enable=_hip.ENABLE_QUAD3), **opt)
		lane, _ = hip_table(nw, tabs, radius, comp, tuning=dict(disable=_hip.DISABLE_QUAD3), **opt)
		for key in q:
			if not key.startswith('_'):
				np.testing.assert_array_equal(q[key], lane[key], err_msg=key)
		soak_compare(q, orc_c.nway_match(tabs, radius, comp, correction='cli' if corr else 'api'), names)  # (goldenutil: near-zero log Bayes factors by their absolute error)
		rows += len(q['ncat'])
		tails[q['_desc']['tail']] = tails.get(q['_desc']['tail'], 0) + 1
		groups = np.bincount(q['P'].astype(np.int64), minlength=n0)
		print('seed %d ok: n0=%d lambda=%.4f %s corr=%d crowd=%d tail=%s rows=%d largest group %d' % (seed, n0, lam, 'sky' if whole_sky else 'patch', corr, crowd,
			q['_desc']['tail'], len(q['ncat']), groups.max()))
	except AssertionError as e:
		bad.append(seed)
		print('seed %d FAILED (n0=%d lambda=%.4f): %s' % (seed, n0, lam, " | ".join(str(e).strip().splitlines()[:12])[:900]))
print('%d configurations, %d rows, tails %s, %d failures %s in %.0f s' % (hi - lo, rows, tails, len(bad), bad, time.time() - t0))
