"""development aid: random sparse 2-way matches through k_tail2 (csrc/tail2.inc: one candidate per lane, second candidates adopted by
spare lanes of the wave, the general code for what does not fit) against the C oracle and, bit for bit, against the general path:
1e3..2e5 primaries, 1e-3..0.3 chance neighbours per primary, 20..100 % of the primaries with a counterpart (100 %: no lane idle by itself),
up to 8 % of them with a second candidate and a few with a third / fourth (the general code in their waves), scalar and per-source errors of
the secondaries, whole sky and flat patches, the script's float32 numerics now and then
    python tools/dev/soak_tail2.py 0 80        (on the GPU box)
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
from goldenutil import cat
from test_full_size import hip_table, compare
from goldenutil import soak_compare

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad, t0, rows, tails, flips = [], time.time(), 0, {}, 0
for seed in range(lo, hi):
	rng = np.random.default_rng(12000 + seed)
	n0 = int(10 ** rng.uniform(3, 5.3))
	radius = float(rng.choice([2.0, 5.0, 10.0, 20.0]))
	lam = 10 ** rng.uniform(-3, -0.5)
	whole_sky = seed % 2 == 0
	if whole_sky:
		area = 41252.96
		pos = lambda n: (rng.uniform(0, 360, n), np.degrees(np.arcsin(rng.uniform(-1, 1, n))))
	else:
		side = float(10 ** rng.uniform(-0.3, 0.8))
		area = side * side
		c_ra, c_dec = rng.uniform(20, 340), rng.uniform(-40, 40 - side)
		pos = lambda n: (c_ra + rng.uniform(0, side, n), c_dec + rng.uniform(0, side, n))
	pra, pdec = pos(n0)
	tabs = [cat('P', pra, pdec, rng.uniform(0.3, radius / 4, n0), area)]
	frac = 1.0 if seed % 4 == 0 else float(rng.uniform(0.2, 0.95))
	crowd = 0 if seed % 5 == 4 else int(n0 * rng.uniform(0.0, 0.08))
	m = int(n0 * frac)
	ns = int(min(2e6, max(2 * n0 + 16, lam * area / (np.pi * (radius / 3600.) ** 2))))
	ra, dec = pos(ns)
	ra[:m] = pra[:m] + rng.normal(0, radius / 5, m) / 3600. / np.maximum(np.cos(np.radians(pdec[:m])), 1e-3)
	dec[:m] = np.clip(pdec[:m] + rng.normal(0, radius / 5, m) / 3600., -90, 90)
	who = rng.choice(n0, size=crowd, replace=False)  # a second candidate for these
	ra[m:m + crowd] = pra[who] + rng.normal(0, radius / 4, crowd) / 3600. / np.maximum(np.cos(np.radians(pdec[who])), 1e-3)
	dec[m:m + crowd] = np.clip(pdec[who] + rng.normal(0, radius / 4, crowd) / 3600., -90, 90)
	extra = 0
	if seed % 3 == 1:  # a third and a fourth candidate for a few primaries, an exact duplicate of a secondary for one
		extra = 6
		t = rng.choice(n0, size=3, replace=False)
		at = m + crowd
		ra[at:at + 4] = pra[t[0]]
		dec[at:at + 4] = np.clip(pdec[t[0]] + np.array([1, -1.3, 2.1, -2.7]) * radius / 12 / 3600., -90, 90)  # (no two at one distance: a tie in p_i is decided by the last bit of a libm)
		ra[at + 4:at + 6] = ra[0]
		dec[at + 4:at + 6] = dec[0]
	order = rng.permutation(ns)
	err = float(rng.uniform(0.2, 1.0)) * np.ones(ns) if seed % 2 else rng.uniform(0.2, 1.0, ns)
	sec = cat('S', (ra[order] % 360) if whole_sky else ra[order], dec[order], err, area)
	if seed % 6 == 2:
		sec['error'] = float(err[0])  # ONE positional error for the whole catalogue (the kernel's constants)
	tabs.append(sec)
	names = ['P', 'S']
	comp = float(rng.choice([1.0, 0.9, 0.6]))
	f32 = seed % 9 == 5
	try:
		q, st = hip_table(nw, tabs, radius, comp, f32_roundtrip=f32)
		g, _ = hip_table(nw, tabs, radius, comp, link_slots=-1, f32_roundtrip=f32)
		for key in q:
			if not key.startswith('_'):
				np.testing.assert_array_equal(q[key], g[key], err_msg=key)
		otabs = [tabs[0], dict(sec, error=np.broadcast_to(np.asarray(sec['error'], dtype=float), (ns,)).copy())]
		o = orc_c.nway_match(otabs, radius, comp, f32_roundtrip=f32)
		# (near-zero log Bayes factors and, with the script's float32 numerics, float32 rounding flips: goldenutil.soak_compare)
		flips += soak_compare(q, o, names, f32=f32)
		rows += len(q['ncat'])
		tails[q['_desc']['tail']] = tails.get(q['_desc']['tail'], 0) + 1
		groups = np.bincount(q['P'].astype(np.int64), minlength=n0)
		print('seed %d ok: n0=%d ns=%d lambda=%.4f %s frac=%.2f crowd=%d extra=%d tail=%s rows=%d groups of 3+: %d, largest %d' % (seed, n0, ns, lam,
			'sky' if whole_sky else 'patch', frac, crowd, extra, q['_desc']['tail'], len(q['ncat']), (groups >= 3).sum(), groups.max()), flush=True)
	except AssertionError as e:
		bad.append(seed)
		print('seed %d FAILED (n0=%d ns=%d lambda=%.4f frac=%.2f crowd=%d): %s' % (seed, n0, ns, lam, frac, crowd, " | ".join(str(e).strip().splitlines()[:12])[:900]), flush=True)
print('%d configurations, %d rows, tails %s, %d failures %s, %d rows excused as float32 rounding flips, in %.0f s' % (hi - lo, rows, tails, len(bad), bad, flips, time.time() - t0))
