"""oracle/healpix.py (own restatement of the HEALPix pixelisation, stands in for the absent
healpy when the reference's all-sky branch is exercised) and the all-sky golden fixture.

What pins the pixelisation: the examples printed in healpy's documentation (quoted below from
the docstrings of ang2pix, nest2ring, ring2nest and get_all_neighbours; arguments that sit
exactly on a pixel border are left out, their answer depends on the rounding of cos(pi/2)),
plus geometric checks that need no second implementation.  healpy itself never ran here.
"""
import numpy as np

from goldenutil import golden, cat, assert_table_matches
from oracle import healpix as hp
from oracle import nway_oracle as orc

pi = np.pi
TIGHT = dict(rtol=1e-9, atol=1e-13)


def test_documented_examples():
	assert hp.ang2pix(16, pi / 2, 0) == 1440
	np.testing.assert_array_equal(hp.ang2pix(16, [pi / 2, pi / 4, 0], [0., pi / 4, 0]), [1440, 427, 0])
	assert [int(hp.ang2pix(n, pi / 2, 0)) for n in (1, 2, 4, 8, 16)] == [4, 12, 72, 336, 1440]
	assert hp.nest2ring(16, 1130) == 1504 and hp.ring2nest(16, 1504) == 1130
	np.testing.assert_array_equal(hp.nest2ring(2, np.arange(10)), [13, 5, 4, 0, 15, 7, 6, 1, 17, 9])
	np.testing.assert_array_equal(hp.ring2nest(2, np.arange(10)), [3, 7, 11, 15, 2, 1, 6, 5, 10, 9])
	np.testing.assert_array_equal(hp.get_all_neighbours(1, 4), [11, 7, 3, -1, 0, 5, 8, -1])
	np.testing.assert_array_equal(hp.get_all_neighbours(1, pi / 2, pi / 2), [8, 4, 0, -1, 1, 6, 9, -1])
	assert abs(hp.nside2resol(128) * 180 * 60 / pi - 27.483891294539248) < 1e-9  # arcmin, as documented


def _sphere(rng, n):
	return np.arccos(rng.uniform(-1, 1, n)), rng.uniform(0, 2 * pi, n)


def test_equal_area_and_numbering():
	rng = np.random.default_rng(5)
	theta, phi = _sphere(rng, 400000)
	for nside in (1, 2, 8):
		npix = hp.nside2npix(nside)
		nest = hp.ang2pix(nside, theta, phi, nest=True)
		ring = hp.ang2pix(nside, theta, phi, nest=False)
		assert nest.min() >= 0 and nest.max() < npix
		counts = np.bincount(nest, minlength=npix)
		expect = len(theta) / npix
		assert np.abs(counts - expect).max() < 6 * np.sqrt(expect)
		np.testing.assert_array_equal(hp.nest2ring(nside, nest), ring)
		# ring numbering runs from north to south: the pixel number is monotonic in the ring's z
		order = np.argsort(ring, kind='stable')
		z = np.cos(theta[order])
		ringz = np.array([z[ring[order] == p].mean() for p in np.unique(ring)])
		assert (np.diff(ringz) < 0.05).all()


def test_neighbours_symmetric_and_adjacent():
	"""q is a neighbour of p iff p is one of q; and crossing a pixel border by a small step
	always lands in a listed neighbour (both numberings, belt and caps)"""
	rng = np.random.default_rng(6)
	for nside in (2, 4, 16):
		npix = hp.nside2npix(nside)
		nb = hp.get_all_neighbours(nside, np.arange(npix), nest=True)
		# three faces meet at 8 points of the sphere; the 3 pixels touching each lack one neighbour
		assert ((nb >= 0).sum(axis=0) >= 7).all() and ((nb < 0).sum() == 24)
		pairs = set()
		for m in range(8):
			for p, q in zip(np.arange(npix), nb[m]):
				if q >= 0:
					pairs.add((int(p), int(q)))
		assert all((q, p) in pairs for p, q in pairs)
		theta, phi = _sphere(rng, 200000)
		own = hp.ang2pix(nside, theta, phi, nest=True)
		around = hp.get_all_neighbours(nside, theta, phi, nest=True)
		np.testing.assert_array_equal(around, nb[:, own])
		step = 0.05 * hp.nside2resol(nside)
		for ang in np.linspace(0, 2 * pi, 7)[:-1]:
			t2 = np.clip(theta + step * np.cos(ang), 1e-9, pi - 1e-9)
			p2 = phi + step * np.sin(ang) / np.sin(theta).clip(0.05)
			moved = hp.ang2pix(nside, t2, p2, nest=True)
			changed = moved != own
			assert changed.any()
			assert ((around == moved[None, :]).any(axis=0) | ~changed).all()


def _tables(g):
	return [cat('ABC'[i], g['ra%d' % i], g['dec%d' % i], g['err%d' % i], g['area'][0]) for i in range(3)]


def test_healpix_branch_equals_pairwise_radius_rule():
	"""after the radius filter the reference's HEALPix candidate set is exactly
	"every present pairwise separation < radius" -- checked here with the literal restatement
	of the branch on inputs that include both poles and the RA seam"""
	g = golden('allsky')
	tabs = _tables(g)
	err = float(g['radius'][0]) / 60 / 60
	for k, key in ((2, 'w2'), (3, 'w3')):
		rt = [(t['ra'], t['dec']) for t in tabs[:k]]
		lit = orc.crossproduct_healpix_literal(rt, err)
		assert len(lit) == int(g[key + '_crossproduct_nrows'][0])
		post = orc.create_match_table(tabs[:k], float(g['radius'][0]), tuples=lit)[1]
		direct = orc.enumerate_tuples(rt, err, orc.SPHERE, float(g['radius'][0]))
		np.testing.assert_array_equal(post, direct)
		np.testing.assert_array_equal(direct, g[key + '_idx'])


def test_allsky_golden_tables():
	"""the reference's own all-sky run (its HEALPix branch over oracle/healpix.py) against the
	oracle's all-sky enumeration, the full probability table included"""
	g = golden('allsky')
	tabs = _tables(g)
	radius, c = float(g['radius'][0]), float(g['completeness'][0])
	t = orc.nway_match(tabs[:2], radius, c, literal_groups=True)
	assert_table_matches(t, g, 'w2_', ['A', 'B'], **TIGHT)
	t = orc.nway_match(tabs, radius, c, literal_groups=True)
	assert_table_matches(t, g, 'w3_', ['A', 'B', 'C'], **TIGHT)
	# the script (nway.py executed by make_script_golden.py) on the same tables: float32 separations, correction loop
	from goldenutil import script_golden, assert_script_correction
	gs = script_golden()
	# (at the contract's 1e-6, not tighter: near the poles the last bit of a float64 separation differs between numpy's loops over
	# the script's big-endian columns and the oracle's native ones, and where it decides which float32 value the separation becomes
	# a p_i moves by up to 1.3e-7 -- two rows of 14 031)
	ts = orc.nway_match(tabs, radius, c, correction='cli', f32_roundtrip=True)
	assert_table_matches(ts, gs, 'allsky_w3_script_', ['A', 'B', 'C'], rtol=1e-6, atol=1e-13)
	assert_script_correction(ts, gs, 'allsky_w3_', rtol=1e-6)
	ts = orc.nway_match(tabs[:2], radius, c, correction='cli', f32_roundtrip=True)
	assert_table_matches(ts, gs, 'allsky_w2_script_', ['A', 'B'], rtol=1e-6, atol=1e-13)
