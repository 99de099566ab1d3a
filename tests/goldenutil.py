"""helpers shared by the parity tests: golden-vector loading and table comparison"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
if ROOT not in sys.path:
	sys.path.insert(0, ROOT)


def golden(name):
	return np.load(os.path.join(GOLDEN, name + '.npz'))


def cat(name, ra, dec, error, area):
	return dict(name=name, ra=np.array(ra, dtype=float), dec=np.array(dec, dtype=float),
		error=np.array(error, dtype=float), area=float(area), mags=[], maghists=[], magnames=[])


def ell_tables():
	"""tests/elltest/randomcat{X,R,O}.fits of the reference, rebuilt exactly from the
	micro-degree integers stored in ell_inputs.npz."""
	g = golden('ell_inputs')
	names = [str(n) for n in g['names']]
	X = cat(names[0], g['X_ra_u'] / 1e6, g['X_dec_u'] / 1e6, g['X_err_u'] / 1e6, g['area'][0])
	R = cat(names[1], g['R_ra_u'] / 1e6, g['R_dec_u'] / 1e6, g['R_err_u'] / 1e6, g['area'][1])
	O = cat(names[2], g['O_ra_u'] / 1e6, g['O_dec_u'] / 1e6, 0.1 * np.ones(len(g['O_ra_u'])), g['area'][2])
	return X, R, O


def xmm_tables():
	"""real COSMOS_XMM + the seeded uniform stand-ins for the missing OPT/IRAC catalogues
	(same recipe as tests/golden/make_golden.py:gen_xmm)."""
	g = golden('xmm_inputs')
	s = golden('xmm_syn')
	rng = np.random.RandomState(int(s['seed'][0]))
	n_opt, n_irac = int(s['n_opt'][0]), int(s['n_irac'][0])
	lo_ra, hi_ra, lo_dec, hi_dec = s['box']
	opt_ra = rng.uniform(lo_ra, hi_ra, size=n_opt)
	opt_dec = rng.uniform(lo_dec, hi_dec, size=n_opt)
	irac_ra = rng.uniform(lo_ra, hi_ra, size=n_irac)
	irac_dec = rng.uniform(lo_dec, hi_dec, size=n_irac)
	X = cat('XMM', g['RA'], g['DEC'], g['pos_err'].astype(float), 2.0)
	O = cat('OPT', opt_ra, opt_dec, 0.1 * np.ones(n_opt), 2.0)
	I = cat('IRAC', irac_ra, irac_dec, 0.5 * np.ones(n_irac), 2.0)
	return X, O, I


def mag_tables(maghist=None):
	"""XMM x seeded OPT stand-in with a magnitude column (recipe of make_golden.py:gen_mag)"""
	g = golden('xmm_inputs')
	m = golden('mag')
	rng = np.random.RandomState(int(m['seed'][0]))
	n_opt, k = int(m['n_opt'][0]), int(m['n_true'][0])
	opt_ra = rng.uniform(149.35, 150.87, size=n_opt)
	opt_dec = rng.uniform(1.47, 2.96, size=n_opt)
	mag = rng.normal(24.0, 1.5, size=n_opt)
	slots = rng.choice(n_opt, size=k, replace=False)
	opt_ra[slots] = g['RA'][:k] + rng.normal(0, 0.5, size=k) / 3600. / np.cos(np.radians(g['DEC'][:k]))
	opt_dec[slots] = g['DEC'][:k] + rng.normal(0, 0.5, size=k) / 3600.
	mag[slots] = rng.normal(21.0, 1.0, size=k)
	mag[rng.choice(n_opt, size=3000, replace=False)] = -99
	mag[rng.choice(n_opt, size=500, replace=False)] = np.nan
	np.testing.assert_allclose([opt_ra.sum(), opt_dec.sum(), np.nansum(mag), float(np.isnan(mag).sum())], m['checksum'], rtol=0, atol=0)
	X = cat('XMM', g['RA'], g['DEC'], g['pos_err'].astype(float), 2.0)
	O = cat('OPT', opt_ra, opt_dec, 0.1 * np.ones(n_opt), 2.0)
	O['mags'] = [mag]
	O['magnames'] = ['MAG']
	O['maghists'] = [maghist]
	return [X, O]


def mag3_tables():
	"""XMM x OPT x IRAC stand-ins with three magnitude columns (recipe of
	make_golden.py:mag3_catalogues; the shape of BASELINE configs[1])"""
	g = golden('xmm_inputs')
	m = golden('mag3')
	n_opt, n_irac, k_opt, k_irac = [int(v) for v in m['sizes']]
	rng = np.random.RandomState(int(m['seed'][0]))
	cols = {}
	for name, n, k, sig, m0 in (('OPT', n_opt, k_opt, 0.3, 24.0), ('IRAC', n_irac, k_irac, 0.6, 22.5)):
		ra = rng.uniform(149.35, 150.87, size=n)
		dec = rng.uniform(1.47, 2.96, size=n)
		mag = rng.normal(m0, 1.5, size=n)
		slots = rng.choice(n, size=k, replace=False)
		ra[slots] = g['RA'][:k] + rng.normal(0, sig, size=k) / 3600. / np.cos(np.radians(g['DEC'][:k]))
		dec[slots] = g['DEC'][:k] + rng.normal(0, sig, size=k) / 3600.
		mag[slots] = rng.normal(m0 - 3.0, 1.0, size=k)
		mag[rng.choice(n, size=n // 50, replace=False)] = -99
		mag[rng.choice(n, size=n // 300, replace=False)] = np.nan
		cols[name] = [ra, dec, mag]
		if name == 'OPT':
			colour = mag + rng.normal(0.0, 0.7, size=n)
			colour[slots] -= 0.8
			colour[~np.isfinite(mag) | (mag == -99)] = -99
			cols[name].append(colour)
	np.testing.assert_allclose([cols['OPT'][0].sum(), cols['OPT'][1].sum(), np.nansum(cols['OPT'][2]), np.nansum(cols['OPT'][3]),
		cols['IRAC'][0].sum(), cols['IRAC'][1].sum(), np.nansum(cols['IRAC'][2])], m['checksum'], rtol=0, atol=0)
	X = cat('XMM', g['RA'], g['DEC'], g['pos_err'].astype(float), 2.0)
	O = cat('OPT', cols['OPT'][0], cols['OPT'][1], 0.1 * np.ones(n_opt), 2.0)
	O['mags'], O['magnames'], O['maghists'] = [cols['OPT'][2], cols['OPT'][3]], ['R', 'I'], [None, None]
	I = cat('IRAC', cols['IRAC'][0], cols['IRAC'][1], 0.5 * np.ones(n_irac), 2.0)
	I['mags'], I['magnames'], I['maghists'] = [cols['IRAC'][2]], ['CH1'], [None]
	return [X, O, I]


def magmix_tables():
	"""3-way with magnitude columns on the primary (supplied histogram) and both secondaries
	(learned); arrays stored in tests/golden/magmix.npz"""
	g = golden('magmix')
	tabs = []
	for i, name in enumerate('PAB'):
		t = cat(name, g['ra%d' % i], g['dec%d' % i], g['err%d' % i], float(g['area'][0]))
		t['mags'], t['magnames'], t['maghists'] = [g['mag%d' % i].copy()], ['F' if i == 0 else 'M'], [None]
		tabs.append(t)
	h = g['hist0']
	tabs[0]['maghists'] = [(h[0][:-1], h[0][1:], h[1][:-1], h[2][:-1])]
	return tabs


def fuzz_cases():
	"""the randomized configurations of tests/golden/fuzz.npz (make_golden.py:gen_fuzz):
	yields (tag, tables, radius, completeness, options, golden)"""
	g = golden('fuzz')
	for seed in range(int(g['nseeds'][0])):
		tag = 'f%d_' % seed
		k = int(g[tag + 'k'][0])
		tabs = [cat('C%d' % i, g[tag + 'ra%d' % i], g[tag + 'dec%d' % i], g[tag + 'err%d' % i], float(g[tag + 'area%d' % i][0])) for i in range(k)]
		comp = g[tag + 'completeness']
		comp = float(comp[0]) if len(comp) == 1 else comp
		ratio, min_prob, unrelated = g[tag + 'opts']
		opts = dict(prob_ratio_secondary=float(ratio), min_prob=float(min_prob), consider_unrelated_associations=bool(unrelated))
		if tag + 'mag' in g.files:
			h = g[tag + 'maghist']
			tabs[-1]['mags'], tabs[-1]['magnames'] = [g[tag + 'mag'].copy()], ['M']
			tabs[-1]['maghists'] = [(h[0][:-1], h[0][1:], h[1][:-1], h[2][:-1])]
		yield tag, tabs, float(g[tag + 'radius'][0]), comp, opts, g


def idx_hash(idx):
	idx = np.asarray(idx).astype(np.int64)
	w = np.arange(1, len(idx) + 1, dtype=np.uint64)
	h = np.uint64(0)
	for c in range(idx.shape[1]):
		h = h + ((idx[:, c] + 2).astype(np.uint64) * np.uint64(1000003 + 7919 * c) * w).sum(dtype=np.uint64)
	return h


FLOATCOLS = ['Separation_max', 'dist_bayesfactor_uncorrected', 'dist_bayesfactor', 'dist_post',
	'p_single', 'prob_has_match', 'prob_this_match']

# Tolerances of the parity contract (BASELINE.json north_star: 1e-6 relative on floating
# columns, match_flag/index columns bit-identical).  The absolute term covers columns that
# are differences of O(1) numbers (p_any = 1 - 10**x) or exactly-zero separations, where a
# relative error is undefined at the 1e-16 rounding level of either implementation.
RTOL = 1e-6
ATOL = 1e-12
# The log Bayes factors are sums of O(10) terms of either sign; one that happens to come out near
# zero (|log_bf| ~ 1e-5 on one row in a million, seen in tools/dev/soak_mid.py) carries the same
# ~1e-11 absolute rounding as its neighbours, which is then no longer 1e-6 of its value.  For these
# logarithmic columns the absolute error is the meaningful one: 1e-9 dex.
ATOL_LOG = 1e-9
LOGCOLS = ('dist_bayesfactor_uncorrected', 'dist_bayesfactor')


def atol_for(column, atol=ATOL):
	"""absolute tolerance of a column under the PRODUCT contract; tighter caller-supplied values
	(oracle against reference) are left alone"""
	return max(atol, ATOL_LOG) if (column in LOGCOLS and atol >= ATOL) else atol


def script_golden():
	"""tests/golden/script_api.npz: the reference's SCRIPT (nway.py, executed by make_script_golden.py) on the inputs of the API
	fixtures -- what ``unrelated_associations='cli', f32_roundtrip=True`` has to reproduce"""
	return golden('script_api')


# a golden column the script only ever holds in float32 (its separations after the trip through the FITS 'E' columns) is stored
# as float32: the value it is compared with may differ from it by the rounding of the STORED number, 6e-8 relative
F32_STORED_RTOL = 1.5e-7


def _rtol_for(golden_array, rtol):
	return max(rtol, F32_STORED_RTOL) if golden_array.dtype == np.float32 else rtol


def assert_script_correction(table, gs, prefix, rtol=RTOL, atol=0.):
	"""rows changed by the script's unrelated-association loop (nway.py:366-420) and by how much: the difference of two runs of
	the script (with and without --ignore-unrelated-associations), both in float64"""
	delta = np.asarray(table['dist_bayesfactor']) - np.asarray(table['dist_bayesfactor_uncorrected'])
	np.testing.assert_array_equal(np.flatnonzero(delta != 0), gs[prefix + 'cli_changed_rows'])
	np.testing.assert_allclose(delta[delta != 0], gs[prefix + 'cli_correction'], rtol=rtol, atol=atol)


def assert_table_matches(table, g, prefix, names, rows=None, rtol=RTOL, atol=ATOL, soak=False):
	"""compare a result table (dict of arrays) with golden arrays stored under ``prefix``: the
	contract of the north star, 1e-6 relative (1e-12 absolute) on every floating column.  ``soak``:
	the wider absolute tolerance of the logarithmic columns (ATOL_LOG) -- for randomized
	oracle-against-HIP runs only, never for a fixture generated by the reference"""
	k = len(names)
	sel = (lambda a: a) if rows is None else (lambda a: np.asarray(a)[rows])
	idx = np.stack([sel(table[n]) for n in names], axis=1)
	np.testing.assert_array_equal(idx, g[prefix + 'idx'])
	np.testing.assert_array_equal(sel(table['ncat']), g[prefix + 'ncat'])
	np.testing.assert_array_equal(sel(table['match_flag']), g[prefix + 'match_flag'])
	for i in range(k):
		for j in range(i + 1, k):
			want = g[prefix + 'sep_%d_%d' % (i, j)]
			np.testing.assert_allclose(sel(table['Separation_%s_%s' % (names[i], names[j])]),
				want.astype(float), rtol=_rtol_for(want, rtol), atol=1e-9, equal_nan=True)
	for c in FLOATCOLS:
		want = g[prefix + c]
		np.testing.assert_allclose(sel(table[c]), want.astype(float), rtol=_rtol_for(want, rtol), atol=(atol_for(c, atol) if soak else atol), err_msg=c)


def assert_checksums_match(table, g, prefix, names, rtol=1e-9):
	k = len(names)
	idx = np.stack([table[n] for n in names], axis=1)
	assert len(idx) == int(g[prefix + 'nrows'][0])
	np.testing.assert_array_equal(np.bincount(idx[:, 0], minlength=len(g[prefix + 'rows_per_primary'])),
		g[prefix + 'rows_per_primary'])
	assert idx_hash(idx) == g[prefix + 'idx_hash'][0]
	np.testing.assert_array_equal(np.bincount(table['match_flag'], minlength=3), g[prefix + 'flag_counts'])
	np.testing.assert_array_equal(np.bincount(table['ncat'], minlength=k + 1), g[prefix + 'ncat_counts'])
	for c in FLOATCOLS:
		np.testing.assert_allclose(np.sum(table[c], dtype=float), g[prefix + 'sum_' + c][0], rtol=rtol, err_msg=c)
	for i in range(k):
		for j in range(i + 1, k):
			np.testing.assert_allclose(np.nansum(table['Separation_%s_%s' % (names[i], names[j])]),
				g[prefix + 'sum_sep_%d_%d' % (i, j)][0], rtol=rtol)


def soak_compare(t, o, names, f32=False):
	"""HIP table against an oracle table in the randomised soaks (tools/dev/soak_*.py) -- the product contract with the two
	allowances that only show up when millions of random rows go by, and only there:
	* a log Bayes factor that happens to come out near zero (|log_bf| < 1e-3) carries the absolute rounding of its neighbours
	  (ATOL_LOG), which is then more than 1e-6 of its value;
	* ``f32`` (the script's numerics, f32_roundtrip): the separations pass through float32 (fastskymatch.py:328), so a float64
	  separation within a rounding error of the midpoint of two float32 values is rounded the other way by the other libm -- one
	  float32 ulp (6e-8) of the separation, ~1e-6 of a posterior, about one row in a million.  Such rows (at most three per
	  million, none beyond 1e-5) are counted and returned, not failed.
	Index columns and ncat stay bit-identical throughout, match_flag too unless two rows of a primary tie in p_i to within rounding
	(checked row by row below).  Returns the number of excused rows."""
	assert len(t['ncat']) == len(o['ncat'])
	for n in names:
		np.testing.assert_array_equal(t[n], o[n])
	np.testing.assert_array_equal(t['ncat'], o['ncat'])
	flags = np.flatnonzero(np.asarray(t['match_flag']) != np.asarray(o['match_flag']))
	if len(flags):
		# identical unless two p_i of one primary are equal to within rounding (the allowance of tests/test_hip_fuzz.py: a tie of the
		# INPUT -- e.g. two sources placed symmetrically about a primary -- decided by the last bit of two separations)
		prim, pi = np.asarray(o[names[0]]), np.asarray(o['prob_this_match'], dtype=float)
		for r in flags:
			mine = np.sort(pi[prim == prim[r]])
			tie = np.isclose(mine, mine[-1], rtol=1e-12).sum() > 1 or np.isclose(mine, 0.5 * mine[-1], rtol=1e-12).any()
			assert tie, 'match_flag differs without a rounding-level tie (row %d)' % r
	excused = np.zeros(len(o['ncat']), dtype=bool)
	near_zero = np.abs(np.asarray(o['dist_bayesfactor'])) < 1e-3
	cols = ['Separation_%s_%s' % (names[i], names[j]) for i in range(len(names)) for j in range(i + 1, len(names))]
	for c in cols + ['Separation_max', 'dist_bayesfactor', 'dist_post', 'p_single', 'prob_has_match', 'prob_this_match']:
		a, b = np.asarray(t[c], dtype=float), np.asarray(o[c], dtype=float)
		atol = 1e-9 if c.startswith('Separation') else ATOL
		tol = atol + RTOL * np.abs(b)
		if c == 'dist_bayesfactor':
			tol = np.where(near_zero, ATOL_LOG + RTOL * np.abs(b), tol)
		bad = ~((np.abs(a - b) <= tol) | (np.isnan(a) & np.isnan(b)))
		if bad.any() and f32:
			mild = bad & (np.abs(a - b) <= atol + 1e-5 * np.abs(b))
			excused |= mild
			bad &= ~mild
		assert not bad.any(), '%s: %d rows beyond tolerance, worst relative %.3g' % (c, bad.sum(), np.max(np.abs(a - b)[bad] / np.maximum(np.abs(b[bad]), 1e-300)))
	assert excused.sum() <= 3 + 3e-6 * len(excused), '%d rows excused as float32 rounding flips: too many' % excused.sum()
	return int(excused.sum())
