#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference (pure Python, v4.7.1) is imported through ``ref_harness`` and
executed on
  * the known-answer inputs of its own unit tests (tests/bayesdistance_test.py:12-32,
    tests/fastskymatch_test.py:16-29),
  * its shipped fixtures tests/elltest/randomcat{X,R,O}.fits (2-way and 3-way),
  * its shipped doc/COSMOS_XMM.fits against seeded uniform stand-ins for the two
    catalogues that are missing from the checkout (.MISSING_LARGE_BLOBS),
  * small adversarial tables (negative declination cells, lone primaries, ties),
  * all-sky tables (both poles, the RA = 0/360 seam) through the reference's HEALPix branch with
    oracle/healpix.py standing in for the absent healpy (see ref_harness.py: pins the branch
    logic, not healpy's pixel numbering).
Outputs are .npz files holding INPUTS and EXPECTED OUTPUTS only (data, no code).
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from ref_harness import load_reference, REFERENCE  # noqa: E402
from nway_amd import _fits  # noqa: E402

warnings.simplefilter('ignore')
ref = load_reference(healpix=True)
LOG = ref.NullOutputLogger()
raw_crossproduct = ref.match.crossproduct.func

FLOATCOLS = ['Separation_max', 'dist_bayesfactor_uncorrected', 'dist_bayesfactor', 'dist_post',
	'p_single', 'prob_has_match', 'prob_this_match']


def save(name, **arrays):
	path = os.path.join(HERE, name + '.npz')
	np.savez_compressed(path, **arrays)
	print('%-14s %8.1f KB  %s' % (name, os.path.getsize(path) / 1024., ' '.join(sorted(arrays))[:100]))


def microdeg(x):
	"""store 6-decimal coordinates exactly as integers (x == i / 1e6 bit for bit)"""
	i = np.round(np.asarray(x, dtype=float) * 1e6).astype(np.int64)
	assert (i / 1e6 == x).all()
	return i


def cat(name, ra, dec, error, area):
	return dict(name=name, ra=np.array(ra, dtype=float), dec=np.array(dec, dtype=float),
		error=np.array(error, dtype=float), area=area, mags=[], maghists=[], magnames=[])


def run(tables, radius, completeness, **kw):
	tables = [dict(t, ra=t['ra'].copy(), dec=t['dec'].copy(), error=t['error'].copy()) for t in tables]
	return ref.nway_match(tables, match_radius=radius, prior_completeness=completeness, logger=LOG, **kw)


def table_arrays(res, names, prefix=''):
	out = {}
	k = len(names)
	out[prefix + 'idx'] = np.stack([res[n].values for n in names], axis=1).astype(np.int32)
	for i in range(k):
		for j in range(i + 1, k):
			out[prefix + 'sep_%d_%d' % (i, j)] = res['Separation_%s_%s' % (names[i], names[j])].values
	for c in FLOATCOLS:
		out[prefix + c] = res[c].values
	out[prefix + 'ncat'] = res['ncat'].values.astype(np.int8)
	out[prefix + 'match_flag'] = res['match_flag'].values.astype(np.int8)
	return out


def checksums(res, names, prefix=''):
	"""order-sensitive integer checksum of the index table + float column sums"""
	out = {}
	idx = np.stack([res[n].values for n in names], axis=1).astype(np.int64)
	w = np.arange(1, len(idx) + 1, dtype=np.uint64)
	h = np.uint64(0)
	for c in range(idx.shape[1]):
		h = h + ((idx[:, c] + 2).astype(np.uint64) * np.uint64(1000003 + 7919 * c) * w).sum(dtype=np.uint64)
	out[prefix + 'idx_hash'] = np.array([h], dtype=np.uint64)
	out[prefix + 'nrows'] = np.array([len(idx)])
	out[prefix + 'rows_per_primary'] = np.bincount(idx[:, 0]).astype(np.int32)
	for c in FLOATCOLS:
		out[prefix + 'sum_' + c] = np.array([res[c].values.sum()])
	k = len(names)
	for i in range(k):
		for j in range(i + 1, k):
			out[prefix + 'sum_sep_%d_%d' % (i, j)] = np.array([np.nansum(res['Separation_%s_%s' % (names[i], names[j])].values)])
	out[prefix + 'flag_counts'] = np.bincount(res['match_flag'].values, minlength=3)
	out[prefix + 'ncat_counts'] = np.bincount(res['ncat'].values, minlength=k + 1)
	return out


def subset_rows(res, names, step, prefix=''):
	"""full rows of every ``step``-th primary"""
	prim = res[names[0]].values
	mask = (prim % step) == 0
	sub = res[mask]
	out = table_arrays(sub, names, prefix + 'sub_')
	out[prefix + 'sub_step'] = np.array([step])
	out[prefix + 'sub_rows'] = np.flatnonzero(mask).astype(np.int64)
	return out


# ---------------------------------------------------------------------------
def gen_kat_math():
	bd = ref.bayesdist
	sep = np.array([0., 0.1, 0.2, 0.3, 0.4, 0.5])
	out = dict(sep=sep)
	out['log_bf2'] = np.array([bd.log_bf2(p, 0.1, 0.2) for p in sep])
	out['log_bf_n2'] = np.array([bd.log_bf([[None, p]], [0.1, 0.2]) for p in sep])
	out['log_bf3'] = np.array([bd.log_bf3(p, p, p, 0.1, 0.2, 0.3) for p in sep])
	out['log_bf_n3'] = np.array([bd.log_bf([[None, p, p], [p, None, p], [p, p, None]], [0.1, 0.2, 0.3]) for p in sep])
	out['log_bf_n1'] = np.atleast_1d(bd.log_bf([[None]], [0.5]))
	out['log_arcsec2rad'] = np.array([bd.log_arcsec2rad])
	# vectorised 4-way call with distinct separations/sigmas
	rng = np.random.RandomState(7)
	n = 257
	s4 = [rng.uniform(0.05, 3, size=n) for _ in range(4)]
	p4 = [[rng.uniform(0, 5, size=n) if i < j else np.nan * np.ones(n) for j in range(4)] for i in range(4)]
	out['n4_sigma'] = np.array(s4)
	out['n4_sep'] = np.array([[p4[i][j] for j in range(4)] for i in range(4)])
	out['n4_log_bf'] = bd.log_bf(p4, s4)
	prior = 10**rng.uniform(-12, -0.01, size=n)
	lbf = rng.uniform(-30, 30, size=n)
	out['post_prior'] = prior
	out['post_logbf'] = lbf
	out['posterior'] = bd.posterior(prior, lbf)
	out['log_posterior'] = bd.log_posterior(prior, lbf)
	out['unnormalised_log_posterior'] = bd.unnormalised_log_posterior(prior, lbf, 2)
	out['posterior_scalar'] = np.array([bd.posterior(1e-3, 2.5), bd.log_posterior(1e-3, 2.5)])
	# dist: inputs of tests/fastskymatch_test.py:16-29
	out['dist_scalar'] = np.array([ref.match.dist((53.15964508, -27.92927742), (53.15953445, -27.9313736))])
	ra = np.array([53.14784241, 53.14784241, 53.14749908, 53.16559982, 53.19423676, 53.1336441])
	dec = np.array([-27.79363823, -27.79363823, -27.81790352, -27.79622459, -27.70860672, -27.76327515])
	ra2 = np.array([53.14907837, 53.14907837, 53.1498642, 53.16150284, 53.19681549, 53.13626862])
	dec2 = np.array([-27.79297447, -27.79297447, -27.81404877, -27.79223251, -27.71365929, -27.76314735])
	out['dist_ra'], out['dist_dec'], out['dist_ra2'], out['dist_dec2'] = ra, dec, ra2, dec2
	out['dist_array'] = ref.match.dist((ra, dec), (ra2, dec2))
	# dist over the whole sphere incl. poles, antipodes, identical points, RA wrap
	n = 4096
	a_ra = rng.uniform(0, 360, size=n)
	a_dec = np.degrees(np.arcsin(rng.uniform(-1, 1, size=n)))
	b_ra = a_ra + rng.normal(0, 1, size=n) * 10**rng.uniform(-7, 2, size=n)
	b_dec = np.clip(a_dec + rng.normal(0, 1, size=n) * 10**rng.uniform(-7, 2, size=n), -90, 90)
	a_dec[:4] = [90, -90, 0, 45]
	b_dec[:4] = [-90, -90, 0, 45]
	b_ra[2:4] = a_ra[2:4]
	b_ra[4] = a_ra[4] + 360
	out['sph_a_ra'], out['sph_a_dec'], out['sph_b_ra'], out['sph_b_dec'] = a_ra, a_dec, b_ra, b_dec
	out['sph_dist'] = ref.match.dist((a_ra, a_dec), (b_ra, b_dec))
	save('kat_math', **out)


def load_elltest():
	X = _fits.read_table(os.path.join(REFERENCE, 'tests/elltest/randomcatX.fits'))
	R = _fits.read_table(os.path.join(REFERENCE, 'tests/elltest/randomcatR.fits'))
	O = _fits.read_table(os.path.join(REFERENCE, 'tests/elltest/randomcatO.fits'))
	return X, R, O


def gen_ell():
	X, R, O = load_elltest()
	inputs = dict(
		X_ra_u=microdeg(X.data['RA']), X_dec_u=microdeg(X.data['DEC']), X_err_u=microdeg(X.data['pos_err']),
		R_ra_u=microdeg(R.data['RA']), R_dec_u=microdeg(R.data['DEC']), R_err_u=microdeg(R.data['pos_err']),
		O_ra_u=microdeg(O.data['RA']).astype(np.int32), O_dec_u=microdeg(O.data['DEC']).astype(np.int32),
		area=np.array([X.header['SKYAREA'], R.header['SKYAREA'], O.header['SKYAREA']]),
		names=np.array([X.name, R.name, O.name]),
	)
	save('ell_inputs', **inputs)
	tX = cat(X.name, X.data['RA'], X.data['DEC'], X.data['pos_err'], X.header['SKYAREA'])
	tR = cat(R.name, R.data['RA'], R.data['DEC'], R.data['pos_err'], R.header['SKYAREA'])
	tO = cat(O.name, O.data['RA'], O.data['DEC'], 0.1 * np.ones(len(O.data)), O.header['SKYAREA'])
	# --- 2-way
	cp = raw_crossproduct([(tX['ra'], tX['dec']), (tO['ra'], tO['dec'])], 10. / 60 / 60, LOG)
	out = dict(crossproduct=cp.astype(np.int32), radius=np.array([10.]))
	res = run([tX, tO], 10., 1.0)
	names = [X.name, O.name]
	out.update(table_arrays(res, names, 'c10_'))
	out.update(checksums(res, names, 'c10_'))
	res9 = run([tX, tO], 10., 0.9)
	out.update(checksums(res9, names, 'c09_'))
	f1 = res9['match_flag'].values == 1
	out.update(dict(('c09_best_' + c, res9[c].values[f1]) for c in FLOATCOLS))
	out['c09_match_flag'] = res9['match_flag'].values.astype(np.int8)
	resm = run([tX, tO], 10., 0.9, min_prob=0.01, prob_ratio_secondary=0.25)
	out.update(table_arrays(resm, names, 'trunc_'))
	save('ell2', **out)
	# --- 3-way
	cp3 = raw_crossproduct([(tX['ra'], tX['dec']), (tR['ra'], tR['dec']), (tO['ra'], tO['dec'])], 10. / 60 / 60, LOG)
	names3 = [X.name, R.name, O.name]
	w = np.arange(1, len(cp3) + 1, dtype=np.uint64)
	h = np.uint64(0)
	for c in range(3):
		h = h + ((cp3[:, c] + 2).astype(np.uint64) * np.uint64(1000003 + 7919 * c) * w).sum(dtype=np.uint64)
	out = dict(crossproduct_nrows=np.array([len(cp3)]), crossproduct_hash=np.array([h], dtype=np.uint64),
		crossproduct_rows_per_primary=np.bincount(cp3[:, 0]).astype(np.int32), radius=np.array([10.]))
	res3 = run([tX, tR, tO], 10., 1.0)
	out.update(checksums(res3, names3, 'c10_'))
	out.update(subset_rows(res3, names3, 13, 'c10_'))
	# (the script's unrelated-association correction on these files: make_script_golden.py, which executes nway.py itself)
	save('ell3', **out)


def gen_xmm():
	XMM = _fits.read_table(os.path.join(REFERENCE, 'doc/COSMOS_XMM.fits'))
	d = XMM.data
	inputs = dict(ID=d['ID'], RA=d['RA'], DEC=d['DEC'], pos_err=d['pos_err'], area=np.array([XMM.header['SKYAREA']]),
		table_name=np.array([XMM.name]))
	save('xmm_inputs', **inputs)
	# seeded stand-ins for the missing COSMOS_OPTICAL / COSMOS_IRAC (same sizes as the real ones)
	rng = np.random.RandomState(42)
	n_opt, n_irac = 560536, 345512
	opt_ra = rng.uniform(149.35, 150.87, size=n_opt)
	opt_dec = rng.uniform(1.47, 2.96, size=n_opt)
	irac_ra = rng.uniform(149.35, 150.87, size=n_irac)
	irac_dec = rng.uniform(1.47, 2.96, size=n_irac)
	tX = cat('XMM', d['RA'], d['DEC'], d['pos_err'].astype(float), 2.0)
	tO = cat('OPT', opt_ra, opt_dec, 0.1 * np.ones(n_opt), 2.0)
	tI = cat('IRAC', irac_ra, irac_dec, 0.5 * np.ones(n_irac), 2.0)
	res = run([tX, tO], 20., 0.9)
	names = ['XMM', 'OPT']
	out = dict(seed=np.array([42]), n_opt=np.array([n_opt]), n_irac=np.array([n_irac]),
		box=np.array([149.35, 150.87, 1.47, 2.96]), radius=np.array([20.]), completeness=np.array([0.9]))
	out.update(table_arrays(res, names, 'w2_'))
	out.update(checksums(res, names, 'w2_'))
	res3 = run([tX, tO, tI], 20., 0.9)
	names3 = ['XMM', 'OPT', 'IRAC']
	out.update(checksums(res3, names3, 'w3_'))
	out.update(subset_rows(res3, names3, 29, 'w3_'))
	save('xmm_syn', **out)


def gen_edge():
	out = {}
	# (1) cells straddling Dec = 0 (int() truncation toward zero), 3-way, with lone primaries
	rng = np.random.RandomState(3)
	n0, n1, n2 = 60, 900, 700
	r = 30.
	def box(n):
		return rng.uniform(20.0, 20.12, size=n), rng.uniform(-0.06, 0.06, size=n)
	a, b, c = box(n0), box(n1), box(n2)
	a[0][:3] = [25., 26., 27.]  # three primaries far away from everything: groups of one row
	t0 = cat('A', a[0], a[1], rng.uniform(1, 4, size=n0), 0.0144)
	t1 = cat('B', b[0], b[1], rng.uniform(0.5, 2, size=n1), 0.0144)
	t2 = cat('C', c[0], c[1], 1.0 * np.ones(n2), 0.0144)
	cp = raw_crossproduct([(t0['ra'], t0['dec']), (t1['ra'], t1['dec']), (t2['ra'], t2['dec'])], r / 60 / 60, LOG)
	res = run([t0, t1, t2], r, np.array([1.0, 0.8, 0.6]))
	names = ['A', 'B', 'C']
	for i, t in enumerate((t0, t1, t2)):
		out['neg_ra%d' % i], out['neg_dec%d' % i], out['neg_err%d' % i] = t['ra'], t['dec'], t['error']
	out['neg_area'] = np.array([0.0144])
	out['neg_radius'] = np.array([r])
	out['neg_completeness'] = np.array([1.0, 0.8, 0.6])
	out['neg_crossproduct'] = cp.astype(np.int32)
	out.update(table_arrays(res, names, 'neg_'))
	# (2) exact ties: two secondaries mirrored in RA about the primary (bit-equal separations are
	#     not guaranteed; whatever the reference says is the expected answer) + a duplicate secondary
	p_ra, p_dec = np.array([100.0, 100.5]), np.array([10.0, 10.0])
	s_ra = np.array([100.0 + 1e-4, 100.0 - 1e-4, 100.5 + 2e-4, 100.5 + 2e-4, 100.5 - 3e-4])
	s_dec = np.array([10.0, 10.0, 10.0, 10.0, 10.0 + 1e-4])
	tp = cat('P', p_ra, p_dec, np.array([0.5, 0.7]), 1.0)
	ts = cat('S', s_ra, s_dec, 0.2 * np.ones(5), 1.0)
	res = run([tp, ts], 5., 0.95)
	out['tie_p_ra'], out['tie_p_dec'], out['tie_p_err'] = tp['ra'], tp['dec'], tp['error']
	out['tie_s_ra'], out['tie_s_dec'], out['tie_s_err'] = ts['ra'], ts['dec'], ts['error']
	out['tie_radius'] = np.array([5.]); out['tie_completeness'] = np.array([0.95])
	out.update(table_arrays(res, ['P', 'S'], 'tie_'))
	# (3) a hopeless single candidate: p_i = 1 although log_bf is hugely negative
	tp = cat('P', [50.0], [-30.0], [0.01], 1.0)
	ts = cat('S', [50.0 + 1.2e-3], [-30.0], [0.01], 1.0)
	res = run([tp, ts], 10., 0.9)
	out['hop_p'] = np.array([50.0, -30.0, 0.01]); out['hop_s'] = np.array([50.0 + 1.2e-3, -30.0, 0.01])
	out['hop_radius'] = np.array([10.]); out['hop_completeness'] = np.array([0.9])
	out.update(table_arrays(res, ['P', 'S'], 'hop_'))
	# (4) 4-way, small, for the generic k expansion
	rng = np.random.RandomState(11)
	tabs = []
	for i, n in enumerate((25, 160, 140, 120)):
		ra = rng.uniform(200.0, 200.05, size=n)
		dec = rng.uniform(30.0, 30.05, size=n)
		tabs.append(cat('T%d' % i, ra, dec, rng.uniform(0.5, 3., size=n), 0.0025))
		out['k4_ra%d' % i], out['k4_dec%d' % i], out['k4_err%d' % i] = tabs[-1]['ra'], tabs[-1]['dec'], tabs[-1]['error']
	cp = raw_crossproduct([(t['ra'], t['dec']) for t in tabs], 25. / 60 / 60, LOG)
	res = run(tabs, 25., 0.7)
	out['k4_area'] = np.array([0.0025]); out['k4_radius'] = np.array([25.]); out['k4_completeness'] = np.array([0.7])
	out['k4_crossproduct_nrows'] = np.array([len(cp)])
	out['k4_crossproduct_rows_per_primary'] = np.bincount(cp[:, 0], minlength=25).astype(np.int32)
	out.update(table_arrays(res, ['T0', 'T1', 'T2', 'T3'], 'k4_'))
	save('edge', **out)


if __name__ == '__main__':
	which = sys.argv[1:] or ['kat', 'ell', 'xmm', 'edge']
	if 'kat' in which:
		gen_kat_math()
	if 'edge' in which:
		gen_edge()
	if 'ell' in which:
		gen_ell()
	if 'xmm' in which:
		gen_xmm()


def gen_mag():
	"""magnitude priors: XMM x seeded OPT stand-in with a seeded magnitude column in which true
	counterparts are brighter; auto histograms by radius, by posterior, and a supplied histogram"""
	XMM = _fits.read_table(os.path.join(REFERENCE, 'doc/COSMOS_XMM.fits'))
	d = XMM.data
	rng = np.random.RandomState(77)
	n_opt = 120000
	opt_ra = rng.uniform(149.35, 150.87, size=n_opt)
	opt_dec = rng.uniform(1.47, 2.96, size=n_opt)
	mag = rng.normal(24.0, 1.5, size=n_opt)
	# give 1500 XMM sources a bright counterpart close by
	k = 1500
	slots = rng.choice(n_opt, size=k, replace=False)
	opt_ra[slots] = d['RA'][:k] + rng.normal(0, 0.5, size=k) / 3600. / np.cos(np.radians(d['DEC'][:k]))
	opt_dec[slots] = d['DEC'][:k] + rng.normal(0, 0.5, size=k) / 3600.
	mag[slots] = rng.normal(21.0, 1.0, size=k)
	mag[rng.choice(n_opt, size=3000, replace=False)] = -99
	mag[rng.choice(n_opt, size=500, replace=False)] = np.nan
	# the catalogue is regenerated from the seed by tests/goldenutil.py:mag_tables (same recipe);
	# only checksums are stored
	out = dict(seed=np.array([77]), n_opt=np.array([n_opt]), n_true=np.array([k]),
		checksum=np.array([opt_ra.sum(), opt_dec.sum(), np.nansum(mag), float(np.isnan(mag).sum())]))

	def tables(maghist):
		tX = cat('XMM', d['RA'], d['DEC'], d['pos_err'].astype(float), 2.0)
		tO = cat('OPT', opt_ra, opt_dec, 0.1 * np.ones(n_opt), 2.0)
		tO['mags'] = [mag.copy()]
		tO['magnames'] = ['MAG']
		tO['maghists'] = [maghist]
		return [tX, tO]
	cwd = os.getcwd()
	import tempfile
	os.chdir(tempfile.mkdtemp(prefix='nwaymag_'))
	try:
		for prefix, kw in (('rad_', dict(mag_include_radius=4.0)), ('post_', dict())):
			res = ref.nway_match(tables(None), match_radius=20., prior_completeness=0.9, store_mag_hists=False, logger=LOG, **kw)
			out.update(table_arrays(res, ['XMM', 'OPT'], prefix))
			out[prefix + 'bias'] = res['bias_OPT_MAG'].values
		# histogram file written by the reference + the run that consumes it
		res = ref.nway_match(tables(None), match_radius=20., prior_completeness=0.9, store_mag_hists=True, mag_include_radius=4.0, logger=LOG)
		out['hist_text'] = np.frombuffer(open('OPT_MAG_fit.txt', 'rb').read(), dtype=np.uint8)
		lo, hi, hs, ha = np.loadtxt('OPT_MAG_fit.txt').transpose()
		res = ref.nway_match(tables((lo, hi, hs, ha)), match_radius=20., prior_completeness=0.9, store_mag_hists=False, logger=LOG)
		out.update(table_arrays(res, ['XMM', 'OPT'], 'file_'))
		out['file_bias'] = res['bias_OPT_MAG'].values
	finally:
		os.chdir(cwd)
	# the interpolants the reference builds with scipy, on fixed inputs
	mw = ref.magnitudeweights
	a = rng.normal(22, 2, size=4000)
	s = rng.normal(20, 1, size=300)
	w = rng.uniform(0.2, 1, size=300)
	bins, hs, ha = mw.adaptive_histograms(a, s, weights=w)
	out['ah_all'], out['ah_sel'], out['ah_w'] = a, s, w
	out['ah_bins'], out['ah_hist_sel'], out['ah_hist_all'] = bins, hs, ha
	x = np.r_[np.linspace(bins[0] - 1, bins[-1] + 1, 501), bins, np.nan]
	out['ff_x'] = x
	out['ff_y'] = mw.fitfunc_histogram(bins, hs, ha)(x)
	save('mag', **out)


def gen_ellmath():
	"""elliptical-error helpers of bayesdistance.py:92-240 on seeded inputs"""
	bd = ref.bayesdist
	rng = np.random.RandomState(5)
	n = 300
	a, b = rng.uniform(0.1, 5, size=n), rng.uniform(0.1, 5, size=n)
	phi = rng.uniform(0, np.pi, size=n)
	sx, sy, rho = bd.convert_from_ellipse(a, b, phi)
	out = dict(a=a, b=b, phi=phi, sigma_x=sx, sigma_y=sy, rho=rho)
	inv = bd.make_invcovmatrix(sx, sy, rho)
	cov = bd.make_covmatrix(sx, sy, rho)
	out['inv'] = np.array(inv)
	out['cov'] = np.array(cov)
	v = rng.normal(0, 1, size=(2, n))
	a2, b2, phi2 = rng.uniform(0.1, 5, size=n), rng.uniform(0.1, 5, size=n), rng.uniform(0, np.pi, size=n)
	e2 = bd.convert_from_ellipse(a2, b2, phi2)
	out['v'] = v
	out['a2'], out['b2'], out['phi2'] = a2, b2, phi2
	out['vABv'] = bd.apply_vABv(v, inv, bd.make_invcovmatrix(*e2))
	a3, b3, phi3 = rng.uniform(0.1, 5, size=n), rng.uniform(0.1, 5, size=n), rng.uniform(0, np.pi, size=n)
	e3 = bd.convert_from_ellipse(a3, b3, phi3)
	out['a3'], out['b3'], out['phi3'] = a3, b3, phi3
	dra = rng.normal(0, 2, size=(3, n))
	ddec = rng.normal(0, 2, size=(3, n))
	out['dra'], out['ddec'] = dra, ddec
	nan = np.nan * np.ones(n)
	sra = [[nan, dra[0], dra[1]], [nan, nan, dra[2]], [nan, nan, nan]]
	sdec = [[nan, ddec[0], ddec[1]], [nan, nan, ddec[2]], [nan, nan, nan]]
	out['log_bf_ell3'] = bd.log_bf_elliptical(sra, sdec, [(sx, sy, rho), e2, e3])
	out['log_bf_ell2'] = bd.log_bf_elliptical([[nan, dra[0]], [nan, nan]], [[nan, ddec[0]], [nan, nan]], [(sx, sy, rho), e2])
	save('ellmath', **out)


if __name__ == '__main__' and ('mag' in sys.argv[1:] or not sys.argv[1:]):
	gen_mag()
if __name__ == '__main__' and ('ellmath' in sys.argv[1:] or not sys.argv[1:]):
	gen_ellmath()


def _scatter(rng, ra, dec, sigma_arcsec):
	"""positions displaced by an isotropic gaussian of the given width; valid at the poles"""
	lon, lat = np.radians(ra), np.radians(dec)
	v = np.stack([np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)], axis=1)
	v = v + rng.normal(size=v.shape) * np.radians(sigma_arcsec / 3600.)
	v /= np.sqrt((v ** 2).sum(axis=1))[:, None]
	return np.degrees(np.arctan2(v[:, 1], v[:, 0])) % 360., np.degrees(np.arcsin(np.clip(v[:, 2], -1, 1)))


def gen_allsky():
	"""3-way and 2-way all-sky matches; takes the HEALPix branch (fastskymatch.py:99-160)"""
	rng = np.random.RandomState(77)
	radius = 20.
	npole, nseam, nfree = 300, 300, 600
	ra = np.concatenate([rng.uniform(0, 360, npole), rng.uniform(0, 360, npole),
		np.mod(rng.normal(0, 0.02, nseam), 360.), rng.uniform(0, 360, nfree)])
	dec = np.concatenate([90 - np.abs(rng.normal(0, 0.5, npole)), -90 + np.abs(rng.normal(0, 0.5, npole)),
		rng.uniform(-60, 60, nseam), np.degrees(np.arcsin(rng.uniform(-1, 1, nfree)))])
	ra[0], dec[0] = 0.0, 90.0       # exactly on the pole
	ra[npole], dec[npole] = 123.0, -90.0
	ra[2 * npole] = 0.0             # exactly on the seam
	ra[2 * npole + 1] = 359.9999999
	n0 = len(ra)
	def secondary(frac, sigma, nfield, nrandom):
		has = np.flatnonzero(rng.uniform(size=n0) < frac)
		parts = [_scatter(rng, ra[has], dec[has], sigma)]
		for _ in range(nfield):
			parts.append(_scatter(rng, ra, dec, 25.))
		parts.append((rng.uniform(0, 360, nrandom), np.degrees(np.arcsin(rng.uniform(-1, 1, nrandom)))))
		r, d = np.concatenate([q[0] for q in parts]), np.concatenate([q[1] for q in parts])
		order = rng.permutation(len(r))
		return r[order], d[order]
	b_ra, b_dec = secondary(0.8, 3., 6, 3000)
	c_ra, c_dec = secondary(0.6, 2., 4, 2000)
	area = 41252.96
	ta = cat('A', ra, dec, rng.uniform(1, 4, size=n0), area)
	tb = cat('B', b_ra, b_dec, 0.5 * np.ones(len(b_ra)), area)
	tc = cat('C', c_ra, c_dec, 1.0 * np.ones(len(c_ra)), area)
	out = dict(radius=np.array([radius]), completeness=np.array([0.9]), area=np.array([area]))
	for i, t in enumerate((ta, tb, tc)):
		out['ra%d' % i], out['dec%d' % i], out['err%d' % i] = t['ra'], t['dec'], t['error']
	cp = raw_crossproduct([(t['ra'], t['dec']) for t in (ta, tb)], radius / 60 / 60, LOG)
	out['w2_crossproduct_nrows'] = np.array([len(cp)])
	res = run([ta, tb], radius, 0.9)
	out.update(table_arrays(res, ['A', 'B'], 'w2_'))
	cp = raw_crossproduct([(t['ra'], t['dec']) for t in (ta, tb, tc)], radius / 60 / 60, LOG)
	out['w3_crossproduct_nrows'] = np.array([len(cp)])
	res = run([ta, tb, tc], radius, 0.9)
	out.update(table_arrays(res, ['A', 'B', 'C'], 'w3_'))
	print('allsky: 2-way %d rows, 3-way %d rows (pre-filter %d)' % (len(out['w2_idx']), len(out['w3_idx']), len(cp)))
	save('allsky', **out)


if __name__ == '__main__' and ('allsky' in sys.argv[1:] or not sys.argv[1:]):
	gen_allsky()


def mag3_catalogues(xmm_ra, xmm_dec, seed, n_opt, n_irac, k_opt, k_irac):
	"""seeded OPT / IRAC stand-ins with magnitude columns (two in OPT, one in IRAC) in which the
	true counterparts of the first XMM sources are brighter.  tests/goldenutil.py:mag3_tables
	repeats this recipe; keep them identical."""
	rng = np.random.RandomState(seed)
	cols = {}
	for name, n, k, sig, m0 in (('OPT', n_opt, k_opt, 0.3, 24.0), ('IRAC', n_irac, k_irac, 0.6, 22.5)):
		ra = rng.uniform(149.35, 150.87, size=n)
		dec = rng.uniform(1.47, 2.96, size=n)
		mag = rng.normal(m0, 1.5, size=n)
		slots = rng.choice(n, size=k, replace=False)
		ra[slots] = xmm_ra[:k] + rng.normal(0, sig, size=k) / 3600. / np.cos(np.radians(xmm_dec[:k]))
		dec[slots] = xmm_dec[:k] + rng.normal(0, sig, size=k) / 3600.
		mag[slots] = rng.normal(m0 - 3.0, 1.0, size=k)
		mag[rng.choice(n, size=n // 50, replace=False)] = -99
		mag[rng.choice(n, size=n // 300, replace=False)] = np.nan
		cols[name] = [ra, dec, mag]
		if name == 'OPT':
			colour = mag + rng.normal(0.0, 0.7, size=n)  # a second, correlated band
			colour[slots] -= 0.8
			colour[~np.isfinite(mag) | (mag == -99)] = -99
			cols[name].append(colour)
	return cols


def gen_mag3():
	"""BASELINE configs[1] in shape: XMM x OPT x IRAC with magnitude priors learned from the data
	(posterior-selected, the default), three magnitude columns over two catalogues"""
	XMM = _fits.read_table(os.path.join(REFERENCE, 'doc/COSMOS_XMM.fits'))
	d = XMM.data
	seed, n_opt, n_irac, k_opt, k_irac = 91, 150000, 90000, 1500, 1300
	cols = mag3_catalogues(d['RA'], d['DEC'], seed, n_opt, n_irac, k_opt, k_irac)
	tX = cat('XMM', d['RA'], d['DEC'], d['pos_err'].astype(float), 2.0)
	tO = cat('OPT', cols['OPT'][0], cols['OPT'][1], 0.1 * np.ones(n_opt), 2.0)
	tO['mags'], tO['magnames'], tO['maghists'] = [cols['OPT'][2].copy(), cols['OPT'][3].copy()], ['R', 'I'], [None, None]
	tI = cat('IRAC', cols['IRAC'][0], cols['IRAC'][1], 0.5 * np.ones(n_irac), 2.0)
	tI['mags'], tI['magnames'], tI['maghists'] = [cols['IRAC'][2].copy()], ['CH1'], [None]
	out = dict(seed=np.array([seed]), sizes=np.array([n_opt, n_irac, k_opt, k_irac]),
		checksum=np.array([cols['OPT'][0].sum(), cols['OPT'][1].sum(), np.nansum(cols['OPT'][2]), np.nansum(cols['OPT'][3]),
			cols['IRAC'][0].sum(), cols['IRAC'][1].sum(), np.nansum(cols['IRAC'][2])]))
	names = ['XMM', 'OPT', 'IRAC']
	biases = ['bias_OPT_R', 'bias_OPT_I', 'bias_IRAC_CH1']
	cwd = os.getcwd()
	import tempfile
	os.chdir(tempfile.mkdtemp(prefix='nwaymag3_'))
	try:
		res = ref.nway_match([tX, tO, tI], match_radius=20., prior_completeness=0.9, store_mag_hists=False, logger=LOG)
	finally:
		os.chdir(cwd)
	out.update(checksums(res, names, 'm3_'))
	out.update(subset_rows(res, names, 23, 'm3_'))
	out['m3_columns'] = np.array(list(res.columns))            # the frame's layout: names in order,
	out['m3_dtypes'] = np.array([str(res[c].dtype) for c in res.columns])  # and their dtypes
	mask = (res['XMM'].values % 23) == 0
	for b in biases:
		v = res[b].values
		out['m3_sum_' + b] = np.array([v.sum()])
		out['m3_sub_' + b] = v[mask]
	# adaptive_histograms with weights that are exactly zero (repeated knots in the cumulative
	# weight axis): which knot supplies a quantile is scipy's / numpy.interp's choice
	rng = np.random.RandomState(5)
	a = rng.normal(22, 2, size=3000)
	s = rng.normal(20, 1, size=200)
	w = rng.randint(1, 65, size=200) / 64.  # sums are exact: the repeated knots are bit-equal
	order = np.argsort(s)
	w[order[-3:]] = 0.0   # the brightest-weight tail carries no weight: the last knots repeat
	w[order[:2]] = 0.0    # ... and the first ones
	w[order[90:95]] = 0.0  # ... and a run in the middle
	bins, hs, ha = ref.magnitudeweights.adaptive_histograms(a, s, weights=w)
	out['ah0_all'], out['ah0_sel'], out['ah0_w'] = a, s, w
	out['ah0_bins'], out['ah0_hist_sel'], out['ah0_hist_all'] = bins, hs, ha
	# ... and the IRAC column's own selection from the run above: its running weight sum exceeds the
	# total by an ulp just before the last knot, i.e. the knots interp1d receives are not sorted
	idx = res['IRAC'].values
	defined = idx != -1
	post = res['dist_post'].values
	rows, first = np.unique(idx[(post > 0.9) & defined], return_index=True)
	wsel = post[defined][first]
	msel = tI['mags'][0][rows]
	ok = np.isfinite(msel)
	a1 = tI['mags'][0][np.isfinite(tI['mags'][0])][::150]
	c1 = np.cumsum(wsel[ok][np.argsort(msel[ok])]) / np.sum(wsel[ok])
	assert (np.diff(np.r_[c1[:-1], 1.0]) < 0).any(), 'the fixture no longer has unsorted knots'
	bins, hs, ha = ref.magnitudeweights.adaptive_histograms(a1, msel[ok], weights=wsel[ok])
	out['ah1_all'], out['ah1_sel'], out['ah1_w'] = a1, msel[ok], wsel[ok]
	out['ah1_bins'], out['ah1_hist_sel'], out['ah1_hist_all'] = bins, hs, ha
	print('mag3: %d rows, flags %s, bias sums %s' % (len(res), np.bincount(res['match_flag'].values), [float(out['m3_sum_' + b][0]) for b in biases]))
	save('mag3', **out)


if __name__ == '__main__' and ('mag3' in sys.argv[1:] or not sys.argv[1:]):
	gen_mag3()


def gen_kway():
	"""4- and 5-way matches on small clustered tables: the generic-k enumeration, the 2^(k-1)
	presence patterns of the Bayes factor / prior, and the script's unrelated-association correction
	for k > 3 (rows with ncat <= k - 2 augmented by two or more of their missing catalogues)"""
	out = {}
	for tag, sizes, radius, comp, seed in (('k4c', (25, 160, 140, 120), 25., 0.7, 11), ('k5', (12, 25, 22, 20, 18), 20., np.array([1.0, 0.9, 0.8, 0.7, 0.6]), 19),
			('k6', (8, 12, 11, 10, 9, 9), 20., 0.8, 23), ('k8', (5, 7, 7, 6, 6, 6, 5, 5), 18., np.array([1.0, 0.95, 0.9, 0.85, 0.8, 0.75, 0.7, 0.65]), 29)):
		rng = np.random.RandomState(seed)
		span = 0.05 if tag == 'k4c' else 0.02
		tabs = []
		for i, n in enumerate(sizes):
			ra = rng.uniform(200.0, 200.0 + span, size=n)
			dec = rng.uniform(30.0, 30.0 + span, size=n)
			tabs.append(cat('T%d' % i, ra, dec, rng.uniform(0.5, 3., size=n), span**2))
			out['%s_ra%d' % (tag, i)], out['%s_dec%d' % (tag, i)], out['%s_err%d' % (tag, i)] = tabs[-1]['ra'], tabs[-1]['dec'], tabs[-1]['error']
		names = [t['name'] for t in tabs]
		res = run(tabs, radius, comp)
		out[tag + '_area'] = np.array([span**2]); out[tag + '_radius'] = np.array([radius]); out[tag + '_completeness'] = np.atleast_1d(comp)
		out.update(table_arrays(res, names, tag + '_'))
		print('%s: %d rows, ncat %s' % (tag, len(res), np.bincount(res['ncat'].values)))
	save('kway', **out)


if __name__ == '__main__' and ('kway' in sys.argv[1:] or not sys.argv[1:]):
	gen_kway()


def gen_kmulti():
	"""5- and 6-way matches with SEVERAL LINKS PER CATALOGUE for a few hundred primaries -- the case the one-lane tuple walk
	of the HIP path got wrong for three rounds (two links in each of two or more catalogues with k >= 5) while the tiny
	kway tables passed.  Primaries on a lattice 72 arcsec apart (radius 10 arcsec: groups never touch); every secondary
	catalogue gives a primary 0, 1 or 2 sources (probabilities 0.25 / 0.35 / 0.4) scattered by N(0, 2.5 arcsec) per axis, so that
	some pairs of secondaries exceed the radius; plus a sprinkle of field sources.  The reference's nway_match runs these
	in seconds (nwaylib/__init__.py:31-120)."""
	out = {}
	for tag, k, nprim, radius, comp, seed in (('m5', 5, 320, 10., 0.85, 101), ('m6', 6, 180, 10., np.array([1.0, 0.95, 0.9, 0.85, 0.8, 0.75]), 102)):
		rng = np.random.RandomState(seed)
		side = int(np.ceil(np.sqrt(nprim)))
		gx, gy = np.meshgrid(np.arange(side), np.arange(side))
		pdec = 20.0 + gy.ravel()[:nprim] * 0.02
		pra = 130.0 + gx.ravel()[:nprim] * 0.02 / np.cos(np.radians(20.3))
		span = side * 0.02
		area = (span + 0.02)**2
		tabs = [cat('T0', pra, pdec, rng.uniform(0.8, 2.0, size=nprim), area)]
		multi = np.zeros(nprim, dtype=int)
		for c in range(1, k):
			ra, dec = [], []
			for i in range(nprim):
				m = rng.choice(3, p=[0.25, 0.35, 0.4])
				multi[i] += (m == 2)
				for _ in range(m):
					dec.append(pdec[i] + rng.normal(0, 2.5) / 3600.)
					ra.append(pra[i] + rng.normal(0, 2.5) / 3600. / np.cos(np.radians(pdec[i])))
			nfield = nprim // 2
			ra += list(rng.uniform(pra.min() - 0.005, pra.max() + 0.005, size=nfield))
			dec += list(rng.uniform(pdec.min() - 0.005, pdec.max() + 0.005, size=nfield))
			order = rng.permutation(len(ra))  # (catalogue order is not primary order)
			tabs.append(cat('T%d' % c, np.array(ra)[order], np.array(dec)[order], rng.uniform(0.2, 1.2, size=len(ra)), area))
		for i, t in enumerate(tabs):
			out['%s_ra%d' % (tag, i)], out['%s_dec%d' % (tag, i)], out['%s_err%d' % (tag, i)] = t['ra'], t['dec'], t['error']
		names = [t['name'] for t in tabs]
		res = run(tabs, radius, comp)
		out[tag + '_area'] = np.array([area]); out[tag + '_radius'] = np.array([radius]); out[tag + '_completeness'] = np.atleast_1d(comp)
		out.update(table_arrays(res, names, tag + '_'))
		groups = np.bincount(res[names[0]].values)
		print('%s: %d rows, ncat %s, largest group %d, primaries with two sources in >= 2 catalogues: %d' % (
			tag, len(res), np.bincount(res['ncat'].values), groups.max(), (multi >= 2).sum()))
		assert (multi >= 2).sum() >= 100
	save('kmulti', **out)


if __name__ == '__main__' and ('kmulti' in sys.argv[1:] or not sys.argv[1:]):
	gen_kmulti()


def fuzz_case(seed):
	"""one small randomized configuration (inputs + the options of nway_match)"""
	rng = np.random.RandomState(1000 + seed)
	k = int(rng.choice([2, 2, 3, 3, 4]))
	where = ['flat', 'flat', 'flat', 'north', 'seam', 'south', 'high'][seed % 7]
	radius = float(rng.choice([3., 8., 15., 30.]))
	span = radius / 3600. * rng.uniform(6, 14)   # box of a few search radii: dense groups
	n0 = int(rng.randint(3, 40))
	sizes = [n0] + [int(rng.randint(10, 120 if k < 4 else 60)) for _ in range(k - 1)]
	if where == 'flat':
		c_ra, c_dec = rng.uniform(20, 340), rng.uniform(-40, 40)
		if seed % 3 == 0:
			c_dec = rng.uniform(-0.3, 0.3) * span  # straddle the equator
	elif where == 'high':
		c_ra, c_dec = rng.uniform(20, 340), rng.choice([-1, 1]) * rng.uniform(50, 80)
	elif where == 'seam':
		c_ra, c_dec = 0.0, rng.uniform(-30, 30)
	else:
		c_ra, c_dec = rng.uniform(0, 360), (90.0 if where == 'north' else -90.0)
	tabs = []
	for i, n in enumerate(sizes):
		if where in ('north', 'south'):
			d = np.abs(rng.normal(0, span, size=n))
			dec = 90 - d if where == 'north' else -90 + d
			ra = rng.uniform(0, 360, size=n)
		else:
			dec = c_dec + rng.uniform(-span, span, size=n) / 2
			ra = np.mod(c_ra + rng.uniform(-span, span, size=n) / 2 / np.cos(np.radians(c_dec)), 360.)
		if i > 0:  # counterparts of some primaries
			m = min(n, n0)
			has = rng.uniform(size=m) < 0.6
			pr, pd = _scatter(rng, tabs[0]['ra'][:m][has], tabs[0]['dec'][:m][has], radius / 4)
			ra[:m][has], dec[:m][has] = pr, pd
		err = rng.uniform(0.2, radius / 3, size=n) if rng.uniform() < 0.7 else rng.uniform(0.2, radius / 3) * np.ones(n)
		tabs.append(cat('C%d' % i, ra, dec, err, (span * 1.0)**2 if where not in ('north', 'south') else np.pi * (2 * span)**2))
	comp = float(rng.uniform(0.5, 1.0)) if rng.uniform() < 0.5 else np.r_[1.0, rng.uniform(0.4, 1.0, size=k - 1)]
	opts = dict(prob_ratio_secondary=float(rng.choice([0.5, 0.2, 0.8])), min_prob=float(rng.choice([0., 0., 0.03])),
		consider_unrelated_associations=bool(rng.uniform() < 0.8))
	if seed % 4 == 1:  # a supplied magnitude histogram on the last catalogue
		mag = rng.normal(22, 2, size=sizes[-1])
		mag[rng.uniform(size=sizes[-1]) < 0.1] = -99
		mag[rng.uniform(size=sizes[-1]) < 0.05] = np.nan
		edges = np.sort(rng.uniform(16, 28, size=7))
		hs, ha = rng.uniform(0, 1, size=6), rng.uniform(0, 1, size=6)
		hs[rng.randint(6)] = 0.0
		ha[rng.randint(6)] = 0.0
		tabs[-1]['mags'], tabs[-1]['magnames'], tabs[-1]['maghists'] = [mag], ['M'], [(edges[:-1], edges[1:], hs, ha)]
	return tabs, radius, comp, opts


def gen_fuzz(nseeds=35):
	"""small randomized configurations through the reference: k = 2..4, flat cells and the HEALPix
	branch (poles, seam, high declination; oracle/healpix.py standing in for healpy), scalar and
	vector completeness, secondary ratio, min_prob, supplied magnitude histograms"""
	out = dict(nseeds=np.array([nseeds]))
	cwd = os.getcwd()
	import tempfile
	os.chdir(tempfile.mkdtemp(prefix='nwayfuzz_'))
	try:
		for seed in range(nseeds):
			tabs, radius, comp, opts = fuzz_case(seed)
			tag = 'f%d_' % seed
			for i, t in enumerate(tabs):
				out[tag + 'ra%d' % i], out[tag + 'dec%d' % i], out[tag + 'err%d' % i] = t['ra'].copy(), t['dec'].copy(), t['error'].copy()
				out[tag + 'area%d' % i] = np.array([t['area']])
			out[tag + 'k'] = np.array([len(tabs)])
			out[tag + 'radius'] = np.array([radius])
			out[tag + 'completeness'] = np.atleast_1d(comp)
			out[tag + 'opts'] = np.array([opts['prob_ratio_secondary'], opts['min_prob'], float(opts['consider_unrelated_associations'])])
			if tabs[-1]['mags']:
				out[tag + 'mag'] = tabs[-1]['mags'][0].copy()
				out[tag + 'maghist'] = np.array([np.r_[h, np.nan] if len(h) == 6 else h for h in (np.r_[tabs[-1]['maghists'][0][0], tabs[-1]['maghists'][0][1][-1]],) + tuple(tabs[-1]['maghists'][0][2:])])
			names = [t['name'] for t in tabs]
			try:
				res = ref.nway_match(tabs, match_radius=radius, prior_completeness=comp, store_mag_hists=False, logger=LOG, **opts)
			except ref.EmptyResultException:
				out[tag + 'empty'] = np.array([1])
				print('fuzz %d: empty' % seed)
				continue
			out.update(table_arrays(res, names, tag))
			if tabs[-1]['mags']:
				out[tag + 'bias'] = res['bias_%s_M' % names[-1]].values
			print('fuzz %2d: k=%d %s rows=%d flags=%s' % (seed, len(tabs), ['flat', 'flat', 'flat', 'north', 'seam', 'south', 'high'][seed % 7], len(res), np.bincount(res['match_flag'].values, minlength=3)))
	finally:
		os.chdir(cwd)
	save('fuzz', **out)


if __name__ == '__main__' and ('fuzz' in sys.argv[1:] or not sys.argv[1:]):
	gen_fuzz()


def magmix_tables(seed=2024):
	"""3-way, 400 primaries; magnitude columns on the PRIMARY (supplied histogram) and on both
	secondaries (learned).  tests/goldenutil.py:magmix_tables stores nothing: the arrays are saved."""
	rng = np.random.RandomState(seed)
	n0, n1, n2 = 400, 3000, 2500
	span = 0.2
	p_ra, p_dec = rng.uniform(40, 40 + span, n0), rng.uniform(-5, -5 + span, n0)
	tabs = [cat('P', p_ra, p_dec, rng.uniform(0.5, 2.0, n0), span**2)]
	tabs[0]['mags'] = [rng.normal(18, 1, n0)]
	tabs[0]['magnames'] = ['F']
	edges = np.linspace(14, 22, 9)
	tabs[0]['maghists'] = [(edges[:-1], edges[1:], rng.uniform(0.05, 1, 8), rng.uniform(0.05, 1, 8))]
	for name, n, frac, sig, m0 in (('A', n1, 0.75, 0.4, 23.0), ('B', n2, 0.6, 0.8, 21.0)):
		ra, dec = rng.uniform(40, 40 + span, n), rng.uniform(-5, -5 + span, n)
		mag = rng.normal(m0, 1.2, n)
		has = np.flatnonzero(rng.uniform(size=n0) < frac)
		pr, pd = _scatter(rng, p_ra[has], p_dec[has], sig)
		ra[:len(has)], dec[:len(has)] = pr, pd
		mag[:len(has)] = rng.normal(m0 - 2.5, 0.8, len(has))
		mag[rng.choice(n, n // 40, replace=False)] = -99
		mag[rng.choice(n, n // 100, replace=False)] = np.nan
		order = rng.permutation(n)
		t = cat(name, ra[order], dec[order], (0.3 if name == 'A' else 0.6) * np.ones(n), span**2)
		t['mags'], t['magnames'], t['maghists'] = [mag[order]], ['M'], [None]
		tabs.append(t)
	return tabs


def gen_magmix():
	out = {}
	tabs = magmix_tables()
	for i, t in enumerate(tabs):
		out['ra%d' % i], out['dec%d' % i], out['err%d' % i], out['mag%d' % i] = t['ra'], t['dec'], t['error'], t['mags'][0].copy()
	out['area'] = np.array([tabs[0]['area']])
	out['hist0'] = np.array([np.r_[tabs[0]['maghists'][0][0], tabs[0]['maghists'][0][1][-1]], np.r_[tabs[0]['maghists'][0][2], np.nan], np.r_[tabs[0]['maghists'][0][3], np.nan]])
	names = ['P', 'A', 'B']
	cwd = os.getcwd()
	import tempfile
	os.chdir(tempfile.mkdtemp(prefix='nwaymagmix_'))
	try:
		for tag, kw in (('rad_', dict(mag_include_radius=1.5, mag_exclude_radius=6.0)), ('post_', dict(magauto_post_single_minvalue=0.7))):
			res = ref.nway_match(magmix_tables(), match_radius=12., prior_completeness=np.array([1.0, 0.8, 0.7]), store_mag_hists=False, logger=LOG, **kw)
			out.update(table_arrays(res, names, tag))
			for b in ('bias_P_F', 'bias_A_M', 'bias_B_M'):
				out[tag + b] = res[b].values
			print('magmix %s %d rows flags %s' % (tag, len(res), np.bincount(res['match_flag'].values)))
	finally:
		os.chdir(cwd)
	save('magmix', **out)


if __name__ == '__main__' and ('magmix' in sys.argv[1:] or not sys.argv[1:]):
	gen_magmix()


def gen_sparse():
	"""sparse fields (chance neighbours per primary << 1): the inputs on which the product takes its
	fused sparse kernels -- 2-, 3- and 4-way, API and script numerics, flat cells and (same
	tables moved to Dec 60..70) the HEALPix branch"""
	rng = np.random.RandomState(314)
	n0, ns = 1500, 12000
	out = dict(radius=np.array([6.]), completeness=np.array([1.0, 0.9, 0.8, 0.7]))
	p_ra, p_dec = rng.uniform(100, 110, n0), rng.uniform(-5, 5, n0)
	tabs = [cat('P', p_ra, p_dec, rng.uniform(0.3, 1.5, n0), 100.)]
	for name, frac, sig in (('A', 0.8, 0.5), ('B', 0.6, 0.8), ('C', 0.5, 1.0)):
		ra, dec = rng.uniform(100, 110, ns), rng.uniform(-5, 5, ns)
		has = np.flatnonzero(rng.uniform(size=n0) < frac)
		pr, pd = _scatter(rng, p_ra[has], p_dec[has], sig)
		ra[:len(has)], dec[:len(has)] = pr, pd
		# a few primaries with two or three counterparts in the same catalogue
		extra = has[:25]
		er, ed = _scatter(rng, p_ra[extra], p_dec[extra], 1.5)
		ra[len(has):len(has) + 25], dec[len(has):len(has) + 25] = er, ed
		order = rng.permutation(ns)
		tabs.append(cat(name, ra[order], dec[order], rng.uniform(0.2, 0.6, ns), 100.))
	for i, t in enumerate(tabs):
		out['ra%d' % i], out['dec%d' % i], out['err%d' % i] = t['ra'], t['dec'], t['error']
	names = [t['name'] for t in tabs]
	for shift, where in ((0.0, 'flat'), (65.0, 'high')):
		moved = [cat(t['name'], t['ra'], t['dec'] + shift, t['error'], t['area']) for t in tabs]
		for k in (2, 3, 4):
			tag = '%s%d_' % (where, k)
			comp = out['completeness'][:k]
			res = run(moved[:k], 6., comp)
			out.update(table_arrays(res, names[:k], tag))
			print('sparse %s k=%d: %d rows, ncat %s' % (where, k, len(res), np.bincount(res['ncat'].values)))
	save('sparse', **out)


if __name__ == '__main__' and ('sparse' in sys.argv[1:] or not sys.argv[1:]):
	gen_sparse()


def gen_ellflow():
	"""The script's flow for ELLIPTICAL position errors end to end (``file :major:minor:angle``;
	nway.py:52-88 error columns, :303-305 offset matrices, :327-360 main pass with
	log_bf_elliptical, :366-420 correction loop with its elliptical branch :402-411, then the
	posterior / group statistics of :425-586), on three seeded catalogues.

	A transcription of those lines driven with the reference's OWN functions
	(_create_match_table, convert_from_ellipse, log_bf_elliptical, unnormalised_log_posterior,
	posterior, _compute_final_probabilities) -- the script itself needs astropy.io.fits.  The two
	astropy calls inside dist3d (fastskymatch.py:61-67: SkyOffsetFrame(origin=a), transform_to)
	are replaced by the rotation astropy documents for that frame (oracle/elliptical_oracle.py:
	offsets; origin to (0, 0), no roll): this pins the BRANCH LOGIC and the numerics around the
	offsets (float32 'E' columns, numpy's float32 length / unit vector in the main pass, float64 in
	the correction loop), not astropy -- as allsky.npz does for healpy."""
	sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), 'oracle'))
	import elliptical_oracle as eo
	bd = ref.bayesdist
	rng = np.random.RandomState(31)
	radius, completeness, ratio = 15.0, 0.9, 0.5
	half = 0.12
	area = (2 * half)**2
	sizes = [400, 3000, 2500]
	names = ['X', 'O', 'I']
	cats = []
	for c, n in enumerate(sizes):
		ra = np.round(rng.uniform(150.0 - half, 150.0 + half, size=n), 7)
		dec = np.round(rng.uniform(2.0 - half, 2.0 + half, size=n), 7)
		if c > 0:
			m = int(0.7 * sizes[0])
			ra[:m] = np.round(cats[0]['ra'][:m] + rng.normal(0, 1.2, size=m) / 3600. / np.cos(np.radians(cats[0]['dec'][:m])), 7)
			dec[:m] = np.round(cats[0]['dec'][:m] + rng.normal(0, 1.2, size=m) / 3600., 7)
		major = np.round(rng.uniform(0.5, 3.0, size=n), 4)
		minor = np.round(major * rng.uniform(0.3, 1.0, size=n), 4)
		angle = np.round(rng.uniform(0, 180, size=n), 3)
		cats.append(dict(name=names[c], ra=ra, dec=dec, major=major, minor=minor, angle=angle))
	k = len(cats)
	tables = [cat(t['name'], t['ra'], t['dec'], np.ones(len(t['ra'])), area) for t in cats]
	table, resultstable, separations, _ = ref._create_match_table([dict(t, ra=t['ra'].copy(), dec=t['dec'].copy(), error=t['error'].copy()) for t in tables],
		radius, logger=LOG)
	idx = resultstable
	nrows = len(idx)

	def merged(c, col):  # a column of catalogue c in the merged table: -99 where the source is absent (fastskymatch.py:272-279)
		out = np.array(cats[c][col][idx[:, c]], dtype=float)
		out[idx[:, c] == -1] = -99
		return out
	# nway.py:52-66: per catalogue (sigma_ra, sigma_dec, rho) on the rows of the merged table
	errors = []
	for c in range(k):
		rho = (merged(c, 'angle') - 90) / 180 * np.pi
		errors.append(bd.convert_from_ellipse(merged(c, 'major'), merged(c, 'minor'), rho))
	# fastskymatch.py:298-331: offsets in the frame of the LATER catalogue's source, stored as 'E' columns
	nan = np.ones(nrows) * np.nan
	sep_ra = [[nan for _ in range(k)] for _ in range(k)]
	sep_dec = [[nan for _ in range(k)] for _ in range(k)]
	for i in range(k):
		a_ra, a_dec = merged(i, 'ra'), merged(i, 'dec')
		for j in range(i):
			b_ra, b_dec = merged(j, 'ra'), merged(j, 'dec')
			col_ra, col_dec = eo.offsets(a_ra, a_dec, b_ra, b_dec)
			col_ra, col_dec = np.array(col_ra), np.array(col_dec)
			for col in (col_ra, col_dec):
				col[a_ra == -99] = np.nan
				col[b_ra == -99] = np.nan
			# nway.py:283-305 make_separation_table_matrix: [ti][tj] (ti < tj) = column Separation_{tj}_{ti}_ra
			sep_ra[j][i] = (col_ra * 60 * 60).astype(np.float32)
			sep_dec[j][i] = (col_dec * 60 * 60).astype(np.float32)
	dens, dens_plus = ref._compute_source_densities(tables, logger=LOG)
	comp = np.array([1.0] + [completeness**(1. / (k - 1)) for _ in range(1, k)])  # nway.py:209
	# nway.py:327-360
	log_bf = np.zeros(nrows) * np.nan
	prior = np.zeros(nrows) * np.nan
	for case in range(2**(k - 1)):
		table_mask = np.array([True] + [(case // 2**(ti)) % 2 == 0 for ti in range(k - 1)])
		mask = True
		for i in range(1, k):
			if table_mask[i]:
				mask = np.logical_and(mask, ~np.isnan(separations[0][i]))
			else:
				mask = np.logical_and(mask, np.isnan(separations[0][i]))
		errors_selected = [(era[mask], edec[mask], ephi[mask]) for (era, edec, ephi), m in zip(errors, table_mask) if m]
		sra = [[cell[mask] for cell, m in zip(row, table_mask) if m] for row, m in zip(sep_ra, table_mask) if m]
		sdec = [[cell[mask] for cell, m in zip(row, table_mask) if m] for row, m in zip(sep_dec, table_mask) if m]
		log_bf[mask] = bd.log_bf_elliptical(sra, sdec, errors_selected)
		prior[mask] = dens[0] * np.prod(comp[table_mask]) / np.prod(dens_plus[table_mask])
	assert np.isfinite(prior).all() and np.isfinite(log_bf).all()
	uncorrected = log_bf.copy()
	# nway.py:366-420 (elliptical branch :402-411)
	ncat = table['ncat'].values
	prim = idx[:, 0]
	starts = np.flatnonzero(np.r_[True, prim[1:] != prim[:-1]])
	ends = np.r_[starts[1:], nrows]
	group_of = np.repeat(np.arange(len(starts)), ends - starts)
	for i in np.where(ncat <= k - 2)[0]:
		missing_cats = [c for c, sep in enumerate(separations[0]) if np.isnan(sep[i])]
		best_logpost = 0
		g = group_of[i]
		for j in range(starts[g], ends[g]):
			if not (ncat[j] > 2):
				continue
			augmented_cats = [c for c in missing_cats if not np.isnan(separations[0][c][j])]
			if len(augmented_cats) >= 2:
				prior_j = dens[augmented_cats[0]] / np.prod(dens_plus[augmented_cats])
				sra = [[[sep_ra[c][c2][j]] for c2 in augmented_cats] for c in augmented_cats]
				sdec = [[[sep_dec[c][c2][j]] for c2 in augmented_cats] for c in augmented_cats]
				errors_selected = [([errors[c][0][j]], [errors[c][1][j]], [errors[c][2][j]]) for c in augmented_cats]
				log_bf_j = bd.log_bf_elliptical(np.array(sra), np.array(sdec), np.array(errors_selected))
				logpost_j = bd.unnormalised_log_posterior(prior_j, log_bf_j, len(augmented_cats))[0]
				if logpost_j > best_logpost:
					best_logpost = logpost_j
		if best_logpost > 0:
			log_bf[i] += best_logpost
	assert (log_bf != uncorrected).sum() > 20, 'the correction loop should change some rows of this fixture'
	# nway.py:425-586 == __init__.py:399-461
	t2 = table.assign(dist_bayesfactor_uncorrected=uncorrected, dist_bayesfactor=log_bf, dist_post=bd.posterior(prior, log_bf))
	res = ref._compute_final_probabilities(tables, t2, ratio, prior, log_bf, logger=LOG)
	out = dict(radius=np.array([radius]), completeness=np.array([completeness]), area=np.array([area]))
	for c, t in enumerate(cats):
		for col in ('ra', 'dec', 'major', 'minor', 'angle'):
			out['in%d_%s' % (c, col)] = t[col]
	out.update(table_arrays(res, names))
	for i in range(k):
		for j in range(i):
			out['off_ra_%d_%d' % (j, i)] = sep_ra[j][i]
			out['off_dec_%d_%d' % (j, i)] = sep_dec[j][i]
	out['changed_rows'] = np.flatnonzero(log_bf != uncorrected)
	save('ell_flow', **out)


if __name__ == '__main__' and ('ellflow' in sys.argv[1:] or not sys.argv[1:]):
	gen_ellflow()


def gen_fitshdr():
	"""header cards (80-character records, text) of the BINTABLE extensions of two of the reference's own data
	files, written by a foreign FITS writer (STIL / TOPCAT): what tests/test_fits_columns.py holds the own
	writer's cards against, byte for byte"""
	import json
	out = {}
	for key, path in (('COSMOS_XMM', 'doc/COSMOS_XMM.fits'), ('randomcatX', 'tests/elltest/randomcatX.fits')):
		raw = open(os.path.join(REFERENCE, path), 'rb').read()
		hdus, at = [], 0
		while at < len(raw) and len(hdus) < 2:
			cards = []
			while True:
				card = raw[at:at + 80].decode('ascii')
				at += 80
				cards.append(card)
				if card.startswith('END'):
					break
			at = (at + 2879) // 2880 * 2880
			hdr = dict((c[:8].strip(), c[10:30].strip()) for c in cards if c[8:10] == '= ')
			nbytes = abs(int(hdr.get('BITPIX', 8))) // 8
			naxis = int(hdr.get('NAXIS', 0))
			for i in range(1, naxis + 1):
				nbytes *= int(hdr['NAXIS%d' % i])
			if naxis == 0:
				nbytes = 0
			at += (nbytes + 2879) // 2880 * 2880
			hdus.append(cards)
		out[key] = hdus[1]
	with open(os.path.join(HERE, 'foreign_fits_headers.json'), 'w') as f:
		json.dump(out, f, indent=1)
	print('foreign_fits_headers.json', dict((k, len(v)) for k, v in out.items()))


if __name__ == '__main__' and ('fitshdr' in sys.argv[1:] or not sys.argv[1:]):
	gen_fitshdr()
