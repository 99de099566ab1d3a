"""Pin the CPU oracle (oracle/nway_oracle.py) against golden vectors produced by the
reference itself (tests/golden/make_golden.py) and the reference's known answers."""
import os
import sys

import numpy as np
import pytest

from goldenutil import (ROOT, golden, ell_tables, xmm_tables, assert_table_matches,
	assert_checksums_match, idx_hash, cat, script_golden, assert_script_correction)

sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import nway_oracle as orc  # noqa: E402
import nway_oracle_c as orc_c  # noqa: E402

# the oracle repeats the reference's numpy operations in the same order, on the same
# libm: it is expected to agree far below the product tolerance
TIGHT = dict(rtol=1e-12, atol=1e-13)


def test_known_answers():
	g = golden('kat_math')
	# constants quoted in SURVEY.md section 8c (computed with the reference)
	assert orc.log_bf2(0.3, 0.1, 0.2) == pytest.approx(11.840045223967955, rel=1e-14)
	assert orc.log_bf3(.3, .3, .3, .1, .2, .3) == pytest.approx(23.61118582441539, rel=1e-14)
	assert orc.log_arcsec2rad == g['log_arcsec2rad'][0] == pytest.approx(12.236916089485012, rel=1e-15)
	assert orc.posterior(1e-3, 2.5) == pytest.approx(0.24043574366935122, rel=1e-14)
	assert orc.log_posterior(1e-3, 2.5) == pytest.approx(-0.6190009687639575, rel=1e-14)
	d = orc.dist((53.15964508, -27.92927742), (53.15953445, -27.9313736))
	assert d == g['dist_scalar'][0] == pytest.approx(0.002098457623965017, rel=1e-13)


def test_log_bf_family():
	"""tests/bayesdistance_test.py:12-32 of the reference, on the oracle"""
	g = golden('kat_math')
	for i, psi in enumerate(g['sep']):
		a = orc.log_bf([[None, psi]], [0.1, 0.2])
		np.testing.assert_almost_equal(orc.log_bf2(psi, 0.1, 0.2), a)
		assert a == g['log_bf_n2'][i]
		b = orc.log_bf([[None, psi, psi], [psi, None, psi], [psi, psi, None]], [0.1, 0.2, 0.3])
		np.testing.assert_almost_equal(orc.log_bf3(psi, psi, psi, 0.1, 0.2, 0.3), b)
		assert b == g['log_bf_n3'][i]
	assert orc.log_bf([[None]], [0.5]) == g['log_bf_n1'][0] == 0.0
	p4 = g['n4_sep']
	got = orc.log_bf([[p4[i][j] for j in range(4)] for i in range(4)], list(g['n4_sigma']))
	np.testing.assert_array_equal(got, g['n4_log_bf'])


def test_posterior_and_dist_arrays():
	g = golden('kat_math')
	np.testing.assert_array_equal(orc.posterior(g['post_prior'], g['post_logbf']), g['posterior'])
	np.testing.assert_array_equal(orc.log_posterior(g['post_prior'], g['post_logbf']), g['log_posterior'])
	np.testing.assert_array_equal(orc.unnormalised_log_posterior(g['post_prior'], g['post_logbf'], 2), g['unnormalised_log_posterior'])
	np.testing.assert_array_equal(orc.dist((g['dist_ra'], g['dist_dec']), (g['dist_ra2'], g['dist_dec2'])), g['dist_array'])
	np.testing.assert_array_equal(orc.dist((g['sph_a_ra'], g['sph_a_dec']), (g['sph_b_ra'], g['sph_b_dec'])), g['sph_dist'])


def test_crossproduct_literal_equals_closed_form_small():
	"""the bucket/itertools restatement and the closed-form cell predicate agree"""
	g = golden('edge')
	tabs = [(g['neg_ra%d' % i], g['neg_dec%d' % i]) for i in range(3)]
	err = float(g['neg_radius'][0]) / 60. / 60
	lit = orc.crossproduct_literal(tabs, err)
	np.testing.assert_array_equal(lit, g['neg_crossproduct'])
	np.testing.assert_array_equal(orc.crossproduct(tabs, err), g['neg_crossproduct'])


def test_ell2_crossproduct_and_table():
	X, R, O = ell_tables()
	g = golden('ell2')
	cp = orc.crossproduct([(X['ra'], X['dec']), (O['ra'], O['dec'])], 10. / 60 / 60)
	np.testing.assert_array_equal(cp, g['crossproduct'])
	names = [X['name'], O['name']]
	t = orc.nway_match([X, O], 10., 1.0, literal_groups=True)
	assert_table_matches(t, g, 'c10_', names, **TIGHT)
	assert_checksums_match(t, g, 'c10_', names)
	# vectorised group statistics agree with the literal loop
	t2 = orc.nway_match([X, O], 10., 1.0)
	assert_table_matches(t2, g, 'c10_', names, **TIGHT)
	t9 = orc.nway_match([X, O], 10., 0.9)
	assert_checksums_match(t9, g, 'c09_', names)
	np.testing.assert_array_equal(t9['match_flag'], g['c09_match_flag'])
	f1 = t9['match_flag'] == 1
	np.testing.assert_allclose(t9['prob_has_match'][f1], g['c09_best_prob_has_match'], **TIGHT)
	# SURVEY 8c known answer: p_any of primary 0
	assert t9['prob_has_match'][0] == pytest.approx(0.12830303519644448, rel=1e-12)
	tt = orc.nway_match([X, O], 10., 0.9, prob_ratio_secondary=0.25, min_prob=0.01)
	assert_table_matches(tt, g, 'trunc_', names, **TIGHT)


def test_ell3():
	X, R, O = ell_tables()
	g = golden('ell3')
	tabs = [(t['ra'], t['dec']) for t in (X, R, O)]
	cp = orc.crossproduct(tabs, 10. / 60 / 60)
	assert len(cp) == int(g['crossproduct_nrows'][0]) == 1831619
	assert idx_hash(cp) == g['crossproduct_hash'][0]
	np.testing.assert_array_equal(np.bincount(cp[:, 0]), g['crossproduct_rows_per_primary'])
	names = [X['name'], R['name'], O['name']]
	t = orc.nway_match([X, R, O], 10., 1.0)
	assert len(t['ncat']) == 450435
	assert_checksums_match(t, g, 'c10_', names)
	assert_table_matches(t, g, 'c10_sub_', names, rows=g['c10_sub_rows'], **TIGHT)
	# the SCRIPT on the same three files (nway.py executed by make_script_golden.py): its float32 separations, its
	# unrelated-association loop (nway.py:366-420), everything downstream
	gs = script_golden()
	ts = orc.nway_match([X, R, O], 10., 1.0, correction='cli', f32_roundtrip=True)
	assert_script_correction(ts, gs, 'ell3_', rtol=1e-9)
	assert len(gs['ell3_cli_changed_rows']) == 48 and gs['ell3_cli_correction'].sum() == pytest.approx(23.9030205232378, rel=1e-12)
	assert_checksums_match(ts, gs, 'ell3_script_', names)
	assert_table_matches(ts, gs, 'ell3_script_sub_', names, rows=gs['ell3_script_sub_rows'], **TIGHT)
	# the same loop on float64 separations (the product's unrelated_associations='cli' without f32_roundtrip) is run by no
	# reference code; it follows the script's to within the float32 rounding of the separations
	tc = orc.nway_match([X, R, O], 10., 1.0, correction='cli')
	assert_script_correction(tc, gs, 'ell3_', rtol=5e-6, atol=2e-6)


def test_xmm_standins():
	X, O, I = xmm_tables()
	g = golden('xmm_syn')
	t = orc.nway_match([X, O], 20., 0.9)
	assert len(t['ncat']) == 44909
	assert_table_matches(t, g, 'w2_', ['XMM', 'OPT'], **TIGHT)
	t3 = orc.nway_match([X, O, I], 20., 0.9)
	assert len(t3['ncat']) == 449459
	assert_checksums_match(t3, g, 'w3_', ['XMM', 'OPT', 'IRAC'])
	assert_table_matches(t3, g, 'w3_sub_', ['XMM', 'OPT', 'IRAC'], rows=g['w3_sub_rows'], **TIGHT)
	# the script on COSMOS_XMM.fits and the two stand-ins
	gs = script_golden()
	ts = orc.nway_match([X, O, I], 20., 0.9, correction='cli', f32_roundtrip=True)
	assert_script_correction(ts, gs, 'xmm_w3_', rtol=1e-9)
	assert_checksums_match(ts, gs, 'xmm_w3_script_', ['XMM', 'OPT', 'IRAC'])
	assert_table_matches(ts, gs, 'xmm_w3_script_sub_', ['XMM', 'OPT', 'IRAC'], rows=gs['xmm_w3_script_sub_rows'], **TIGHT)


def test_edge_cases():
	g = golden('edge')
	# negative-declination cells, 3-way, vector completeness, lone primaries
	tabs = [cat('ABC'[i], g['neg_ra%d' % i], g['neg_dec%d' % i], g['neg_err%d' % i], g['neg_area'][0]) for i in range(3)]
	t = orc.nway_match(tabs, float(g['neg_radius'][0]), g['neg_completeness'], literal_groups=True)
	assert_table_matches(t, g, 'neg_', ['A', 'B', 'C'], **TIGHT)
	# the three far-away primaries are groups of one row: p_any = 0, p_i = 0, flag 1
	for p in range(3):
		rows = np.flatnonzero(t['A'] == p)
		assert len(rows) == 1 and t['match_flag'][rows[0]] == 1 and t['prob_has_match'][rows[0]] == 0
	tc = orc.nway_match(tabs, float(g['neg_radius'][0]), g['neg_completeness'], correction='cli', f32_roundtrip=True)
	assert_script_correction(tc, script_golden(), 'neg_w3_', rtol=1e-10)
	# ties / duplicates
	tp = cat('P', g['tie_p_ra'], g['tie_p_dec'], g['tie_p_err'], 1.0)
	ts = cat('S', g['tie_s_ra'], g['tie_s_dec'], g['tie_s_err'], 1.0)
	t = orc.nway_match([tp, ts], float(g['tie_radius'][0]), float(g['tie_completeness'][0]), literal_groups=True)
	assert_table_matches(t, g, 'tie_', ['P', 'S'], **TIGHT)
	# hopeless single candidate
	tp = cat('P', [g['hop_p'][0]], [g['hop_p'][1]], [g['hop_p'][2]], 1.0)
	ts = cat('S', [g['hop_s'][0]], [g['hop_s'][1]], [g['hop_s'][2]], 1.0)
	t = orc.nway_match([tp, ts], float(g['hop_radius'][0]), float(g['hop_completeness'][0]), literal_groups=True)
	assert_table_matches(t, g, 'hop_', ['P', 'S'], **TIGHT)
	assert t['prob_this_match'][1] == 1.0 and t['match_flag'][1] == 1
	# 4-way
	tabs = [cat('T%d' % i, g['k4_ra%d' % i], g['k4_dec%d' % i], g['k4_err%d' % i], g['k4_area'][0]) for i in range(4)]
	cp = orc.crossproduct([(x['ra'], x['dec']) for x in tabs], float(g['k4_radius'][0]) / 60 / 60)
	assert len(cp) == int(g['k4_crossproduct_nrows'][0])
	np.testing.assert_array_equal(np.bincount(cp[:, 0], minlength=25), g['k4_crossproduct_rows_per_primary'])
	t = orc.nway_match(tabs, float(g['k4_radius'][0]), float(g['k4_completeness'][0]), literal_groups=True)
	assert_table_matches(t, g, 'k4_', ['T0', 'T1', 'T2', 'T3'], **TIGHT)


def test_four_and_five_way_with_script_correction():
	"""generic k: presence patterns, vector completeness, and nway.py:366-420 for k > 3"""
	g = golden('kway')
	for tag, k in (('k4c', 4), ('k5', 5), ('k6', 6), ('k8', 8)):
		names = ['T%d' % i for i in range(k)]
		tabs = [cat(names[i], g['%s_ra%d' % (tag, i)], g['%s_dec%d' % (tag, i)], g['%s_err%d' % (tag, i)], g[tag + '_area'][0]) for i in range(k)]
		comp = g[tag + '_completeness']
		comp = float(comp[0]) if len(comp) == 1 else comp
		t = orc.nway_match(tabs, float(g[tag + '_radius'][0]), comp, literal_groups=True)
		assert_table_matches(t, g, tag + '_', names, **TIGHT)
		gs = script_golden()
		tc = orc.nway_match(tabs, float(g[tag + '_radius'][0]), comp, correction='cli')
		assert_script_correction(tc, gs, tag + '_', rtol=5e-6, atol=2e-6)  # float64 separations: no reference run has them (see the elltest case)
		tcc = orc_c.nway_match(tabs, float(g[tag + '_radius'][0]), comp, correction='cli')
		np.testing.assert_allclose(tcc['dist_bayesfactor'], tc['dist_bayesfactor'], rtol=1e-12)
		np.testing.assert_array_equal(tcc['match_flag'], tc['match_flag'])
		for oracle in (orc, orc_c):
			ts = oracle.nway_match(tabs, float(g[tag + '_radius'][0]), comp, correction='cli', f32_roundtrip=True)
			assert_table_matches(ts, gs, tag + '_script_', names, **TIGHT)
			assert_script_correction(ts, gs, tag + '_', rtol=1e-10)


def kmulti_cases():
	g = golden('kmulti')
	for tag, k in (('m5', 5), ('m6', 6)):
		names = ['T%d' % i for i in range(k)]
		tabs = [cat(names[i], g['%s_ra%d' % (tag, i)], g['%s_dec%d' % (tag, i)], g['%s_err%d' % (tag, i)], g[tag + '_area'][0]) for i in range(k)]
		comp = g[tag + '_completeness']
		yield tag, names, tabs, float(g[tag + '_radius'][0]), (float(comp[0]) if len(comp) == 1 else comp), g


def test_five_and_six_way_with_several_links_per_catalogue():
	"""the reference's nway_match on 5- and 6-way tables where a few hundred primaries have two sources in two or more
	catalogues (groups of up to 243 rows): both oracles, also the script's correction loop"""
	for tag, names, tabs, radius, comp, g in kmulti_cases():
		assert np.bincount(g[tag + '_idx'][:, 0]).max() >= 90
		for oracle in (orc, orc_c):
			t = oracle.nway_match(tabs, radius, comp)
			# (1e-10: a last-bit difference of a sine or cosine -- numpy's vector loops against its scalar ones, glibc in the C
			# restatement -- is 5e-12 of a separation of a few arcsec)
			assert_table_matches(t, g, tag + '_', names, rtol=1e-10, atol=1e-13)
			ts = oracle.nway_match(tabs, radius, comp, correction='cli', f32_roundtrip=True)
			assert_table_matches(ts, script_golden(), tag + '_script_', names, rtol=1e-10, atol=1e-13)
			assert_script_correction(ts, script_golden(), tag + '_', rtol=1e-9)


def test_randomized_configurations():
	"""35 small random configurations run through the reference (flat cells and its HEALPix
	branch at the poles, the seam and high declination; k = 2..4; completeness, ratio, min_prob)"""
	from goldenutil import fuzz_cases
	n = 0
	for tag, tabs, radius, comp, opts, g in fuzz_cases():
		if tabs[-1]['mags']:
			continue  # magnitude priors are host logic of the product, not part of the oracle
		names = [t['name'] for t in tabs]
		t = orc.nway_match(tabs, radius, comp, prob_ratio_secondary=opts['prob_ratio_secondary'], min_prob=opts['min_prob'],
			correction='api', literal_groups=True)
		assert_table_matches(t, g, tag, names, **TIGHT)
		# ... and with the script's numerics and correction loop (float32 separations), both oracles
		for oracle in (orc, orc_c):
			ts = oracle.nway_match(tabs, radius, comp, prob_ratio_secondary=opts['prob_ratio_secondary'], correction='cli', f32_roundtrip=True)
			assert_table_matches(ts, script_golden(), tag + 'script_', names, **TIGHT)
		n += 1
	assert n >= 20


def test_sparse_fields():
	"""sparse 2-/3-/4-way tables, flat cells and HEALPix branch, API and script numerics"""
	g = golden('sparse')
	names = ['P', 'A', 'B', 'C']
	for shift, where in ((0.0, 'flat'), (65.0, 'high')):
		tabs = [cat(names[i], g['ra%d' % i], g['dec%d' % i] + shift, g['err%d' % i], 100.) for i in range(4)]
		for k in (2, 3, 4):
			tag = '%s%d_' % (where, k)
			comp = g['completeness'][:k]
			# sub-arcsecond separations at Dec 65: the difference in the Vincenty numerator
			# (fastskymatch.py:40-41) cancels to ~1e-6 of its terms, so one ulp in a sine -- numpy's
			# SIMD loops round differently from libm on array tails -- is 1e-10 of the result
			tol = TIGHT if where == 'flat' else dict(rtol=1e-9, atol=1e-13)
			t = orc.nway_match(tabs[:k], 6., comp, literal_groups=True)
			assert_table_matches(t, g, tag, names[:k], **tol)
			for oracle in (orc, orc_c):
				ts = oracle.nway_match(tabs[:k], 6., comp, correction='cli', f32_roundtrip=True)
				# the script's float32 separations: where a float64 separation lies within an ulp of the midpoint of two float32
				# values, the last bit of a sine decides which one it becomes (one row of high3: 2.6e-9 of its log_bf, 3.9e-8 of a p_i)
				assert_table_matches(ts, script_golden(), tag + 'script_', names[:k], **(TIGHT if where == 'flat' and oracle is orc else dict(rtol=1e-9 if where == 'flat' else 1e-7, atol=1e-13)))


def test_sphere_scheme_equals_bruteforce():
	"""all-sky inputs (reference: HEALPix branch, not executable here): the oracle's sweep is
	checked against an O(N^2) evaluation of its own definition incl. poles and the RA seam"""
	rng = np.random.RandomState(5)
	n0, n1, n2 = 300, 2500, 2000
	def sph(n):
		return rng.uniform(0, 360, size=n), np.degrees(np.arcsin(rng.uniform(-1, 1, size=n)))
	a, b, c = sph(n0), sph(n1), sph(n2)
	# clusters at both poles and across RA = 0/360
	for (ra, dec), n in ((a, 40), (b, 300), (c, 300)):
		ra[:n] = rng.uniform(0, 360, size=n); dec[:n] = 90 - np.abs(rng.normal(0, 0.3, size=n))
		ra[n:2 * n] = rng.uniform(0, 360, size=n); dec[n:2 * n] = -90 + np.abs(rng.normal(0, 0.3, size=n))
		ra[2 * n:3 * n] = rng.normal(0, 0.2, size=n) % 360; dec[2 * n:3 * n] = rng.normal(10, 0.2, size=n)
	radius = 600.
	assert orc.choose_scheme([a, b, c], radius / 3600) == orc.SPHERE
	rows = orc.enumerate_tuples([a, b, c], radius / 3600, orc.SPHERE, radius)
	# brute force
	def sep(x, i, y, j):
		return orc.dist((x[0][i], x[1][i]), (y[0][j], y[1][j])) * 60 * 60
	P, S1 = np.meshgrid(np.arange(n0), np.arange(n1), indexing='ij')
	m1 = sep(a, P, b, S1) < radius
	P, S2 = np.meshgrid(np.arange(n0), np.arange(n2), indexing='ij')
	m2 = sep(a, P, c, S2) < radius
	expect = set()
	for p in range(n0):
		l1 = [-1] + list(np.flatnonzero(m1[p]))
		l2 = [-1] + list(np.flatnonzero(m2[p]))
		for s1 in l1:
			for s2 in l2:
				if s1 >= 0 and s2 >= 0 and not sep(b, s1, c, s2) < radius:
					continue
				expect.add((p, s1, s2))
	expect = np.array(sorted(expect))
	np.testing.assert_array_equal(rows, expect)
	assert (rows[:, 1] >= 0).sum() > 50 and ((rows[:, 1] >= 0) & (rows[:, 2] >= 0)).sum() > 10


def test_script_numerics_float32_round_trip():
	"""nway.py reads the separations back from a float32 FITS column before log_bf
	squares them and before its correction loop; script_api.npz holds what the script itself computes
	on edge.npz's tables (p_i moves by up to 3e-5 relative against the float64 API: far outside the
	1e-6 contract, so the script's numerics are a mode of their own)"""
	g, e = script_golden(), golden('edge')
	tabs = [cat('ABC'[i], e['neg_ra%d' % i], e['neg_dec%d' % i], e['neg_err%d' % i], e['neg_area'][0]) for i in range(3)]
	radius = float(e['neg_radius'][0])
	for oracle in (orc, orc_c):
		kw = dict(literal_groups=True) if oracle is orc else {}
		t = oracle.nway_match(tabs, radius, e['neg_completeness'], correction='cli', f32_roundtrip=True, **kw)
		assert_table_matches(t, g, 'neg_w3_script_', ['A', 'B', 'C'], **TIGHT)
		t = oracle.nway_match(tabs[:2], radius, e['neg_completeness'][:2], correction='cli', f32_roundtrip=True, **kw)
		assert_table_matches(t, g, 'neg_w2_script_', ['A', 'B'], **TIGHT)
	api = orc.nway_match(tabs, radius, e['neg_completeness'], correction='cli')
	assert np.abs(api['prob_this_match'] - g['neg_w3_script_prob_this_match']).max() > 1e-7
	assert np.nanmax(np.abs(api['prob_this_match'] / np.maximum(g['neg_w3_script_prob_this_match'], 1e-300) - 1)[g['neg_w3_script_prob_this_match'] > 0]) > 1e-6
