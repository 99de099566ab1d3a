"""elliptical-error helpers of nway_amd.bayesdistance (host numpy) against values computed
by the reference (tests/golden/ellmath.npz) and the limits its own tests check
(tests/bayesdistance_test.py:34-145 of the reference)."""
import numpy as np
import pytest

from goldenutil import golden


def test_convert_and_matrices_match_reference():
	from nway_amd import bayesdistance as bd
	g = golden('ellmath')
	sx, sy, rho = bd.convert_from_ellipse(g['a'], g['b'], g['phi'])
	np.testing.assert_allclose(sx, g['sigma_x'], rtol=1e-14)
	np.testing.assert_allclose(sy, g['sigma_y'], rtol=1e-14)
	np.testing.assert_allclose(rho, g['rho'], rtol=1e-12, atol=1e-15)
	np.testing.assert_allclose(np.array(bd.make_invcovmatrix(sx, sy, rho)), g['inv'], rtol=1e-12)
	np.testing.assert_allclose(np.array(bd.make_covmatrix(sx, sy, rho)), g['cov'], rtol=1e-12)
	e2 = bd.convert_from_ellipse(g['a2'], g['b2'], g['phi2'])
	got = bd.apply_vABv(g['v'], bd.make_invcovmatrix(sx, sy, rho), bd.make_invcovmatrix(*e2))
	np.testing.assert_allclose(got, g['vABv'], rtol=1e-12)
	assert (got >= 0).all()
	bd.assert_possemdef(bd.make_invcovmatrix(sx, sy, rho))
	bd.assert_possemdef(bd.matrix_add(bd.make_invcovmatrix(sx, sy, rho), bd.make_invcovmatrix(*e2)))
	with pytest.raises(AssertionError):
		bd.assert_possemdef(((np.array([1.0]), np.array([3.0])), (np.array([3.0]), np.array([1.0]))))


def test_ellipse_limits():
	from nway_amd import bayesdistance as bd
	rng = np.random.RandomState(1)
	s1, s2, ang = rng.uniform(1, 100, 100), rng.uniform(1, 100, 100), rng.uniform(0, 180, 100)
	sx, sy, rho = bd.convert_from_ellipse(s1, s1, ang)  # circular
	np.testing.assert_allclose(rho, 0, atol=1e-12)
	np.testing.assert_allclose(sx, s1)
	np.testing.assert_allclose(sy, s1)
	sx, sy, rho = bd.convert_from_ellipse(s1, s2, 0)  # aligned
	np.testing.assert_allclose(rho, 0, atol=1e-12)
	np.testing.assert_allclose(sy, s1)
	np.testing.assert_allclose(sx, s2)
	sx, sy, rho = bd.convert_from_ellipse(s1, s2, np.pi / 2)  # rotated by 90 degrees
	np.testing.assert_allclose(sx, s1)
	np.testing.assert_allclose(sy, s2)
	# inverse really inverts, determinant multiplies
	A = bd.make_covmatrix(s1, s2, 0.3)
	I = bd.matrix_multiply(A, bd.matrix_invert(A))
	np.testing.assert_allclose(I[0][0], 1)
	np.testing.assert_allclose(I[0][1], 0, atol=1e-9)
	np.testing.assert_allclose(bd.matrix_det(bd.make_invcovmatrix(s1, s2, 0.3)), 1 / bd.matrix_det(A), rtol=1e-10)
	u = bd.vector_normalised((np.array([3.0, 0.0]), np.array([4.0, 0.0])))
	np.testing.assert_allclose(u[0], [0.6, 2**-0.5])
	np.testing.assert_allclose(bd.apply_vector_left((1.0, 2.0), ((1.0, 2.0), (3.0, 4.0))), (7.0, 10.0))
	np.testing.assert_allclose(bd.apply_vector_right(((1.0, 2.0), (3.0, 4.0)), (1.0, 2.0)), (5.0, 11.0))


@pytest.mark.gpu
def test_log_bf_elliptical_gpu():
	from nway_amd import bayesdistance as bd
	g = golden('ellmath')
	n = len(g['a'])
	nan = np.nan * np.ones(n)
	e1 = bd.convert_from_ellipse(g['a'], g['b'], g['phi'])
	e2 = bd.convert_from_ellipse(g['a2'], g['b2'], g['phi2'])
	e3 = bd.convert_from_ellipse(g['a3'], g['b3'], g['phi3'])
	dra, ddec = g['dra'], g['ddec']
	got = bd.log_bf_elliptical([[nan, dra[0], dra[1]], [nan, nan, dra[2]], [nan, nan, nan]],
		[[nan, ddec[0], ddec[1]], [nan, nan, ddec[2]], [nan, nan, nan]], [e1, e2, e3])
	np.testing.assert_allclose(got, g['log_bf_ell3'], rtol=1e-9)
	got = bd.log_bf_elliptical([[nan, dra[0]], [nan, nan]], [[nan, ddec[0]], [nan, nan]], [e1, e2])
	np.testing.assert_allclose(got, g['log_bf_ell2'], rtol=1e-9)
	# elliptical reduces to circular (tests/bayesdistance_test.py:149-203 of the reference)
	one = np.ones(1)
	ell = bd.log_bf_elliptical([[None, one]], [[None, 0 * one]], [bd.convert_from_ellipse(0.1 * one, 0.1 * one, 0), bd.convert_from_ellipse(0.2 * one, 0.2 * one, 0)])
	np.testing.assert_allclose(ell, bd.log_bf([[None, one]], [0.1 * one, 0.2 * one]), rtol=1e-9)


@pytest.mark.gpu
def test_cli_elliptical_columns_reduce_to_circular(tmp_path, monkeypatch):
	"""`:a:b:phi` and `:ra_err:dec_err` error columns (nway.py:52-88): with a == b the elliptical
	run must reproduce the circular one; the per-axis separation columns appear"""
	from goldenutil import ell_tables
	from nway_amd import _fits, cli
	monkeypatch.chdir(tmp_path)
	X, R, O = ell_tables()
	rng = np.random.RandomState(4)
	for t in (X, R):
		n = len(t['ra'])
		e = t['error']
		_fits.write_table('%s.fits' % t['name'], [('ID', 'J', np.arange(1, n + 1)), ('RA', 'D', t['ra']), ('DEC', 'D', t['dec']),
			('pos_err', 'D', e), ('a', 'D', e), ('b', 'D', e), ('phi', 'D', rng.uniform(0, 180, n)),
			('a2', 'D', e * 1.5), ('b2', 'D', e * 0.7)], t['name'], table_header={'SKYAREA': t['area']})
	n = len(O['ra'])
	_fits.write_table('OPT.fits', [('ID', 'J', np.arange(1, n + 1)), ('RA', 'D', O['ra']), ('DEC', 'D', O['dec'])], 'OPT', table_header={'SKYAREA': O['area']})
	base = ['--radius', '10', '--min-prob', '0.01']
	assert cli.main(base + ['CHANDRA.fits', ':pos_err', 'XMM.fits', ':pos_err', 'OPT.fits', '0.1', '--out', 'circ.fits']) == 0
	assert cli.main(base + ['CHANDRA.fits', ':a:b:phi', 'XMM.fits', ':a:b', 'OPT.fits', '0.1', '--out', 'ell.fits']) == 0
	circ, ell = _fits.read_table('circ.fits'), _fits.read_table('ell.fits')
	for c in ('Separation_OPT_CHANDRA_ra', 'Separation_OPT_CHANDRA_dec', 'Separation_OPT_XMM_ra', 'Separation_XMM_CHANDRA_dec'):
		assert c in ell.names
	assert len(circ.data) == len(ell.data)
	np.testing.assert_array_equal(circ.data['OPT_ID'], ell.data['OPT_ID'])
	np.testing.assert_array_equal(circ.data['match_flag'], ell.data['match_flag'])
	for c in ('dist_bayesfactor', 'dist_bayesfactor_corrected', 'dist_post', 'p_any', 'p_i'):
		np.testing.assert_allclose(ell.data[c], circ.data[c], rtol=2e-4, atol=1e-6, err_msg=c)
	# per-axis offsets are consistent with the great-circle separation
	both = ell.data['OPT_ID'] != -99
	sep = np.hypot(ell.data['Separation_OPT_CHANDRA_ra'][both], ell.data['Separation_OPT_CHANDRA_dec'][both])
	np.testing.assert_allclose(sep, ell.data['Separation_OPT_CHANDRA'][both], rtol=1e-4, atol=1e-4)
	# genuinely elliptical errors run and change the answer
	assert cli.main(base + ['CHANDRA.fits', ':a2:b2:phi', 'OPT.fits', '0.1', '--out', 'ell2.fits']) == 0
	e2 = _fits.read_table('ell2.fits')
	assert len(e2.data) > 100 and np.isfinite(e2.data['p_i']).all()


def test_elliptical_oracle_against_reference_values():
	"""oracle/elliptical_oracle.py (numpy restatement) against values computed with the reference"""
	import os
	import sys
	sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
	import elliptical_oracle as eo
	from nway_amd import bayesdistance as bd
	g = golden('ellmath')
	nan = np.nan * np.ones(len(g['a']))
	e = [bd.convert_from_ellipse(g['a' + s], g['b' + s], g['phi' + s]) for s in ('', '2', '3')]
	dra, ddec = g['dra'], g['ddec']
	got = eo.log_bf_elliptical([[nan, dra[0], dra[1]], [nan, nan, dra[2]], [nan, nan, nan]],
		[[nan, ddec[0], ddec[1]], [nan, nan, ddec[2]], [nan, nan, nan]], e)
	np.testing.assert_allclose(got, g['log_bf_ell3'], rtol=1e-12)
	got = eo.log_bf_elliptical([[nan, dra[0]], [nan, nan]], [[nan, ddec[0]], [nan, nan]], e[:2])
	np.testing.assert_allclose(got, g['log_bf_ell2'], rtol=1e-12)
	# offsets: along a meridian / the equator they are the coordinate differences; their length
	# is the great-circle separation for small offsets
	lon, lat = eo.offsets(np.array([10.0, 10.0]), np.array([0.0, 20.0]), np.array([10.5, 10.0]), np.array([0.0, 20.25]))
	np.testing.assert_allclose(lon, [-0.5, 0.0], atol=1e-12)
	np.testing.assert_allclose(lat, [0.0, -0.25], atol=1e-12)
	assert np.isnan(eo.offsets(np.array([-99.0]), np.array([-99.0]), np.array([1.0]), np.array([1.0]))[0]).all()


@pytest.mark.gpu
def test_offsets_gpu_against_oracle():
	"""nwayhip_offsets (the two offset columns of dist3d) against the numpy restatement, over the
	whole sphere incl. the poles, the seam and absent sources"""
	import os
	import sys
	sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
	import elliptical_oracle as eo
	from nway_amd import elliptical, fastskymatch
	rng = np.random.RandomState(8)
	n = 5000
	a_ra = rng.uniform(0, 360, n)
	a_dec = np.degrees(np.arcsin(rng.uniform(-1, 1, n)))
	b_ra = a_ra + rng.normal(0, 1, n) * 10**rng.uniform(-6, 1.5, n)
	b_dec = np.clip(a_dec + rng.normal(0, 1, n) * 10**rng.uniform(-6, 1.5, n), -90, 90)
	a_dec[:3] = [90, -90, 89.9999]
	a_ra[3], b_ra[3] = 359.9999, 0.0001
	b_ra[4] = b_dec[4] = -99
	a_ra[5] = a_dec[5] = -99
	lon, lat = elliptical.offsets(a_ra, a_dec, b_ra, b_dec)
	elon, elat = eo.offsets(a_ra, a_dec, b_ra, b_dec)
	np.testing.assert_allclose(lon, elon, rtol=1e-9, atol=1e-12, equal_nan=True)
	np.testing.assert_allclose(lat, elat, rtol=1e-9, atol=1e-12, equal_nan=True)
	assert np.isnan(lon[4]) and np.isnan(lat[5])
	# small offsets: their length is the great-circle separation (dist3d's first column)
	sep, dra, ddec = fastskymatch.dist3d((a_ra[6:], a_dec[6:]), (b_ra[6:], b_dec[6:]))
	small = sep < 0.01
	np.testing.assert_allclose(np.hypot(dra[small], ddec[small]), sep[small], rtol=1e-5)
	np.testing.assert_array_equal(dra, lon[6:])


@pytest.mark.gpu
def test_elliptical_correction_against_row_by_row_oracle():
	"""nway_amd.elliptical.unrelated_associations (one device evaluation per sub-association, per
	primary maxima) against the script's row-by-row loop restated in oracle/elliptical_oracle.py,
	on a 4-way table with random ellipses"""
	import os
	import sys
	sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
	import elliptical_oracle as eo
	import nway_amd
	from nway_amd import elliptical
	from goldenutil import cat
	g = golden('kway')
	k = 4
	tabs = [cat('T%d' % i, g['k4c_ra%d' % i], g['k4c_dec%d' % i], g['k4c_err%d' % i], g['k4c_area'][0]) for i in range(k)]
	res = nway_amd.run_match(tabs, 25., 0.7, logger=nway_amd.NullOutputLogger())
	idx = [res.to_host('idx', c).astype(np.int64) for c in range(k)]
	ncat = res.to_host('ncat').astype(np.int64)
	nrows = len(ncat)
	rng = np.random.RandomState(12)
	sep_ra = [[None] * k for _ in range(k)]
	sep_dec = [[None] * k for _ in range(k)]
	for i in range(k):
		for j in range(i):
			def coords(c):
				return np.where(idx[c] >= 0, tabs[c]['ra'][idx[c]], -99.), np.where(idx[c] >= 0, tabs[c]['dec'][idx[c]], -99.)
			dra, ddec = elliptical.offsets(*(coords(i) + coords(j)))
			sep_ra[j][i], sep_dec[j][i] = dra * 3600, ddec * 3600
	errors = []
	for c in range(k):
		n = len(tabs[c]['ra'])
		a, b, phi = rng.uniform(0.5, 3, n), rng.uniform(0.5, 3, n), rng.uniform(0, np.pi, n)
		sx, sy, rho = nway_amd.bayesdist.convert_from_ellipse(a, b, phi)
		pick = lambda v: np.where(idx[c] >= 0, v[idx[c]], -99.)
		errors.append((pick(sx), pick(sy), np.where(idx[c] >= 0, rho[idx[c]], 0.0)))
	dens = np.array([len(t['ra']) / t['area'] * 41252.96 for t in tabs])
	dens_plus = np.array([(len(t['ra']) + 1) / t['area'] * 41252.96 for t in tabs])
	dens_plus[0] = dens[0]
	base = elliptical.log_bf_table(k, idx, sep_ra, sep_dec, errors)
	assert np.isfinite(base).all()
	got = elliptical.unrelated_associations(k, idx, ncat, sep_ra, sep_dec, errors, dens, dens_plus, base)
	want = eo.unrelated_associations(k, idx, ncat, sep_ra, sep_dec, errors, dens, dens_plus, base)
	assert (want != base).sum() > 100
	np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9)
	res.plan.close()


@pytest.mark.gpu
def test_offsets_known_answers():
	"""nwayhip_offsets against answers worked out by hand from the rotation astropy documents for
	SkyOffsetFrame (the origin goes to (0, 0), no roll: R = R_y(-dec_a) R_z(ra_a)), i.e. NOT taken
	from oracle/elliptical_oracle.py.  For b seen from a: x' = cos(dec_a) cos(dec_b) cos(dra) +
	sin(dec_a) sin(dec_b), y' = cos(dec_b) sin(dra), z' = -sin(dec_a) cos(dec_b) cos(dra) +
	cos(dec_a) sin(dec_b); lon' = atan2(y', x'), lat' = asin(z'); dist3d returns (0 - lon', 0 - lat')
	(fastskymatch.py:62-67)."""
	from nway_amd import elliptical
	cases = [
		# a_ra, a_dec, b_ra, b_dec, d_lon, d_lat
		(10.0, 20.0, 10.0, 21.0, 0.0, -1.0),      # same meridian: x' = cos 1, z' = sin 1
		(30.0, 0.0, 31.5, 0.0, -1.5, 0.0),        # on the equator: lon' = 1.5
		(0.0, 89.0, 180.0, 89.0, 0.0, -2.0),      # across the pole: x' = cos 2, z' = sin 2 -- straight north by 2 deg
		(359.5, 0.0, 0.5, 0.0, -1.0, 0.0),        # across the RA seam
		(0.5, 0.0, 359.5, 0.0, 1.0, 0.0),
		(0.0, 0.0, 90.0, 45.0, -90.0, -45.0),     # x' = 0, y' = cos 45, z' = sin 45
		(0.0, 90.0, 0.0, 89.0, 0.0, 1.0),         # from the pole: z' = -cos 89 -> lat' = -1
		(200.0, -35.0, 200.0, -35.0, 0.0, 0.0),   # the same point
		(120.0, 60.0, 300.0, -60.0, 180.0, 0.0),  # the antipode: x' = -1, y' = 0 -> lon' = 180 (either sign of 180 is the same angle)
	]
	a_ra, a_dec, b_ra, b_dec, want_lon, want_lat = [np.array(c) for c in zip(*cases)]
	lon, lat = elliptical.offsets(a_ra, a_dec, b_ra, b_dec)
	np.testing.assert_allclose(lat, want_lat, rtol=0, atol=1e-11)
	dlon = (lon - want_lon + 180.0) % 360.0 - 180.0
	np.testing.assert_allclose(dlon, 0.0, rtol=0, atol=1e-11)
	# absent sources (-99) give NaN in both columns (fastskymatch.py:55-58)
	lon, lat = elliptical.offsets(np.array([-99.0, 10.0]), np.array([-99.0, 0.0]), np.array([10.0, -99.0]), np.array([0.0, -99.0]))
	assert np.isnan(lon).all() and np.isnan(lat).all()
	# an offset of one arcsecond due east at declination 60: lon' = atan2(cos 60 sin(dra), ...) -> dra cos(dec) to first order
	lon, lat = elliptical.offsets(15.0, 60.0, 15.0 + 1 / 3600., 60.0)
	assert abs(lon * 3600 + 0.5) < 1e-6 and abs(lat * 3600) < 1e-5


def _ellflow_inputs(g):
	k = 3
	cats = [dict((col, g['in%d_%s' % (c, col)]) for col in ('ra', 'dec', 'major', 'minor', 'angle')) for c in range(k)]
	return k, cats, float(g['radius'][0]), float(g['completeness'][0]), float(g['area'][0])


def test_elliptical_oracle_reproduces_the_scripts_flow():
	"""oracle/elliptical_oracle.py: script_flow against tests/golden/ell_flow.npz -- the script's
	``:major:minor:angle`` branch (nway.py:52-88, 303-305, 327-360, 366-420) evaluated with the
	REFERENCE's functions (make_golden.py: gen_ellflow; dist3d's two astropy calls replaced by the
	documented rotation: pins the branch logic and its float32 / float64 numerics, not astropy)"""
	import os
	import sys
	sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
	import elliptical_oracle as eo
	g = golden('ell_flow')
	k, cats, radius, completeness, area = _ellflow_inputs(g)
	idx = [g['idx'][:, c].astype(np.int64) for c in range(k)]
	sizes = [len(c['ra']) for c in cats]
	dens = np.array([n / area * (4 * np.pi * (180 / np.pi)**2) for n in sizes])
	dens_plus = np.array([(n + 1) / area * (4 * np.pi * (180 / np.pi)**2) for n in sizes])
	dens_plus[0] = dens[0]
	comp = np.array([1.0] + [completeness**(1. / (k - 1))] * (k - 1))
	sep_ra, sep_dec, unc, cor, prior = eo.script_flow(k, idx, g['ncat'].astype(np.int64), [(c['ra'], c['dec']) for c in cats],
		[(c['major'], c['minor'], c['angle']) for c in cats], dens, dens_plus, comp)
	for i in range(k):
		for j in range(i):
			np.testing.assert_array_equal(sep_ra[j][i], g['off_ra_%d_%d' % (j, i)])
			np.testing.assert_array_equal(sep_dec[j][i], g['off_dec_%d_%d' % (j, i)])
	np.testing.assert_allclose(unc, g['dist_bayesfactor_uncorrected'], rtol=1e-12, atol=1e-12)
	np.testing.assert_allclose(cor, g['dist_bayesfactor'], rtol=1e-12, atol=1e-12)
	np.testing.assert_array_equal(np.flatnonzero(cor != unc), g['changed_rows'])
	assert len(g['changed_rows']) > 20


@pytest.mark.gpu
def test_cli_elliptical_flow_golden(tmp_path, monkeypatch):
	"""nway.py with ``:major:minor:angle`` error columns on three catalogues, FITS in, FITS out, against
	tests/golden/ell_flow.npz (see the oracle test above for what the fixture pins): rows, offsets,
	uncorrected and corrected Bayes factors, posteriors, group statistics, match_flag"""
	from nway_amd import _fits, cli
	g = golden('ell_flow')
	k, cats, radius, completeness, area = _ellflow_inputs(g)
	names = ['X', 'O', 'I']
	monkeypatch.chdir(tmp_path)
	argv = ['--radius', '%g' % radius, '--prior-completeness', '%g' % completeness]
	for name, c in zip(names, cats):
		n = len(c['ra'])
		_fits.write_table('%s.fits' % name, [('ID', 'J', np.arange(n)), ('RA', 'D', c['ra']), ('DEC', 'D', c['dec']),
			('major', 'D', c['major']), ('minor', 'D', c['minor']), ('angle', 'D', c['angle'])], name, table_header={'SKYAREA': area})
		argv += ['%s.fits' % name, ':major:minor:angle']
	assert cli.main(argv + ['--out', 'out.fits']) == 0
	d = _fits.read_table('out.fits').data
	m = len(g['ncat'])
	assert len(d) == m
	for c, name in enumerate(names):
		ids = np.asarray(d[name + '_ID'], dtype=np.int64)
		np.testing.assert_array_equal(np.where(ids == -99, -1, ids), g['idx'][:, c])
	np.testing.assert_array_equal(np.asarray(d['ncat'], dtype=np.int64), g['ncat'])
	f32 = lambda x: np.asarray(x, dtype=np.float32)
	for i in range(k):
		for j in range(i):
			for axis, key in (('ra', 'off_ra_%d_%d'), ('dec', 'off_dec_%d_%d')):
				got = np.asarray(d['Separation_%s_%s_%s' % (names[i], names[j], axis)], dtype=float)
				# (float32 columns of offsets that the device and numpy evaluate with different libm: one float32 ulp, or 1e-6 arcsec near zero)
				np.testing.assert_allclose(got, g[key % (j, i)].astype(float), rtol=2.4e-7, atol=1e-6, equal_nan=True, err_msg=key % (j, i))
	# the table's floating columns are float32 ('E'): the contract's 1e-6 relative, on top of the float32 rounding
	for col, key in (('dist_bayesfactor', 'dist_bayesfactor_uncorrected'), ('dist_bayesfactor_corrected', 'dist_bayesfactor'), ('dist_post', 'dist_post'),
			('p_single', 'p_single'), ('p_any', 'prob_has_match'), ('p_i', 'prob_this_match')):
		np.testing.assert_allclose(np.asarray(d[col], dtype=float), f32(g[key]).astype(float), rtol=2e-6, atol=1e-9, err_msg=col)
	np.testing.assert_array_equal(np.asarray(d['match_flag'], dtype=np.int64), g['match_flag'])
	changed = np.flatnonzero(np.asarray(d['dist_bayesfactor_corrected']) != np.asarray(d['dist_bayesfactor']))
	assert len(changed) > 20 and set(changed) <= set(g['changed_rows'])


def _astropyconsistent_inputs():
	"""the positions of the reference's test_dist_astropyconsistent (tests/fastskymatch_test.py:74-82): 3 998 declinations between 0 and 90,
	a shift of 1e-5 degrees in right ascension"""
	dec = np.linspace(0, 90, 4000)[1:-1]
	ra = np.zeros_like(dec)
	return ra, dec, ra + 1e-5, dec + 0.0


def test_offsets_of_the_oracle_as_the_reference_tests_astropys():
	"""tests/fastskymatch_test.py:74-106 asserts of astropy's offset frame that the length of (na.lon - nb.lon, na.lat - nb.lat) equals the
	separation to 7 decimals (degrees); the same assertion on the stand-in's rotation (oracle/elliptical_oracle.py: offsets) with the
	separation of the oracle's dist -- a property of the reference's own test that the restatement of dist3d has to have"""
	import os
	import sys
	sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
	import elliptical_oracle as eo
	import nway_oracle as orc
	a_ra, a_dec, b_ra, b_dec = _astropyconsistent_inputs()
	dra, ddec = eo.offsets(a_ra, a_dec, b_ra, b_dec)
	d = orc.dist((a_ra, a_dec), (b_ra, b_dec))
	np.testing.assert_array_almost_equal(d, (dra**2 + ddec**2)**0.5, decimal=7)
	# (and far beyond 7 decimals: at 1e-5 degrees the tangent plane is exact to ~1e-15 relative)
	np.testing.assert_allclose((dra**2 + ddec**2)**0.5, d, rtol=1e-9)
	assert (dra < 0).all()  # a minus b, b to the east


@pytest.mark.gpu
def test_offsets_on_the_device_as_the_reference_tests_astropys():
	"""the same assertion (tests/fastskymatch_test.py:74-106) on nwayhip_offsets and the device's dist"""
	import nway_amd
	from nway_amd import elliptical
	a_ra, a_dec, b_ra, b_dec = _astropyconsistent_inputs()
	dra, ddec = elliptical.offsets(a_ra, a_dec, b_ra, b_dec)
	d = nway_amd.fastskymatch.dist((a_ra, a_dec), (b_ra, b_dec))
	np.testing.assert_array_almost_equal(d, (dra**2 + ddec**2)**0.5, decimal=7)
	np.testing.assert_allclose((dra**2 + ddec**2)**0.5, d, rtol=1e-9)
