"""host-side histogram logic (nway_amd.magnitudeweights) against values produced by the
reference's scipy-based implementation (tests/golden/mag.npz)"""
import numpy as np

from goldenutil import golden, mag_tables


def test_adaptive_histograms_match_reference():
	from nway_amd import magnitudeweights as mw
	g = golden('mag')
	bins, hs, ha = mw.adaptive_histograms(g['ah_all'], g['ah_sel'], weights=g['ah_w'])
	np.testing.assert_allclose(bins, g['ah_bins'], rtol=1e-14)
	np.testing.assert_allclose(hs, g['ah_hist_sel'], rtol=1e-12)
	np.testing.assert_allclose(ha, g['ah_hist_all'], rtol=1e-12)


def test_step_function_matches_interp1d_zero():
	from nway_amd import magnitudeweights as mw
	g = golden('mag')
	f = mw.fitfunc_histogram(g['ah_bins'], g['ah_hist_sel'], g['ah_hist_all'])
	np.testing.assert_array_equal(f(g['ff_x']), g['ff_y'])
	assert np.isnan(f(np.array([g['ah_bins'][0] - 1e-9, g['ah_bins'][-1] + 1e-9, np.nan]))).all()
	np.testing.assert_array_equal(mw.ratio([1., 2., 0.], [2., 0., 0.]), [0.5, 100., 100.])


def test_adaptive_histograms_with_zero_weights():
	"""repeated knots in the cumulative weight axis (weights that are exactly 0): the quantile
	borders follow numpy.interp, which scipy's interp1d delegates to"""
	from nway_amd import magnitudeweights as mw
	g = golden('mag3')
	# ah1: a real selection whose running weight sum passes the total by an ulp before the last
	# knot -- interp1d sorts its knots before evaluating
	for case in ('ah0', 'ah1'):
		bins, hs, ha = mw.adaptive_histograms(g[case + '_all'], g[case + '_sel'], weights=g[case + '_w'])
		np.testing.assert_array_equal(bins, g[case + '_bins'])
		np.testing.assert_allclose(hs, g[case + '_hist_sel'], rtol=1e-13)
		np.testing.assert_allclose(ha, g[case + '_hist_all'], rtol=1e-13)


def test_mag3_tables_regenerate_exactly():
	from goldenutil import mag3_tables
	X, O, I = mag3_tables()  # asserts the stored checksums
	assert len(O['mags']) == 2 and len(I['mags']) == 1


def test_mag_tables_regenerate_exactly():
	X, O = mag_tables()
	assert len(O['ra']) == 120000 and np.isnan(O['mags'][0]).sum() == 500


def test_fraction_known_answer():
	"""magnitudeweights.py:26-42 (a helper nothing calls; expected values from the reference run
	in the build container)"""
	from nway_amd import magnitudeweights as mw
	edges = np.array([10, 11, 12.5, 13, 15.])
	got = mw.fraction(edges, np.array([.1, .5, .3, .1]), np.array([.2, 0, .4, .4]))
	np.testing.assert_allclose(got, [4 / 3., 1.0, 2.0, 2 / 3.], rtol=1e-14)
