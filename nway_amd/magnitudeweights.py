"""Adaptive histograms of a property (typically a magnitude) for "secure counterparts"
versus "everything else", and the step function that turns their ratio into a prior weight.

Interface of nwaylib/magnitudeweights.py (``ratio`` :18-23, ``fitfunc_histogram`` :74-87,
``adaptive_histograms`` :90-118, ``plot_fit`` :46-71).  Host-side numpy: this runs once per
magnitude column on catalogue-sized arrays and is not part of the device hot path; the
per-row lookup of the resulting step function is the HIP kernel ``k_bias``
(``nwayhip_bias_lookup``).  scipy is not needed: the two interpolants the reference builds
with ``scipy.interpolate.interp1d`` are restated with numpy (searchsorted).
"""
from __future__ import division, print_function

import numpy


def ratio(hist_sel, hist_all):
	"""selected / all per bin; bins without any "all" entries get the large weight 100"""
	hist_sel = numpy.asarray(hist_sel, dtype=float)
	hist_all = numpy.asarray(hist_all, dtype=float)
	with numpy.errstate(divide='ignore', invalid='ignore'):
		return numpy.where(hist_all == 0, 100, hist_sel / hist_all)


def fraction(bin_mag, hist_sel, hist_all):
	"""selected / all per bin, rescaled so that its mean over the bins with data (weighted by bin
	width x "all" density) is 1; bins without data get 1 (magnitudeweights.py:26-42 of the
	reference, where nothing calls it either)"""
	bin_mag = numpy.asarray(bin_mag, dtype=float)
	hist_sel = numpy.asarray(hist_sel, dtype=float)
	hist_all = numpy.asarray(hist_all, dtype=float)
	filled = hist_all > 0
	per_bin = hist_sel[filled] / hist_all[filled]
	weight = numpy.diff(bin_mag)[filled] * hist_all[filled]
	mean = (per_bin * weight).sum() / weight.sum()
	assert mean != 0
	out = numpy.ones(len(hist_all))
	out[filled] = per_bin / mean
	assert numpy.isfinite(out).all() and (out > 0).all(), out
	return out


class StepFunction(object):
	"""Zero-order interpolant over bin edges, like
	``interp1d(edges, list(values) + [values[-1]], kind='zero', bounds_error=False)``:
	value of the bin that contains x (left-closed), the last value at x == edges[-1],
	NaN outside [edges[0], edges[-1]] and for NaN input."""

	def __init__(self, edges, values):
		self.edges = numpy.asarray(edges, dtype=float)
		self.values = numpy.asarray(values, dtype=float)
		if len(self.values) != len(self.edges) - 1:
			raise ValueError('need one value per bin')

	def __call__(self, x):
		x = numpy.asarray(x, dtype=float)
		out = numpy.full(x.shape, numpy.nan)
		with numpy.errstate(invalid='ignore'):
			inside = (x >= self.edges[0]) & (x <= self.edges[-1])
		b = numpy.searchsorted(self.edges, x[inside], side='right') - 1
		out[inside] = self.values[numpy.minimum(b, len(self.values) - 1)]
		return out


def fitfunc_histogram(bin_mag, hist_sel, hist_all):
	"""the biasing function: ratio of the two histograms as a step function of the property"""
	return StepFunction(bin_mag, ratio(hist_sel, hist_all))


def _linear_interpolant(x, y, x_new):
	"""piecewise-linear y(x_new) exactly as ``interp1d(x, y)`` evaluates it for 1-D float data:
	scipy first orders the knots by x with a stable sort (``assume_sorted=False``) and then hands
	the evaluation to ``numpy.interp``.  Both steps matter for the cumulative weight axis of
	``adaptive_histograms``: a running sum divided by the total can exceed 1 by an ulp just before
	the last knot (which is set to exactly 1), so the knots are NOT sorted, and weights that
	are 0 (or lost to rounding) repeat knots, of which numpy.interp takes the last.  A plain
	searchsorted restatement moved the uppermost bin border of one magnitude column of a 3-way
	match and, through it, one match_flag of 62 960."""
	x = numpy.asarray(x, dtype=float)
	y = numpy.asarray(y, dtype=float)
	order = numpy.argsort(x, kind='mergesort')
	return numpy.interp(numpy.asarray(x_new, dtype=float), x[order], y[order])


def adaptive_histograms(mag_all, mag_sel, weights=None):
	"""Density histograms of ``mag_sel`` (weighted) and ``mag_all`` on common bins whose borders
	are 15 quantiles of the weighted ``mag_sel`` distribution, extended by one bin on either
	side if ``mag_all`` reaches further.  Returns (bins, hist_sel, hist_all)."""
	mag_all = numpy.asarray(mag_all, dtype=float)
	mag_sel = numpy.asarray(mag_sel, dtype=float)
	if weights is None:
		weights = numpy.ones(len(mag_sel))
	weights = numpy.asarray(weights, dtype=float)
	assert len(weights) == len(mag_sel), (len(weights), len(mag_sel))
	order = numpy.argsort(mag_sel)
	sel_sorted = mag_sel[order]
	cumulative = numpy.cumsum(weights[order]) / numpy.sum(weights)
	cumulative[0] = 0
	cumulative[-1] = 1
	borders = numpy.unique(_linear_interpolant(cumulative, sel_sorted, numpy.linspace(0, 1, 15)))
	lo, hi = numpy.nanmin(mag_all), numpy.nanmax(mag_all)
	if borders[-1] < hi:
		borders = numpy.asarray(list(borders) + [hi + 1])
	if borders[0] > lo:
		borders = numpy.asarray([lo - 1] + list(borders))
	hist_sel, bins = numpy.histogram(mag_sel, bins=borders, density=True, weights=weights)
	hist_all, bins = numpy.histogram(mag_all, bins=bins, density=True)
	return bins, hist_sel, hist_all


def plot_fit(bin_mag, hist_sel, hist_all, func, name):
	"""Diagnostic plot ``<name>_fit.pdf`` of the two histograms and their ratio.  Cosmetic;
	skipped silently when matplotlib is unavailable."""
	try:
		import matplotlib
		matplotlib.use('Agg')
		import matplotlib.pyplot as plt
	except Exception:
		return
	bin_mag = numpy.asarray(bin_mag)
	grid = numpy.linspace(bin_mag.min(), bin_mag.max(), 400)
	fig, (top, bottom) = plt.subplots(2, 1)
	top.step(bin_mag[:-1], hist_all, where='post', label='all')
	top.step(bin_mag[:-1], hist_sel, where='post', label='selected')
	top.legend(loc='best')
	top.set_ylabel('normalized weight')
	top.set_xlim(grid.min(), grid.max())
	bottom.step(bin_mag[:-1], ratio(hist_sel, hist_all), where='post', label='ratio')
	bottom.plot(grid, func(grid), '-', label='fit')
	bottom.legend(loc='best')
	bottom.set_ylabel('normalized weight')
	bottom.set_xlabel(name)
	bottom.set_xlim(grid.min(), grid.max())
	bottom.set_yscale('log')
	fig.savefig(name.replace(':', '_') + '_fit.pdf', bbox_inches='tight')
	plt.close(fig)
