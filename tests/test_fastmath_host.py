"""nway_amd/csrc/fastmath.inc on the host (CPU test): the short roads of the row kernels' elementary functions against 50-digit
arithmetic.  The functions are IEEE operations in a fixed order, so what is checked here is what the kernels compute
(tests/test_fastmath.py compares the device with this build bit for bit).  Reference: numpy.sin / cos / arctan2 / hypot of
fastskymatch.py:26-47, numpy.log / log10 / 10**x of bayesdistance.py:18-86."""
import numpy as np
import pytest

mp = pytest.importorskip('mpmath')  # (50-digit arithmetic; comes with the image's sympy)

from fastmath_util import arguments, host_eval

mp.mp.dps = 50


def ulps(got, exact):
	worst = 0.0
	for g, w in zip(got, exact):
		w = mp.mpf(w)
		if w == 0:
			assert g == 0
			continue
		u = mp.mpf(float(np.spacing(abs(float(w))))) if abs(w) > 3e-308 else mp.mpf(5e-324)
		worst = max(worst, float(abs(mp.mpf(float(g)) - w) / u))
	return worst


def long_road(got, want):
	"""the long road is the C library's function (the device library's on the GPU): numpy's own, vectorised one within an ulp"""
	with np.errstate(all='ignore'):
		same = (got == want) | (np.isnan(got) & np.isnan(want)) | (np.abs(got - want) <= np.spacing(np.abs(want)))
	assert same.all()


def test_sincos_short_road():
	x = arguments(4000)[0][0]
	x = x[np.isfinite(x) & (np.abs(x) <= 1.6e6)]
	s, c = host_eval(0, x)
	es = [mp.sin(mp.mpf(float(v))) for v in x]
	ec = [mp.cos(mp.mpf(float(v))) for v in x]
	# two-term reduction: absolute error below 2e-26 |n| + an ulp of the result -- a relative ulp except within ~1e-9 of a multiple of pi/2
	for got, exact in ((s, es), (c, ec)):
		for g, w, v in zip(got, exact, x):
			tol = float(np.spacing(abs(float(w)))) + 2e-26 * (1 + abs(v))
			assert abs(mp.mpf(float(g)) - w) <= tol, (v, g, w)
	away = np.array([abs(float(mp.sin(2 * mp.mpf(float(v))))) > 1e-6 for v in x])  # not next to a multiple of pi/2
	assert ulps(s[away], np.array(es, dtype=object)[away]) < 1.0
	assert ulps(c[away], np.array(ec, dtype=object)[away]) < 1.0
	assert host_eval(0, [0.0])[0][0] == 0.0 and host_eval(0, [0.0])[1][0] == 1.0


def test_sincos_beyond_the_domain_is_nan():
	s, c = host_eval(0, [1.7e6, -1e300, np.inf, np.nan])
	assert np.isnan(s).all() and np.isnan(c).all()


def test_atan2_and_hypot():
	y, x = arguments(4000)[1]
	short = (y >= 0) & (y * 16 <= x) & (x > 0) & np.isfinite(x) & np.isfinite(y)
	got = host_eval(1, y, x)
	assert ulps(got[short], [mp.atan2(mp.mpf(float(a)), mp.mpf(float(b))) for a, b in zip(y[short], x[short])]) < 1.0
	long_road(got[~short], np.arctan2(y[~short], x[~short]))
	a, b = arguments(4000)[2]
	fin = np.isfinite(a) & np.isfinite(b)
	got = host_eval(2, a, b)
	assert ulps(got[fin], [mp.sqrt(mp.mpf(float(p)) ** 2 + mp.mpf(float(q)) ** 2) for p, q in zip(a[fin], b[fin])]) < 1.3
	long_road(got[~fin], np.hypot(a[~fin], b[~fin]))


@pytest.mark.parametrize('fn', [3, 4])
def test_logarithms(fn):
	x = arguments(4000)[fn][0]
	pos = (x > 0) & np.isfinite(x)
	got = host_eval(fn, x)
	f = mp.log if fn == 3 else mp.log10
	assert ulps(got[pos], [f(mp.mpf(float(v))) for v in x[pos]]) < 1.0
	with np.errstate(all='ignore'):
		want = (np.log if fn == 3 else np.log10)(x[~pos])
	np.testing.assert_array_equal(got[~pos], want)  # -inf, inf, NaN as numpy has them
	if fn == 4:
		assert (host_eval(4, [10.0, 100.0, 1e-5, 1.0]) == [1.0, 2.0, -5.0, 0.0]).all()


def test_exp10():
	x = arguments(4000)[5][0]
	fin = np.isfinite(x) & (x > -307) & (x < 308.25)
	got = host_eval(5, x)
	assert ulps(got[fin], [mp.power(10, mp.mpf(float(v))) for v in x[fin]]) < 1.2
	with np.errstate(all='ignore'):
		want = 10.0 ** x[~fin]
	rest = got[~fin]
	with np.errstate(all='ignore'):  # overflow, underflow, NaN as numpy has them; the smallest results within two steps
		assert ((rest == want) | (np.isnan(rest) & np.isnan(want)) | (np.abs(rest - want) <= 2 * np.spacing(np.abs(want)))).all()
	assert (host_eval(5, [0.0, 1.0, 2.0, 3.0, 22.0]) == [1.0, 10.0, 100.0, 1000.0, 1e22]).all()


def test_divisions_by_a_literal_are_the_true_quotients():
	"""x / 180 * pi and x * 180 / pi in three operations per division (fastmath.inc: nw_div_k): numpy's own expressions, bit for bit"""
	x = arguments(500000, seed=8)[6][0]
	got = host_eval(6, x)
	want = x / 180 * np.pi
	assert ((got.view(np.uint64) == want.view(np.uint64)) | (np.isnan(got) & np.isnan(want))).all()
	x = arguments(500000, seed=9)[7][0]
	got = host_eval(7, x)
	want = x * 180 / np.pi
	assert ((got.view(np.uint64) == want.view(np.uint64)) | (np.isnan(got) & np.isnan(want))).all()
