"""Budavari & Szalay (2008) N-way Bayes factors and posteriors, evaluated on the GPU.

Interface of nwaylib/bayesdistance.py.  ``log_bf`` (:64-86), ``posterior`` (:26-32),
``log_posterior`` (:18-23) and ``unnormalised_log_posterior`` (:35-39) run as HIP
kernels (k_log_bf, k_posterior in csrc/nwayhip.hip); there is no CPU fallback for them.
``log_bf2`` / ``log_bf3`` are the closed-form 2- and 3-catalogue expressions (eq. 16/17
of the paper) that the reference keeps as independent cross-checks of ``log_bf``
(tests/bayesdistance_test.py:12-32); they stay independent host formulas here for the
same purpose.  Everything is log10; separations and errors in arcsec.
"""
from __future__ import division, print_function

import ctypes

import numpy
from numpy import e, log, log10, pi

from . import _hip

# ln of the number of arcsec per radian
log_arcsec2rad = log(3600 * 180 / pi)


def _on_device(arrays):
	"""broadcast host scalars/arrays to one shape, upload as flat float64 tensors"""
	arrs = numpy.broadcast_arrays(*[numpy.asarray(a, dtype=float) for a in arrays])
	shape = arrs[0].shape
	device = _hip.require_device()
	dev = [_hip.to_device(numpy.ascontiguousarray(a).reshape(-1), device) for a in arrs]
	return dev, shape, device


def _finish(out, shape):
	res = _hip.to_host(out).reshape(shape)
	return res if shape else float(res)


def _posterior_like(mode, prior, log_bf_value):
	(pr, lb), shape, device = _on_device([prior, log_bf_value])
	t = _hip.torch()
	n = int(pr.shape[0])
	out = t.empty(n, dtype=t.float64, device=device)
	_hip.check(_hip.load().nwayhip_posterior(mode, _hip.ptr(pr), _hip.ptr(lb), n, _hip.ptr(out), _hip.current_stream_ptr(device)))
	return _finish(out, shape)


def posterior(prior, log_bf):
	"""posterior probability against the hypothesis that the sources are unrelated"""
	return _posterior_like(0, prior, log_bf)


def log_posterior(prior, log_bf):
	"""log10 of ``posterior``"""
	return _posterior_like(1, prior, log_bf)


def unnormalised_log_posterior(prior, log_bf, ncat):
	"""log_bf + log10(prior); ``ncat`` is accepted and ignored, as in the reference"""
	return _posterior_like(2, prior, log_bf)


def log_bf2(psi, s1, s2):
	"""closed form for two catalogues (eq. 16): separation psi, errors s1, s2"""
	var = s1 * s1 + s2 * s2
	return (log(2) + 2 * log_arcsec2rad - log(var) - psi * psi / 2 / var) * log10(e)


def log_bf3(p12, p23, p31, s1, s2, s3):
	"""closed form for three catalogues (eq. 17)"""
	v1, v2, v3 = s1 * s1, s2 * s2, s3 * s3
	det = v1 * v2 + v2 * v3 + v3 * v1
	quad = v3 * p12**2 + v1 * p23**2 + v2 * p31**2
	return (log(4) + 4 * log_arcsec2rad - log(det) - quad / 2 / det) * log10(e)


def log_bf(p, s):
	"""log10 multi-way Bayes factor (eq. 18).

	p: n x n nested sequence of separations; only entries with i < j are read (the rest may
	be None / NaN), s: sequence of n positional errors.  Entries broadcast against each other.
	"""
	n = len(s)
	if n < 1 or n > _hip.MAXCAT:
		raise ValueError('log_bf supports 1..%d catalogues' % _hip.MAXCAT)
	pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
	(dev, shape, device) = _on_device([s[i] for i in range(n)] + [p[i][j] for i, j in pairs])
	t = _hip.torch()
	nrows = int(dev[0].shape[0])
	sig = (ctypes.c_void_p * n)(*[d.data_ptr() for d in dev[:n]])
	sep = (ctypes.c_void_p * (n * n))()
	for (i, j), d in zip(pairs, dev[n:]):
		sep[i * n + j] = d.data_ptr()
	out = t.empty(nrows, dtype=t.float64, device=device)
	_hip.check(_hip.load().nwayhip_log_bf(n, nrows, sep, sig, _hip.ptr(out), _hip.current_stream_ptr(device)))
	return _finish(out, shape)


# ---------------------------------------------------------------------------------------
# Elliptical / asymmetric position errors (reference: bayesdistance.py:92-240).
# The 2x2 matrix helpers (entries may be arrays: one matrix per table row) are part of the
# reference's importable surface and stay plain numpy expressions; ``log_bf_elliptical`` itself --
# rescaling each separation with the error along its direction, then the circular formula -- is
# one device kernel.
# ---------------------------------------------------------------------------------------

def assert_possemdef(M):
	"""raise AssertionError unless the symmetric 2x2 matrix M (entries may be arrays) is
	positive semi-definite"""
	(a, b), (_, d) = M
	tr = a + d
	det = a * d - b * b
	degenerate = numpy.isclose(tr**2, 4 * det)
	if numpy.all(degenerate):
		return
	disc = tr**2 - 4 * det
	assert not numpy.any(numpy.logical_and(~degenerate, disc < 0)), (tr, det, M)
	root = numpy.sqrt(numpy.where(degenerate, 0, disc))
	for ev in ((tr + root) / 2, (tr - root) / 2):
		assert numpy.all(numpy.where(degenerate, 0, ev) >= 0), ('negative eigenvalue', ev, M)


def matrix_add(A, B):
	return tuple(tuple(x + y for x, y in zip(ra, rb)) for ra, rb in zip(A, B))


def matrix_multiply(A, B):
	(a, b), (c, d) = A
	(e, f), (g, h) = B
	return (a * e + b * g, a * f + b * h), (c * e + d * g, c * f + d * h)


def matrix_det(A):
	(a, b), (c, d) = A
	return a * d - b * c


def matrix_invert(A):
	(a, b), (c, d) = A
	scale = 1.0 / matrix_det(A)
	assert numpy.all(scale > 0)
	return (scale * d, -scale * b), (-scale * c, scale * a)


def apply_vector_right(A, b):
	"""A b"""
	(a11, a12), (a21, a22) = A
	return a11 * b[0] + a12 * b[1], a21 * b[0] + a22 * b[1]


def apply_vector_left(a, B):
	"""a^T B"""
	(b11, b12), (b21, b22) = B
	return a[0] * b11 + a[1] * b21, a[0] * b12 + a[1] * b22


def vector_multiply(a, b):
	return a[0] * b[0] + a[1] * b[1]


def vector_normalised(v):
	"""unit vector along v; (1, 1)/sqrt(2) where v vanishes"""
	length = (v[0]**2 + v[1]**2)**0.5
	return tuple(numpy.where(length == 0, 2**-0.5, comp / (length + 1e-300)) for comp in v)


def apply_vABv(v, A, B):
	"""v^T (A + B) v"""
	return vector_multiply(v, apply_vector_right(matrix_add(A, B), v))


def make_covmatrix(sigma_x, sigma_y, rho=0):
	off = rho * sigma_x * sigma_y
	return (sigma_x**2, off), (off, sigma_y**2)


def make_invcovmatrix(sigma_x, sigma_y, rho=0):
	scale = 1.0 / (sigma_x**2 * sigma_y**2 * (1 - rho**2))
	off = scale * -rho * sigma_x * sigma_y
	return (scale * sigma_y**2, off), (off, scale * sigma_x**2)


def convert_from_ellipse(a, b, phi):
	"""(sigma_x, sigma_y, rho) of an error ellipse with semi-axes a, b rotated by phi
	(radians), e.g. Pineau+16 eq. 8-10"""
	s2, c2 = numpy.sin(phi)**2, numpy.cos(phi)**2
	sigma_x = (a**2 * s2 + b**2 * c2)**0.5
	sigma_y = (a**2 * c2 + b**2 * s2)**0.5
	rho = numpy.cos(phi) * numpy.sin(phi) * (a**2 - b**2) / (sigma_x * sigma_y)
	return sigma_x, sigma_y, rho


def log_bf_elliptical(separations_ra, separations_dec, pos_errors, f32_offsets=None):
	"""log10 Bayes factor for elliptical errors: pos_errors = list of (sigma_ra, sigma_dec, rho)
	per catalogue; separations given per axis (n x n nested sequences, entries with i < j read).
	Each pair's separation is rescaled by the ratio of the circularised to the directional error,
	then the circular formula applies (bayesdistance.py:207-240).  One device kernel
	(``nwayhip_log_bf_elliptical``); entries broadcast against each other.
	f32_offsets: None = like numpy, i.e. float32 arithmetic for the length and the unit vector of the
	offsets exactly when every offset entry given is a float32 array (what the script hands over from
	its FITS 'E' columns, nway.py:303-305, 346-354); True / False force it."""
	n = len(pos_errors)
	if n < 1 or n > _hip.MAXCAT:
		raise ValueError('log_bf_elliptical supports 1..%d catalogues' % _hip.MAXCAT)
	pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
	columns = [pos_errors[c][m] for c in range(n) for m in range(3)]
	offs = [separations_ra[i][j] for i, j in pairs] + [separations_dec[i][j] for i, j in pairs]
	if f32_offsets is None:
		f32_offsets = len(offs) > 0 and all(getattr(o, 'dtype', None) == numpy.float32 for o in offs)
	columns += offs
	(dev, shape, device) = _on_device(columns)
	t = _hip.torch()
	nrows = int(dev[0].shape[0])
	sx = (ctypes.c_void_p * n)(*[dev[3 * c].data_ptr() for c in range(n)])
	sy = (ctypes.c_void_p * n)(*[dev[3 * c + 1].data_ptr() for c in range(n)])
	rho = (ctypes.c_void_p * n)(*[dev[3 * c + 2].data_ptr() for c in range(n)])
	sra = (ctypes.c_void_p * (n * n))()
	sdec = (ctypes.c_void_p * (n * n))()
	for m, (i, j) in enumerate(pairs):
		sra[i * n + j] = dev[3 * n + m].data_ptr()
		sdec[i * n + j] = dev[3 * n + len(pairs) + m].data_ptr()
	out = t.empty(nrows, dtype=t.float64, device=device)
	_hip.check(_hip.load().nwayhip_log_bf_elliptical(n, nrows, sra, sdec, sx, sy, rho, _hip.ptr(out), 1 if f32_offsets else 0, _hip.current_stream_ptr(device)))
	return _finish(out, shape)
