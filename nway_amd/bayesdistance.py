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
	res = out.cpu().numpy().reshape(shape)
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
