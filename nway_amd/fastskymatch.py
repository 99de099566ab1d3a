"""Candidate enumeration and great-circle separations on the GPU.

Interface of nwaylib/fastskymatch.py: ``dist`` (:26-47), ``crossproduct`` (:92-218),
``get_tablekeys`` (:77-80).  The arithmetic runs in libnwayhip (k_dist, and the
register/sweep/pairs/expand kernels); this module only moves columns to the device and
mirrors argument conventions.  No CPU fallback.
"""
from __future__ import division, print_function

import ctypes

import numpy

from . import _hip


def dist(apos, bpos):
	"""Angular separation in degrees between positions (ra, dec) in degrees; scalars or
	equal-length arrays, as fastskymatch.py:26-47.  Computed in float64 on the device
	(float32 inputs are widened first; the reference would keep float32)."""
	(a_ra, a_dec), (b_ra, b_dec) = apos, bpos
	arrs = numpy.broadcast_arrays(*[numpy.asarray(x, dtype=float) for x in (a_ra, a_dec, b_ra, b_dec)])
	shape = arrs[0].shape
	device = _hip.require_device()
	t = _hip.torch()
	dev = [_hip.to_device(numpy.ascontiguousarray(a).reshape(-1), device) for a in arrs]
	n = int(dev[0].shape[0])
	out = t.empty(n, dtype=t.float64, device=device)
	_hip.check(_hip.load().nwayhip_dist(_hip.ptr(dev[0]), _hip.ptr(dev[1]), _hip.ptr(dev[2]), _hip.ptr(dev[3]), n,
		_hip.ptr(out), _hip.current_stream_ptr(device)))
	res = out.cpu().numpy().reshape(shape)
	return res if shape else float(res)


def dist3d(apos, bpos):
	"""(separation, d_ra, d_dec) in degrees: great-circle separation (device) and the
	tangent-plane offsets of b in the frame centred on a (fastskymatch.py:50-74; host numpy
	restatement of astropy's SkyOffsetFrame, parity unpinned).  -99 marks absent sources."""
	from . import elliptical
	(a_ra, a_dec), (b_ra, b_dec) = apos, bpos
	nan = lambda x: numpy.where(numpy.asarray(x, dtype=float) == -99, numpy.nan, numpy.asarray(x, dtype=float))
	separation = dist((nan(a_ra), nan(a_dec)), (nan(b_ra), nan(b_dec)))
	dra, ddec = elliptical.offsets(a_ra, a_dec, b_ra, b_dec)
	return separation, dra, ddec


def get_tablekeys(table, name, tablename=''):
	"""column of ``table`` called ``name`` (case-insensitive), else the first one starting
	with it (fastskymatch.py:77-80)"""
	names = list(table.dtype.names)
	exact = [k for k in names if k.upper() == name]
	prefixed = [k for k in names if k.upper().startswith(name)]
	found = exact or prefixed
	assert len(found) > 0, 'ERROR: No "%s" column found in input catalogue "%s". Only have: %s' % (name, tablename, ', '.join(names))
	return found[0]


def crossproduct(radectables, err, logger=None, pairwise_errs=[]):
	"""All candidate tuples, int64 (M, k), lexicographically sorted, -1 = no counterpart.

	radectables: list of (ra, dec) arrays in degrees, primary first; err: cell size /
	search radius in DEGREES (fastskymatch.py:92).

	Flat-cell inputs (:94-98): exactly the reference's pre-filter set -- every tuple whose
	present members' cells ``int(ra/err), int(dec/err)`` span <= 1 in both axes.
	All-sky inputs: the reference's HEALPix pre-filter set is not reproduced; the tuples
	returned are those that survive the subsequent radius filter (all present pairwise
	separations < err), which is what every caller keeps (__init__.py:180).
	"""
	if pairwise_errs:
		raise NotImplementedError('pairwise_errs: the reference implementation of --prefilter-pair '
			'drops every tuple containing both catalogues (fastskymatch.py:203); not reproduced')
	from . import choose_scheme, run_match
	tables = [(numpy.asarray(ra, dtype=float), numpy.asarray(dec, dtype=float)) for ra, dec in radectables]
	scheme = choose_scheme(tables, err)
	if logger is not None:
		logger.log('matching: hashing on the GPU (%s cells)' % ('flat' if scheme == _hip.SCHEME_FLAT else 'all-sky'))
	match_tables = [dict(name='T%d' % i, ra=ra, dec=dec, error=1.0, area=41252.96) for i, (ra, dec) in enumerate(tables)]
	res = run_match(match_tables, err * 60 * 60, radius_filter=(scheme != _hip.SCHEME_FLAT), finalize=False,
		scheme=scheme, logger=logger, err_deg=err)
	out = numpy.stack([res.to_host('idx', c).astype(numpy.int64) for c in range(len(tables))], axis=1)
	res.plan.close()
	if logger is not None:
		logger.log('matching: %6d unique matches from cartesian product.' % len(out))
	return out


# the reference wraps crossproduct in a joblib disk cache and exposes the raw function as .func
crossproduct.func = crossproduct
