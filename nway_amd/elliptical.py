"""Asymmetric and elliptical position errors for the command-line driver
(``file :ra_err:dec_err`` and ``file :major:minor:angle``; nway.py:52-88, 303-305, 346-354,
402-411, with ``dist3d`` of fastskymatch.py:50-74 and ``log_bf_elliptical`` of
bayesdistance.py:207-240).

The candidate enumeration, separations, radius filter and priors are the device pipeline's
as for circular errors; only the Bayes factor differs: per pair the tangent-plane offsets
(d_ra, d_dec; device kernel ``nwayhip_offsets``) are rescaled by the error along their direction
and the circular formula is evaluated (device kernel ``nwayhip_log_bf_elliptical``).  This module
only selects rows per presence pattern and gathers the error columns.

Parity status: ``log_bf_elliptical`` pinned (tests/golden/ellmath.npz); the offsets UNPINNED --
the reference computes them with astropy's SkyOffsetFrame, which is absent here (and unpinned
upstream); ``nwayhip_offsets`` is the rotation of the sphere that puts the first position at the
origin, checked against oracle/elliptical_oracle.py and for consistency with the separation.
"""
from __future__ import division, print_function

import numpy

from . import _hip
from . import bayesdistance as bayesdist


def offsets(a_ra, a_dec, b_ra, b_dec):
	"""(d_lon, d_lat) in degrees of position b seen from the offset frame centred on a, with the
	sign convention of fastskymatch.py:66-67 (a minus b).  -99 marks an absent source -> NaN.
	Device kernel ``nwayhip_offsets``."""
	arrs = numpy.broadcast_arrays(*[numpy.asarray(x, dtype=float) for x in (a_ra, a_dec, b_ra, b_dec)])
	shape = arrs[0].shape
	device = _hip.require_device()
	t = _hip.torch()
	dev = [_hip.to_device(numpy.ascontiguousarray(a).reshape(-1), device) for a in arrs]
	n = int(dev[0].shape[0])
	d_lon = t.empty(n, dtype=t.float64, device=device)
	d_lat = t.empty(n, dtype=t.float64, device=device)
	_hip.check(_hip.load().nwayhip_offsets(_hip.ptr(dev[0]), _hip.ptr(dev[1]), _hip.ptr(dev[2]), _hip.ptr(dev[3]), n,
		_hip.ptr(d_lon), _hip.ptr(d_lat), _hip.current_stream_ptr(device)))
	lon, lat = _hip.to_host(d_lon).reshape(shape), _hip.to_host(d_lat).reshape(shape)
	return (lon, lat) if shape else (float(lon), float(lat))


def error_triplets(tables, table_names, pos_errors, idx_columns):
	"""(sigma_ra, sigma_dec, rho) per catalogue on the rows of the match table
	(nway.py:25-98): fixed value, ``:col``, ``:ra_err:dec_err`` or ``:major:minor:angle``."""
	out = []
	nrows = len(idx_columns[0])
	for t, name, spec, idx in zip(tables, table_names, pos_errors, idx_columns):
		if spec[0] != ':':
			e = float(spec) * numpy.ones(nrows)
			out.append((e, e, numpy.zeros(nrows)))
			continue
		keys = spec[1:].split(':')
		absent = idx < 0

		def column(key):
			col = numpy.array(numpy.asarray(t.data[key], dtype=float)[idx])
			col[absent] = -99
			return col
		if len(keys) == 3:
			angle = (column(keys[2]) - 90) / 180 * numpy.pi
			out.append(bayesdist.convert_from_ellipse(column(keys[0]), column(keys[1]), angle))
		elif len(keys) == 2:
			out.append((column(keys[0]), column(keys[1]), numpy.zeros(nrows)))
		else:
			c = column(keys[0])
			out.append((c, c, numpy.zeros(nrows)))
	return out


def log_bf_table(k, idx_columns, sep_ra, sep_dec, errors):
	"""log10 Bayes factor of every table row: for each presence pattern the elliptical formula
	on the present catalogues (nway.py:330-360)"""
	nrows = len(idx_columns[0])
	log_bf = numpy.zeros(nrows) * numpy.nan
	present = [idx >= 0 for idx in idx_columns]
	for pattern in range(1 << (k - 1)):
		cats = [0] + [c for c in range(1, k) if (pattern >> (c - 1)) & 1]
		mask = numpy.ones(nrows, dtype=bool)
		for c in range(1, k):
			mask &= present[c] if c in cats else ~present[c]
		if not mask.any():
			continue
		if len(cats) == 1:
			log_bf[mask] = 0.0
			continue
		sra = [[sep_ra[a][b][mask] if a < b else None for b in cats] for a in cats]
		sdec = [[sep_dec[a][b][mask] if a < b else None for b in cats] for a in cats]
		errs = [tuple(e[mask] for e in errors[c]) for c in cats]
		log_bf[mask] = bayesdist.log_bf_elliptical(sra, sdec, errs)  # (float32 offsets: numpy's float32 length and unit vector)
	return log_bf


def unrelated_associations(k, idx_columns, ncat, sep_ra, sep_dec, errors, dens, dens_plus, log_bf):
	"""the script's correction (nway.py:366-420) with the elliptical Bayes factor: a row lacking two
	or more catalogues (ncat <= k - 2) gains max(0, best), the best log posterior of a
	sub-association of its MISSING catalogues found among the richer rows (ncat > 2) of the same
	primary.  One device evaluation per sub-association pattern over all rows that contain it;
	the host only selects rows and takes per-primary maxima."""
	log_bf = log_bf.copy()
	nrows = len(ncat)
	prim = idx_columns[0]
	starts = numpy.flatnonzero(numpy.r_[True, prim[1:] != prim[:-1]])
	group = numpy.repeat(numpy.arange(len(starts)), numpy.diff(numpy.r_[starts, nrows]))
	code = numpy.zeros(nrows, dtype=numpy.int64)  # bit c-1: secondary catalogue c is present
	for c in range(1, k):
		code |= (idx_columns[c] >= 0).astype(numpy.int64) << (c - 1)
	everything = (1 << (k - 1)) - 1
	rich = ncat > 2
	cand = ncat <= k - 2
	if not cand.any() or not rich.any():
		return log_bf
	value = {}  # sub-association -> its log posterior on the rich rows that contain it
	for sub in range(1, everything + 1):
		cats = [c for c in range(1, k) if (sub >> (c - 1)) & 1]
		sel = numpy.flatnonzero(rich & ((code & sub) == sub))
		if len(cats) < 2 or len(sel) == 0:
			continue
		sra = [[sep_ra[a][b][sel] if a < b else None for b in cats] for a in cats]
		sdec = [[sep_dec[a][b][sel] if a < b else None for b in cats] for a in cats]
		errs = [tuple(e[sel] for e in errors[c]) for c in cats]
		v = numpy.full(nrows, -numpy.inf)
		# (the script gathers one row's offsets into a numpy.array next to float64 NaN placeholders, nway.py:402-408:
		# float32 VALUES, float64 arithmetic)
		v[sel] = numpy.atleast_1d(bayesdist.log_bf_elliptical(sra, sdec, errs, f32_offsets=False)) + numpy.log10(dens[cats[0]] / numpy.prod(dens_plus[cats]))
		value[sub] = v
	missing = everything & ~code
	for pattern in numpy.unique(missing[cand]):
		shared = code & pattern  # what a richer row can contribute to a row with this pattern
		u = numpy.full(nrows, -numpy.inf)
		for sub in numpy.unique(shared[rich]):
			if int(sub) in value:
				rows = rich & (shared == sub)
				u[rows] = value[int(sub)][rows]
		best = numpy.maximum.reduceat(u, starts)
		rows = cand & (missing == pattern)
		gain = best[group[rows]]
		log_bf[rows] += numpy.where(gain > 0, gain, 0.0)
	return log_bf
