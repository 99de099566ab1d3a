"""Asymmetric and elliptical position errors for the command-line driver
(``file :ra_err:dec_err`` and ``file :major:minor:angle``; nway.py:52-88, 303-305, 346-354,
402-411, with ``dist3d`` of fastskymatch.py:50-74 and ``log_bf_elliptical`` of
bayesdistance.py:207-240).

The candidate enumeration, separations, radius filter and priors are the device pipeline's
as for circular errors; only the Bayes factor differs: per pair the tangent-plane offsets
(d_ra, d_dec) are rescaled by the error along their direction and the circular formula is
evaluated on the device (``bayesdistance.log_bf``).

Parity status: UNPINNED.  The reference computes the offsets with astropy's SkyOffsetFrame,
which is absent here (and unpinned upstream); ``offsets`` restates that frame (rotation of the
sphere that puts the first position at the origin) and is checked for self-consistency only.
"""
from __future__ import division, print_function

import numpy

from . import bayesdistance as bayesdist


def offsets(a_ra, a_dec, b_ra, b_dec):
	"""(d_lon, d_lat) in degrees of position b seen from the offset frame centred on a, with the
	sign convention of fastskymatch.py:66-67 (a minus b).  -99 marks an absent source -> NaN."""
	a_ra = numpy.where(a_ra == -99, numpy.nan, numpy.asarray(a_ra, dtype=float))
	a_dec = numpy.where(a_dec == -99, numpy.nan, numpy.asarray(a_dec, dtype=float))
	b_ra = numpy.where(b_ra == -99, numpy.nan, numpy.asarray(b_ra, dtype=float))
	b_dec = numpy.where(b_dec == -99, numpy.nan, numpy.asarray(b_dec, dtype=float))
	dlon = numpy.radians(b_ra - a_ra)
	lat0, lat = numpy.radians(a_dec), numpy.radians(b_dec)
	x1 = numpy.cos(lat) * numpy.cos(dlon)
	y1 = numpy.cos(lat) * numpy.sin(dlon)
	z1 = numpy.sin(lat)
	x = x1 * numpy.cos(lat0) + z1 * numpy.sin(lat0)
	z = -x1 * numpy.sin(lat0) + z1 * numpy.cos(lat0)
	with numpy.errstate(invalid='ignore'):
		lon_b = numpy.degrees(numpy.arctan2(y1, x))
		lat_b = numpy.degrees(numpy.arcsin(numpy.clip(z, -1, 1)))
	return -lon_b, -lat_b


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
		log_bf[mask] = bayesdist.log_bf_elliptical(sra, sdec, errs)
	return log_bf


def unrelated_associations(k, idx_columns, ncat, sep_ra, sep_dec, errors, dens, dens_plus, log_bf):
	"""the script's correction (nway.py:366-420) with the elliptical Bayes factor, vectorised
	over the rows of each primary"""
	log_bf = log_bf.copy()
	prim = idx_columns[0]
	starts = numpy.flatnonzero(numpy.r_[True, prim[1:] != prim[:-1]])
	ends = numpy.r_[starts[1:], len(prim)]
	present = numpy.stack([idx >= 0 for idx in idx_columns], axis=1)
	for lo, hi in zip(starts, ends):
		rows = numpy.arange(lo, hi)
		cand = rows[ncat[rows] <= k - 2]
		rich = rows[ncat[rows] > 2]
		if len(cand) == 0 or len(rich) == 0:
			continue
		for i in cand:
			missing = [c for c in range(1, k) if not present[i, c]]
			# group the richer rows by which of the missing catalogues they contain
			sub = present[rich][:, missing]
			best = 0.0
			for pat in numpy.unique(sub, axis=0):
				aug = [c for c, on in zip(missing, pat) if on]
				if len(aug) < 2:
					continue
				sel = rich[(sub == pat).all(axis=1)]
				sra = [[sep_ra[a][b][sel] if a < b else None for b in aug] for a in aug]
				sdec = [[sep_dec[a][b][sel] if a < b else None for b in aug] for a in aug]
				errs = [tuple(e[sel] for e in errors[c]) for c in aug]
				lb = numpy.atleast_1d(bayesdist.log_bf_elliptical(sra, sdec, errs))
				logpost = lb + numpy.log10(dens[aug[0]] / numpy.prod(dens_plus[aug]))
				best = max(best, float(logpost.max()))
			if best > 0:
				log_bf[i] += best
	return log_bf
