"""Candidate enumeration and great-circle separations on the GPU.

Interface of nwaylib/fastskymatch.py: ``dist`` (:26-47), ``crossproduct`` (:92-218),
``get_tablekeys`` (:77-80).  The arithmetic runs in libnwayhip (k_dist, and the
register/sweep/pairs/expand kernels); this module only moves columns to the device and
mirrors argument conventions.  No CPU fallback.
"""
from __future__ import division, print_function


import numpy

from . import _hip


def dist(apos, bpos):
	"""Angular separation in degrees between positions (ra, dec) in degrees; scalars or
	equal-length arrays, as fastskymatch.py:26-47.  The dtype follows the inputs like numpy's
	(SURVEY A.8): where numpy's promotion of the four arguments gives float32 -- float32 arrays,
	possibly next to Python scalars, which are weak -- the separation is evaluated in float32 and
	returned as float32 (k_dist_f32); anything else in float64."""
	(a_ra, a_dec), (b_ra, b_dec) = apos, bpos
	device = _hip.require_device()
	t = _hip.torch()
	args = (a_ra, a_dec, b_ra, b_dec)
	# numpy.result_type treats Python scalars as weak (NEP 50); 0-d arrays and numpy scalars carry their own dtype
	promoted = numpy.result_type(*[x if type(x) in (int, float) else numpy.asarray(x) for x in args])
	given = [numpy.asarray(x) for x in args]
	if promoted == numpy.float32 and any(g.ndim > 0 for g in given):
		arrs = numpy.broadcast_arrays(*[g.astype(numpy.float32) for g in given])
		shape = arrs[0].shape
		dev = [_hip.to_device(numpy.ascontiguousarray(a).reshape(-1), device, t.float32) for a in arrs]
		n = int(dev[0].shape[0])
		out = t.empty(n, dtype=t.float32, device=device)
		_hip.check(_hip.load().nwayhip_dist_f32(_hip.ptr(dev[0]), _hip.ptr(dev[1]), _hip.ptr(dev[2]), _hip.ptr(dev[3]), n,
			_hip.ptr(out), _hip.current_stream_ptr(device)))
		return _hip.to_host(out).reshape(shape)
	arrs = numpy.broadcast_arrays(*[numpy.asarray(x, dtype=float) for x in (a_ra, a_dec, b_ra, b_dec)])
	shape = arrs[0].shape
	dev = [_hip.to_device(numpy.ascontiguousarray(a).reshape(-1), device) for a in arrs]
	n = int(dev[0].shape[0])
	out = t.empty(n, dtype=t.float64, device=device)
	_hip.check(_hip.load().nwayhip_dist(_hip.ptr(dev[0]), _hip.ptr(dev[1]), _hip.ptr(dev[2]), _hip.ptr(dev[3]), n,
		_hip.ptr(out), _hip.current_stream_ptr(device)))
	res = _hip.to_host(out).reshape(shape)
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


def get_healpix_resolution_degrees(nside):
	"""0.7 x the mean pixel spacing sqrt(4 pi / (12 nside^2)) of a HEALPix map, in degrees: the
	largest search radius the reference trusts a pixel and its neighbours to contain
	(fastskymatch.py:83-88).  Kept for callers of the module; the device pipeline hashes all-sky
	catalogues into its own declination-band cells and never pixelises with HEALPix."""
	return 0.7 * (numpy.sqrt(numpy.pi / 3.) / nside) / numpy.pi * 180


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


# ---------------------------------------------------------------------------------------
# FITS-flavoured front end used by the script (fastskymatch.py:222-363).  astropy is not a
# dependency: columns are light ``Column`` records and files are written by nway_amd._fits.
# ---------------------------------------------------------------------------------------

class Column(object):
	"""name / FITS format / array triple (the subset of ``pyfits.Column`` the callers use)"""

	def __init__(self, name, format, array):
		self.name = name
		self.format = format
		self.array = numpy.asarray(array)

	def __repr__(self):
		return 'Column(%r, %r, %d rows)' % (self.name, self.format, len(self.array))


class TableHDU(object):
	"""what ``fits_from_columns`` returns: ``.data`` is a structured array, ``.columns`` the
	Column list, ``.header`` a dict; ``writeto`` stores it as the first extension of a FITS file"""

	def __init__(self, columns, extname=''):
		from . import _fits
		self.columns = list(columns)
		self.header = {'EXTNAME': extname}
		self.primary_header = {}
		dtype = []
		for c in self.columns:
			dtype.append((c.name, numpy.dtype({'E': 'f4', 'D': 'f8', 'I': 'i2', 'J': 'i4', 'K': 'i8', 'L': 'b1', 'B': 'u1'}.get(c.format[-1], c.array.dtype))))
		self.data = numpy.zeros(len(self.columns[0].array) if self.columns else 0, dtype=dtype)
		for c in self.columns:
			with numpy.errstate(invalid='ignore', over='ignore'):
				self.data[c.name] = c.array
		self._fits = _fits

	@property
	def name(self):
		return self.header.get('EXTNAME', '')

	def writeto(self, filename, overwrite=True, **kwargs):
		extra = dict((k, v) for k, v in self.header.items() if k != 'EXTNAME')
		self._fits.write_table(filename, [(c.name, c.format, c.array) for c in self.columns], self.name,
			primary_header=self.primary_header, table_header=extra, overwrite=overwrite)


def fits_from_columns(columns):
	return TableHDU(columns)


def wraptable2fits(cat_columns, extname):
	"""table HDU named ``extname`` from Columns (or an existing TableHDU)"""
	hdu = cat_columns if isinstance(cat_columns, TableHDU) else TableHDU(cat_columns)
	hdu.header['EXTNAME'] = extname
	return hdu


def array2fits(table, extname):
	"""structured array -> table HDU with every column stored as 32-bit float"""
	return wraptable2fits([Column(n, 'E', table[n]) for n in table.dtype.names], extname)


def match_multiple(tables, table_names, err, fits_formats, logger=None, circular=True, pairwise_errs=[]):
	"""Cartesian product of all possible matches within ``err`` DEGREES of each other
	(fastskymatch.py:228-342): returns (results, cat_columns, header) where results is a
	structured index array with one field per table (-1 = absent), cat_columns holds every
	input column as ``{table}_{column}`` (-99 where absent), the ``Separation_{later}_{earlier}``
	columns in arcsec ('E'), ``Separation_max`` and ``ncat``, and header names the coordinate
	columns.  ``circular=False`` adds the per-axis ``_ra`` / ``_dec`` offsets of ``dist3d``."""
	from . import run_match, NullOutputLogger
	logger = logger or NullOutputLogger()
	if pairwise_errs:
		raise NotImplementedError('pairwise_errs is not supported (see crossproduct)')
	logger.log('')
	logger.log('matching with %f arcsec radius' % (err * 60 * 60))
	logger.log('matching: %6d naive possibilities' % numpy.prod([float(len(t)) for t in tables]))
	ra_keys = [get_tablekeys(t, 'RA', tablename=n) for t, n in zip(tables, table_names)]
	dec_keys = [get_tablekeys(t, 'DEC', tablename=n) for t, n in zip(tables, table_names)]
	logger.log('    using RA  columns: %s' % ', '.join(ra_keys))
	logger.log('    using DEC columns: %s' % ', '.join(dec_keys))
	match_tables = [dict(name=n, ra=numpy.asarray(t[rk], dtype=float), dec=numpy.asarray(t[dk], dtype=float), error=1.0, area=41252.96)
		for t, n, rk, dk in zip(tables, table_names, ra_keys, dec_keys)]
	res = run_match(match_tables, err * 60 * 60, finalize=False, logger=logger, err_deg=err)
	k = len(tables)
	idx = [res.to_host('idx', c).astype(numpy.int64) for c in range(k)]
	results = numpy.zeros(res.nrows, dtype=[(n, numpy.int64) for n in table_names])
	cat_columns = []
	for t, n, fmts, ix in zip(tables, table_names, fits_formats, idx):
		results[n] = ix
		rows = t[ix]
		for colname, fmt in zip(t.dtype.names, fmts):
			col = numpy.array(rows[colname])
			try:
				col[ix == -1] = -99
			except Exception as e:
				logger.log('   setting "%s_%s" to -99 failed (%d affected; column format "%s"): %s' % (n, colname, (ix == -1).sum(), fmt, e))
			cat_columns.append(Column('%s_%s' % (n, colname), fmt, col))
	header = dict(COLS_RA=' '.join('%s_%s' % (n, rk) for n, rk in zip(table_names, ra_keys)),
		COLS_DEC=' '.join('%s_%s' % (n, dk) for n, dk in zip(table_names, dec_keys)))
	logger.log('    adding angular separation columns')
	pair_index = dict((p, i) for i, p in enumerate(_hip.pair_columns(k)))
	for i in range(k):
		for j in range(i):
			name = 'Separation_%s_%s' % (table_names[i], table_names[j])
			cat_columns.append(Column(name, 'E', res.to_host('sep', pair_index[(j, i)])))
			if not circular:
				from . import elliptical
				coords = lambda c: (numpy.where(idx[c] >= 0, match_tables[c]['ra'][idx[c]], -99.), numpy.where(idx[c] >= 0, match_tables[c]['dec'][idx[c]], -99.))
				dra, ddec = elliptical.offsets(*(coords(i) + coords(j)))
				cat_columns.append(Column(name + '_ra', 'E', dra * 60 * 60))
				cat_columns.append(Column(name + '_dec', 'E', ddec * 60 * 60))
	cat_columns.append(Column('Separation_max', 'E', res.to_host('sep_max')))
	cat_columns.append(Column('ncat', 'I', res.to_host('ncat').astype(numpy.int64)))
	logger.log('matching: %6d matches after filtering by search radius' % res.nrows)
	logger.log('')
	res.plan.close()
	return results, cat_columns, header
