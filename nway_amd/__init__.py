"""nway_amd -- MI355X-native implementation of nway's match-probability hot path.

Drop-in for ``nwaylib`` (JohannesBuchner/nway v4.7.1): same function names, arguments,
output columns and exceptions as nwaylib/__init__.py:31-120, with the candidate
enumeration, separations, Bayes factors, posteriors and per-primary statistics computed
by hand-written HIP kernels (nway_amd/csrc/nwayhip.hip) through a C ABI
(include/nwayhip.h).  There is no CPU fallback: without the HIP library or a GPU every
compute call raises ``NwayHipError``.

File:line citations in this package point into the reference checkout.
"""
from __future__ import division, print_function

from collections import OrderedDict

import numpy

from . import _hip
from ._hip import NwayHipError
from .logger import NormalLogger, NullOutputLogger

__version__ = '4.7.1'
__hip_backend__ = 'libnwayhip/gfx950 ABI %d' % _hip.ABI_VERSION


class UndersampledException(Exception):
	pass


class EmptyResultException(Exception):
	pass


default_logger = NormalLogger()

AREA_TOTAL = 4 * numpy.pi * (180 / numpy.pi)**2


# ---------------------------------------------------------------------------------------
# host-side scalars of the path (cheap, O(k) or O(N) once)
# ---------------------------------------------------------------------------------------

def choose_scheme(radectables, err):
	"""Flat-cell vs all-sky decision of fastskymatch.py:94-98 (made on the host columns)."""
	for ra, dec in radectables:
		ra = numpy.asarray(ra)
		dec = numpy.asarray(dec)
		if not (err < 1 and bool((ra > 10 * err).all()) and bool((ra < 360 - 10 * err).all()) and bool((numpy.abs(dec) < 45).all())):
			return _hip.SCHEME_SPHERE
	return _hip.SCHEME_FLAT


def _densities_from_sizes(names, sizes, areas, logger):
	"""nu_c = N_c / area_c * whole sky, nu+_c = (N_c + 1) / area_c * whole sky, nu+_0 = nu_0."""
	dens, dens_plus = [], []
	for i, (name, n, area) in enumerate(zip(names, sizes, areas)):
		area = area * 1.0
		density = n / area * AREA_TOTAL
		logger.log('%s "%s" (%d), density gives %.2e objects on entire sky' % ('Primary catalogue' if i == 0 else 'Catalogue', name, n, density))
		dens.append(density)
		dens_plus.append((n + 1) / area * AREA_TOTAL)
	dens_plus[0] = dens[0]
	return numpy.array(dens), numpy.array(dens_plus)


def _compute_source_densities(match_tables, logger):
	"""nu_c and nu+_c of the catalogues, __init__.py:199-217."""
	return _densities_from_sizes([t['name'] for t in match_tables], [len(t['ra']) for t in match_tables],
		[t['area'] for t in match_tables], logger)


def _completeness_vector(prior_completeness, ncats):
	"""scalar c -> [1, c**(1/(k-1)), ...]; vectors are checked (__init__.py:224-229)."""
	if numpy.shape(prior_completeness) == ():
		prior_completeness = numpy.array([1.0] + [float(prior_completeness)**(1. / (ncats - 1)) for _ in range(1, ncats)])
	prior_completeness = numpy.asarray(prior_completeness, dtype=float)
	if len(prior_completeness) != ncats:
		raise Exception('Prior completeness needs one value per catalog. Received "%s".' % prior_completeness)
	assert prior_completeness[0] == 1.0
	return prior_completeness


def _prior_table(dens, dens_plus, completeness):
	"""prior of every presence pattern (bit c-1 <=> catalogue c present), __init__.py:254."""
	ncats = len(dens)
	table = numpy.zeros(1 << (ncats - 1))
	for pattern in range(1 << (ncats - 1)):
		mask = numpy.array([True] + [bool((pattern >> (c - 1)) & 1) for c in range(1, ncats)])
		table[pattern] = dens[0] * numpy.prod(completeness[mask]) / numpy.prod(dens_plus[mask])
	assert numpy.isfinite(table).all(), (dens, dens_plus, completeness)
	return table


def _estimate_capacities(sizes, areas, radius_arcsec, scheme, radius_filter):
	"""first guess of the link / row capacities from the catalogue densities; the engine
	reports exact needs and the run is repeated if the guess was too small"""
	n0 = sizes[0]
	r_deg = radius_arcsec / 3600.
	lam = []
	for n, area in zip(sizes[1:], areas[1:]):
		patch = (numpy.pi if radius_filter else 9.0) * r_deg**2
		lam.append(n * min(1.0, patch / max(area, 1e-12)))
	cap_pairs = int(2.0 * n0 * max(lam + [0.0]) + 4 * n0 + 65536)
	rows = float(n0)
	for l in lam:
		rows *= (1.0 + l)
	cap_rows = int(2.0 * rows + 4 * n0 + 65536)
	limit = (1 << 31) - 4096
	return min(cap_pairs, limit), min(cap_rows, limit)


# ---------------------------------------------------------------------------------------
# the device run
# ---------------------------------------------------------------------------------------

class MatchResult(object):
	"""Device-resident match table of one run (columns are torch tensors of length M)."""

	def __init__(self, plan, status, names):
		self.plan = plan
		self.status = status
		self.names = names
		self.nrows = int(status[_hip.ST_ROWS])

	def column(self, name, index=None):
		self.plan.check_live()
		col = self.plan.cols[name]
		if index is not None:
			col = col[index]
		return col[:self.nrows]

	def to_host(self, name, index=None):
		return _hip.to_host(self.column(name, index))


def run_match(match_tables, match_radius, prior_completeness=1.0, prob_ratio_secondary=0.5,
		radius_filter=True, correction=_hip.CORRECTION_NONE, finalize=True, scheme=None, device=None,
		logger=None, sphere_cell_factor=0.0, bitmap_bits=0, err_deg=None, link_slots=0, table_slots=0, f32_roundtrip=False, lean=False,
		tuning=None):
	"""Upload the catalogues and run the whole HIP pipeline once; returns a MatchResult.
	lean: see ``_hip.MatchPlan`` (columns nobody reads afterwards are not materialised).
	tuning: see ``_hip.make_params`` (tests force a path with it; never changes a result)."""
	logger = logger or NullOutputLogger()
	device = _hip.require_device(device)
	ncats = len(match_tables)
	if ncats < 2 or ncats > _hip.MAXCAT:
		raise ValueError('between 2 and %d catalogues can be matched, got %d' % (_hip.MAXCAT, ncats))
	err = match_radius / 60. / 60 if err_deg is None else err_deg  # __init__.py:128
	# all columns go up behind one synchronisation; the flat-cell condition (fastskymatch.py:94-98) is then read off the
	# device columns (one small kernel per catalogue, one read-back) instead of four passes over every host column
	cats = _hip.DeviceCatalogue.from_columns([(t['ra'], t['dec'], numpy.asarray(t['error'], dtype=float)) for t in match_tables], device)
	if scheme is None:
		scheme = _hip.scheme_from_extents(_hip.catalogue_extents(cats), err)
	if scheme == _hip.SCHEME_FLAT:
		logger.log('matching: using fast flat-sky approximation for this match')
	else:
		logger.log('matching: all-sky cell scheme (replaces the reference\'s healpix hashing)')
	dens, dens_plus = _compute_source_densities(match_tables, logger=logger)
	completeness = _completeness_vector(prior_completeness, ncats)
	params = _hip.make_params(ncats, scheme, float(match_radius), err, dens, dens_plus,
		_prior_table(dens, dens_plus, completeness), prob_ratio_secondary=prob_ratio_secondary,
		radius_filter=radius_filter, correction=correction, finalize=finalize,
		sphere_cell_factor=sphere_cell_factor, bitmap_bits=bitmap_bits, link_slots=link_slots, table_slots=table_slots, f32_roundtrip=f32_roundtrip,
		tuning=tuning)
	sizes = [c.n for c in cats]
	cap_pairs, cap_rows = _estimate_capacities(sizes, [t['area'] * 1.0 for t in match_tables], match_radius, scheme, radius_filter)
	plan, status = _hip.run_plan(sizes, params, cats, cap_pairs, cap_rows, device, lean=lean)
	return MatchResult(plan, status, [t['name'] for t in match_tables])


def _create_match_table(match_tables, match_radius, logger):
	"""Candidate table of __init__.py:123-196: (DataFrame, resultstable, separations, errors)."""
	import pandas
	res = run_match(match_tables, match_radius, finalize=False, logger=logger)
	names = res.names
	k = len(names)
	resultstable = numpy.stack([res.to_host('idx', c).astype(numpy.int64) for c in range(k)], axis=1)
	keys, columns = list(names), [resultstable[:, c] for c in range(k)]
	nan = numpy.ones(res.nrows) * numpy.nan
	separations = [[nan for _ in range(k)] for _ in range(k)]
	for p, (i, j) in enumerate(_hip.pair_columns(k)):
		col = res.to_host('sep', p)
		separations[i][j] = col
		keys.append('Separation_%s_%s' % (names[i], names[j]))
		columns.append(col)
	keys.append('Separation_max')
	columns.append(res.to_host('sep_max'))
	keys.append('ncat')
	columns.append(res.to_host('ncat').astype(numpy.int64))
	logger.log('matching: %6d matches after filtering by search radius' % res.nrows)
	errors = []
	for c, t in enumerate(match_tables):
		e = numpy.asarray(t['error'])
		errors.append(e[resultstable[:, c]])
	table = pandas.DataFrame(OrderedDict(zip(keys, columns)))
	return table, resultstable, separations, errors


def nway_match(match_tables, match_radius, prior_completeness,
	mag_include_radius=None, mag_exclude_radius=None, magauto_post_single_minvalue=0.9,
	prob_ratio_secondary=0.5,
	min_prob=0., consider_unrelated_associations=True,
	store_mag_hists=True,
	logger=default_logger, unrelated_associations='api', device=None, f32_roundtrip=False, tuning=None):
	"""Same contract as ``nwaylib.nway_match`` (__init__.py:31-120).

	match_tables: list of dicts with name, ra, dec (deg), error (arcsec), area (deg^2),
		mags, magnames, maghists.
	match_radius: arcsec.  prior_completeness: scalar or one value per catalogue (first = 1).
	Returns a pandas DataFrame with the reference's columns.

	unrelated_associations (extension): 'api' reproduces nwaylib.nway_match, whose correction
	never changes a value (__init__.py:276-282); 'cli' applies the working correction of
	the script (nway.py:366-423).
	f32_roundtrip (extension): numerics of the script, whose separations pass through a FITS
	float32 column before log_bf squares them (fastskymatch.py:328, bayesdistance.py:84); the
	importable API (the default here) is float64 throughout.
	tuning (extension): which kernels run (``_hip.make_params``); what the parity tests force paths with.
	"""
	import pandas
	if mag_exclude_radius is None:
		mag_exclude_radius = mag_include_radius
	if mag_include_radius is not None:
		if mag_include_radius >= match_radius:
			logger.warn('WARNING: magnitude radius is very large (>= matching radius). Consider using a smaller value.')
	names = [t['name'] for t in match_tables]
	k = len(match_tables)
	has_mags = any(len(t.get('mags', [])) > 0 for t in match_tables)
	correction = _hip.CORRECTION_CLI if (consider_unrelated_associations and unrelated_associations == 'cli') else _hip.CORRECTION_NONE

	if len(match_tables[0]['ra']) == 0:
		raise EmptyResultException('No matches.')  # nothing can create a bucket (fastskymatch.py:131)
	logger.log('Computing distance-based probabilities ...')
	res = run_match(match_tables, match_radius, prior_completeness, prob_ratio_secondary,
		correction=correction, finalize=not has_mags, device=device, logger=logger, f32_roundtrip=f32_roundtrip, lean=not has_mags, tuning=tuning)
	if not res.nrows > 0:
		raise EmptyResultException('No matches.')
	logger.log('matching: %6d matches after filtering by search radius' % res.nrows)

	if not has_mags:
		# the whole table with one transfer, the DataFrame over views of the page-locked buffer it arrived in (no copy on the host)
		logger.log('')
		logger.log('Computing final probabilities ...')
		pairs = _hip.pair_columns(k)
		f64 = ['sep%d' % i for i in range(len(pairs))] + ['sep_max', 'log_bf', 'log_bf_corrected', 'dist_post', 'p_single', 'p_any', 'p_i']
		idx, (ncat, flag), fc = res.plan.download_table(res.nrows, f64)
		cols = OrderedDict()
		for c in range(k):
			cols[names[c]] = idx[c]
		for i, (a, b) in enumerate(pairs):
			cols['Separation_%s_%s' % (names[a], names[b])] = fc[i]
		np_ = len(pairs)
		cols['Separation_max'] = fc[np_]
		cols['ncat'] = ncat
		cols['dist_bayesfactor_uncorrected'] = fc[np_ + 1]
		cols['dist_bayesfactor'] = fc[np_ + 2]
		cols['dist_post'] = fc[np_ + 3]
		cols['p_single'] = fc[np_ + 4]
		cols['match_flag'] = flag
		cols['prob_has_match'] = fc[np_ + 5]
		cols['prob_this_match'] = fc[np_ + 6]
		table = pandas.DataFrame(cols, copy=False)
		res.plan.release()
		return _truncate_table(table, min_prob, logger=logger)

	cols = OrderedDict()
	for c in range(k):
		cols[names[c]] = res.to_host('idx', c).astype(numpy.int64)
	for p, (i, j) in enumerate(_hip.pair_columns(k)):
		cols['Separation_%s_%s' % (names[i], names[j])] = res.to_host('sep', p)
	cols['Separation_max'] = res.to_host('sep_max')
	cols['ncat'] = res.to_host('ncat').astype(numpy.int64)
	cols['dist_bayesfactor_uncorrected'] = res.to_host('log_bf')
	cols['dist_bayesfactor'] = res.to_host('log_bf_corrected')
	cols['dist_post'] = res.to_host('dist_post')

	from . import magpriors
	total = magpriors.apply_magnitude_biasing(match_tables, cols, res, mag_include_radius, mag_exclude_radius,
		magauto_post_single_minvalue, store_mag_hists, logger=logger)  # (adds the bias_* columns)
	logger.log('')
	logger.log('Computing final probabilities ...')
	stats = magpriors.final_probabilities_device(res, total, prob_ratio_secondary)
	cols['p_single'] = stats['p_single']
	cols['match_flag'] = stats['match_flag'].astype(numpy.int64)
	cols['prob_has_match'] = stats['p_any']
	cols['prob_this_match'] = stats['p_i']
	table = pandas.DataFrame(cols, copy=False)  # (one frame over the host arrays, made once)
	res.plan.release()
	return _truncate_table(table, min_prob, logger=logger)


def _truncate_table(table, min_prob, logger):
	"""drop rows with p_i < min_prob (rows with NaN stay), __init__.py:464-471"""
	if min_prob > 0:
		mask = ~(table['prob_this_match'] < min_prob)
		logger.log('    cutting away %d (below p_i minimum)' % (len(mask) - mask.sum()))
		table = table[mask]
	return table


from . import bayesdistance as bayesdist  # noqa: E402
from . import fastskymatch as match  # noqa: E402
