"""Command-line driver with the interface of the reference's ``nway.py`` script
(argparse surface nway.py:110-158, FITS in / FITS out), running the match on the GPU.

Differences between the script and the API that are kept:
separation columns are named ``Separation_{later}_{earlier}``, every input column is copied
as ``{TABLE}_{column}`` with -99 for absent counterparts, floats are written as 'E', the
unrelated-association correction is the working one (nway.py:366-423), the auto-histogram
selection indexes its weights by the selected rows (nway.py:471).

The script's numerics are reproduced as well (``f32_roundtrip``: its separations pass through a
float32 FITS column before log_bf squares them).  Pinned to the reference's script ITSELF: tests/golden/script_cli.npz /
.json hold the tables /root/reference/nway.py wrote for eleven command lines (tests/golden/make_script_golden.py executes it under an
I/O-only stand-in for astropy.io.fits), tests/test_cli_script_golden.py compares this module's output files with them.

Asymmetric / elliptical error columns (``:ra_err:dec_err``, ``:major:minor:angle``) use the
device pipeline for the candidates and nway_amd/elliptical.py (device kernels for the offsets
and the elliptical Bayes factor) in float64; the offsets' parity is unpinned: the reference
needs astropy's SkyOffsetFrame.

Not reproduced: ``--prefilter-pair`` (broken upstream, fastskymatch.py:203).
"""
from __future__ import division, print_function

import argparse
import sys

import numpy

from . import _fits, _hip, magnitudeweights
from . import fastskymatch as match

DESCRIPTION = """Multiway association between astrometric catalogue. Use --help for usage.

Example: nway.py --radius 10 --prior-completeness 0.95 --mag GOODS:mag_H auto --mag IRAC:mag_irac1 auto cdfs4Ms_srclist_v3.fits :Pos_error CANDELS_irac1.fits 0.5 gs_short.fits 0.1 --out=out.fits
"""


class HelpfulParser(argparse.ArgumentParser):
	def error(self, message):
		sys.stderr.write('error: %s\n' % message)
		self.print_help()
		sys.exit(2)


def build_parser():
	p = HelpfulParser(description=DESCRIPTION, formatter_class=argparse.ArgumentDefaultsHelpFormatter,
		epilog='GPU (MI355X) build of nway; reference: Johannes Buchner (C) 2013-2025')
	p.add_argument('--radius', type=float, required=True, help='exclusive search radius in arcsec for initial matching')
	p.add_argument('--mag-radius', default=None, type=float,
		help='search radius for building the magnitude histogram of target sources. If not set, the Bayesian posterior is used.')
	p.add_argument('--mag-auto-minprob', default=0.9, type=float,
		help='minimum posterior probability (default: 0.9) for the magnitude histogram of secure target sources. Used in the Bayesian procedure.')
	p.add_argument('--mag-exclude-radius', default=None, type=float,
		help='exclusion radius for building the magnitude histogram of field sources. If not set, --mag-radius is used.')
	p.add_argument('--prior-completeness', metavar='COMPLETENESS', default='1', type=str,
		help='expected matching completeness of sources (prior)')
	p.add_argument('--ignore-unrelated-associations', dest='consider_unrelated_associations', action='store_false',
		help='Ignore in the calculation source pairings unrelated to the primary source (not recommended)')
	p.set_defaults(consider_unrelated_associations=True)
	p.add_argument('--mag', metavar='MAGCOLUMN+MAGFILE', type=str, nargs=2, action='append', default=[],
		help='name of <table>:<column> for magnitude biasing, and filename for magnitude histogram '
		'(use auto for auto-computation within mag-radius). Example: --mag GOODS:mag_H auto --mag IRAC:mag_irac1 irac_histogram.txt')
	p.add_argument('--acceptable-prob', metavar='PROB', type=float, default=0.5,
		help='ratio limit up to which secondary solutions are flagged')
	p.add_argument('--min-prob', type=float, default=0,
		help='lowest probability allowed in final catalogue. If 0, no trimming is performed (default).')
	p.add_argument('--out', metavar='OUTFILE', help='output file name', required=True)
	p.add_argument('catalogues', type=str, nargs='+',
		help='input catalogue fits files and position errors. Example: cdfs4Ms_srclist_v3.fits :Pos_error CANDELS_irac1.fits 0.5 gs_short.fits 0.1')
	p.add_argument('--prefilter-pair', metavar='CATNAME1 CATNAME2 radius', type=str, nargs=3, action='append', default=[],
		help='accepted for compatibility; not supported (its reference implementation drops every tuple containing both catalogues)')
	return p


def resolve_errors(tables, table_names, pos_errors, match_radius_arcsec):
	"""positional error (arcsec) per catalogue from the ``file error`` pairs: a fixed value or
	``:column`` (nway.py:25-98, circular cases)"""
	errors = []
	for t, name, spec in zip(tables, table_names, pos_errors):
		if spec[0] != ':':
			value = float(spec)
			print('    Position error for "%s": using fixed value %f' % (name, value))
			if value > match_radius_arcsec:
				print('WARNING: Given separation error for "%s" is larger than the match radius! Increase --radius to >> %s' % (name, value))
			errors.append(value * numpy.ones(len(t.data)))
			continue
		keys = spec[1:].split(':')
		if len(keys) > 3:
			raise AssertionError('Invalid column specifier: %s' % spec)
		for k in keys:
			assert k in t.data.dtype.names, 'ERROR: Position error column "%s" not in table "%s". Have these columns: %s' % (k, name, ', '.join(t.data.dtype.names))
		if len(keys) > 1:
			for key, meaning in zip(keys, ['ra_error', 'dec_error', 'ell_angle']):
				print('    Position error for "%s": found column %s (for %s): Values are [%f..%f]' % (name, key, meaning, t.data[key].min(), t.data[key].max()))
			errors.append(None)  # asymmetric / elliptical: handled by nway_amd.elliptical
			continue
		col = numpy.asarray(t.data[keys[0]], dtype=float)
		print('    Position error for "%s": found column %s (for ra_error): Values are [%f..%f]' % (name, keys[0], col.min(), col.max()))
		if col.min() <= 0:
			print('WARNING: Some separation errors in "%s" are 0! This will give invalid results (%d rows).' % (keys[0], (col <= 0).sum()))
		if col.max() > match_radius_arcsec:
			print('WARNING: Some separation errors in "%s" are larger than the match radius! Increase --radius to >> %s' % (keys[0], col.max()))
		errors.append(col)
	return errors


def cli_magnitude_bias(mag, magfile, table_names, tables, idx_columns, sep_max, post, mag_include_radius, mag_exclude_radius,
		minprob, match_radius_arcsec):
	"""histogram + step function of one ``--mag T:COL file|auto`` option, script flavour
	(nway.py:434-516); returns (column name, table number, StepFunction, magnitude column)"""
	from . import magpriors
	table_name, col_name = mag.split(':', 1)
	ti = table_names.index(table_name)
	mag_all = numpy.array(tables[ti].data[col_name], dtype=float)
	mag_all[mag_all == -99] = numpy.nan
	col = '%s_%s' % (table_name, col_name)
	if magfile == 'auto':
		if mag_include_radius is not None:
			if mag_include_radius >= match_radius_arcsec:
				print('WARNING: magnitude radius is very large (>= matching radius). Consider using a smaller value.')
			secure, plausible, weights = sep_max < mag_include_radius, sep_max < mag_exclude_radius, numpy.ones(len(sep_max))
		else:
			secure, plausible, weights = post > minprob, post > 0.01, post
		target, target_weights, field, n_plausible = magpriors.secure_and_field_sources(idx_columns[ti], mag_all, secure, plausible,
			weights, 'script', mag)
		print('    magnitude histogram of column "%s": %d secure matches, %d insecure matches and %d secure non-matches of %d total entries (%d valid)'
			% (col, len(target), n_plausible, field.sum(), len(mag_all), numpy.isfinite(mag_all).sum()))
		bins, hist_sel, hist_all = magnitudeweights.adaptive_histograms(mag_all[field], target, weights=target_weights)
		filename = mag.replace(':', '_') + '_fit.txt'
		print('    magnitude histogram stored to "%s".' % filename)
		magpriors.write_histogram(filename, bins, hist_sel, hist_all)
		if len(target) < 100:
			print('ERROR: too few secure matches to make a good histogram. If you are sure you want to use this poorly sampled histogram, replace "auto" with the filename. You can also decrease the mag-auto-minprob parameter.')
			sys.exit(1)
	else:
		print('    magnitude histogramming: using histogram from "%s" for column "%s"' % (magfile, col))
		bins_lo, bins_hi, hist_sel, hist_all = numpy.loadtxt(magfile).transpose()
		bins = numpy.array(list(bins_lo) + [bins_hi[-1]])
	func = magnitudeweights.fitfunc_histogram(bins, hist_sel, hist_all)
	magnitudeweights.plot_fit(bins, hist_sel, hist_all, func, mag)
	return col, ti, func, mag_all


def main(argv=None):
	argv_from_sys = argv is None
	argv = list(sys.argv[1:] if argv is None else argv)
	args = build_parser().parse_args(argv)
	print('NWAY arguments:')
	filenames = args.catalogues[::2]
	pos_errors = args.catalogues[1::2]
	if len(filenames) != len(pos_errors) or len(filenames) < 2:
		raise SystemExit('error: catalogues must be given as pairs of <file> <position error>, at least two')
	print('    catalogues: ', ', '.join(filenames))
	print('    position errors/columns: ', ', '.join(pos_errors))
	if args.prefilter_pair:
		raise NotImplementedError('--prefilter-pair is not supported (see module docstring)')

	tables, table_names, sizes, areas = [], [], [], []
	for fitsname in filenames:
		t = _fits.read_table(fitsname, 1)
		assert 'SKYAREA' in t.header, 'file "%s", table "%s" does not have a field "SKYAREA", which should contain the area of the catalogue in square degrees' % (fitsname, t.name)
		tables.append(t)
		table_names.append(t.name)
		sizes.append(len(t.data))
		areas.append(t.header['SKYAREA'] * 1.0)
		print('      from catalogue "%s" (%d), density gives %.2e on entire sky' % (t.name, len(t.data), len(t.data) / areas[-1] * (4 * numpy.pi * (180 / numpy.pi)**2)))
	k = len(tables)
	if ':' in args.prior_completeness:
		completeness = numpy.array([1.0] + [float(pc) for pc in args.prior_completeness.split(':')])
		if len(completeness) != k:
			raise Exception('Prior completeness needs one value per catalog, like "%s". Received "%s".' % (':'.join(['0.9'] * (k - 1)), args.prior_completeness))
	else:
		completeness = numpy.array([1.0] + [float(args.prior_completeness)**(1. / (k - 1)) for _ in range(1, k)])
	mag_include_radius = args.mag_radius
	mag_exclude_radius = args.mag_exclude_radius if args.mag_exclude_radius is not None else mag_include_radius
	assert 0 < args.mag_auto_minprob <= 1, 'probability should be between 0 and 1'
	print('    magnitude columns: ', ', '.join([c for c, _ in args.mag]))
	for mag, magfile in args.mag:
		table_name, col_name = mag.split(':', 1)
		assert table_name in table_names, 'table name specified for magnitude ("%s") unknown. Known tables: %s' % (table_name, ', '.join(table_names))
		names = tables[table_names.index(table_name)].data.dtype.names
		assert col_name in names, 'column name specified for magnitude ("%s") unknown. Known columns in table "%s": %s' % (mag, table_name, ', '.join(names))

	print('Computing distance-based probabilities ...')
	print('  finding position error columns ...')
	errors = resolve_errors(tables, table_names, pos_errors, args.radius)
	print('  finding position columns ...')
	ra_keys = [match.get_tablekeys(t.data, 'RA', tablename=n) for t, n in zip(tables, table_names)]
	dec_keys = [match.get_tablekeys(t.data, 'DEC', tablename=n) for t, n in zip(tables, table_names)]
	print('    using RA  columns: %s' % ', '.join(ra_keys))
	print('    using DEC columns: %s' % ', '.join(dec_keys))
	print('  building primary_id index ...')
	primary_id_key = match.get_tablekeys(tables[0].data, 'ID', tablename=table_names[0])
	ids = tables[0].data[primary_id_key]
	assert len(numpy.unique(ids)) == len(ids), "ERROR: ID column '%s' in primary catalog contains duplicates." % primary_id_key
	primary_id_key = '%s_%s' % (table_names[0], primary_id_key)

	import nway_amd
	simple_errors = all(e is not None for e in errors)
	match_tables = [dict(name=n, ra=numpy.asarray(t.data[rk], dtype=float), dec=numpy.asarray(t.data[dk], dtype=float),
		error=(e if simple_errors else 1.0), area=a)
		for t, n, rk, dk, e, a in zip(tables, table_names, ra_keys, dec_keys, errors, areas)]
	print('  computing probabilities ...')
	correction = _hip.CORRECTION_CLI if (args.consider_unrelated_associations and simple_errors) else _hip.CORRECTION_NONE
	# the script's numerics: cells from radius / 60 / 60 degrees, the separation filter
	# against that value * 60 * 60 (fastskymatch.py:336; may differ from --radius in the last bit),
	# separations squared in float32 after their trip through the FITS 'E' column
	err_deg = args.radius / 60. / 60
	res = nway_amd.run_match(match_tables, err_deg * 60 * 60, completeness, args.acceptable_prob, correction=correction,
		finalize=(not args.mag) and simple_errors, logger=nway_amd.NullOutputLogger(), err_deg=err_deg, f32_roundtrip=True)
	assert res.nrows > 0, 'No matches.'
	print('matching: %6d matches after filtering by search radius' % res.nrows)
	idx_columns = [res.to_host('idx', c).astype(numpy.int64) for c in range(k)]

	columns = []  # (name, tform, array)
	for t, name, idx in zip(tables, table_names, idx_columns):
		missing = idx == -1
		gathered = t.data[idx]  # -1 picks the last row; overwritten below
		for colname, fmt in zip(t.data.dtype.names, t.formats):
			col = numpy.array(gathered[colname])
			try:
				col[missing] = -99
			except Exception as e:
				print('   setting "%s_%s" to -99 failed (%d affected; column format "%s"): %s' % (name, colname, missing.sum(), fmt, e))
			columns.append(('%s_%s' % (name, colname), fmt, col))
	pair_index = dict((p, n) for n, p in enumerate(_hip.pair_columns(k)))
	sep_ra = [[None] * k for _ in range(k)]
	sep_dec = [[None] * k for _ in range(k)]
	for i in range(k):
		for j in range(i):
			columns.append(('Separation_%s_%s' % (table_names[i], table_names[j]), 'E', res.to_host('sep', pair_index[(j, i)])))
			if not simple_errors:
				# tangent-plane offsets, frame centred on the later catalogue's source (fastskymatch.py:306-312)
				from . import elliptical
				def coords(c):
					m = idx_columns[c]
					ra = numpy.where(m >= 0, match_tables[c]['ra'][m], -99.)
					dec = numpy.where(m >= 0, match_tables[c]['dec'][m], -99.)
					return ra, dec
				dra, ddec = elliptical.offsets(*(coords(i) + coords(j)))
				# the script reads the offsets back from FITS 'E' columns (fastskymatch.py:329-331,
				# nway.py:303-305): what log_bf_elliptical sees are float32 values
				sep_ra[j][i] = (dra * 60 * 60).astype(numpy.float32)
				sep_dec[j][i] = (ddec * 60 * 60).astype(numpy.float32)
				columns.append(('Separation_%s_%s_ra' % (table_names[i], table_names[j]), 'E', sep_ra[j][i]))
				columns.append(('Separation_%s_%s_dec' % (table_names[i], table_names[j]), 'E', sep_dec[j][i]))
	sep_max = res.to_host('sep_max')
	ncat = res.to_host('ncat').astype(numpy.int64)
	columns.append(('Separation_max', 'E', sep_max))
	columns.append(('ncat', 'I', ncat))
	prior = res.to_host('prior')
	if simple_errors:
		log_bf_uncorrected = res.to_host('log_bf')
		log_bf = res.to_host('log_bf_corrected')
	else:
		from . import elliptical
		triplets = elliptical.error_triplets(tables, table_names, pos_errors, idx_columns)
		log_bf_uncorrected = elliptical.log_bf_table(k, idx_columns, sep_ra, sep_dec, triplets)
		log_bf = log_bf_uncorrected
		if args.consider_unrelated_associations and k >= 3:
			dens = numpy.array([n / a * (4 * numpy.pi * (180 / numpy.pi)**2) for n, a in zip(sizes, areas)])
			dens_plus = numpy.array([(n + 1) / a * (4 * numpy.pi * (180 / numpy.pi)**2) for n, a in zip(sizes, areas)])
			dens_plus[0] = dens[0]
			log_bf = elliptical.unrelated_associations(k, idx_columns, ncat, sep_ra, sep_dec, triplets, dens, dens_plus, log_bf_uncorrected)
	columns.append(('dist_bayesfactor', 'E', log_bf_uncorrected))
	if args.consider_unrelated_associations:
		if (ncat <= k - 2).any():
			print('    correcting for unrelated associations ...')
			columns.append(('dist_bayesfactor_corrected', 'E', log_bf))
		else:
			print('      correcting for unrelated associations ... not necessary')
	if simple_errors:
		post = res.to_host('dist_post')
	else:
		from . import bayesdistance
		post = bayesdistance.posterior(prior, log_bf)
	columns.append(('dist_post', 'E', post))

	biases = []
	if args.mag:
		print()
		print('Incorporating magnitude biases ...')
		t = _hip.torch()
		lib = _hip.load()
		device = res.plan.device
		total = res.column('log_bf_corrected').clone() if simple_errors else _hip.to_device(log_bf, device)
		for mag, magfile in args.mag:
			print('    magnitude bias "%s" ...' % mag)
			col, ti, func, mag_all = cli_magnitude_bias(mag, magfile, table_names, tables, idx_columns, sep_max, post,
				mag_include_radius, mag_exclude_radius, args.mag_auto_minprob, args.radius)
			d_bias = t.empty(res.nrows, dtype=t.float64, device=device)
			# named: a temporary tensor would be released (and its block handed to the next
			# allocation) as soon as its address had been taken
			d_mag = _hip.to_device(mag_all, device)
			d_edges = _hip.to_device(func.edges, device)
			d_ratio = _hip.to_device(func.values, device)
			_hip.check(lib.nwayhip_bias_lookup(res.nrows, _hip.ptr(res.column('idx', ti)), _hip.ptr(d_mag), len(func.edges),
				_hip.ptr(d_edges), _hip.ptr(d_ratio), _hip.ptr(total), _hip.ptr(d_bias), _hip.current_stream_ptr(device)))
			biases.append(col)
			columns.append(('bias_%s' % col, 'E', _hip.to_host(d_bias)))
		print()
		print('Computing final probabilities ...')
		from . import magpriors
		stats = magpriors.final_probabilities_device(res, total, args.acceptable_prob)
		p_single, p_any, p_i, flag = stats['p_single'], stats['p_any'], stats['p_i'], stats['match_flag']
	elif not simple_errors:
		print()
		print('Computing final probabilities ...')
		from . import magpriors
		stats = magpriors.final_probabilities_device(res, _hip.to_device(log_bf, res.plan.device), args.acceptable_prob)
		p_single, p_any, p_i, flag = stats['p_single'], stats['p_any'], stats['p_i'], stats['match_flag']
	else:
		print()
		print('Computing final probabilities ...')
		p_single, p_any, p_i, flag = res.to_host('p_single'), res.to_host('p_any'), res.to_host('p_i'), res.to_host('match_flag')
	print('    grouping by column "%s" and flagging ...' % primary_id_key)
	columns.append(('p_single', 'E', p_single))
	columns.append(('p_any', 'E', p_any))
	columns.append(('p_i', 'E', p_i))
	columns.append(('match_flag', 'I', flag.astype(numpy.int64)))
	res.plan.close()

	if args.min_prob > 0:
		mask = ~(p_i < args.min_prob)
		print('    cutting away %d (below p_i minimum)' % (len(mask) - mask.sum()))
		columns = [(n, f, a[mask]) for n, f, a in columns]

	if not filenames[0].endswith('shifted.fits'):
		print_calibration_hint(filenames, args, argv)

	print()
	print('creating output FITS file ...')
	header = dict(METHOD='NWAY multi-way matching', INPUT=', '.join(filenames), TABLES=', '.join(table_names),
		BIASING=', '.join(biases), NWAYCMD=' '.join([sys.argv[0] if argv_from_sys else 'nway.py'] + argv))
	header['COLS_RA'] = ' '.join('%s_%s' % (n, rk) for n, rk in zip(table_names, ra_keys))
	header['COLS_DEC'] = ' '.join('%s_%s' % (n, dk) for n, dk in zip(table_names, dec_keys))
	header['COL_PRIM'] = primary_id_key
	header['COLS_ERR'] = ' '.join('%s_%s' % (n, e) for n, e in zip(table_names, pos_errors))
	comments = ['argument %s: %s' % (key, value) for key, value in vars(args).items()]  # the namespace's order, like nway.py:645
	print('    writing "%s" (%d rows, %d columns) ...' % (args.out, len(columns[0][2]), len(columns)))
	_fits.write_table(args.out, columns, 'NWAYMATCH', primary_header=header, comments=comments, overwrite=True)
	return 0


def print_calibration_hint(filenames, args, argv):
	"""how to calibrate a p_any cut-off with a fake catalogue (nway.py:596-632)"""
	print()
	print()
	print('  You can calibrate a p_any cut-off with the following steps:')
	print('   1) Create a offset catalogue to simulate random sky positions:')
	shiftfile = filenames[0].replace('.fits', '').replace('.FITS', '') + '-fake.fits'
	shiftoutfile = args.out + '-fake.fits'
	print('      nway-create-fake-catalogue.py --radius %d %s %s' % (args.radius * 2, filenames[0], shiftfile))
	print('   2) Match the offset catalogue in the same way as this run:')
	new, i = ['nway.py'], 0
	while i < len(argv):
		v = argv[i]
		if v == filenames[0]:
			new.append(shiftfile)
		elif v == '--mag':
			new += [v, argv[i + 1], argv[i + 1].replace(':', '_') + '_fit.txt' if argv[i + 2] == 'auto' else argv[i + 2]]
			i += 2
		elif v == '--out':
			new += [v, shiftoutfile]
			i += 1
		elif v.startswith('--out='):
			new.append('--out=' + shiftoutfile)
		else:
			new.append(v)
		i += 1
	print('      ' + ' '.join(new))
	print('   3) determining the p_any cutoff that corresponds to a false-detection rate')
	print('      nway-calibrate-cutoff.py %s %s' % (args.out, shiftoutfile))
	print()


if __name__ == '__main__':
	sys.exit(main())
