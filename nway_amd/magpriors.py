"""Magnitude (or any other property) priors on top of the distance-based match table.

Restates nwaylib/__init__.py:304-396 (``_apply_magnitude_biasing``): per magnitude column
build or take the selected/all histograms on the host (catalogue-sized numpy work, once),
then look every table row up in the resulting step function ON THE DEVICE
(``nwayhip_bias_lookup``) and run the per-primary statistics with the biased totals
(``nwayhip_group_stats``, __init__.py:399-461).

The API flavour of the selection logic is reproduced literally, including its indexing of
the weights by the "defined" rows (``__init__.py:337``; the script uses the selected rows,
``nway.py:471``) -- outputs must equal ``nwaylib.nway_match``.
"""
from __future__ import division, print_function

import numpy

from . import _hip
from . import magnitudeweights


def secure_and_field_sources(idx, magnitudes, secure, plausible, weights, flavour, label):
	"""Which sources of a catalogue enter the two histograms of one magnitude column.

	idx: the catalogue's index column of the match table (-1 = absent); secure / plausible: row
	masks (a secure counterpart; a counterpart that cannot be ruled out); weights: one per row.
	Returns (magnitudes of the secure counterparts, their weights, mask of the field sources,
	number of plausible sources).  Field = every source with a finite magnitude that is not a
	plausible counterpart of anything.

	The weight of a secure counterpart is looked up at the position of its first secure row --
	a position among the rows that HAVE a counterpart in ``nwaylib.nway_match``
	(__init__.py:337) and among the SECURE rows in the script (nway.py:471).  Both are kept, as
	is the script's demand for at least two counterparts (nway.py:477) where the API is content
	with one (__init__.py:343)."""
	present = idx != -1
	secure = numpy.logical_and(secure, present)
	plausible = numpy.logical_and(plausible, present)
	pool = weights[secure] if flavour == 'script' else weights[present]
	sources, first_row = numpy.unique(idx[secure], return_index=True)
	source_weights = pool[first_row]
	assert len(sources) > (1 if flavour == 'script' else 0), 'No magnitude values within radius for "%s".' % label
	target = magnitudes[sources]
	usable = ~numpy.logical_or(numpy.isnan(target), numpy.isinf(target))
	plausible_sources = numpy.unique(idx[plausible])
	field = ~numpy.logical_or(numpy.isnan(magnitudes), numpy.isinf(magnitudes))
	field[plausible_sources] = False
	return target[usable], source_weights[usable], field, len(plausible_sources)


def write_histogram(filename, bins, hist_sel, hist_all):
	"""the four-column text file of __init__.py:369-373 / nway.py:497-501"""
	with open(filename, 'wb') as f:
		f.write(b'# lo hi selected others\n')
		numpy.savetxt(f, numpy.transpose([bins[:-1], bins[1:], hist_sel, hist_all]), fmt=['%10.5f'] * 4)


def apply_magnitude_biasing(match_tables, table, res, mag_include_radius, mag_exclude_radius,
		magauto_post_single_minvalue, store_mag_hists, logger):
	"""``table``: the columns of the match table so far (an ordered mapping name -> host array; the ``bias_*`` columns are added
	to it -- the DataFrame is made once, at the end, by the caller); returns the device tensor ``total`` = dist_bayesfactor + sum
	of log10 biases"""
	from . import UndersampledException
	t = _hip.torch()
	lib = _hip.load()
	device = res.plan.device
	nrows = res.nrows
	total = res.column('log_bf_corrected').clone()
	for i, tab in enumerate(match_tables):
		for magvals, maghist, magname in zip(tab['mags'], tab['maghists'], tab['magnames']):
			col = '%s_%s' % (tab['name'], magname)
			mag = '%s:%s' % (tab['name'], magname)
			logger.log('Incorporating bias "%s" ...' % mag)
			mag_all = magvals
			mag_all[mag_all == -99] = numpy.nan  # in place, like the reference (:319)
			if maghist is None:
				if mag_include_radius is not None:
					sep_max = table['Separation_max']
					secure, plausible, weights = sep_max < mag_include_radius, sep_max < mag_exclude_radius, numpy.ones(nrows)
				else:
					post = table['dist_post']
					secure, plausible, weights = post > magauto_post_single_minvalue, post > 0.01, post
				target, target_weights, field, n_plausible = secure_and_field_sources(table[tab['name']], mag_all,
					secure, plausible, weights, 'api', mag)
				logger.log('magnitude histogram of column "%s": %d secure matches, %d insecure matches and %d secure non-matches of %d total entries (%d valid)'
					% (col, len(target), n_plausible, field.sum(), len(mag_all), numpy.isfinite(mag_all).sum()))
				bins, hist_sel, hist_all = magnitudeweights.adaptive_histograms(mag_all[field], target, weights=target_weights)
				if store_mag_hists:
					filename = mag.replace(':', '_') + '_fit.txt'
					logger.log('magnitude histogram stored to "%s".' % filename)
					write_histogram(filename, bins, hist_sel, hist_all)
				if len(target) < 100:
					raise UndersampledException('ERROR: too few secure matches (%d) to make a good histogram. If you are sure you want to use this poorly sampled histogram, replace "auto" with the filename. You can also decrease the mag-auto-minprob parameter.' % len(target))
			else:
				logger.log('magnitude histogramming: using user-supplied histogram for "%s"' % (col))
				bins_lo, bins_hi, hist_sel, hist_all = maghist
				bins = numpy.array(list(bins_lo) + [bins_hi[-1]])
			func = magnitudeweights.fitfunc_histogram(bins, hist_sel, hist_all)
			if store_mag_hists:
				magnitudeweights.plot_fit(bins, hist_sel, hist_all, func, mag)
			# per-row lookup on the device: weight = log10(func(mag[idx])), undefined -> 0
			# (one transfer for the three: _hip.upload_columns)
			d_mag, d_edges, d_ratio = _hip.upload_columns([numpy.where(numpy.isfinite(mag_all), mag_all, numpy.nan), func.edges, func.values], device)
			d_bias = t.empty(nrows, dtype=t.float64, device=device)
			_hip.check(lib.nwayhip_bias_lookup(nrows, _hip.ptr(res.column('idx', i)), _hip.ptr(d_mag), len(func.edges),
				_hip.ptr(d_edges), _hip.ptr(d_ratio), _hip.ptr(total), _hip.ptr(d_bias), _hip.current_stream_ptr(device)))
			table['bias_%s' % col] = _hip.to_host(d_bias)
	return total


def final_probabilities_device(res, total, prob_ratio_secondary):
	"""p_single, p_any, p_i, match_flag from the biased totals (device), as host arrays"""
	lib = _hip.load()
	cols = res.plan.cols
	n_groups = res.plan.sizes[0]
	_hip.check(lib.nwayhip_group_stats(res.nrows, n_groups, _hip.ptr(cols['group_start']), _hip.ptr(total), _hip.ptr(cols['prior']),
		prob_ratio_secondary, _hip.ptr(cols['p_single']), _hip.ptr(cols['p_any']), _hip.ptr(cols['p_i']), _hip.ptr(cols['match_flag']),
		_hip.current_stream_ptr(res.plan.device)))
	return dict(p_single=res.to_host('p_single'), p_any=res.to_host('p_any'), p_i=res.to_host('p_i'),
		match_flag=res.to_host('match_flag'))
