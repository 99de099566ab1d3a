"""The command line of nway.py (nway.py:110-158 of the reference; option list as printed in its
doc/logs/help): same options, same defaults, same positional grammar -- checked without a GPU."""
import pytest

from nway_amd import cli

REFERENCE_OPTIONS = ['--acceptable-prob', '--help', '--ignore-unrelated-associations', '--mag', '--mag-auto-minprob',
	'--mag-exclude-radius', '--mag-radius', '--min-prob', '--out', '--prefilter-pair', '--prior-completeness', '--radius']


def test_option_surface():
	p = cli.build_parser()
	have = sorted(o for a in p._actions for o in a.option_strings if o.startswith('--'))
	assert have == sorted(REFERENCE_OPTIONS)


def test_defaults_and_grammar():
	p = cli.build_parser()
	a = p.parse_args(['--radius', '20', '--out', 'o.fits', 'X.fits', ':pos_err', 'O.fits', '0.1'])
	assert a.radius == 20.0 and a.out == 'o.fits'
	assert a.catalogues == ['X.fits', ':pos_err', 'O.fits', '0.1']
	assert a.consider_unrelated_associations is True
	assert a.min_prob == 0. and a.acceptable_prob == 0.5 and a.mag_auto_minprob == 0.9
	assert a.mag == [] and a.mag_radius is None and a.mag_exclude_radius is None
	b = p.parse_args(['--radius', '5', '--out', 'o.fits', '--ignore-unrelated-associations', '--mag', 'O:mag', 'auto',
		'--mag', 'I:m1', 'file.txt', '--prior-completeness', '0.9:0.8', 'X.fits', '1', 'O.fits', '0.1', 'I.fits', '0.5'])
	assert b.consider_unrelated_associations is False
	assert b.mag == [['O:mag', 'auto'], ['I:m1', 'file.txt']]
	with pytest.raises(SystemExit):
		p.parse_args(['X.fits', '1'])  # --radius and --out are required


# public names of the reference's modules (nwaylib/*.py of v4.7.1: every def / class / constant a
# caller can import); the drop-in package must offer each of them under the same module path
REFERENCE_API = {
	'nwaylib': ['nway_match', '__version__', 'EmptyResultException', 'UndersampledException', 'bayesdist', 'match', 'magnitudeweights',
		'NormalLogger', 'NullOutputLogger'],
	'nwaylib.fastskymatch': ['dist', 'dist3d', 'get_tablekeys', 'get_healpix_resolution_degrees', 'crossproduct', 'match_multiple',
		'fits_from_columns', 'wraptable2fits', 'array2fits'],
	'nwaylib.bayesdistance': ['log_arcsec2rad', 'log_posterior', 'posterior', 'unnormalised_log_posterior', 'log_bf2', 'log_bf3', 'log_bf',
		'assert_possemdef', 'matrix_add', 'matrix_multiply', 'matrix_det', 'matrix_invert', 'apply_vector_right', 'apply_vector_left',
		'vector_multiply', 'vector_normalised', 'apply_vABv', 'make_covmatrix', 'make_invcovmatrix', 'convert_from_ellipse', 'log_bf_elliptical'],
	'nwaylib.magnitudeweights': ['ratio', 'fraction', 'plot_fit', 'fitfunc_histogram', 'adaptive_histograms'],
	'nwaylib.logger': ['FakeProgressBar', 'NullOutputLogger', 'NormalLogger'],
	'nwaylib.progress': ['bar', 'kwargs_overwrite_true'],
}


def test_python_surface_is_complete():
	import importlib
	for module, names in REFERENCE_API.items():
		for prefix in ('nwaylib', 'nway_amd'):
			m = importlib.import_module(module.replace('nwaylib', prefix, 1))
			missing = [n for n in names if not hasattr(m, n)]
			assert not missing, '%s lacks %s' % (m.__name__, missing)
