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
