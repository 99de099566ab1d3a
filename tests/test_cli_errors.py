"""How the command line refuses defective input: the exception type and the message of the REFERENCE'S SCRIPT (tests/golden/script_errors.json,
made by executing /root/reference/nway.py on the same defective catalogues: tests/golden/make_script_golden.py: gen_errors), repeated by
nway_amd/cli.py.  Every one of these checks comes before the match, so no GPU is needed."""
import json
import os

import numpy as np
import pytest

from goldenutil import GOLDEN

CASES = json.load(open(os.path.join(GOLDEN, 'script_errors.json')))


def write_inputs():
	"""the catalogues of make_script_golden.error_inputs, written with the product's own FITS writer"""
	from nway_amd import _fits
	rng = np.random.RandomState(8)
	n = 30
	ra, dec = 150 + rng.uniform(0, 0.01, n), 2 + rng.uniform(0, 0.01, n)
	cols = [('ID', 'J', np.arange(n)), ('RA', 'D', ra), ('DEC', 'D', dec), ('pos_err', 'D', np.full(n, 0.5)), ('MAG', 'D', rng.normal(20, 1, n))]
	_fits.write_table('good_a.fits', cols, 'A', table_header={'SKYAREA': 0.01})
	_fits.write_table('good_b.fits', [('ID', 'J', np.arange(n)), ('RA', 'D', ra + 1e-4), ('DEC', 'D', dec), ('MAG', 'D', rng.normal(22, 1, n))], 'B',
		table_header={'SKYAREA': 0.01})
	_fits.write_table('noarea.fits', cols, 'NA')
	_fits.write_table('dupid.fits', [('ID', 'J', np.zeros(n, dtype=int))] + cols[1:], 'DUP', table_header={'SKYAREA': 0.01})


@pytest.mark.parametrize('tag', sorted(CASES))
def test_cli_refuses_like_the_script(tag, tmp_path, monkeypatch, capsys):
	from nway_amd import cli
	monkeypatch.chdir(tmp_path)
	write_inputs()
	want = CASES[tag]
	kinds = dict(AssertionError=AssertionError, Exception=Exception, SystemExit=SystemExit)
	with pytest.raises(kinds.get(want['type'], Exception)) as caught:
		cli.main(list(want['argv']))
	assert type(caught.value).__name__ == want['type']
	# (the script formats one message with the file's path as given; everything else is the same text)
	assert str(caught.value) == want['message'], (str(caught.value), want['message'])
