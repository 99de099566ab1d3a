"""The FITS BINTABLE reader / writer of the CLI (nway_amd/_fits.py) on the column kinds an
astropy-written catalogue brings along (the CLI copies every input column to its output,
fastskymatch.py:271-282): vector columns keep their repeat count, unsigned integers travel with
the TZERO convention, column types that are not decoded are skipped instead of failing the read."""
import os
import warnings

import numpy as np

from goldenutil import ROOT  # noqa: F401  (puts the repo on sys.path)
from nway_amd import _fits


def test_vector_unsigned_and_string_columns_round_trip(tmp_path):
	n = 7
	cols = [('id', 'J', np.arange(n, dtype=np.uint32) + 4000000000), ('flux', '3E', np.arange(3 * n, dtype=float).reshape(n, 3)),
		('ra', 'D', np.linspace(0, 1, n)), ('name', '4A', np.array(['a', 'bb', 'ccc', 'dddd', 'e', 'f', 'g'])),
		('big', 'K', np.arange(n, dtype=np.uint64) + (1 << 63)), ('small', 'I', np.arange(n, dtype=np.int16) - 3)]
	f1 = str(tmp_path / 'a.fits')
	_fits.write_table(f1, cols, 'CAT')
	t = _fits.read_table(f1)
	assert t.header['TZERO1'] == 2 ** 31 and t.formats[1] == '3E'
	for name, tform, arr in cols:
		got = t.data[name]
		if tform.endswith('A'):
			got = np.char.decode(got, 'ascii')
		np.testing.assert_array_equal(got, arr, err_msg=name)
	assert t.data['id'].dtype == np.uint32 and t.data['big'].dtype == np.uint64
	# what the CLI does with its input columns: read, write again under the same formats
	f2 = str(tmp_path / 'b.fits')
	_fits.write_table(f2, [(nme, fmt, t.data[nme]) for nme, fmt in zip(t.names, t.formats)], 'CAT')
	t2 = _fits.read_table(f2)
	for nme in t.names:
		np.testing.assert_array_equal(t2.data[nme], t.data[nme], err_msg=nme)


def test_undecoded_column_types_are_skipped(tmp_path):
	# a table with a complex column between two ordinary ones, assembled by hand
	n = 4
	row = np.dtype([('ra', '>f8'), ('z', 'V8'), ('dec', '>f8')])
	data = np.zeros(n, dtype=row)
	data['ra'] = np.arange(n)
	data['dec'] = -np.arange(n)
	cards = [_fits._card('XTENSION', 'BINTABLE'), _fits._card('BITPIX', 8), _fits._card('NAXIS', 2), _fits._card('NAXIS1', row.itemsize),
		_fits._card('NAXIS2', n), _fits._card('PCOUNT', 0), _fits._card('GCOUNT', 1), _fits._card('TFIELDS', 3), _fits._card('EXTNAME', 'X'),
		_fits._card('TTYPE1', 'ra'), _fits._card('TFORM1', 'D'), _fits._card('TTYPE2', 'z'), _fits._card('TFORM2', 'C'),
		_fits._card('TTYPE3', 'dec'), _fits._card('TFORM3', 'D')]
	primary = [_fits._card('SIMPLE', True), _fits._card('BITPIX', 8), _fits._card('NAXIS', 0), _fits._card('EXTEND', True)]
	raw = data.tobytes()
	path = str(tmp_path / 'c.fits')
	with open(path, 'wb') as f:
		f.write(_fits._header_bytes(primary))
		f.write(_fits._header_bytes(cards))
		f.write(raw + b'\0' * (_fits._pad(len(raw)) - len(raw)))
	with warnings.catch_warnings(record=True) as w:
		warnings.simplefilter('always')
		t = _fits.read_table(path)
	assert t.names == ['ra', 'dec'] and any('not read' in str(x.message) for x in w)
	np.testing.assert_array_equal(t.data['ra'], np.arange(n))
	np.testing.assert_array_equal(t.data['dec'], -np.arange(n))
