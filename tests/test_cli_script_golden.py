"""The product's command line (nway.py -> nway_amd/cli.py) against what the REFERENCE'S SCRIPT wrote.

tests/golden/script_cli.{npz,json} were made by executing /root/reference/nway.py itself
(tests/golden/make_script_golden.py, build container, an I/O-only stand-in for astropy.io.fits): for
every command line below the table it wrote -- column names, order, TFORMs, every computed
column as stored (float32 'E' / int16 'I'), the header keys of the primary HDU, the COMMENT
cards, the ``*_fit.txt`` histogram files.  The product is given the same command line on the
same files (the reference's own tests/elltest catalogues and doc/COSMOS_XMM.fits, byte for byte
under tests/golden/elltest/) and has to write the same table: index and flag columns bit-equal,
floating columns within 1e-6 relative of the float32 values (BASELINE.json's contract).
"""
import json
import os
import shutil

import numpy as np
import pytest

from goldenutil import GOLDEN, golden, mag3_tables

pytestmark = pytest.mark.gpu

META = json.load(open(os.path.join(GOLDEN, 'script_cli.json')))
ELL = ['ell2_minprob', 'ell2', 'ell3_minprob', 'ell3_ignore', 'ell3_opts', 'ell3']
MAG = ['mag_post', 'mag_rad', 'mag_rad_excl', 'mag_minprob', 'mag_file']
# elliptical / asymmetric position errors (script_ell.*: the script run with the coordinate-frame stand-in of fits_standin.py -- what these
# pin is the script's branch around fastskymatch.dist3d, not astropy's frames)
META_ELLIPSE = json.load(open(os.path.join(GOLDEN, 'script_ell.json')))
ELLIPSE = ['rot3', 'asym2', 'mixed3']
RTOL, ATOL = 1e-6, 1e-12
# the elliptical branch computes with float32 OFFSET columns which the device and numpy evaluate with different libm (one float32 ulp, or 1e-6
# arcsec near zero); the Bayes factors computed FROM those float32 values inherit it: the contract's 1e-6 relative on top of that rounding
OFFSET_RTOL, OFFSET_ATOL = 2.4e-7, 1e-6
ELLIPSE_RTOL, ELLIPSE_ATOL = 2e-6, 1e-9


def stage_inputs(tag, tmp_path):
	from nway_amd import _fits
	if tag in ELLIPSE:
		flow = golden('ell_flow')
		for c, name in enumerate(['X', 'O', 'I']):
			col = lambda key: flow['in%d_%s' % (c, key)]
			_fits.write_table(str(tmp_path / (name + '.fits')), [('ID', 'J', np.arange(len(col('ra')))), ('RA', 'D', col('ra')), ('DEC', 'D', col('dec')),
				('major', 'D', col('major')), ('minor', 'D', col('minor')), ('angle', 'D', col('angle'))], name, table_header={'SKYAREA': float(flow['area'][0])})
		return
	if tag in ELL:
		for f in ('randomcatX.fits', 'randomcatR.fits', 'randomcatO.fits'):
			shutil.copy(os.path.join(GOLDEN, 'elltest', f), str(tmp_path / f))
		return
	shutil.copy(os.path.join(GOLDEN, 'elltest', 'COSMOS_XMM.fits'), str(tmp_path / 'COSMOS_XMM.fits'))
	X, O, I = mag3_tables()
	_fits.write_table(str(tmp_path / 'OPT.fits'), [('ID', 'J', np.arange(len(O['ra']))), ('RA', 'D', O['ra']), ('DEC', 'D', O['dec']),
		('R', 'D', O['mags'][0]), ('I', 'D', O['mags'][1])], 'OPT', table_header={'SKYAREA': O['area']})
	_fits.write_table(str(tmp_path / 'IRAC.fits'), [('ID', 'J', np.arange(len(I['ra']))), ('RA', 'D', I['ra']), ('DEC', 'D', I['dec']),
		('CH1', 'D', I['mags'][0])], 'IRAC', table_header={'SKYAREA': I['area']})
	if tag == 'mag_file':
		# the histogram the script stored in its posterior run, handed back as a file (nway.py:501-504)
		with open(str(tmp_path / 'mag_post_OPT_R_fit.txt'), 'w') as f:
			f.write(META['mag_post']['histogram_files']['OPT_R_fit.txt'])


def compare_float(got, want, name, rtol=RTOL, atol=ATOL):
	got, want = np.asarray(got, dtype=float), np.asarray(want, dtype=float)
	np.testing.assert_array_equal(np.isnan(got), np.isnan(want), err_msg=name)
	np.testing.assert_allclose(got, want, rtol=rtol, atol=atol, equal_nan=True, err_msg=name)


def check_input_copies(out, tmp_path, meta):
	"""every input column copied as {TABLE}_{column}, -99 where the catalogue has no source in the row (fastskymatch.py:265-282)"""
	from nway_amd import _fits
	for filename, table_name in zip(meta['primary_header']['INPUT'].split(', '), meta['primary_header']['TABLES'].split(', ')):
		t = _fits.read_table(str(tmp_path / filename))
		ids = np.asarray(t.data['ID'])
		order = np.argsort(ids)
		got_id = np.asarray(out.data[table_name + '_ID'])
		present = got_id != -99
		rows = order[np.searchsorted(ids[order], got_id[present])]
		np.testing.assert_array_equal(ids[rows], got_id[present])
		for col in t.data.dtype.names:
			got = np.asarray(out.data['%s_%s' % (table_name, col)])
			np.testing.assert_array_equal(got[present], np.asarray(t.data[col])[rows], err_msg=col)
			np.testing.assert_array_equal(got[~present], -99, err_msg=col)


@pytest.mark.parametrize('tag', ELL + MAG + ELLIPSE)
def test_cli_writes_what_the_script_writes(tag, tmp_path, monkeypatch):
	from nway_amd import _fits, cli
	meta, g = (META_ELLIPSE[tag], golden('script_ell')) if tag in ELLIPSE else (META[tag], golden('script_cli'))
	monkeypatch.chdir(tmp_path)
	stage_inputs(tag, tmp_path)
	assert cli.main(list(meta['argv'])) == 0
	outfile = tag + '.fits'
	out = _fits.read_table(outfile)
	# layout: names, order, storage types
	assert out.name == meta['extname'] == 'NWAYMATCH'
	assert out.names == meta['columns']
	assert out.formats == meta['formats']
	assert len(out.data) == meta['nrows']
	# header of the primary HDU (nway.py:640-647, fastskymatch.py:287-290,353-354)
	head = _fits.read_header(outfile, 0)
	assert ('DATE' in head) == meta['has_date']
	for key, want in meta['primary_header'].items():
		if key == 'NWAYCMD':
			assert head[key].split()[1:] == want.split()[1:]  # all but the path of the script itself
		else:
			assert head[key] == want, key
	squeeze = lambda texts: ''.join(texts).replace(' ', '')
	assert squeeze(head['_COMMENTS']) == squeeze(meta['comments'])
	# the computed columns
	inputs = set()
	for table_name in meta['primary_header']['TABLES'].split(', '):
		inputs.update(n for n in meta['columns'] if n.startswith(table_name + '_'))
	computed = [n for n in meta['columns'] if n not in inputs or n.endswith('_ID')]
	rows = g[tag + '/rows'] if (tag + '/rows') in g.files else slice(None)
	for name in computed:
		want = g['%s/%s' % (tag, name)]
		got = np.asarray(out.data[name])[rows]
		if want.dtype.kind in 'iu':
			np.testing.assert_array_equal(got, want, err_msg=name)
		else:
			assert np.asarray(out.data[name]).dtype == np.float32
			if tag in ELLIPSE and (name.endswith('_ra') or name.endswith('_dec')):
				compare_float(got, want, name, OFFSET_RTOL, OFFSET_ATOL)
			elif tag in ELLIPSE:
				compare_float(got, want, name, ELLIPSE_RTOL, ELLIPSE_ATOL)
			else:
				compare_float(got, want, name)
	if (tag + '/rows') in g.files:
		np.testing.assert_array_equal(out.data['match_flag'], g[tag + '/all/match_flag'])
		np.testing.assert_array_equal(out.data['ncat'], g[tag + '/all/ncat'])
		for name in computed:
			key = '%s/sum/%s' % (tag, name)
			if key in g.files:
				np.testing.assert_allclose(np.nansum(np.asarray(out.data[name]), dtype=float), g[key][0], rtol=1e-6, err_msg=name)
	check_input_copies(out, tmp_path, meta)
	# the histograms the script stores next to its output (nway.py:491-497)
	for name, text in meta.get('histogram_files', {}).items():
		want = np.loadtxt(text.splitlines())
		got = np.loadtxt(name)
		assert open(name).readline() == text.splitlines(True)[0]
		np.testing.assert_allclose(got, want, rtol=0, atol=1.001e-5, err_msg=name)


def test_nway_py_as_a_program(tmp_path):
	"""``python nway.py ...`` itself (the drop-in's entry script, a process of its own) on the reference's test catalogues: the table the
	reference's script wrote, and NWAYCMD = the command line as typed (nway.py:644)"""
	import subprocess
	import sys
	from goldenutil import ROOT
	from nway_amd import _fits
	meta, g = META['ell2'], golden('script_cli')
	stage_inputs('ell2', tmp_path)
	script = os.path.join(ROOT, 'nway.py')
	res = subprocess.run([sys.executable, script] + list(meta['argv']), cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
	assert res.returncode == 0, res.stderr[-2000:]
	assert 'matches after filtering by search radius' in res.stdout and 'creating output FITS file' in res.stdout
	out = _fits.read_table(str(tmp_path / 'ell2.fits'))
	assert out.names == meta['columns'] and len(out.data) == meta['nrows']
	np.testing.assert_array_equal(out.data['match_flag'], g['ell2/match_flag'])
	compare_float(out.data['p_i'], g['ell2/p_i'], 'p_i')
	head = _fits.read_header(str(tmp_path / 'ell2.fits'), 0)
	assert head['NWAYCMD'] == ' '.join([script] + list(meta['argv']))


@pytest.mark.parametrize('tag', ['mm_rot3', 'mm_circ3'])
def test_match_multiple_returns_what_it_returned_to_the_script(tag, tmp_path):
	"""``fastskymatch.match_multiple`` (fastskymatch.py:228-342) against what the reference's returned to the script at nway.py:267
	(tests/golden/script_mm.*: recorded from the script's globals): the index array field by field, the columns up to ``ncat`` in the
	same order with the same TFORMs, their values (input copies and indices exactly; the 'E' separations as float32, one ulp for the
	device's libm; the per-axis offsets of circular=False as in the table test above), the two header keys"""
	from nway_amd import _fits, NullOutputLogger
	from nway_amd.fastskymatch import match_multiple
	meta = json.load(open(os.path.join(GOLDEN, 'script_mm.json')))[tag]
	g = golden('script_mm')
	stage_inputs('rot3', tmp_path)
	tables = [_fits.read_table(str(tmp_path / (n + '.fits'))) for n in meta['table_names']]
	results, columns, header = match_multiple([t.data for t in tables], meta['table_names'], meta['radius_deg'], [t.formats for t in tables],
		logger=NullOutputLogger(), circular=meta['circular'])
	for n in meta['table_names']:
		np.testing.assert_array_equal(results[n], g['%s/results/%s' % (tag, n)], err_msg=n)
	assert [c.name for c in columns] == meta['columns']
	assert [c.format for c in columns] == meta['formats']
	for c in columns:
		want = g['%s/col/%s' % (tag, c.name)]
		if want.dtype.kind in 'iu':
			np.testing.assert_array_equal(np.asarray(c.array, dtype=np.int64), want.astype(np.int64), err_msg=c.name)
		elif c.format == 'D':
			np.testing.assert_array_equal(np.asarray(c.array, dtype=float), want, err_msg=c.name)
		else:
			assert want.dtype == np.float32
			with np.errstate(invalid='ignore'):
				got = np.asarray(c.array, dtype=float).astype(np.float32)
			if c.name.endswith('_ra') or c.name.endswith('_dec'):
				compare_float(got, want, c.name, OFFSET_RTOL, OFFSET_ATOL)
			else:
				compare_float(got, want, c.name, 2.4e-7, 0.0)
	for key in ('COLS_RA', 'COLS_DEC'):
		assert header[key] == meta['header'][key]
	assert sorted(header) == ['COLS_DEC', 'COLS_RA']
