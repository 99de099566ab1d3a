"""Minimal FITS binary-table reader/writer (numpy only).

The reference reads its catalogues with ``astropy.io.fits`` (nway.py:180-198,
fastskymatch.py:222-225,345-363).  astropy is not part of this image and is not
guaranteed on the GPU box, so the host side carries its own implementation of
the small subset of the FITS standard that nway catalogues use: one BINTABLE
extension with scalar columns of TFORM L/B/I/J/K/E/D/A.

This is host-side I/O, not part of the device hot path.
"""
from __future__ import annotations

import datetime
import re

import numpy

BLOCK = 2880
CARD = 80

# TFORM letter -> (big-endian numpy dtype, byte width)
_TFORM = {
	'L': ('S1', 1), 'B': ('u1', 1), 'I': ('>i2', 2), 'J': ('>i4', 4),
	'K': ('>i8', 8), 'E': ('>f4', 4), 'D': ('>f8', 8),
}
# native numpy kind/itemsize -> TFORM letter
_DTYPE2TFORM = {
	('b', 1): 'L', ('u', 1): 'B', ('i', 1): 'I', ('i', 2): 'I', ('i', 4): 'J',
	('i', 8): 'K', ('u', 2): 'J', ('u', 4): 'K', ('f', 4): 'E', ('f', 8): 'D',
}


class FitsError(Exception):
	pass


def _parse_value(raw):
	raw = raw.strip()
	if raw.startswith("'"):
		# string value: up to closing quote (doubled quotes are escapes)
		m = re.match(r"'((?:[^']|'')*)'", raw)
		return m.group(1).replace("''", "'").rstrip() if m else raw.strip("'").rstrip()
	if '/' in raw:
		raw = raw.split('/', 1)[0].strip()
	if raw == 'T':
		return True
	if raw == 'F':
		return False
	if raw == '':
		return None
	try:
		return int(raw)
	except ValueError:
		pass
	try:
		return float(raw.replace('D', 'E'))
	except ValueError:
		return raw


def _read_header(buf, off):
	"""Parse header cards starting at byte ``off``; returns (ordered dict, new offset)."""
	header = {}
	comments = []
	while True:
		blk = buf[off:off + BLOCK]
		if len(blk) < BLOCK:
			raise FitsError('truncated FITS header')
		off += BLOCK
		done = False
		for i in range(0, BLOCK, CARD):
			card = blk[i:i + CARD].decode('ascii', errors='replace')
			key = card[:8].strip()
			if key == 'END':
				done = True
				break
			if key in ('COMMENT', 'HISTORY'):
				comments.append(card[8:].rstrip())
			elif card[8:10] == '= ':
				header[key] = _parse_value(card[10:])
		if done:
			break
	header['_COMMENTS'] = comments
	return header, off


def _data_size(header):
	naxis = header.get('NAXIS', 0)
	if naxis == 0:
		return 0
	size = abs(header['BITPIX']) // 8
	for i in range(1, naxis + 1):
		size *= header['NAXIS%d' % i]
	size = (size + header.get('PCOUNT', 0)) * header.get('GCOUNT', 1)
	return size


def _pad(n):
	return (n + BLOCK - 1) // BLOCK * BLOCK


class Table(object):
	"""A BINTABLE HDU: ``data`` (native-endian structured array), ``header`` dict,
	``name`` (EXTNAME), ``formats`` (TFORM strings per column)."""

	def __init__(self, data, header, formats):
		self.data = data
		self.header = header
		self.formats = formats
		self.name = header.get('EXTNAME', '')

	@property
	def names(self):
		return list(self.data.dtype.names)


def read_table(filename, hdu=1):
	"""Read binary-table extension number ``hdu`` (1 = first extension)."""
	with open(filename, 'rb') as f:
		buf = f.read()
	off = 0
	index = 0
	while off < len(buf):
		header, off = _read_header(buf, off)
		size = _data_size(header)
		if index == hdu:
			if header.get('XTENSION') != 'BINTABLE':
				raise FitsError('HDU %d of "%s" is not a BINTABLE' % (hdu, filename))
			return _decode_table(buf[off:off + size], header)
		off += _pad(size)
		index += 1
	raise FitsError('"%s" has no HDU %d' % (filename, hdu))


def _decode_table(raw, header):
	nrows = header['NAXIS2']
	width = header['NAXIS1']
	fields = []
	formats = []
	for i in range(1, header['TFIELDS'] + 1):
		name = header['TTYPE%d' % i]
		tform = header['TFORM%d' % i].strip()
		m = re.match(r'^(\d*)([LBIJKEDA])', tform)
		if not m:
			raise FitsError('unsupported TFORM "%s" for column "%s"' % (tform, name))
		rep = int(m.group(1)) if m.group(1) else 1
		letter = m.group(2)
		formats.append(tform)
		if letter == 'A':
			fields.append((name, 'S%d' % rep))
		elif rep == 1:
			fields.append((name, _TFORM[letter][0]))
		else:
			fields.append((name, _TFORM[letter][0], (rep,)))
	be = numpy.dtype(fields)
	if be.itemsize != width:
		raise FitsError('row width mismatch: header says %d, columns give %d' % (width, be.itemsize))
	table = numpy.frombuffer(raw, dtype=be, count=nrows)
	native = numpy.dtype([(d[0],) + ((numpy.dtype(d[1]).newbyteorder('='),) + tuple(d[2:])) for d in fields])
	out = numpy.empty(nrows, dtype=native)
	for d in fields:
		out[d[0]] = table[d[0]]
	return Table(out, header, formats)


def _card(key, value, comment=''):
	if isinstance(value, bool):
		v = '%20s' % ('T' if value else 'F')
	elif isinstance(value, (int, numpy.integer)):
		v = '%20d' % value
	elif isinstance(value, (float, numpy.floating)):
		v = '%20s' % repr(float(value)).upper()
	else:
		s = str(value).replace("'", "''")
		v = "'%-8s'" % s
		if len(v) > 70:
			v = v[:69] + "'"
	card = '%-8s= %s' % (key[:8], v)
	if comment:
		card += ' / ' + comment
	return ('%-80s' % card)[:80]


def _header_bytes(cards):
	cards = list(cards) + ['%-80s' % 'END']
	raw = ''.join(cards).encode('ascii', errors='replace')
	return raw + b' ' * (_pad(len(raw)) - len(raw))


def tform_of(array):
	a = numpy.asarray(array)
	if a.dtype.kind in 'SU':
		return '%dA' % max(1, a.dtype.itemsize // (4 if a.dtype.kind == 'U' else 1))
	try:
		return _DTYPE2TFORM[(a.dtype.kind, a.dtype.itemsize)]
	except KeyError:
		raise FitsError('cannot store dtype %s in a FITS table' % a.dtype)


def write_table(filename, columns, extname, primary_header=None, table_header=None, comments=None, overwrite=True):
	"""Write one BINTABLE.

	columns: list of (name, tform, array).  tform is a FITS TFORM string
	('E', 'D', 'I', 'J', 'K', 'nA'); arrays are cast to it, which is how the
	reference's ``pyfits.Column(format='E', array=float64)`` behaves
	(nway.py:361,427,529,581-586).
	"""
	import os
	if os.path.exists(filename) and not overwrite:
		raise FitsError('"%s" exists' % filename)
	nrows = len(columns[0][2]) if columns else 0
	fields = []
	for name, tform, arr in columns:
		m = re.match(r'^(\d*)([LBIJKEDA])', tform)
		rep = int(m.group(1)) if m.group(1) else 1
		letter = m.group(2)
		if letter == 'A':
			fields.append((name, 'S%d' % rep))
		else:
			fields.append((name, _TFORM[letter][0]))
	be = numpy.dtype(fields)
	data = numpy.zeros(nrows, dtype=be)
	for (name, tform, arr), fld in zip(columns, fields):
		arr = numpy.asarray(arr)
		if len(arr) != nrows:
			raise FitsError('column "%s" has %d rows, expected %d' % (name, len(arr), nrows))
		if fld[1].startswith('S') and arr.dtype.kind == 'U':
			arr = numpy.char.encode(arr, 'ascii')
		with numpy.errstate(invalid='ignore', over='ignore'):
			data[name] = arr.astype(be[name], copy=False)

	now = datetime.datetime.now().replace(microsecond=0).isoformat()
	pcards = [_card('SIMPLE', True, 'Standard FITS format'), _card('BITPIX', 8), _card('NAXIS', 0),
		_card('EXTEND', True), _card('DATE', now), _card('ANALYSIS', 'NWAY matching')]
	for k, v in (primary_header or {}).items():
		pcards.append(_card(k, v))
	for c in (comments or []):
		c = str(c)
		while True:
			pcards.append(('COMMENT ' + c[:72] + ' ' * 80)[:80])
			c = c[72:]
			if not c:
				break
	tcards = [_card('XTENSION', 'BINTABLE', 'binary table extension'), _card('BITPIX', 8),
		_card('NAXIS', 2), _card('NAXIS1', be.itemsize), _card('NAXIS2', nrows),
		_card('PCOUNT', 0), _card('GCOUNT', 1), _card('TFIELDS', len(columns)),
		_card('EXTNAME', extname)]
	for i, (name, tform, arr) in enumerate(columns, 1):
		tcards.append(_card('TTYPE%d' % i, name))
		tcards.append(_card('TFORM%d' % i, tform))
	for k, v in (table_header or {}).items():
		tcards.append(_card(k, v))
	raw = data.tobytes()
	with open(filename, 'wb') as f:
		f.write(_header_bytes(pcards))
		f.write(_header_bytes(tcards))
		f.write(raw)
		f.write(b'\0' * (_pad(len(raw)) - len(raw)))
