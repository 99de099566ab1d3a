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
	last = None  # the keyword a CONTINUE card (long-string convention) carries on
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
			elif key == 'CONTINUE' and last is not None and isinstance(header[last], str):
				# a string value too long for one card: every piece but the last ends in '&' (what astropy writes for
				# the script's NWAYCMD, nway.py:644)
				head = header[last]
				header[last] = (head[:-1] if head.endswith('&') else head) + str(_parse_value(card[8:]))
			elif card[8:10] == '= ':
				header[key] = _parse_value(card[10:])
				last = key
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
	fields = []     # every column, for the row layout
	kept = []       # (field, tform, scale, zero) of the columns that are handed on
	# bytes per element of the column types that are not decoded (bit, complex, descriptors): such a
	# column is skipped with a warning instead of making the whole catalogue unreadable
	other = {'X': None, 'C': 8, 'M': 16, 'P': 8, 'Q': 16}
	for i in range(1, header['TFIELDS'] + 1):
		name = header['TTYPE%d' % i]
		tform = header['TFORM%d' % i].strip()
		m = re.match(r'^(\d*)([LBIJKEDAXCMPQ])', tform)
		if not m:
			raise FitsError('unsupported TFORM "%s" for column "%s"' % (tform, name))
		rep = int(m.group(1)) if m.group(1) else 1
		letter = m.group(2)
		if letter in other:
			nbytes = (rep + 7) // 8 if letter == 'X' else rep * other[letter]
			if letter in 'PQ':
				nbytes = other[letter] * (1 if m.group(1) == '' else min(rep, 1))
			fields.append(('_skip%d' % i, 'V%d' % nbytes))
			import warnings
			warnings.warn('column "%s" (TFORM %s) is not read' % (name, tform))
			continue
		if letter == 'A':
			fld = (name, 'S%d' % rep)
		elif rep == 1:
			fld = (name, _TFORM[letter][0])
		else:
			fld = (name, _TFORM[letter][0], (rep,))
		fields.append(fld)
		kept.append((fld, tform, header.get('TSCAL%d' % i, 1), header.get('TZERO%d' % i, 0)))
	be = numpy.dtype(fields)
	if be.itemsize != width:
		raise FitsError('row width mismatch: header says %d, columns give %d' % (width, be.itemsize))
	table = numpy.frombuffer(raw, dtype=be, count=nrows)
	native = []
	for fld, tform, scale, zero in kept:
		base = numpy.dtype(fld[1]).newbyteorder('=')
		if (scale != 1 or zero != 0) and base.kind in 'iu':
			# scaled integers: unsigned columns are stored with TZERO = 2^(bits - 1) (the FITS convention,
			# what astropy writes for IDs); anything else becomes float64 like astropy's
			unsigned = scale == 1 and zero == 2 ** (8 * base.itemsize - 1)
			signed_byte = scale == 1 and base == numpy.dtype('u1') and zero == -128  # the other standard convention: 'B' + TZERO = -128 is int8
			base = numpy.dtype('u%d' % base.itemsize) if unsigned else (numpy.dtype('i1') if signed_byte else numpy.dtype('f8'))
		native.append((fld[0], base) + tuple(fld[2:]))
	out = numpy.empty(nrows, dtype=numpy.dtype(native))
	formats = []
	for (fld, tform, scale, zero), nat in zip(kept, native):
		col = table[fld[0]]
		if numpy.dtype(nat[1]) == numpy.dtype('i1') and numpy.dtype(fld[1]) == numpy.dtype('u1'):
			out[fld[0]] = (col.astype(numpy.int16) - 128).astype(numpy.int8)
		elif numpy.dtype(nat[1]).kind == 'u' and numpy.dtype(fld[1]).kind == 'i':
			if numpy.dtype(nat[1]).itemsize == 8:
				out[fld[0]] = col.astype(numpy.int64).view(numpy.uint64) ^ numpy.uint64(1 << 63)  # + 2^63 mod 2^64
			else:
				out[fld[0]] = (col.astype(numpy.int64) + int(zero)).astype(nat[1])
		elif scale != 1 or zero != 0:
			out[fld[0]] = col * scale + zero
		else:
			out[fld[0]] = col
		formats.append(tform if numpy.dtype(nat[1]).kind != 'f' or numpy.dtype(fld[1]).kind == 'f' else re.sub(r'[BIJK]', 'D', tform))
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
		v = '%-20s' % v  # fixed format: the value field ends in column 30 (what astropy and STIL write)
	card = '%-8s= %s' % (key[:8], v)
	if comment:
		card += ' / ' + comment
	return ('%-80s' % card)[:80]


def _string_cards(key, value):
	"""a string value as one card, or -- longer than a card holds -- as a card and CONTINUE cards (pieces of 67
	characters, all but the last ending in '&': the long-string convention astropy follows)"""
	text = str(value).replace("'", "''")
	if len(text) <= 68:
		return [_card(key, value)]
	# pieces of at most 67 characters, none ending inside an escaped quote: a piece that ends in an odd run of quotes would leave
	# "...'&'" on its card and "'..." on the next -- a reader (this one, astropy) ends the string at the lone quote
	pieces, at = [], 0
	while at < len(text):
		end = min(at + 67, len(text))
		if end < len(text):
			run = len(text[at:end]) - len(text[at:end].rstrip("'"))
			if run % 2 == 1:
				end -= 1
		pieces.append(text[at:end])
		at = end
	cards = []
	for n, piece in enumerate(pieces):
		body = "'%s%s'" % (piece, '&' if n + 1 < len(pieces) else '')
		cards.append(('%-80s' % (('%-8s= ' % key[:8] if n == 0 else 'CONTINUE  ') + body))[:80])
	return cards


def read_header(filename, hdu=0):
	"""the header of HDU number ``hdu`` as a dict (COMMENT / HISTORY texts under '_COMMENTS')"""
	with open(filename, 'rb') as f:
		buf = f.read()
	off = 0
	for index in range(hdu + 1):
		header, off = _read_header(buf, off)
		off += _pad(_data_size(header))
	return header


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
	tzero = {}
	for name, tform, arr in columns:
		m = re.match(r'^(\d*)([LBIJKEDA])', tform)
		rep = int(m.group(1)) if m.group(1) else 1
		letter = m.group(2)
		if letter == 'A':
			fields.append((name, 'S%d' % rep))
		elif rep == 1:
			fields.append((name, _TFORM[letter][0]))
		else:
			fields.append((name, _TFORM[letter][0], (rep,)))  # a vector column keeps its repeat count
		a = numpy.asarray(arr)
		if a.dtype.kind == 'u' and letter in 'IJK' and a.dtype.itemsize == numpy.dtype(_TFORM[letter][0]).itemsize:
			tzero[name] = 2 ** (8 * a.dtype.itemsize - 1)  # the FITS convention for unsigned integers
		if a.dtype == numpy.int8 and letter == 'B':
			tzero[name] = -128  # and for signed bytes
	be = numpy.dtype(fields)
	data = numpy.zeros(nrows, dtype=be)
	for (name, tform, arr), fld in zip(columns, fields):
		arr = numpy.asarray(arr)
		if len(arr) != nrows:
			raise FitsError('column "%s" has %d rows, expected %d' % (name, len(arr), nrows))
		if fld[1].startswith('S') and arr.dtype.kind == 'U':
			arr = numpy.char.encode(arr, 'ascii')
		if name in tzero:
			arr = (arr.astype(numpy.int64) - tzero[name]) if arr.dtype.itemsize < 8 else (arr ^ numpy.uint64(1 << 63)).view(numpy.int64)
			if tzero[name] == -128:
				arr = arr.astype(numpy.uint8)
		with numpy.errstate(invalid='ignore', over='ignore'):
			data[name] = arr.astype(be[name].base if be[name].subdtype else be[name], copy=False)

	now = datetime.datetime.now().replace(microsecond=0).isoformat()
	pcards = [_card('SIMPLE', True, 'Standard FITS format'), _card('BITPIX', 8), _card('NAXIS', 0),
		_card('EXTEND', True), _card('DATE', now), _card('ANALYSIS', 'NWAY matching')]
	for k, v in (primary_header or {}).items():
		pcards += _string_cards(k, v) if isinstance(v, str) else [_card(k, v)]
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
		if name in tzero:
			tcards.append(_card('TZERO%d' % i, tzero[name]))
	for k, v in (table_header or {}).items():
		tcards += _string_cards(k, v) if isinstance(v, str) else [_card(k, v)]  # (a long string: CONTINUE cards, as in the primary header)
	raw = data.tobytes()
	with open(filename, 'wb') as f:
		f.write(_header_bytes(pcards))
		f.write(_header_bytes(tcards))
		f.write(raw)
		f.write(b'\0' * (_pad(len(raw)) - len(raw)))
