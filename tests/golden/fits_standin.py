"""An I/O-only stand-in for ``astropy.io.fits`` -- BUILD CONTAINER ONLY (listed in .gpurunignore).

astropy is not installed in this image, and the reference's command-line script
(/root/reference/nway.py) cannot start without it.  This module supplies exactly the part of
``astropy.io.fits`` that the script and ``nwaylib.fastskymatch.match_multiple`` touch, so that
``tests/golden/make_script_golden.py`` can EXECUTE the reference's script (``runpy``) and record
what it writes.  It reads and writes FITS binary tables and carries arrays; it holds no
arithmetic of the match: no separation, no Bayes factor, no histogram, no flag.  Everything it
does is a change of container or of storage type that astropy performs for the script:

* ``open(name)[1]`` -> an HDU with ``.name`` (EXTNAME), ``.header`` (mapping), ``.columns``
  (objects with ``.name`` / ``.format`` = the TFORM string) and ``.data`` = a numpy structured
  array with BIG-ENDIAN fields, as astropy's FITS_rec presents an unscaled table: a column read
  as ``table['RA']`` is a view ('>f8'), so in-place edits reach the table (nway.py:447 marks
  -99 magnitudes as NaN that way) and a float32 'E' column stays float32 in arithmetic.
* ``Column(name=, format=, array=)``: astropy converts the array to the column's storage type
  WHEN THE COLUMN IS MADE (astropy/io/fits/column.py: ``_convert_to_valid_data_type`` ->
  ``_convert_array``: the same object if the dtype already matches, else ``array.astype``).
  Consequences the script relies on, reproduced here and stated because they cannot be
  verified against astropy in this image:
    - ``Column('dist_bayesfactor', 'E', log_bf)`` (nway.py:361) holds a float32 COPY taken
      before the correction loop changes ``log_bf`` in place (nway.py:420): the column keeps the
      uncorrected values; ``dist_bayesfactor_corrected`` (nway.py:421) is a second copy.
    - a column whose array already has the storage type (an input 'D' column gathered from a
      catalogue, fastskymatch.py:281) is NOT copied.
* ``BinTableHDU.from_columns(ColDefs)`` builds a new big-endian record array from the columns
  (a cast to the TFORM type: THE float32 trip of the separations, fastskymatch.py:328 ->
  nway.py:286,303); ``.data['x']`` of it is what the script computes with.
* ``PrimaryHDU``, ``HDUList``, ``Header.add_comment/update``, ``HDUList.writeto``: a real FITS
  file is written (long strings with the CONTINUE convention astropy uses, long comments split
  over COMMENT cards).

Independent of nway_amd/_fits.py (the product's reader/writer) on purpose: the fixtures must
not inherit a reading error of the code they test.

One section at the end is NOT I/O and says so: the few names of ``astropy.coordinates`` /
``astropy.units`` that ``fastskymatch.dist3d`` uses, for the script runs through the elliptical
branch (``install(coordinates=True)``).
"""
import builtins
import os
import sys
import types

import numpy as np

_REC = {'L': 'i1', 'B': 'u1', 'I': 'i2', 'J': 'i4', 'K': 'i8', 'E': 'f4', 'D': 'f8'}


def _parse_tform(tform):
	t = str(tform).strip()
	i = 0
	while i < len(t) and t[i].isdigit():
		i += 1
	repeat = int(t[:i]) if i else 1
	code = t[i]
	return repeat, code


class Header(object):
	"""ordered keyword -> value mapping with COMMENT cards (what the script uses of astropy's Header)"""

	def __init__(self, cards=()):
		self._keys = []
		self._values = {}
		self.comments = []
		for k, v in cards:
			self[k] = v

	def __contains__(self, key):
		return str(key).upper() in self._values

	def __getitem__(self, key):
		return self._values[str(key).upper()]

	def __setitem__(self, key, value):
		key = str(key).upper()
		if key not in self._values:
			self._keys.append(key)
		self._values[key] = value

	def get(self, key, default=None):
		return self._values.get(str(key).upper(), default)

	def keys(self):
		return list(self._keys)

	def items(self):
		return [(k, self._values[k]) for k in self._keys]

	def update(self, mapping):
		for k, v in dict(mapping).items():
			self[k] = v

	def add_comment(self, text):
		self.comments.append(str(text))


def _parse_card_value(text):
	text = text.strip()
	if text.startswith("'"):
		# string: up to the closing quote ('' = an embedded quote)
		out, i = [], 1
		while i < len(text):
			if text[i] == "'":
				if i + 1 < len(text) and text[i + 1] == "'":
					out.append("'")
					i += 2
					continue
				break
			out.append(text[i])
			i += 1
		return ''.join(out).rstrip()
	value = text.split('/')[0].strip()
	if value == 'T':
		return True
	if value == 'F':
		return False
	if value == '':
		return None
	try:
		return int(value)
	except ValueError:
		return float(value.replace('D', 'E'))


def _read_header(buf, pos):
	header = Header()
	last = None
	while True:
		block = buf[pos:pos + 2880]
		if len(block) < 2880:
			raise IOError('truncated FITS header')
		pos += 2880
		done = False
		for c in range(36):
			card = block[c * 80:(c + 1) * 80].decode('ascii')
			key = card[:8].strip()
			if key == 'END':
				done = True
				break
			if key == 'CONTINUE' and last is not None:
				prev = header[last]
				if isinstance(prev, str) and prev.endswith('&'):
					prev = prev[:-1]
				header[last] = prev + _parse_card_value(card[8:])
				continue
			if key == 'COMMENT':
				header.comments.append(card[8:].rstrip())
				continue
			if card[8:10] != '= ':
				continue
			header[key] = _parse_card_value(card[10:])
			last = key
		if done:
			break
	return header, pos


class _ColumnInfo(object):
	def __init__(self, name, format):
		self.name = name
		self.format = format


class _TableHDU(object):
	def __init__(self, header, data, columns):
		self.header = header
		self.data = data
		self.columns = columns

	@property
	def name(self):
		return self.header.get('EXTNAME', '')


class PrimaryHDU(object):
	def __init__(self):
		self.header = Header()
		self.data = None


class HDUList(list):
	def writeto(self, filename, overwrite=False):
		if os.path.exists(filename) and not overwrite:
			raise OSError('File %s already exists' % filename)
		with builtins.open(filename, 'wb') as f:
			f.write(_encode_header([('SIMPLE', True), ('BITPIX', 8), ('NAXIS', 0), ('EXTEND', True)] + self[0].header.items(),
				self[0].header.comments))
			for hdu in self[1:]:
				_write_table(f, hdu)


def open(filename):
	"""every HDU of a FITS file; binary tables become _TableHDU (other extensions: header only)"""
	with builtins.open(filename, 'rb') as f:
		buf = f.read()
	hdus = HDUList()
	pos = 0
	while pos < len(buf):
		header, pos = _read_header(buf, pos)
		naxis = header.get('NAXIS', 0)
		nbytes = abs(header.get('BITPIX', 8)) // 8
		if naxis:
			for a in range(1, naxis + 1):
				nbytes *= header['NAXIS%d' % a]
		else:
			nbytes = 0
		nbytes = (nbytes + header.get('PCOUNT', 0)) * header.get('GCOUNT', 1) if naxis else 0
		if header.get('XTENSION', '').strip() == 'BINTABLE':
			fields, infos = [], []
			for c in range(1, header['TFIELDS'] + 1):
				name, tform = header['TTYPE%d' % c], header['TFORM%d' % c]
				repeat, code = _parse_tform(tform)
				if code == 'A':
					dt = 'S%d' % repeat
				elif code in _REC:
					dt = '>' + _REC[code] if _REC[code][1] != '1' else _REC[code]
					if repeat != 1:
						dt = (dt, (repeat,))
				else:
					raise NotImplementedError('TFORM %r' % tform)
				for scale in ('TSCAL%d' % c, 'TZERO%d' % c):
					if scale in header and header[scale] not in (0, 1, 0.0, 1.0):
						raise NotImplementedError('scaled column %s' % name)
				fields.append((name, dt))
				infos.append(_ColumnInfo(name, tform.strip()))
			dtype = np.dtype(fields)
			assert dtype.itemsize == header['NAXIS1'], (dtype.itemsize, header['NAXIS1'])
			data = np.frombuffer(buf, dtype=dtype, count=header['NAXIS2'], offset=pos).copy()
			hdus.append(_TableHDU(header, data, infos))
		else:
			p = PrimaryHDU()
			p.header = header
			hdus.append(p)
		pos += (nbytes + 2879) // 2880 * 2880
	return hdus


def _convert_array(array, dtype):
	"""astropy/io/fits/column.py:_convert_array -- the same object when nothing changes"""
	if array.dtype == dtype:
		return array
	if array.dtype.itemsize == dtype.itemsize and not (np.issubdtype(array.dtype, np.number) and np.issubdtype(dtype, np.number)):
		return array.view(dtype)
	return array.astype(dtype)


class Column(object):
	def __init__(self, name=None, format=None, array=None):
		self.name = name
		self.format = str(format).strip()
		repeat, code = _parse_tform(self.format)
		if array is not None:
			array = np.asarray(array)
			if code in _REC and code != 'L':
				# "preserve byte order of the original array" (column.py): byteorder + recformat
				order = array.dtype.byteorder if array.dtype.byteorder in '<>' else '='
				array = _convert_array(array, np.dtype(order + _REC[code]))
			elif code == 'A':
				array = np.asarray(array)
			else:
				raise NotImplementedError('Column format %r' % self.format)
		self.array = array


class ColDefs(list):
	"""a list of columns; made from a table HDU it lists that table's columns over its data
	(astropy: ColDefs._init_from_table -- fastskymatch.py:346 is handed an HDU by nway.py:639)"""

	def __init__(self, columns):
		if isinstance(columns, _TableHDU):
			hdu = columns
			columns = [Column(name=c.name, format=c.format, array=hdu.data[c.name]) for c in hdu.columns]
		list.__init__(self, columns)


class BinTableHDU(_TableHDU):
	@staticmethod
	def from_columns(columns):
		fields = []
		for c in columns:
			repeat, code = _parse_tform(c.format)
			if code == 'A':
				dt = 'S%d' % repeat
			else:
				dt = '>' + _REC[code] if _REC[code][1] != '1' else _REC[code]
				if repeat != 1:
					dt = (dt, (repeat,))
			fields.append((c.name, dt))
		n = len(columns[0].array) if len(columns) else 0
		data = np.zeros(n, dtype=np.dtype(fields))
		for c in columns:
			assert len(c.array) == n, (c.name, len(c.array), n)
			data[c.name] = c.array  # the cast to the storage type
		header = Header()
		return _TableHDU(header, data, [_ColumnInfo(c.name, c.format) for c in columns])


def _format_value(value):
	if isinstance(value, (bool, np.bool_)):
		return '%20s' % ('T' if value else 'F')
	if isinstance(value, (int, np.integer)):
		return '%20d' % value
	if isinstance(value, (float, np.floating)):
		return '%20s' % repr(float(value)).upper()
	raise TypeError(type(value))


def _encode_header(items, comments=()):
	cards = []
	for key, value in items:
		if isinstance(value, str):
			text = value.replace("'", "''")
			if len(text) <= 68:
				cards.append('%-8s= %-20s' % (key, "'%-8s'" % text))
			else:
				# the CONTINUE long-string convention: pieces of 67 characters, all but the last end in '&'
				pieces = [text[i:i + 67] for i in range(0, len(text), 67)]
				for n, piece in enumerate(pieces):
					amp = '&' if n + 1 < len(pieces) else ''
					cards.append(('%-8s= ' % key if n == 0 else 'CONTINUE  ') + "'%s%s'" % (piece, amp))
		else:
			cards.append('%-8s= %s' % (key, _format_value(value)))
	for text in comments:
		for i in range(0, max(len(text), 1), 72):
			cards.append('COMMENT ' + text[i:i + 72])
	cards.append('END')
	raw = ''.join('%-80s' % c[:80] for c in cards)
	raw += ' ' * (-len(raw) % 2880)
	return raw.encode('ascii')


def _write_table(f, hdu):
	data = hdu.data
	items = [('XTENSION', 'BINTABLE'), ('BITPIX', 8), ('NAXIS', 2), ('NAXIS1', data.dtype.itemsize), ('NAXIS2', len(data)),
		('PCOUNT', 0), ('GCOUNT', 1), ('TFIELDS', len(hdu.columns))]
	for n, c in enumerate(hdu.columns):
		items.append(('TTYPE%d' % (n + 1), c.name))
		items.append(('TFORM%d' % (n + 1), c.format))
	items += [(k, v) for k, v in hdu.header.items() if k not in dict(items)]
	f.write(_encode_header(items, hdu.header.comments))
	raw = data.tobytes()
	f.write(raw)
	f.write(b'\0' * (-len(raw) % 2880))


def writeto(filename, data, header=None, overwrite=False):
	"""signature only: nwaylib/progress.py:11 inspects it for the name of the overwrite argument"""
	raise NotImplementedError


def write_catalogue(filename, extname, skyarea, columns):
	"""an input catalogue for the script: ``columns`` = [(name, TFORM, array)].  Not part of astropy's
	surface; used by the generator to put seeded synthetic catalogues on disk."""
	hdu = BinTableHDU.from_columns(ColDefs([Column(name=n, format=t, array=a) for n, t, a in columns]))
	hdu.header['EXTNAME'] = extname
	hdu.header['SKYAREA'] = float(skyarea)
	HDUList([PrimaryHDU(), hdu]).writeto(filename, overwrite=True)


# ---------------------------------------------------------------------------------------------
# astropy.coordinates / astropy.units, as far as fastskymatch.dist3d (:50-74) uses them -- the ONE part of this file that is not
# I/O: the script's elliptical branch calls SkyCoord(ra, dec, frame="icrs", unit="deg"), SkyOffsetFrame(origin=a),
# transform_to, .lon / .lat, separation, .to(u.degree).value.  The rotation into the offset frame is the restatement of what
# astropy documents for that frame (oracle/elliptical_oracle.py: offsets -- origin to (0, 0), no roll; pinned by hand-computed
# known answers, tests/test_elliptical_helpers.py), the separation is the reference's own Vincenty formula (fastskymatch.dist,
# the expression astropy's angular_separation documents).  What a script run under this stand-in pins is therefore the SCRIPT --
# its error columns, its float32 trips, its branch logic, its output table -- around these two functions, not astropy itself.
class _Degrees(object):
	def __init__(self, value):
		self.value = np.asarray(value, dtype=float)

	def __sub__(self, other):
		return _Degrees(self.value - other.value)

	def to(self, unit):
		assert unit is DEGREE, unit
		return self


class _UnitDegree(object):
	pass


DEGREE = _UnitDegree()


class SkyOffsetFrame(object):
	def __init__(self, origin):
		self.origin = origin


class _OffsetPosition(object):
	def __init__(self, lon, lat):
		self.lon, self.lat = _Degrees(lon), _Degrees(lat)


class SkyCoord(object):
	def __init__(self, ra, dec, frame='icrs', unit='deg'):
		assert frame == 'icrs' and unit == 'deg', (frame, unit)
		self.ra, self.dec = np.asarray(ra, dtype=float), np.asarray(dec, dtype=float)

	def transform_to(self, frame):
		sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'oracle'))
		import elliptical_oracle
		o = frame.origin
		# offsets() returns origin minus point in the origin's frame, the origin being (0, 0) there: the point is its negative
		dlon, dlat = elliptical_oracle.offsets(o.ra, o.dec, self.ra, self.dec)
		return _OffsetPosition(-dlon, -dlat)

	def separation(self, other):
		from nwaylib.fastskymatch import dist  # (the reference's; loaded by the time the script gets here)
		return _Degrees(dist((self.ra, self.dec), (other.ra, other.dec)))


def install(healpix=False, coordinates=False):
	"""register this module as astropy.io.fits, plus the empty astropy / healpy stand-ins the import
	of nwaylib needs (tests/golden/ref_harness.py explains them); ``coordinates``: the stand-in above for dist3d"""
	me = sys.modules[__name__]
	io = types.ModuleType('astropy.io')
	io.fits = me
	units = types.ModuleType('astropy.units')
	units.degree = DEGREE
	coords = types.ModuleType('astropy.coordinates')
	coords.SkyCoord = SkyCoord if coordinates else None
	coords.SkyOffsetFrame = SkyOffsetFrame if coordinates else None
	top = types.ModuleType('astropy')
	top.io, top.units, top.coordinates = io, units, coords
	sys.modules.update({'astropy': top, 'astropy.io': io, 'astropy.io.fits': me, 'astropy.units': units, 'astropy.coordinates': coords})
	if healpix:
		sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
		from oracle import healpix as own
		pixelfunc = types.ModuleType('healpy.pixelfunc')
		pixelfunc.nside2resol, pixelfunc.ang2pix, pixelfunc.get_all_neighbours = own.nside2resol, own.ang2pix, own.get_all_neighbours
		hp = types.ModuleType('healpy')
		hp.pixelfunc = pixelfunc
		sys.modules.update({'healpy': hp, 'healpy.pixelfunc': pixelfunc})
	elif 'healpy' not in sys.modules:
		sys.modules['healpy'] = types.ModuleType('healpy')
	return me
