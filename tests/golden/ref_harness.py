"""Import the pure-Python reference (/root/reference/nwaylib) in THIS container.

Only used by ``make_golden.py`` to generate the committed golden vectors; the
reference never travels to the GPU box and nothing under tests/ imports this
module at test time.

astropy and healpy are not installed here, so empty stand-in modules are
registered before the import (the reference touches them only at import time
on the code paths exercised below: fastskymatch.py:12-18,222-223,
progress.py:11).  With the empty healpy stand-in only the flat-cell branch of
``crossproduct`` (fastskymatch.py:123-133) can be executed.  ``load_reference(healpix=True)``
instead registers oracle/healpix.py (an own restatement of the published pixelisation, see
its header) under the three names the HEALPix branch calls (fastskymatch.py:84,139-140): the
fixtures made that way pin the reference's branch LOGIC (bucket filling, the -1 neighbour key,
nside choice, radius filter) but not healpy's pixel numbering, and say so.
"""
import os
import sys
import tempfile
import types

REFERENCE = '/root/reference'


def _stub(name, **attrs):
	m = types.ModuleType(name)
	m.__dict__.update(attrs)
	sys.modules[name] = m
	return m


def load_reference(healpix=False):
	if not os.path.isdir(REFERENCE):
		raise RuntimeError('reference checkout %s not present' % REFERENCE)

	class _BinTable(object):
		@staticmethod
		def from_columns(*a, **k):
			raise NotImplementedError

	fits = _stub('astropy.io.fits', BinTableHDU=_BinTable,
		writeto=lambda filename, data, header=None, overwrite=False: None)
	io = _stub('astropy.io', fits=fits)
	units = _stub('astropy.units')
	coords = _stub('astropy.coordinates', SkyCoord=None, SkyOffsetFrame=None)
	_stub('astropy', io=io, units=units, coordinates=coords)
	if healpix:
		sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
		from oracle import healpix as own
		pixelfunc = _stub('healpy.pixelfunc', nside2resol=own.nside2resol, ang2pix=own.ang2pix,
			get_all_neighbours=own.get_all_neighbours)
		_stub('healpy', pixelfunc=pixelfunc)
	else:
		_stub('healpy')
	import matplotlib
	matplotlib.use('Agg')
	# the import creates ./cache (fastskymatch.py:21-23): do it in a scratch cwd
	cwd = os.getcwd()
	scratch = tempfile.mkdtemp(prefix='nwayref_')
	os.chdir(scratch)
	sys.path.insert(0, REFERENCE)
	try:
		import nwaylib
	finally:
		sys.path.remove(REFERENCE)
		os.chdir(cwd)
	assert nwaylib.__file__.startswith(REFERENCE), nwaylib.__file__
	return nwaylib
