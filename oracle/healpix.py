"""HEALPix pixelisation, numpy restatement of the published algorithm (test infrastructure).

TEST INFRASTRUCTURE ONLY, like everything under ``oracle/``.  The reference's all-sky branch
(fastskymatch.py:83-88, 135-160) calls three functions of the third-party package **healpy**
(``pyproject.toml:12``, version unpinned upstream, absent from this image and from
/root/reference): ``pixelfunc.nside2resol``, ``pixelfunc.ang2pix(nside, theta, phi, nest=True)``
and ``pixelfunc.get_all_neighbours(nside, theta, phi, nest=True)``.  This file restates them from
the published description of the pixelisation (Gorski et al. 2005, ApJ 622, 759, section 4 and
appendix; the face/neighbour bookkeeping of the HEALPix "xyf" representation) so that the
reference's branch can be exercised here.  The GPU path never uses HEALPix.

Parity status: pinned only against the examples printed in healpy's own documentation
(tests/test_healpix_oracle.py lists them) and against geometric self-checks (equal areas,
neighbour symmetry, neighbours found by displacing points across pixel borders).  healpy itself
was never run against it -- "parity unpinned" at the healpy boundary.
"""
from __future__ import division

import numpy
from numpy import pi

# face -> ring/phi offsets of the face's southern corner ("jrll", "jpll" in the HEALPix papers)
_JRLL = numpy.array([2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4])
_JPLL = numpy.array([1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7])

# neighbour order of get_all_neighbours: SW, W, NW, N, NE, E, SE, S
_DX = numpy.array([-1, -1, 0, 1, 1, 1, 0, -1])
_DY = numpy.array([0, 1, 1, 1, 0, -1, -1, -1])

# which face lies in direction (dx, dy) of each of the 12 faces; row = 3*(dy+1) + (dx+1),
# i.e. S, SE, E, SW, centre, NE, W, NW, N.  -1 = no face there (the 8 corners with 7 neighbours).
_FACE_NEXT = numpy.array([
	[8, 9, 10, 11, -1, -1, -1, -1, 10, 11, 8, 9],
	[5, 6, 7, 4, 8, 9, 10, 11, 9, 10, 11, 8],
	[-1, -1, -1, -1, 5, 6, 7, 4, -1, -1, -1, -1],
	[4, 5, 6, 7, 11, 8, 9, 10, 11, 8, 9, 10],
	[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11],
	[1, 2, 3, 0, 0, 1, 2, 3, 5, 6, 7, 4],
	[-1, -1, -1, -1, 7, 4, 5, 6, -1, -1, -1, -1],
	[3, 0, 1, 2, 3, 0, 1, 2, 4, 5, 6, 7],
	[2, 3, 0, 1, -1, -1, -1, -1, 0, 1, 2, 3]])
# coordinate change when stepping onto that face, per face row (north, equator, south):
# bit 1 = mirror x, bit 2 = mirror y, bit 4 = swap x and y
_FACE_TWIST = numpy.array([
	[0, 0, 3], [0, 0, 6], [0, 0, 0], [0, 0, 5], [0, 0, 0], [5, 0, 0], [0, 0, 0], [6, 0, 0], [3, 0, 0]])


def _order(nside):
	nside = int(nside)
	assert nside > 0 and nside & (nside - 1) == 0, 'nside must be a power of two'
	return nside.bit_length() - 1


def nside2npix(nside):
	return 12 * int(nside) ** 2


def nside2resol(nside):
	"""square root of the pixel area, in radians"""
	return numpy.sqrt(4 * pi / nside2npix(nside))


def _spread(v):
	"""bits of v moved to the even positions (v < 2**30)"""
	v = numpy.asarray(v, dtype=numpy.int64)
	out = numpy.zeros_like(v)
	for b in range(30):
		out |= ((v >> b) & 1) << (2 * b)
	return out


def _squeeze(v):
	v = numpy.asarray(v, dtype=numpy.int64)
	out = numpy.zeros_like(v)
	for b in range(30):
		out |= ((v >> (2 * b)) & 1) << b
	return out


def xyf2nest(nside, ix, iy, face):
	return numpy.asarray(face, dtype=numpy.int64) * (int(nside) ** 2) + _spread(ix) + 2 * _spread(iy)


def nest2xyf(nside, pix):
	pix = numpy.asarray(pix, dtype=numpy.int64)
	n2 = int(nside) ** 2
	inface = pix % n2
	return _squeeze(inface), _squeeze(inface >> 1), pix // n2


def xyf2ring(nside, ix, iy, face):
	nside = int(nside)
	ix, iy, face = [numpy.asarray(a, dtype=numpy.int64) for a in (ix, iy, face)]
	nl4 = 4 * nside
	ncap = 2 * nside * (nside - 1)
	npix = 12 * nside * nside
	jr = _JRLL[face] * nside - ix - iy - 1
	north, south = jr < nside, jr > 3 * nside
	nr = numpy.where(north, jr, numpy.where(south, nl4 - jr, nside))
	n_before = numpy.where(north, 2 * nr * (nr - 1), numpy.where(south, npix - 2 * (nr + 1) * nr, ncap + (jr - nside) * nl4))
	kshift = numpy.where(north | south, 0, (jr - nside) & 1)
	jp = (_JPLL[face] * nr + ix - iy + 1 + kshift) // 2
	jp = numpy.where(jp > 4 * nr, jp - 4 * nr, numpy.where(jp < 1, jp + 4 * nr, jp))
	return n_before + jp - 1


def nest2ring(nside, pix):
	return xyf2ring(nside, *nest2xyf(nside, pix))


def _ang2xyf(nside, theta, phi):
	"""face coordinates of the pixel containing colatitude theta, longitude phi (radians)"""
	nside = int(nside)
	order = _order(nside)
	theta, phi = numpy.broadcast_arrays(numpy.asarray(theta, dtype=float), numpy.asarray(phi, dtype=float))
	z = numpy.cos(theta)
	za = numpy.abs(z)
	tt = numpy.mod(phi, 2 * pi) / (pi / 2)
	tt = numpy.where(tt >= 4, 0.0, tt)
	# equatorial belt: the two families of pixel edges are straight lines in (tt, z)
	t1 = nside * (0.5 + tt)
	t2 = nside * z * 0.75
	jp = numpy.floor(t1 - t2).astype(numpy.int64)
	jm = numpy.floor(t1 + t2).astype(numpy.int64)
	ifp, ifm = jp >> order, jm >> order
	face_eq = numpy.where(ifp == ifm, ifp | 4, numpy.where(ifp < ifm, ifp, ifm + 8))
	ix_eq = jm & (nside - 1)
	iy_eq = nside - (jp & (nside - 1)) - 1
	# polar caps
	ntt = numpy.minimum(3, tt.astype(numpy.int64))
	tp = tt - ntt
	tmp = nside * numpy.sqrt(3 * (1 - za))
	jpc = numpy.minimum((tp * tmp).astype(numpy.int64), nside - 1)
	jmc = numpy.minimum(((1.0 - tp) * tmp).astype(numpy.int64), nside - 1)
	up = z >= 0
	face_cap = numpy.where(up, ntt, ntt + 8)
	ix_cap = numpy.where(up, nside - jmc - 1, jpc)
	iy_cap = numpy.where(up, nside - jpc - 1, jmc)
	belt = za <= 2.0 / 3.0
	return (numpy.where(belt, ix_eq, ix_cap), numpy.where(belt, iy_eq, iy_cap), numpy.where(belt, face_eq, face_cap))


def ang2pix(nside, theta, phi, nest=False):
	ix, iy, face = _ang2xyf(nside, theta, phi)
	return xyf2nest(nside, ix, iy, face) if nest else xyf2ring(nside, ix, iy, face)


def _neighbours_xyf(nside, ix, iy, face, nest):
	nside = int(nside)
	ix, iy, face = [numpy.asarray(a, dtype=numpy.int64).reshape(-1) for a in (ix, iy, face)]
	out = numpy.empty((8, len(ix)), dtype=numpy.int64)
	for m in range(8):
		x, y = ix + _DX[m], iy + _DY[m]
		step = numpy.full(len(ix), 4)
		step = step + numpy.where(x < 0, -1, numpy.where(x >= nside, 1, 0)) + numpy.where(y < 0, -3, numpy.where(y >= nside, 3, 0))
		x, y = numpy.mod(x, nside), numpy.mod(y, nside)
		f = _FACE_NEXT[step, face]
		twist = _FACE_TWIST[step, face >> 2]
		x = numpy.where(twist & 1, nside - x - 1, x)
		y = numpy.where(twist & 2, nside - y - 1, y)
		x, y = numpy.where(twist & 4, y, x), numpy.where(twist & 4, x, y)
		fsafe = numpy.maximum(f, 0)
		pix = xyf2nest(nside, x, y, fsafe) if nest else xyf2ring(nside, x, y, fsafe)
		out[m] = numpy.where(f >= 0, pix, -1)
	return out


def get_all_neighbours(nside, theta, phi=None, nest=False):
	"""the 8 neighbours (SW, W, NW, N, NE, E, SE, S; -1 where there is none) of a pixel given
	by number (phi None) or of the pixel containing (theta, phi); shape (8,) or (8, n)"""
	if phi is None:
		pix = numpy.asarray(theta, dtype=numpy.int64)
		shape = pix.shape
		if nest:
			ix, iy, face = nest2xyf(nside, pix.reshape(-1))
		else:
			ix, iy, face = ring2xyf(nside, pix.reshape(-1))
	else:
		theta, phi = numpy.broadcast_arrays(numpy.asarray(theta, dtype=float), numpy.asarray(phi, dtype=float))
		shape = theta.shape
		ix, iy, face = _ang2xyf(nside, theta.reshape(-1), phi.reshape(-1))
	return _neighbours_xyf(nside, ix, iy, face, nest).reshape((8,) + shape)


def ring2xyf(nside, pix):
	"""inverse of xyf2ring by table (small nside only; used for the documentation examples)"""
	nside = int(nside)
	npix = nside2npix(nside)
	allnest = numpy.arange(npix)
	ix, iy, face = nest2xyf(nside, allnest)
	ring = xyf2ring(nside, ix, iy, face)
	inv = numpy.empty(npix, dtype=numpy.int64)
	inv[ring] = allnest
	return nest2xyf(nside, inv[numpy.asarray(pix, dtype=numpy.int64)])


def ring2nest(nside, pix):
	return xyf2nest(nside, *ring2xyf(nside, pix))
