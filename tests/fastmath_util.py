"""The host build of nway_amd/csrc/fastmath.inc (tests/fastmath_host.cpp) and the argument sets the fastmath tests share."""
import ctypes
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_P = ctypes.POINTER(ctypes.c_double)
_lib = None


def host_library():
	"""g++ -O2 -ffp-contract=off of the shim (fma() is the C library's: the hardware's where there is one, exact either way): IEEE operations in the order they are written, fused where the
	source says fma -- the arithmetic of the device build."""
	global _lib
	if _lib is None:
		out = os.path.join(tempfile.mkdtemp(prefix='nway_fastmath_'), 'fastmath_host.so')
		subprocess.check_call(['g++', '-O2', '-ffp-contract=off', '-fPIC', '-shared', '-o', out, os.path.join(HERE, 'fastmath_host.cpp')])
		_lib = ctypes.CDLL(out)
	return _lib


def _ptr(a):
	return a.ctypes.data_as(_P)


def host_eval(fn, x, y=None):
	"""fn as in nwayhip_fastmath_probe; returns one array (two for sincos)."""
	lib = host_library()
	x = np.ascontiguousarray(x, dtype=np.float64)
	n = ctypes.c_long(len(x))
	out = np.empty_like(x)
	if fn == 0:
		out2 = np.empty_like(x)
		lib.fm_sincos(_ptr(x), _ptr(out), _ptr(out2), n)
		return out, out2
	if fn in (1, 2):
		y = np.ascontiguousarray(y, dtype=np.float64)
		(lib.fm_atan2 if fn == 1 else lib.fm_hypot)(_ptr(x), _ptr(y), _ptr(out), n)
		return out
	(lib.fm_log, lib.fm_log10, lib.fm_exp10, lib.fm_radians, lib.fm_degrees)[fn - 3](_ptr(x), _ptr(out), n)
	return out


def arguments(n=20000, seed=3):
	"""fn -> (x, y or None): the arguments a match has, the edges of every short road, and what lies beyond them."""
	rng = np.random.RandomState(seed)
	special = np.array([0.0, -0.0, np.nan, np.inf, -np.inf, 1e-320, -1e-320, 1.0, -1.0])
	a = {}
	a[0] = (np.concatenate([rng.uniform(-np.pi / 2, np.pi / 2, n), rng.uniform(-1e-4, 1e-4, n), rng.uniform(-2 * np.pi, 2 * np.pi, n),
		rng.uniform(-16, 16, n), rng.uniform(-1.6e6, 1.6e6, n), [np.pi / 2, -np.pi / 2, np.pi / 4, 1.6e6, -1.6e6, 1.7e6, 1e300], special]), None)
	ys = np.concatenate([10 ** rng.uniform(-12, -1.3, n), rng.uniform(0, 1 / 16, n), rng.uniform(0, 2, n), rng.uniform(-1, 1, n), special, special])
	xs = np.concatenate([rng.uniform(0.9, 1.0, n), np.ones(n), rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), special, special[::-1]])
	a[1] = (ys, xs)
	hs = np.concatenate([10 ** rng.uniform(-12, -1, n) * rng.choice([-1, 1], n), 10 ** rng.uniform(-200, 200, n), [0.0, 0.0, 3.0, 1e-170, 1e170], special])
	ht = np.concatenate([10 ** rng.uniform(-12, -1, n) * rng.choice([-1, 1], n), 10 ** rng.uniform(-200, 200, n), [0.0, 1e-9, 4.0, 1e-170, 1e170], special[::-1]])
	a[2] = (hs, ht)
	lg = np.concatenate([10 ** rng.uniform(-300, 300, n), rng.uniform(0.5, 2, n), 10 ** rng.uniform(-3, 6, n), [1.0, 2.0, 10.0, 0.1, 1e-5, 5e-324, 1.7976931348623157e308], special])
	a[3] = (lg, None)
	a[4] = (lg, None)
	a[5] = (np.concatenate([rng.uniform(-30, 5, n), rng.uniform(-1, 1, n), rng.uniform(-330, 310, n), [0.0, 1.0, -1.0, 2.0, 308.0, -323.0, 400.0, -400.0, 1e300, -1e300], special]), None)
	a[6] = (np.concatenate([rng.uniform(-360, 720, n), rng.uniform(-90, 90, n), 10 ** rng.uniform(-30, 30, n) * rng.choice([-1, 1], n), [0.0, -0.0, 180.0, 90.0, 360.0, -999.0, np.nan]]), None)
	a[7] = (np.concatenate([rng.uniform(0, np.pi, n), 10 ** rng.uniform(-12, 0.5, n), [0.0, -0.0, np.pi, np.nan]]), None)
	return a
