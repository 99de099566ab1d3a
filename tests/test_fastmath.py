"""The row kernels' elementary functions on the GPU (csrc/fastmath.inc through nwayhip_fastmath_probe): bit for bit what the same
source gives on the host (tests/test_fastmath_host.py checks that against 50-digit arithmetic), the long roads being the device
library's; and within 2 ulp of numpy, the reference's arithmetic (fastskymatch.py:26-47, bayesdistance.py:18-86)."""
import numpy as np
import pytest
import torch

from fastmath_util import arguments, host_eval

pytestmark = pytest.mark.gpu


def device_eval(fn, x, y=None):
	from nway_amd import _hip
	dev = torch.device('cuda', 0)
	tx = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64), device=dev)
	ty = torch.as_tensor(np.ascontiguousarray(y, dtype=np.float64), device=dev) if y is not None else None
	out = torch.empty_like(tx)
	out2 = torch.empty_like(tx) if fn == 0 else None
	_hip.check(_hip.load().nwayhip_fastmath_probe(fn, _hip.ptr(tx), _hip.ptr(ty) if ty is not None else None, len(x), _hip.ptr(out),
		_hip.ptr(out2) if out2 is not None else None, _hip.current_stream_ptr(dev)))
	torch.cuda.synchronize()
	return (out.cpu().numpy(), out2.cpu().numpy()) if fn == 0 else out.cpu().numpy()


def same_bits(a, b):
	a = np.asarray(a)
	b = np.asarray(b)
	nan = np.isnan(a) & np.isnan(b)
	return ((a.view(np.uint64) == b.view(np.uint64)) | nan)


@pytest.mark.parametrize('fn', [0, 1, 2, 3, 4, 5, 6, 7])
def test_device_equals_host_build(fn):
	x, y = arguments(200000, seed=11 + fn)[fn]
	got = device_eval(fn, x, y)
	want = host_eval(fn, x, y)
	if fn == 0:
		assert same_bits(got[0], want[0]).all() and same_bits(got[1], want[1]).all()
		return
	# the short roads: bit for bit.  The long roads of atan2 and hypot are the two libraries' own (device library / glibc): an ulp
	if fn == 1:
		short = (x >= 0) & (x * 16 <= y) & (y > 0) & (y <= 1e300)
	elif fn == 2:
		s = x * x + y * y
		short = ((s >= 1e-279) & (s <= 1e279)) | ((x == 0) & (y == 0))
	else:
		short = np.ones(len(x), dtype=bool)
	bad = ~same_bits(got[short], want[short])
	assert not bad.any(), (x[short][bad][:5], got[short][bad][:5], want[short][bad][:5])
	rest = ~short
	with np.errstate(all='ignore'):
		ok = same_bits(got[rest], want[rest]) | (np.abs(got[rest] - want[rest]) <= 2 * np.spacing(np.abs(want[rest])))
	assert ok.all()


def test_within_two_ulp_of_numpy():
	args = arguments(100000, seed=5)
	with np.errstate(all='ignore'):
		x = args[0][0]
		x = x[np.abs(x) <= 16]
		s, c = device_eval(0, x)
		for got, want in ((s, np.sin(x)), (c, np.cos(x))):
			away = np.abs(np.sin(2 * x)) > 1e-6
			assert (np.abs(got - want)[away] <= 2 * np.spacing(np.abs(want[away]))).all()
			assert (np.abs(got - want) <= 2 * np.spacing(np.abs(want)) + 1e-25).all()
		for fn, ref in ((1, np.arctan2), (2, np.hypot)):
			a, b = args[fn]
			got, want = device_eval(fn, a, b), ref(a, b)
			assert (same_bits(got, want) | (np.abs(got - want) <= 2 * np.spacing(np.abs(want)))).all()
		for fn, ref in ((3, np.log), (4, np.log10), (5, lambda v: 10.0 ** v)):
			a = args[fn][0]
			got, want = device_eval(fn, a), ref(a)
			assert (same_bits(got, want) | (np.abs(got - want) <= 2 * np.spacing(np.abs(want)))).all()
