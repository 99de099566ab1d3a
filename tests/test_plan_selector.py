"""Which kernels a plan runs, on both sides of every threshold of the selector (plan.inc:
nwayhip_plan_create): expected chance neighbours per primary (0.5; the 64-slot cap), the size of
the direct-mapped table (LDS bitmap up to 2^20 positions, folded beyond; the 5e6-registration
limit), the number of catalogues, the correction flavour, the radius filter and the caller's
overrides.  Plan creation needs no GPU: the selector is host code behind the C ABI
(nwayhip_plan_describe)."""
import ctypes
import math

import numpy as np
import pytest

SKY = 4 * math.pi * (180 / math.pi)**2


def poisson_slots(lam, n0):
	"""plan.inc: the smallest s with n0 x P(Poisson(lam) > s) < 1e-3, + 2"""
	p = math.exp(-lam)
	cdf, s = p, 0
	while (1 - cdf) * n0 > 1e-3 and s < 4096:
		s += 1
		p *= lam / s
		cdf += p
	return s + 2


def describe(n, lam, radius=10.0, scheme=None, correction=0, radius_filter=True, link_slots=0, tuning=None):
	"""n: catalogue sizes; lam: expected chance neighbours per primary of every secondary catalogue"""
	from nway_amd import _hip
	lib = _hip.load()
	k = len(n)
	r_deg = radius / 3600.
	per_deg2 = lam / (math.pi * r_deg**2)
	dens = [n[0] / 2.0 * SKY] + [per_deg2 * SKY] * (k - 1)
	scheme = _hip.SCHEME_SPHERE if scheme is None else scheme
	prm = _hip.make_params(k, scheme, radius, r_deg, dens, dens, [1.0] * (1 << (k - 1)), radius_filter=radius_filter, correction=correction,
		link_slots=link_slots, tuning=tuning)
	handle = ctypes.c_void_p(0)
	_hip.check(lib.nwayhip_plan_create(ctypes.byref(handle), ctypes.byref(prm), (ctypes.c_int64 * k)(*n), 1 << 20, 1 << 20))
	try:
		out = (ctypes.c_int32 * _hip.DESC_WORDS)()
		_hip.check(lib.nwayhip_plan_describe(handle, out))
		d = dict(zip(_hip.DESC_NAMES, [int(x) for x in out]))
		d['sweep'] = _hip.SWEEP_NAMES[d['sweep']]
		d['tail'] = _hip.TAIL_NAMES[d['tail']]
		d['split_capable'] = int(lib.nwayhip_plan_split_capable(handle))
		assert d['path'] == int(lib.nwayhip_plan_path(handle)) and d['link_slots'] == int(lib.nwayhip_plan_link_slots(handle))
		return d
	finally:
		lib.nwayhip_plan_destroy(handle)


def test_half_a_chance_neighbour_per_primary():
	"""lambda < 0.5: 8 slots and the one-launch tails; above: Poisson-tail slots and the candidate-parallel tails"""
	a = describe([100000, 10000000], 0.499)
	assert (a['path'], a['link_slots'], a['sweep'], a['tail'], a['split_capable']) == (1, 8, 'lds', 'sparse2', 1)
	b = describe([100000, 10000000], 0.501)
	assert b['link_slots'] == poisson_slots(0.501, 100000) == 10 and b['tail'] == 'dense2' and b['path'] == 1
	a3 = describe([100000, 1000000, 1000000], 0.499)
	assert (a3['tail'], a3['link_slots'], a3['one_sweep'], a3['split_capable']) == ('sparsek', 8, 1, 1)
	b3 = describe([100000, 1000000, 1000000], 0.501)
	assert (b3['tail'], b3['path'], b3['split_capable']) == ('dense3', 1, 0)
	b4 = describe([100000, 1000000, 1000000, 1000000], 0.501)
	assert (b4['tail'], b4['path'], b4['split_capable']) == ('hybrid', 2, 0)


def test_the_slot_cap():
	"""the Poisson quantile over all primaries (one overflowing primary per thousand runs, + 2) up to 128 slots keeps
	the sparse front; one slot more is the general path.  BASELINE configs[0] (1 797 primaries, 27 chance neighbours
	within 20 arcsec) is inside"""
	n0 = 100000
	lams = np.arange(40.0, 90.0, 0.05)
	edge = [l for l in lams if poisson_slots(l, n0) <= 128][-1]
	inside = describe([n0, 10000000], edge, scheme=0)
	assert inside['link_slots'] == poisson_slots(edge, n0) and 126 <= inside['link_slots'] <= 128 and inside['tail'] == 'dense2'
	outside = describe([n0, 10000000], edge + 0.5, scheme=0)
	assert poisson_slots(edge + 0.5, n0) > 128
	assert (outside['path'], outside['link_slots'], outside['sweep'], outside['tail']) == (0, 0, 'general', 'general')
	cosmos = describe([1797, 560536], 27.2, radius=20.0, scheme=0)
	assert (cosmos['path'], cosmos['link_slots'], cosmos['tail']) == (1, poisson_slots(27.2, 1797), 'dense2') and cosmos['link_slots'] == 58
	# fewer primaries need fewer slots at the same density
	assert describe([1000, 10000000], 10.0, scheme=0)['link_slots'] < describe([1000000, 10000000], 10.0, scheme=0)['link_slots']


def test_dense_three_way_tail_up_to_63_slots():
	"""k = 3: the tuple-parallel tail keeps a 64-bit word of validity bits per (primary, link of the first secondary
	catalogue): 63 slots; beyond that the general back end fed from the slots.  BASELINE configs[1] is inside"""
	from nway_amd import _hip
	n = [50000, 500000, 500000]
	lam63 = [l for l in np.arange(20.0, 40.0, 0.01) if poisson_slots(l, n[0]) == 63][-1]
	assert describe(n, lam63, scheme=0)['tail'] == 'dense3'
	assert describe(n, lam63 + 0.5, scheme=0)['tail'] == 'hybrid'
	cosmos = describe([1797, 560536, 345512], 27.2, radius=20.0, scheme=0)
	assert (cosmos['link_slots'], cosmos['tail'], cosmos['path']) == (58, 'dense3', 1)
	assert describe(n, 3.0, scheme=0, tuning=dict(disable=_hip.DISABLE_DENSE3))['tail'] == 'hybrid'
	assert describe(n, 3.0, scheme=0, tuning=dict(disable=_hip.DISABLE_DENSE3 | _hip.DISABLE_HYBRID))['tail'] == 'general'


def test_table_size_decides_the_sweep():
	"""the occupancy bitmap goes to LDS while the expected registrations (1.5 per primary on the sphere, 4 in
	flat cells) stay below 0.3 x 2^20; beyond that the large-table sweep, up to 0.3 x 2^24 registrations"""
	lds_limit = int(0.3 * (1 << 20) / 1.5)
	a = describe([lds_limit, 10000000], 0.1)
	assert (a['sweep'], a['direct_log2'], a['fold_log2']) == ('lds', 20, 0)
	b = describe([lds_limit + 2, 10000000], 0.1)
	assert b['sweep'] == 'big' and b['direct_log2'] == 22 and b['fold_log2'] == 20
	flat_limit = int(0.3 * (1 << 20) / 4)
	assert describe([flat_limit, 10000000], 0.1, scheme=0)['sweep'] == 'lds'
	assert describe([flat_limit + 1, 10000000], 0.1, scheme=0)['sweep'] == 'big'
	big_limit = int(0.3 * (1 << 24) / 1.5)
	assert describe([big_limit, 10000000], 0.1)['sweep'] == 'big'
	over = describe([big_limit + 2, 10000000], 0.1)
	assert (over['path'], over['sweep']) == (0, 'general')
	# small catalogues: at least 8 positions per expected registration, never fewer than 2^12
	assert describe([300, 20000], 0.1)['direct_log2'] == 12
	assert describe([3000, 20000], 0.1)['direct_log2'] == 16  # 4 500 registrations x 8 = 36 000 <= 2^16
	# a dense field stages the survivors' coordinates next to a 2^19-bit fold
	assert describe([400000, 10000000], 10.0, scheme=0)['fold_log2'] == 19


def test_correction_radius_filter_and_overrides():
	from nway_amd import _hip
	n3 = [100000, 1000000, 1000000]
	assert describe(n3, 0.1, correction=_hip.CORRECTION_CLI)['tail'] == 'sparsek'  # the fused tail applies the script's correction itself
	assert describe(n3, 0.1, correction=_hip.CORRECTION_CLI, tuning=dict(disable=_hip.DISABLE_FUSED_CORRECTION))['tail'] == 'hybrid'
	assert describe(n3, 3.0, scheme=0, correction=_hip.CORRECTION_CLI)['tail'] == 'dense3'
	assert describe(n3, 0.1, tuning=dict(disable=_hip.DISABLE_ONE_SWEEP))['one_sweep'] == 0
	# the raw crossproduct (no radius filter) bounds nothing: general path (flat cells only)
	assert describe([1000, 100000], 0.1, scheme=0, radius_filter=False)['path'] == 0
	# the caller's choices win
	assert describe([100000, 10000000], 0.1, link_slots=-1)['path'] == 0
	forced = describe([100000, 10000000], 0.1, link_slots=24)
	assert (forced['link_slots'], forced['tail']) == (24, 'dense2')
	assert describe([100000, 10000000], 0.1, link_slots=200)['link_slots'] == 128
	t = describe([100000, 10000000], 0.1, tuning=dict(direct_log2=21, fold_log2=19))
	assert (t['sweep'], t['direct_log2'], t['fold_log2']) == ('big', 21, 19)


def test_environment_is_ignored_without_the_development_guard(monkeypatch):
	"""a stray NWAYHIP_* variable in a user's environment changes nothing (plan.inc: env_int reads it only under NWAYHIP_DEV=1)"""
	monkeypatch.delenv('NWAYHIP_DEV', raising=False)
	monkeypatch.setenv('NWAYHIP_LINK_SLOTS', '40')
	monkeypatch.setenv('NWAYHIP_DIRECT_LOG2', '22')
	d = describe([100000, 10000000], 0.1)
	assert (d['link_slots'], d['direct_log2'], d['sweep']) == (8, 20, 'lds')


def quad3_deep(n0, lams):
	"""plan.inc: expected primaries of a run with a counterpart and two or more chance candidates in a catalogue"""
	return sum(n0 * (1 - math.exp(-1.25 * lam) * (1 + 1.25 * lam)) for lam in lams)


def test_four_lanes_per_primary_of_a_sparse_three_way_field():
	"""k = 3: k_tail3q where a RUN is unlikely to meet a primary with three candidates in a catalogue (such a primary sends
	the whole run back to k_tailk): expected number of them below 0.1 -- a bound on chance neighbours per primary that
	tightens with the number of primaries; the caller's switches either way; never for other k"""
	from nway_amd import _hip
	n = [100000, 1000000, 1000000]
	lam_edge = math.sqrt(0.1 / (2 * 100000 * 1.25**2 / 2))   # n0 * 2 catalogues * mu^2 / 2 = 0.1
	assert quad3_deep(100000, [lam_edge * 0.98] * 2) < 0.1 < quad3_deep(100000, [lam_edge * 1.02] * 2)
	below, above = describe(n, lam_edge * 0.98), describe(n, lam_edge * 1.02)
	assert (below['tail'], below['link_slots'], below['split_capable']) == ('quad3', 8, 1)
	assert (above['tail'], above['link_slots'], above['split_capable']) == ('sparsek', 8, 1)
	# BASELINE configs[3] (1e5 x 1e6 x 1e6 on the whole sky, 10 arcsec: lambda = 5.9e-4) takes it; ten times the primaries do not
	lam_c4s = 1e6 / SKY * math.pi * (10 / 3600.)**2
	assert describe(n, lam_c4s)['tail'] == 'quad3' and quad3_deep(100000, [lam_c4s] * 2) < 0.06
	assert describe([1000000, 1000000, 1000000], lam_c4s)['tail'] == 'sparsek'
	assert describe([1000, 100000, 100000], 0.005)['tail'] == 'quad3' and describe(n, 0.005)['tail'] == 'sparsek'
	assert describe(n, 0.3, tuning=dict(enable=_hip.ENABLE_QUAD3))['tail'] == 'quad3'
	assert describe(n, 0.0005, tuning=dict(disable=_hip.DISABLE_QUAD3))['tail'] == 'sparsek'
	assert describe(n, 0.0005, link_slots=1)['tail'] == 'sparsek'   # (its lanes read two slots of every primary)
	assert describe(n, 0.0005, correction=_hip.CORRECTION_CLI)['tail'] == 'quad3'
	assert describe(n[:2], 0.0005)['tail'] == 'sparse2' and describe(n + [1000000], 0.0005)['tail'] == 'sparsek'
	# the one-lane walk: three and four catalogues (four with the script's correction: the general back end); five or more: hybrid
	n4 = n + [1000000]
	assert describe(n4, 0.0005, correction=_hip.CORRECTION_CLI)['tail'] == 'hybrid'
	five = describe(n4 + [1000000], 0.0005)
	assert (five['tail'], five['path'], five['link_slots'], five['split_capable']) == ('hybrid', 2, 8, 0)
	assert describe(n4 + [1000000] * 4, 0.0005)['tail'] == 'hybrid'
