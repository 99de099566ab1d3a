"""The sparse front beyond sparse fields (GPU): dense 2-way fields (tens of links per primary:
link slots sized by the Poisson tail, candidate-parallel tail), direct-mapped tables too large
for the LDS of a sweep workgroup (bitmap in L2, folded copy in LDS), k >= 3 with the sparse
front feeding the general back end, many primaries in one cell.  Each against the C
restatement of the oracle and against the general path of the library itself."""
import os
import sys

import numpy as np
import pytest

from goldenutil import ROOT

sys.path.insert(0, os.path.join(ROOT, 'oracle'))
sys.path.insert(0, ROOT)
from test_full_size import hip_table, compare, check_properties  # noqa: E402

pytestmark = pytest.mark.gpu


def patch_tables(rng, sizes, half, errors, centre=(150.0, 2.0), frac=0.8):
	"""catalogues uniform in a square patch; a fraction of the primaries has a counterpart in every secondary"""
	tabs = []
	for c, n in enumerate(sizes):
		ra = rng.uniform(centre[0] - half, centre[0] + half, size=n)
		dec = rng.uniform(centre[1] - half, centre[1] + half, size=n)
		err = errors[c] * np.ones(n) if np.isscalar(errors[c]) else errors[c]
		if c > 0:
			m = int(frac * sizes[0])
			slots = rng.choice(n, size=min(m, n), replace=False)
			m = len(slots)
			psig = tabs[0]['error'][:m]
			dec[slots] = tabs[0]['dec'][:m] + rng.normal(0, 1, size=m) * psig / 3600.
			ra[slots] = tabs[0]['ra'][:m] + rng.normal(0, 1, size=m) * psig / 3600. / np.cos(np.radians(tabs[0]['dec'][:m]))
		tabs.append(dict(name='T%d' % c, ra=ra, dec=dec, error=err, area=(2 * half)**2, mags=[], maghists=[], magnames=[]))
	return tabs


def both_paths(nw, tabs, radius, completeness=0.9, **options):
	import nway_oracle_c as orc_c
	names = [t['name'] for t in tabs]
	t, status = hip_table(nw, tabs, radius, completeness, **options)
	assert int(status[1]) == 0
	check_properties(t, names, len(tabs[0]['ra']))
	o = orc_c.nway_match(tabs, radius, completeness, correction='cli' if options.get('correction') else 'api')
	compare(t, o, names)
	general = dict(options, link_slots=-1)
	general.pop('tuning', None)
	g, _ = hip_table(nw, tabs, radius, completeness, **general)
	assert g['_path'] == 0
	# the same table bit for bit: the dense tails sum a group of up to 64 rows in row order and a larger one by a wave,
	# term for term as the general path's group kernel does (rows.inc: group_sub / group_wave; taild.inc)
	for key in t:
		if not key.startswith('_'):
			np.testing.assert_array_equal(t[key], g[key], err_msg=key)
	return t


def test_dense_two_way_field_stays_on_the_sparse_front():
	import nway_amd as nw
	from nway_amd import _hip
	rng = np.random.default_rng(11)
	tabs = patch_tables(rng, [3000, 300000], 0.21, [rng.uniform(0.3, 1.5, size=3000), 0.1])  # ~10 chance neighbours per primary
	t = both_paths(nw, tabs, 5.0)
	assert t['_path'] == _hip.PATH_SPARSE and t['_link_slots'] > 8
	assert len(t['ncat']) > 10 * 3000


@pytest.mark.parametrize('fold', [19, 20])
def test_dense_two_way_field_with_a_table_beyond_the_lds(fold):
	import nway_amd as nw
	from nway_amd import _hip
	rng = np.random.default_rng(12)
	tabs = patch_tables(rng, [3001, 200001], 0.21, [rng.uniform(0.3, 1.5, size=3001), 0.1], centre=(0.1, -0.05))  # cells of both signs, odd sizes
	t = both_paths(nw, tabs, 5.0, tuning=dict(direct_log2=21, fold_log2=fold))
	assert t['_path'] == _hip.PATH_SPARSE and t['_link_slots'] > 8
	assert t['_desc']['sweep'] == 'big' and t['_desc']['fold_log2'] == fold and t['_desc']['tail'] == 'dense2'


@pytest.mark.parametrize('fold', [19, 20])
def test_all_sky_field_with_a_table_beyond_the_lds(fold):
	import bench
	import nway_amd as nw
	from nway_amd import _hip
	prim, sec = bench.make_workload(20000, 1500001, 5)
	sec = dict(sec, error=0.1 * np.ones(len(sec['ra'])))
	t = both_paths(nw, [prim, sec], 20.0, tuning=dict(direct_log2=22, fold_log2=fold))
	assert t['_path'] == _hip.PATH_SPARSE
	assert t['_desc']['sweep'] == 'big' and t['_desc']['direct_log2'] == 22 and t['_desc']['tail'] == 'sparse2'


def test_dense_three_way_field_tuple_parallel_tail():
	import nway_amd as nw
	from nway_amd import _hip
	rng = np.random.default_rng(13)
	tabs = patch_tables(rng, [3000, 30000, 40000], 0.21, [1.0, 0.1, 0.5])  # ~4 and ~5 chance neighbours per primary
	t = both_paths(nw, tabs, 10.0)
	assert t['_path'] == _hip.PATH_SPARSE and t['_link_slots'] > 8
	assert len(t['ncat']) > 15 * 3000


def test_dense_three_way_field_with_a_crowd(monkeypatch):
	"""a workgroup of the tuple-parallel tail with more rows than its LDS arrays hold"""
	import nway_amd as nw
	from nway_amd import _hip
	rng = np.random.default_rng(16)
	tabs = patch_tables(rng, [600, 20000, 20000], 0.21, [1.0, 0.1, 0.5])
	for c in (1, 2):  # 14 more secondaries of both catalogues around each of the first 40 primaries
		for i in range(40):
			at = 1000 + 14 * i
			tabs[c]['ra'][at:at + 14] = tabs[0]['ra'][i] + rng.normal(0, 2, size=14) / 3600.
			tabs[c]['dec'][at:at + 14] = tabs[0]['dec'][i] + rng.normal(0, 2, size=14) / 3600.
	t = both_paths(nw, tabs, 10.0, link_slots=31)
	assert t['_path'] == _hip.PATH_SPARSE


def test_dense_four_way_field_sparse_front_general_back_end():
	import nway_amd as nw
	from nway_amd import _hip
	rng = np.random.default_rng(17)
	tabs = patch_tables(rng, [2000, 20000, 25000, 15000], 0.21, [1.0, 0.1, 0.5, 0.3])  # 2 .. 3 chance neighbours per primary and catalogue
	t = both_paths(nw, tabs, 10.0)
	assert t['_path'] == _hip.PATH_HYBRID


def test_three_way_with_the_scripts_correction_in_the_fused_tails():
	import bench
	import nway_amd as nw
	from nway_amd import _hip
	rng = np.random.default_rng(14)
	prim, a = bench.make_workload(5000, 200000, 7)
	_, b = bench.make_workload(5000, 150000, 8)
	m = 3000
	b['ra'][:m] = prim['ra'][:m]
	b['dec'][:m] = np.clip(prim['dec'][:m] + rng.normal(0, 0.3, size=m) / 3600., -90, 90)
	tabs = [prim, dict(a, name='A', error=0.1 * np.ones(len(a['ra']))), dict(b, name='B', error=0.5 * np.ones(len(b['ra'])))]
	t = both_paths(nw, tabs, 10.0, correction=_hip.CORRECTION_CLI)
	assert t['_path'] == _hip.PATH_SPARSE and t['_link_slots'] == 8   # k_tailk<3> applies the correction itself
	# a dense field: the tuple-parallel tail does
	rng = np.random.default_rng(19)
	dense = patch_tables(rng, [3000, 30000, 40000], 0.21, [1.0, 0.1, 0.5])
	t = both_paths(nw, dense, 10.0, correction=_hip.CORRECTION_CLI)
	assert t['_path'] == _hip.PATH_SPARSE and t['_link_slots'] > 8
	# ... also where a workgroup has more rows than its LDS arrays
	crowd = patch_tables(rng, [600, 20000, 20000], 0.21, [1.0, 0.1, 0.5])
	for c in (1, 2):
		for i in range(40):
			at = 1000 + 14 * i
			crowd[c]['ra'][at:at + 14] = crowd[0]['ra'][i] + rng.normal(0, 2, size=14) / 3600.
			crowd[c]['dec'][at:at + 14] = crowd[0]['dec'][i] + rng.normal(0, 2, size=14) / 3600.
	t = both_paths(nw, crowd, 10.0, correction=_hip.CORRECTION_CLI, link_slots=31)
	assert t['_path'] == _hip.PATH_SPARSE
	# ... and groups of more than 64 rows whose workgroup's rows DO fit (the best offer of such a primary is found by a wave,
	# taild3.inc): a few primaries with ten sources each in both catalogues, 121 rows
	few = patch_tables(rng, [3000, 30000, 40000], 0.21, [1.0, 0.1, 0.5])
	for c in (1, 2):
		for i in range(5):
			at = 2000 + 10 * i
			few[c]['ra'][at:at + 10] = few[0]['ra'][100 * i] + rng.normal(0, 2, size=10) / 3600.
			few[c]['dec'][at:at + 10] = few[0]['dec'][100 * i] + rng.normal(0, 2, size=10) / 3600.
	t = both_paths(nw, few, 10.0, correction=_hip.CORRECTION_CLI)
	assert t['_path'] == _hip.PATH_SPARSE and t['_link_slots'] > 8
	groups = np.bincount(t['T0'].astype(np.int64))
	assert groups.max() > 64


def test_four_way_with_the_scripts_correction_fused_and_on_the_general_back_end(monkeypatch):
	import bench
	import nway_amd as nw
	from nway_amd import _hip
	rng = np.random.default_rng(20)
	prim, a = bench.make_workload(4000, 150000, 9)
	tabs = [prim, dict(a, name='A', error=0.1 * np.ones(len(a['ra'])))]
	for name, n, seed, sig in (('B', 120000, 10, 0.5), ('C', 100000, 11, 0.3)):
		_, b = bench.make_workload(4000, n, seed)
		m = 2500
		b['ra'][:m] = prim['ra'][:m]
		b['dec'][:m] = np.clip(prim['dec'][:m] + rng.normal(0, 0.3, size=m) / 3600., -90, 90)
		tabs.append(dict(b, name=name, error=sig * np.ones(n)))
	# (round 4: k_tailk<4, true> -- 292 VGPRs and 344 bytes of scratch -- is no longer compiled: four catalogues with the script's
	# correction take k_correct behind the general back end; without the correction the one-lane walk k_tailk<4, false>)
	t = both_paths(nw, tabs, 10.0, correction=_hip.CORRECTION_CLI)
	assert t['_path'] == _hip.PATH_HYBRID
	t = both_paths(nw, tabs, 10.0)
	assert t['_path'] == _hip.PATH_SPARSE and t['_desc']['tail'] == 'sparsek'
	t = both_paths(nw, tabs, 10.0, correction=_hip.CORRECTION_CLI, tuning=dict(disable=_hip.DISABLE_FUSED_CORRECTION))
	assert t['_path'] == _hip.PATH_HYBRID   # the same with k_correct behind the general back end
	# a dense 4-way field with the correction: hybrid
	dense = patch_tables(rng, [1500, 15000, 20000, 12000], 0.21, [1.0, 0.1, 0.5, 0.3])
	t = both_paths(nw, dense, 10.0, correction=_hip.CORRECTION_CLI)
	assert t['_path'] == _hip.PATH_HYBRID


def test_many_primaries_in_one_cell():
	"""more registrations in one cell than a group of the table holds and than one routing pass
	takes: displaced claims, walks, the routing's repeat"""
	import nway_amd as nw
	from nway_amd import _hip
	rng = np.random.default_rng(15)
	tabs = patch_tables(rng, [2000, 100000], 0.21, [0.5, 0.1])
	# 40 primaries within one arcsec of each other, twice; a handful of secondaries among them
	for at, (ra0, dec0) in ((0, (150.03, 2.01)), (40, (149.9, 1.95))):
		tabs[0]['ra'][at:at + 40] = ra0 + rng.uniform(0, 1, size=40) / 3600.
		tabs[0]['dec'][at:at + 40] = dec0 + rng.uniform(0, 1, size=40) / 3600.
		tabs[1]['ra'][at:at + 7] = ra0 + rng.uniform(-2, 3, size=7) / 3600.
		tabs[1]['dec'][at:at + 7] = dec0 + rng.uniform(-2, 3, size=7) / 3600.
	t = both_paths(nw, tabs, 5.0)
	assert t['_path'] == _hip.PATH_SPARSE
	both_paths(nw, tabs, 5.0, link_slots=48)


@pytest.mark.parametrize('k', [2, 3])
@pytest.mark.parametrize('dense', [False, True])
def test_a_row_capacity_that_is_too_small_is_settled_by_one_repeat(monkeypatch, k, dense):
	"""the fused tails count every row they cannot write: the run that overflowed reports the exact
	need, the second one holds (no doubling loop)"""
	import nway_amd as nw
	from nway_amd import _hip
	rng = np.random.default_rng(18)
	sizes = [2000, 200000, 150000][:k] if dense and k == 2 else ([2000, 30000, 25000][:k] if dense else [4000, 50000, 40000][:k])
	tabs = patch_tables(rng, sizes, 0.21 if dense else 3.0, [1.0, 0.1, 0.5][:k])
	radius = 5.0 if k == 2 else 10.0
	want, _ = hip_table(nw, tabs, radius, 0.9)
	assert want['_path'] == _hip.PATH_SPARSE and (want['_link_slots'] > 8) == dense
	roomy = nw._estimate_capacities
	monkeypatch.setattr(nw, '_estimate_capacities', lambda *a, **kw: (roomy(*a, **kw)[0], len(want['ncat']) // 3))
	res = nw.run_match(tabs, radius, 0.9, logger=nw.NullOutputLogger())
	monkeypatch.setattr(nw, '_estimate_capacities', roomy)
	assert res.plan.attempts == 2 and res.plan.path == _hip.PATH_SPARSE
	assert res.nrows == len(want['ncat'])
	np.testing.assert_array_equal(res.to_host('idx', k - 1).astype(np.int64), want[tabs[k - 1]['name']])
	np.testing.assert_array_equal(res.to_host('match_flag').astype(np.int64), want['match_flag'])
	np.testing.assert_array_equal(res.to_host('p_i'), want['prob_this_match'])
	res.plan.close()


@pytest.mark.parametrize('k', [2, 3])
def test_a_cluster_with_more_candidates_than_slots_keeps_the_sparse_front(k):
	"""slots are sized for the mean density; a few primaries in a cluster have more candidates:
	the run that overflowed reports how many, the next one has that many slots (no general path)"""
	import nway_amd as nw
	from nway_amd import _hip
	rng = np.random.default_rng(21)
	tabs = patch_tables(rng, [3000, 40000, 30000][:k], 3.0, [1.0, 0.1, 0.5][:k])   # ~0.01 chance neighbours per primary
	for c in range(1, k):
		for i in range(30):  # a dozen secondaries within a few arcsec of each of the first 30 primaries
			at = 5000 + 12 * i
			tabs[c]['ra'][at:at + 12] = tabs[0]['ra'][i] + rng.normal(0, 1.5, size=12) / 3600.
			tabs[c]['dec'][at:at + 12] = tabs[0]['dec'][i] + rng.normal(0, 1.5, size=12) / 3600.
	res = nw.run_match(tabs, 8.0, 0.9, logger=nw.NullOutputLogger())
	assert res.plan.path == _hip.PATH_SPARSE and res.plan.attempts == 2 and 8 < res.plan.link_slots <= 20
	res.plan.close()
	both_paths(nw, tabs, 8.0)


@pytest.mark.parametrize('k', [2, 3])
def test_fields_with_more_than_64_links_per_primary(k):
	"""~40 chance neighbours per primary: up to 128 slots the sparse front keeps such a field (dense 2-way tail;
	k = 3: the general back end fed from the slots), also with the slots forced to the cap"""
	import nway_amd as nw
	from nway_amd import _hip
	rng = np.random.default_rng(41 + k)
	tabs = patch_tables(rng, [1500] + [66000] * (k - 1), 0.05, [rng.uniform(0.3, 1.5, size=1500)] + [0.1] * (k - 1))
	t = both_paths(nw, tabs, 5.0)
	assert 64 < t['_link_slots'] <= 128 and t['_desc']['tail'] == ('dense2' if k == 2 else 'hybrid')
	assert len(t['ncat']) > 40 * 1500
	both_paths(nw, tabs, 5.0, link_slots=128)


def test_dense_three_way_field_with_tens_of_links_per_catalogue():
	"""the tuple-parallel 3-way tail beyond 31 slots (64-bit validity words): ~20 and ~12 chance neighbours per primary,
	hundreds of tuples each -- the shape of BASELINE configs[1]"""
	import nway_amd as nw
	rng = np.random.default_rng(47)
	tabs = patch_tables(rng, [400, 33000, 20000], 0.05, [rng.uniform(0.5, 1.5, size=400), 0.1, 0.5])
	t = both_paths(nw, tabs, 5.0)
	assert 31 < t['_link_slots'] <= 63 and t['_desc']['tail'] == 'dense3'
	assert len(t['ncat']) > 150 * 400
	t = both_paths(nw, tabs, 5.0, link_slots=63, correction=1)
	assert t['_desc']['tail'] == 'dense3'


@pytest.mark.parametrize('k', [5, 6])
def test_many_catalogues_with_two_links_in_several_of_them(k):
	"""five and six catalogues with two links in several of them for a few hundred primaries -- the input on which the one-lane
	walk k_tailk<K >= 5> (313-512 VGPRs, scratch) carried the first link's separation from the second through rounds 1-3 (found by
	tools/dev/soak_mid.py with SOAK_KMAX=6; the tiny many-catalogue cases of test_hip_fuzz.py never showed it).  Since round 4 the
	walk is compiled for three and four catalogues only: more take the sparse front with the general back end"""
	import nway_amd as nw
	from nway_amd import _hip
	rng = np.random.default_rng(40 + k)
	sizes = [6000] + [2500] * (k - 1)
	tabs = patch_tables(rng, sizes, 1.0, [1.0] + [0.3] * (k - 1), frac=0.35)
	for c in (2, 4, k - 1):  # a second candidate around each of 300 primaries that already have one there
		at = sizes[c] - 1
		for i in range(0, 300):
			tabs[c]['ra'][at] = tabs[0]['ra'][i] + rng.normal(0, 1.5) / 3600.
			tabs[c]['dec'][at] = tabs[0]['dec'][i] + rng.normal(0, 1.5) / 3600.
			at -= 1
	import nway_oracle_c as orc_c
	names = [x['name'] for x in tabs]
	t, status = hip_table(nw, tabs, 5.0, 0.9)
	assert int(status[1]) == 0 and t['_desc']['tail'] == 'hybrid' and t['_path'] == _hip.PATH_HYBRID and t['_desc']['link_slots'] == 8
	compare(t, orc_c.nway_match(tabs, 5.0, 0.9, correction='api'), names)
	g, _ = hip_table(nw, tabs, 5.0, 0.9, link_slots=-1)
	assert g['_path'] == 0
	for key in t:  # (the general path sums a group of more than 64 rows by a wave, the walk row by row: equal to rounding)
		if not key.startswith('_'):
			np.testing.assert_allclose(t[key], g[key], rtol=1e-12, atol=1e-15, equal_nan=True, err_msg=key)
	groups = np.bincount(t['T0'].astype(np.int64), minlength=sizes[0])
	assert (groups >= 18).sum() > 50
