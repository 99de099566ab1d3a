"""The 3-way sparse tail with four lanes per primary (csrc/tail3q.inc; GPU): against the C
restatement of the oracle, the general path and the one-lane-per-primary tail it replaces (k_tailk<3>),
bit for bit -- on a sparse sky (every primary on the fast lanes), on fields where many primaries have two
or more candidates in a catalogue (the walk on the first lane), in flat cells, with the script's
correction, with float32 separations, and at workgroup boundaries."""
import os
import sys

import numpy as np
import pytest

from goldenutil import ROOT

sys.path.insert(0, os.path.join(ROOT, 'oracle'))
sys.path.insert(0, ROOT)
from test_full_size import hip_table  # noqa: E402
from test_dense_paths import both_paths, patch_tables  # noqa: E402

pytestmark = pytest.mark.gpu


def sky_tables(n0, n1, n2, seed, frac=(0.8, 0.6)):
	import bench
	rng = np.random.default_rng(seed)
	prim, a = bench.make_workload(n0, n1, seed)
	_, b = bench.make_workload(n0, n2, seed + 1)
	m = int(frac[1] * n0)
	b['ra'][:m] = prim['ra'][:m]
	b['dec'][:m] = np.clip(prim['dec'][:m] + rng.normal(0, 0.5, size=m) / 3600., -90, 90)
	return [prim, dict(a, name='A', error=0.1 * np.ones(n1)), dict(b, name='B', error=0.5 * np.ones(n2))]


def same_table(x, y):
	for key in x:
		if not key.startswith('_'):
			np.testing.assert_array_equal(x[key], y[key], err_msg=key)


def quad_and_lane(nw, tabs, radius, oracle=True, **options):
	"""the table through k_tail3q (checked against the oracle and the general path) and through k_tailk<3>"""
	from nway_amd import _hip
	tuning = dict(options.pop('tuning', {}))
	if oracle:
		q = both_paths(nw, tabs, radius, tuning=dict(tuning, enable=_hip.ENABLE_QUAD3), **options)
	else:
		q, _ = hip_table(nw, tabs, radius, 0.9, tuning=dict(tuning, enable=_hip.ENABLE_QUAD3), **options)
		g, _ = hip_table(nw, tabs, radius, 0.9, **dict(options, link_slots=-1))
		assert g['_path'] == 0
		same_table(q, g)
	assert q['_desc']['tail'] == 'quad3', q['_desc']
	lane, _ = hip_table(nw, tabs, radius, 0.9, tuning=dict(tuning, disable=_hip.DISABLE_QUAD3), **options)
	assert lane['_desc']['tail'] == 'sparsek', lane['_desc']
	same_table(q, lane)
	return q


@pytest.mark.parametrize('n0', [5000, 64, 65, 1])
def test_sparse_sky_every_primary_on_the_fast_lanes(n0):
	import nway_amd as nw
	tabs = sky_tables(n0, 200000, 150000, 21)
	t = quad_and_lane(nw, tabs, 10.0)
	if n0 >= 5000:
		assert (t['ncat'] == 3).sum() > 0.4 * n0 and (t['ncat'] == 1).sum() == n0


def test_default_selection_by_chance_neighbours():
	import nway_amd as nw
	tabs = sky_tables(3000, 200000, 150000, 22)
	t, _ = hip_table(nw, tabs, 10.0, 0.9)
	assert t['_desc']['tail'] == 'quad3'
	rng = np.random.default_rng(23)
	dense = patch_tables(rng, [2000, 3000, 4000], 0.21, [1.0, 0.1, 0.5])  # ~0.4 and ~0.55 chance neighbours per primary
	t, _ = hip_table(nw, dense, 10.0, 0.9)
	assert t['_desc']['tail'] == 'dense3'
	t, _ = hip_table(nw, dense, 10.0, 0.9, link_slots=8)
	assert t['_desc']['tail'] == 'sparsek'


def crowded(tabs, rng, which, cat, extra, spread=2.0):
	"""`extra` more secondaries of catalogue `cat` around each primary in `which` (overwriting unmatched secondaries at the end of the catalogue)"""
	t = tabs[cat]
	at = len(t['ra']) - 1
	for i in which:
		for _ in range(extra):
			t['ra'][at] = tabs[0]['ra'][i] + rng.normal(0, spread) / 3600. / max(np.cos(np.radians(tabs[0]['dec'][i])), 1e-6)
			t['dec'][at] = np.clip(tabs[0]['dec'][i] + rng.normal(0, spread) / 3600., -90, 90)
			at -= 1


@pytest.mark.parametrize('correction', [False, True])
def test_two_candidates_in_a_catalogue_stay_on_the_four_lanes(correction):
	import nway_amd as nw
	from nway_amd import _hip
	rng = np.random.default_rng(24)
	tabs = sky_tables(5000, 200000, 150000, 24)
	crowded(tabs, rng, range(0, 300), 1, 1)        # (2, 1) and, from 3000 on, (2, 0)
	crowded(tabs, rng, range(200, 500), 2, 1)      # (2, 2) for 200..299, (1, 2) after
	crowded(tabs, rng, range(3900, 4100), 1, 1)    # primaries with and without counterparts
	crowded(tabs, rng, range(4500, 4600), 2, 2)    # (0, 2): no counterpart in either catalogue
	crowded(tabs, rng, range(4500, 4550), 1, 2)    # (2, 2) without counterparts
	t = quad_and_lane(nw, tabs, 10.0, **(dict(correction=_hip.CORRECTION_CLI) if correction else {}))
	groups = np.bincount(t['PRIM'].astype(np.int64), minlength=5000)
	assert groups.max() == 9 and (groups > 4).sum() > 100 and (groups == 1).sum() > 0


@pytest.mark.parametrize('correction', [False, True])
def test_a_primary_catalogue_that_fills_the_chip_takes_the_wider_workgroups(correction):
	"""65 536 primaries and more: workgroups of 512 threads (tail3q.inc: TAILQ_LARGE_FROM); the last workgroup partly filled"""
	import nway_amd as nw
	from nway_amd import _hip
	rng = np.random.default_rng(31)
	tabs = sky_tables(70001, 300000, 250000, 31)
	crowded(tabs, rng, range(0, 400), 1, 1)
	crowded(tabs, rng, range(300, 700), 2, 1)
	crowded(tabs, rng, range(69900, 70001), 2, 1)
	t = quad_and_lane(nw, tabs, 10.0, **(dict(correction=_hip.CORRECTION_CLI) if correction else {}))
	groups = np.bincount(t['PRIM'].astype(np.int64), minlength=70001)
	assert groups.max() >= 6 and (groups == 1).sum() > 0


def test_three_candidates_in_a_catalogue_send_the_run_to_the_walk():
	import nway_amd as nw
	from nway_amd import _hip
	rng = np.random.default_rng(30)
	tabs = sky_tables(5000, 200000, 150000, 30)
	crowded(tabs, rng, [77], 2, 2)
	t, _ = hip_table(nw, tabs, 10.0, 0.9)
	assert t['_desc']['tail'] == 'sparsek'   # (chosen by the run, not by the plan: NWAYHIP_FLAG_QUAD_DEEP)
	lane, _ = hip_table(nw, tabs, 10.0, 0.9, tuning=dict(disable=_hip.DISABLE_QUAD3))
	same_table(t, lane)
	assert np.bincount(t['PRIM'].astype(np.int64)).max() >= 8
	both_paths(nw, tabs, 10.0)


def test_flat_cells_and_pairs_in_different_buckets():
	import nway_amd as nw
	rng = np.random.default_rng(25)
	tabs = patch_tables(rng, [3000, 2500, 2400], 1.0, [1.0, 0.1, 0.5], centre=(0.2, -0.1))  # cells of both signs; radius = cell size
	crowded(tabs, rng, range(2500, 2520), 1, 1, spread=3.0)
	crowded(tabs, rng, range(2510, 2530), 2, 1, spread=3.0)
	t = quad_and_lane(nw, tabs, 5.0)
	assert np.isnan(t['Separation_T1_T2'][(t['T1'] >= 0) & (t['T2'] >= 0)]).sum() == 0


def test_the_scripts_correction_on_the_fast_lanes():
	import nway_amd as nw
	from nway_amd import _hip
	tabs = sky_tables(5000, 200000, 150000, 26)
	t = quad_and_lane(nw, tabs, 10.0, correction=_hip.CORRECTION_CLI)
	plain, _ = hip_table(nw, tabs, 10.0, 0.9)
	assert (t['dist_bayesfactor'] != plain['dist_bayesfactor']).sum() > 100


def test_float32_separations_of_the_script():
	import nway_amd as nw
	tabs = sky_tables(3000, 100000, 80000, 28)
	quad_and_lane(nw, tabs, 10.0, oracle=False, f32_roundtrip=True)
	rng = np.random.default_rng(29)
	flat = patch_tables(rng, [3000, 2500, 2400], 1.0, [1.0, 0.1, 0.5])
	quad_and_lane(nw, flat, 5.0, oracle=False, f32_roundtrip=True)
