"""BASELINE.json-sized runs on the GPU: the bench workload (1e5 x 1e7 uniform sky, 5 arcsec)
and its dense flat-cell variant (same counts in a 6 deg^2 patch) against the C restatement of
the oracle, plus size-independent properties of the table."""
import os
import sys

import numpy as np
import pytest

from goldenutil import ROOT, RTOL, ATOL

sys.path.insert(0, os.path.join(ROOT, 'oracle'))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def check_properties(t, names, n_primary):
	prim = t[names[0]]
	k = len(names)
	# one contiguous, ascending block per primary, starting with its no-counterpart row
	assert (np.diff(prim) >= 0).all() and prim[0] == 0 and prim[-1] == n_primary - 1
	starts = np.flatnonzero(np.r_[True, prim[1:] != prim[:-1]])
	assert len(starts) == n_primary
	for c in range(1, k):
		assert (t[names[c]][starts] == -1).all()
	assert (t['ncat'][starts] == 1).all() and (t['prob_this_match'][starts] == 0).all()
	# lexicographic order inside the blocks
	key = np.stack([t[n] for n in names], axis=1)
	later = np.ones(len(key) - 1, dtype=bool)
	decided = np.zeros(len(key) - 1, dtype=bool)
	for c in range(k):
		d = key[1:, c] - key[:-1, c]
		later &= decided | (d >= 0)
		decided |= d != 0
	assert later.all() and decided.all()
	# probabilities: p_i of the candidates of a primary sum to 1, p_any in [0, 1], one best match
	counts = np.diff(np.r_[starts, len(prim)])
	sums = np.add.reduceat(t['prob_this_match'], starts)
	np.testing.assert_allclose(sums[counts > 1], 1.0, rtol=1e-9)
	assert (t['prob_has_match'] >= -1e-12).all() and (t['prob_has_match'] <= 1).all()
	assert (np.add.reduceat((t['match_flag'] == 1).astype(int), starts) >= 1).all()
	assert (t['Separation_max'] >= 0).all()


def hip_table(nw, tables, radius, completeness, **options):
	from nway_amd import _hip
	res = nw.run_match(tables, radius, completeness, logger=nw.NullOutputLogger(), **options)
	names = res.names
	t = {}
	for c, n in enumerate(names):
		t[n] = res.to_host('idx', c).astype(np.int64)
	for p, (i, j) in enumerate(_hip.pair_columns(len(names))):
		t['Separation_%s_%s' % (names[i], names[j])] = res.to_host('sep', p)
	# (dist_bayesfactor is the corrected column; it equals log_bf unless the script's correction was asked for)
	for src, dst in (('sep_max', 'Separation_max'), ('log_bf_corrected' if options.get('correction') else 'log_bf', 'dist_bayesfactor'), ('dist_post', 'dist_post'),
			('p_single', 'p_single'), ('p_any', 'prob_has_match'), ('p_i', 'prob_this_match')):
		t[dst] = res.to_host(src)
	t['ncat'] = res.to_host('ncat').astype(np.int64)
	t['match_flag'] = res.to_host('match_flag').astype(np.int64)
	status = res.status
	t['_sparse'] = res.plan.sparse
	t['_path'] = res.plan.path
	t['_link_slots'] = res.plan.link_slots
	t['_desc'] = dict(res.plan.description)
	res.plan.close()
	return t, status


def compare(t, o, names):
	assert len(t['ncat']) == len(o['ncat'])
	for n in names:
		np.testing.assert_array_equal(t[n], o[n])
	np.testing.assert_array_equal(t['ncat'], o['ncat'])
	np.testing.assert_array_equal(t['match_flag'], o['match_flag'])
	for i in range(len(names)):
		for j in range(i + 1, len(names)):
			c = 'Separation_%s_%s' % (names[i], names[j])
			np.testing.assert_allclose(t[c], o[c], rtol=RTOL, atol=1e-9, equal_nan=True, err_msg=c)
	for c in ('Separation_max', 'dist_bayesfactor', 'dist_post', 'p_single', 'prob_has_match', 'prob_this_match'):
		np.testing.assert_allclose(t[c], o[c], rtol=RTOL, atol=ATOL, err_msg=c)


def test_bench_workload_full_size():
	import bench
	import nway_amd as nw
	import nway_oracle_c as orc_c
	prim, sec = bench.make_workload(100000, 10000000, 1)
	t, status = hip_table(nw, [prim, sec], 5.0, 0.9)
	assert len(t['ncat']) == 180103
	check_properties(t, ['PRIM', 'SEC'], 100000)
	o = orc_c.nway_match([prim, dict(sec, error=0.1 * np.ones(len(sec['ra'])))], 5.0, 0.9)
	compare(t, o, ['PRIM', 'SEC'])
	# idempotence: a second run gives the identical table
	t2, _ = hip_table(nw, [prim, sec], 5.0, 0.9)
	for key in t:
		np.testing.assert_array_equal(t[key], t2[key])
	# the general path (what denser inputs take) gives the same table
	t3, _ = hip_table(nw, [prim, sec], 5.0, 0.9, link_slots=-1)
	assert t['_sparse'] and not t3['_sparse']
	for key in t:
		if not key.startswith('_'):
			np.testing.assert_array_equal(t[key], t3[key], err_msg=key)


def test_dense_flat_patch_full_size():
	import nway_amd as nw
	import nway_oracle_c as orc_c
	rng = np.random.default_rng(2)
	n0, n1 = 100000, 10000000
	half = 1.23
	def patch(n):
		return rng.uniform(150 - half, 150 + half, size=n), rng.uniform(2 - half, 2 + half, size=n)
	pra, pdec = patch(n0)
	sra, sdec = patch(n1)
	psig = rng.uniform(0.3, 1.5, size=n0)
	ntrue = 80000
	slots = rng.choice(n1, size=ntrue, replace=False)
	sdec[slots] = pdec[:ntrue] + rng.normal(0, 1, size=ntrue) * psig[:ntrue] / 3600.
	sra[slots] = pra[:ntrue] + rng.normal(0, 1, size=ntrue) * psig[:ntrue] / 3600. / np.cos(np.radians(pdec[:ntrue]))
	area = (2 * half)**2
	prim = dict(name='P', ra=pra, dec=pdec, error=psig, area=area, mags=[], maghists=[], magnames=[])
	sec = dict(name='S', ra=sra, dec=sdec, error=0.1 * np.ones(n1), area=area, mags=[], maghists=[], magnames=[])
	assert nw.choose_scheme([(pra, pdec), (sra, sdec)], 5. / 3600) == 0  # flat cells
	t, status = hip_table(nw, [prim, sec], 5.0, 0.9)
	assert len(t['ncat']) > 1000000
	check_properties(t, ['P', 'S'], n0)
	o = orc_c.nway_match([prim, sec], 5.0, 0.9)
	compare(t, o, ['P', 'S'])


def sphere_catalogue(rng, name, n, sigma, parents=None, frac=0.0, psig=None):
	import bench
	ra, dec = bench.uniform_sphere(rng, n)
	if parents is not None:
		m = int(frac * len(parents['ra']))
		slots = rng.choice(n, size=m, replace=False)
		dec[slots] = np.clip(parents['dec'][:m] + rng.normal(0, 1, size=m) * psig[:m] / 3600., -90, 90)
		ra[slots] = (parents['ra'][:m] + rng.normal(0, 1, size=m) * psig[:m] / 3600. / np.maximum(np.cos(np.radians(parents['dec'][:m])), 1e-6)) % 360
	return dict(name=name, ra=ra, dec=dec, error=sigma * np.ones(n), area=bench.SKY_AREA, mags=[], maghists=[], magnames=[])


def test_three_way_uniform_sky_full_size():
	"""BASELINE configs[3] (SURVEY 8d "C4-S"): 3-way 1e5 x 1e6 x 1e6, uniform sky, 10 arcsec, on the
	fused sparse tail k_tailk<3> and on the general path (breadth-first expansion), both against
	the C restatement of the oracle (__init__.py:123-196, fastskymatch.py:94-98,135-160)"""
	import nway_amd as nw
	import nway_oracle_c as orc_c
	rng = np.random.default_rng(3)
	n0, n1 = 100000, 1000000
	psig = np.ones(n0)
	prim = sphere_catalogue(rng, 'P', n0, 1.0)
	a = sphere_catalogue(rng, 'A', n1, 0.1, prim, 0.8, psig)
	b = sphere_catalogue(rng, 'B', n1, 0.5, prim, 0.6, psig)
	tabs = [prim, a, b]
	names = ['P', 'A', 'B']
	o = orc_c.nway_match(tabs, 10.0, 0.9, threads=0)
	assert len(o['ncat']) > 280000 and (o['ncat'] == 3).sum() > 40000
	for slots in (0, -1):
		t, status = hip_table(nw, tabs, 10.0, 0.9, link_slots=slots)
		assert t['_sparse'] == (slots == 0)
		check_properties(t, names, n0)
		compare(t, o, names)


def test_all_sky_five_hundred_thousand_by_hundred_million():
	"""BASELINE configs[4] (SURVEY 8d "C5"): 2-way 5e5 x 1e8, uniform sky, 5 arcsec.  The whole job
	on ONE GPU: size-independent properties of the 9e5-row table; one of the eight primary shards
	the configuration is specified with (62 500 primaries against all 1e8 secondaries) against the C
	restatement of the oracle, and against the same rows of the whole run"""
	import bench
	import nway_amd as nw
	import nway_oracle_c as orc_c
	prim, sec = bench.make_workload(500000, 100000000, 5)
	t, status = hip_table(nw, [prim, sec], 5.0, 0.9)
	assert 880000 < len(t['ncat']) < 940000
	check_properties(t, ['PRIM', 'SEC'], 500000)
	lo, hi = 3 * 62500, 4 * 62500
	shard = dict(prim, ra=prim['ra'][lo:hi], dec=prim['dec'][lo:hi], error=prim['error'][lo:hi], area=prim['area'] * 62500 / 500000.)
	ts, _ = hip_table(nw, [shard, sec], 5.0, 0.9)
	assert ts['_sparse']
	o = orc_c.nway_match([shard, dict(sec, error=0.1 * np.ones(len(sec['ra'])))], 5.0, 0.9, threads=0)
	compare(ts, o, ['PRIM', 'SEC'])
	# the shard's table is the whole run's block of rows (priors do not depend on the primary density)
	rows = (t['PRIM'] >= lo) & (t['PRIM'] < hi)
	assert rows.sum() == len(ts['ncat'])
	np.testing.assert_array_equal(t['PRIM'][rows] - lo, ts['PRIM'])
	np.testing.assert_array_equal(t['SEC'][rows], ts['SEC'])
	np.testing.assert_array_equal(t['match_flag'][rows], ts['match_flag'])
	np.testing.assert_allclose(t['prob_this_match'][rows], ts['prob_this_match'], rtol=RTOL, atol=ATOL)
	np.testing.assert_allclose(t['prob_has_match'][rows], ts['prob_has_match'], rtol=RTOL, atol=ATOL)
