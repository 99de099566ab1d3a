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


def hip_table(nw, tables, radius, completeness):
	res = nw.run_match(tables, radius, completeness, logger=nw.NullOutputLogger())
	names = res.names
	t = {}
	for c, n in enumerate(names):
		t[n] = res.to_host('idx', c).astype(np.int64)
	t['Separation_%s_%s' % (names[0], names[1])] = res.to_host('sep', 0)
	for src, dst in (('sep_max', 'Separation_max'), ('log_bf', 'dist_bayesfactor'), ('dist_post', 'dist_post'),
			('p_single', 'p_single'), ('p_any', 'prob_has_match'), ('p_i', 'prob_this_match')):
		t[dst] = res.to_host(src)
	t['ncat'] = res.to_host('ncat').astype(np.int64)
	t['match_flag'] = res.to_host('match_flag').astype(np.int64)
	status = res.status
	res.plan.close()
	return t, status


def compare(t, o, names):
	assert len(t['ncat']) == len(o['ncat'])
	for n in names:
		np.testing.assert_array_equal(t[n], o[n])
	np.testing.assert_array_equal(t['ncat'], o['ncat'])
	np.testing.assert_array_equal(t['match_flag'], o['match_flag'])
	np.testing.assert_allclose(t['Separation_%s_%s' % tuple(names)], o['Separation_%s_%s' % tuple(names)], rtol=RTOL, atol=1e-9, equal_nan=True)
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
