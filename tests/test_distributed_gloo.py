"""world_size 2, 3 and 8 tests of the sharding logic with the gloo backend on CPU.

The HIP kernels cannot run here, so the per-rank compute is the CPU oracle (test
infrastructure, plugged into the engines' hooks by the subclasses of tests/cpu_engines.py --
nway_amd.distributed itself has no CPU path); what is under test is nway_amd.distributed: row sharding of the
primaries, all-gatherv of uneven secondary slices, global indices and the rank-order
concatenation reproducing the unsharded table exactly."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from goldenutil import ROOT, cat

sys.path.insert(0, os.path.join(ROOT, 'oracle'))


def free_port():
	s = socket.socket()
	s.bind(('127.0.0.1', 0))
	port = s.getsockname()[1]
	s.close()
	return port


def make_catalogues():
	rng = np.random.RandomState(21)
	def patch(n, name, err):
		return cat(name, rng.uniform(30.0, 30.4, size=n), rng.uniform(-0.2, 0.2, size=n), rng.uniform(0.5 * err, err, size=n), 0.16)
	return patch(401, 'A', 3.), patch(6001, 'B', 1.), patch(5000, 'C', 2.)


def worker(rank, world, port, outfile):
	os.environ['MASTER_ADDR'] = '127.0.0.1'
	os.environ['MASTER_PORT'] = str(port)
	dist.init_process_group('gloo', rank=rank, world_size=world)
	try:
		from nway_amd import distributed
		# all-gatherv of uneven (and empty) pieces
		mine = torch.arange(3 * rank, dtype=torch.float64) + 100 * rank
		full, counts = distributed.allgatherv(mine)
		assert counts == [3 * r for r in range(world)]
		expect = torch.cat([torch.arange(3 * r, dtype=torch.float64) + 100 * r for r in range(world)])
		assert torch.equal(full, expect)

		A, B, C = make_catalogues()
		pb = distributed.shard_bounds(len(A['ra']), world)
		bb = [0, 3500, len(B['ra'])] if world == 2 else distributed.shard_bounds(len(B['ra']), world)
		cb = [0, 1200, len(C['ra'])] if world == 2 else distributed.shard_bounds(len(C['ra']), world)
		def rows(t, lo, hi):
			return dict(t, ra=t['ra'][lo:hi], dec=t['dec'][lo:hi], error=t['error'][lo:hi])
		from cpu_engines import OracleShardedMatch
		sm = OracleShardedMatch(rows(A, pb[rank], pb[rank + 1]), [rows(B, bb[rank], bb[rank + 1]), rows(C, cb[rank], cb[rank + 1])],
			20., 0.85, device=torch.device('cpu'))
		assert sm.primary_offset == pb[rank]
		assert [len(f['ra']) for f in sm.full_secondaries] == [len(B['ra']), len(C['ra'])]
		sm.step()
		total = sm.total_rows()
		table = sm.gather_table(dst=0)
		if rank == 0:
			np.savez(outfile, total=total, **table)
	finally:
		dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3, 8])
def test_sharded_equals_unsharded(tmp_path, world):
	import nway_oracle as orc
	outfile = str(tmp_path / 'gathered.npz')
	mp.spawn(worker, args=(world, free_port(), outfile), nprocs=world, join=True)
	got = np.load(outfile)
	A, B, C = make_catalogues()
	want = orc.nway_match([A, B, C], 20., 0.85)
	assert int(got['total']) == len(want['ncat']) > 1000
	for key in want:
		if key.startswith('_'):
			continue
		np.testing.assert_array_equal(got[key], want[key], err_msg=key)


def test_shard_bounds():
	from nway_amd import distributed
	b = distributed.shard_bounds(10, 4)
	assert list(b) == [0, 3, 6, 8, 10]
	assert list(distributed.shard_bounds(2, 4)) == [0, 1, 2, 2, 2]


# ---- secondary-split mode (one job over several ranks): CPU stand-ins for the two device halves in tests/cpu_engines.py ----

def split_worker(rank, world, port, outfile, k):
	os.environ['MASTER_ADDR'] = '127.0.0.1'
	os.environ['MASTER_PORT'] = str(port)
	dist.init_process_group('gloo', rank=rank, world_size=world)
	try:
		from nway_amd import distributed
		A, B, C = make_catalogues()
		if world == 2:
			pb = [0, 150, len(A['ra'])]       # uneven shards of the primaries
			bb = [0, 3500, len(B['ra'])]      # and of the secondary streams
			cb = [0, 1200, len(C['ra'])]
		else:
			pb, bb, cb = [list(distributed.shard_bounds(len(t['ra']), world)) for t in (A, B, C)]
		def rows(t, lo, hi):
			return dict(t, ra=t['ra'][lo:hi], dec=t['dec'][lo:hi], error=t['error'][lo:hi])
		secs = [rows(B, bb[rank], bb[rank + 1]), rows(C, cb[rank], cb[rank + 1])][:k - 1]
		from cpu_engines import OracleSecondarySplitMatch
		sm = OracleSecondarySplitMatch(rows(A, pb[rank], pb[rank + 1]), secs, 20., 0.85, device=torch.device('cpu'))
		assert list(sm.bounds) == pb and sm.primary_offset == pb[rank]
		assert sm.sec_global == [len(B['ra']), len(C['ra'])][:k - 1] and sm.sec_offset == [bb[rank], cb[rank]][:k - 1]
		sm.step()
		total = sm.total_rows()
		table = sm.gather_table(dst=0)
		if rank == 0:
			np.savez(outfile, total=total, **table)
	finally:
		dist.destroy_process_group()


@pytest.mark.parametrize('world,k', [(2, 2), (2, 3), (8, 3)])
def test_secondary_split_equals_unsharded(tmp_path, world, k):
	"""every rank sweeps only its slice of the secondaries against ALL primaries; the candidates are
	routed to the owners of the primaries (all-to-all-v with global indices, uneven slices); the
	rank-order concatenation of the owners' tables is the unsharded table"""
	import nway_oracle as orc
	outfile = str(tmp_path / 'split.npz')
	mp.spawn(split_worker, args=(world, free_port(), outfile, k), nprocs=world, join=True)
	got = np.load(outfile)
	A, B, C = make_catalogues()
	want = orc.nway_match([A, B, C][:k], 20., 0.85)
	assert int(got['total']) == len(want['ncat']) > 400
	for key in want:
		if key.startswith('_'):
			continue
		np.testing.assert_array_equal(got[key], want[key], err_msg=key)


# ---- zone mode (both sides sharded by declination zones; one all-to-all-v of rows at set-up, no collective per step) ----

def zone_worker(rank, world, port, outfile, k, zpr=1):
	os.environ['MASTER_ADDR'] = '127.0.0.1'
	os.environ['MASTER_PORT'] = str(port)
	dist.init_process_group('gloo', rank=rank, world_size=world)
	try:
		from nway_amd import distributed
		# the all-to-all-v of rows by itself: rank r sends r + d + 1 rows to rank d, tagged with (source, destination, number)
		counts = [rank + d + 1 for d in range(world)]
		rows = torch.tensor([[rank, d, i] for d in range(world) for i in range(counts[d])], dtype=torch.float64)
		got = distributed.exchange_rows(rows, counts).numpy()
		want = np.array([[s, rank, i] for s in range(world) for i in range(s + rank + 1)], dtype=float)
		np.testing.assert_array_equal(got, want)

		A, B, C = make_catalogues()
		pb = [0, 150, len(A['ra'])] if world == 2 else distributed.shard_bounds(len(A['ra']), world)   # uneven input shards of every catalogue
		bb = [0, 3500, len(B['ra'])] if world == 2 else distributed.shard_bounds(len(B['ra']), world)
		cb = [0, 1200, len(C['ra'])] if world == 2 else distributed.shard_bounds(len(C['ra']), world)
		def rows_of(t, lo, hi):
			return dict(t, ra=t['ra'][lo:hi], dec=t['dec'][lo:hi], error=t['error'][lo:hi])
		secs = [rows_of(B, bb[rank], bb[rank + 1]), rows_of(C, cb[rank], cb[rank + 1])][:k - 1]
		from cpu_engines import OracleZoneShardedMatch
		zm = OracleZoneShardedMatch(rows_of(A, pb[rank], pb[rank + 1]), secs, 20., 0.85, device=torch.device('cpu'), zones_per_rank=zpr)
		assert zm.primary_offset == pb[rank] and zm.sec_global == [len(B['ra']), len(C['ra'])][:k - 1]
		assert len(zm.edges) == world * zpr - 1 and (np.diff(zm.edges) >= 0).all() and len(zm.zones) == zpr
		# every primary of the job has exactly one owner; the zones' secondaries overlap in the seams only
		own = torch.tensor([sum(len(z['primary']['ra']) for z in zm.zones), sum(len(z['secondaries'][0]['ra']) for z in zm.zones)], dtype=torch.int64)
		dist.all_reduce(own)
		assert int(own[0]) == len(A['ra']) and len(B['ra']) <= int(own[1]) <= (1.2 if world * zpr <= 3 else 1.6) * len(B['ra'])  # (seven seams of +-20 arcsec in a 0.4 degree patch)
		for z in zm.zones:
			assert (np.diff(z['primary_gidx']) > 0).all() and (np.diff(z['sec_gidx'][0]) > 0).all()  # (ascending global indices: the rows' order)
		zm.step()
		total = zm.total_rows()
		table = zm.gather_table(dst=0)
		if rank == 0:
			np.savez(outfile, total=total, zone_rows=zm.local_rows(), **table)
	finally:
		dist.destroy_process_group()


@pytest.mark.parametrize('world,k,zpr', [(2, 2, 1), (2, 3, 1), (3, 3, 1), (8, 3, 1), (2, 3, 3), (1, 2, 4)])
def test_zone_sharded_equals_unsharded(tmp_path, world, k, zpr):
	"""primaries AND secondaries redistributed by declination zones (edges from the summed histogram of the largest secondary
	catalogue; the secondaries inside the seams go to both neighbours); the ranks' tables, concatenated and sorted by primary,
	are the unsharded table -- global indices, row order inside the groups, every value; also with several zones per rank (round 5:
	a rank runs its zones one after the other), down to ONE rank with four zones"""
	import nway_oracle as orc
	outfile = str(tmp_path / 'zones.npz')
	mp.spawn(zone_worker, args=(world, free_port(), outfile, k, zpr), nprocs=world, join=True)
	got = np.load(outfile)
	A, B, C = make_catalogues()
	want = orc.nway_match([A, B, C][:k], 20., 0.85)
	assert int(got['total']) == len(want['ncat']) > 400
	assert 0 < int(got['zone_rows']) <= int(got['total']) and (world == 1 or int(got['zone_rows']) < int(got['total']))
	for key in want:
		if key.startswith('_'):
			continue
		np.testing.assert_array_equal(got[key], want[key], err_msg=key)


@pytest.mark.parametrize('scalar_error', [False, True])
def test_zones_cut_where_resident_equal_the_host_side_cut(scalar_error):
	"""a process alone cuts its zones out of the uploaded columns (torch.bucketize / nonzero where they lie, the histogram of the
	edges likewise); with ranks to exchange with it cuts on the host and packs rows for the all-to-all-v.  Same edges, same rows
	in the same order, same values -- also with NaN and infinite declinations, a scalar error, five zones and three catalogues"""
	from nway_amd import distributed
	from cpu_engines import OracleZoneShardedMatch
	A, B, C = zone_edge_catalogues('lopsided')
	B['dec'][13] = np.inf
	C['dec'][5] = -np.inf
	if scalar_error:
		B = dict(B, error=1.0)

	class HostCut(OracleZoneShardedMatch):
		CUT_WHERE_RESIDENT = False

	made = [cls(A, [B, C], 20., 0.85, device=torch.device('cpu'), zones_per_rank=5, local_only=True) for cls in (OracleZoneShardedMatch, HostCut)]
	here, host = made
	np.testing.assert_array_equal(here.edges, host.edges)
	assert len(here.zones) == len(host.zones) == 5 and here.moved_bytes == host.moved_bytes
	for zh, zo in zip(here.zones, host.zones):
		np.testing.assert_array_equal(zh['primary_gidx'], zo['primary_gidx'])
		for th, to in [(zh['primary'], zo['primary'])] + list(zip(zh['secondaries'], zo['secondaries'])):
			for col in ('ra', 'dec', 'error'):
				np.testing.assert_array_equal(np.asarray(th[col]), np.asarray(to[col]), err_msg=col)
		for gh, go in zip(zh['sec_gidx'], zo['sec_gidx']):
			np.testing.assert_array_equal(gh, go)
	for e in made:
		e.step()
	assert here.total_rows() == host.total_rows() > 300


def zone_edge_worker(rank, world, port, outfile, case):
	os.environ['MASTER_ADDR'] = '127.0.0.1'
	os.environ['MASTER_PORT'] = str(port)
	dist.init_process_group('gloo', rank=rank, world_size=world)
	try:
		from nway_amd import distributed
		from cpu_engines import OracleZoneShardedMatch
		A, B, C = zone_edge_catalogues(case)
		def rows_of(t, b):
			return dict(t, ra=t['ra'][b[rank]:b[rank + 1]], dec=t['dec'][b[rank]:b[rank + 1]], error=t['error'][b[rank]:b[rank + 1]])
		pb, bb, cb = [distributed.shard_bounds(len(t['ra']), world) for t in (A, B, C)]
		if case == 'lopsided':
			pb = [0, 0, len(A['ra']) // 3, len(A['ra'])]  # (a rank that brings no primaries at all)
		zm = OracleZoneShardedMatch(rows_of(A, pb), [rows_of(B, bb), rows_of(C, cb)], 20., 0.85, device=torch.device('cpu'))
		zm.step()
		total = zm.total_rows()
		sizes = torch.tensor([len(zm.zone_primary['ra']), len(zm.zone_secondaries[0]['ra']), len(zm.zone_secondaries[1]['ra'])], dtype=torch.int64)
		allsizes = [torch.zeros_like(sizes) for _ in range(world)]
		dist.all_gather(allsizes, sizes)
		table = zm.gather_table(dst=0)
		if rank == 0:
			np.savez(outfile, total=total, zone_sizes=torch.stack(allsizes).numpy(), **table)
	finally:
		dist.destroy_process_group()


def zone_edge_catalogues(case):
	rng = np.random.RandomState(5)
	def patch(n, name, err, dec_lo, dec_hi):
		return cat(name, rng.uniform(30.0, 30.3, size=n), rng.uniform(dec_lo, dec_hi, size=n), rng.uniform(0.5 * err, err, size=n), 0.09)
	if case == 'one_declination':
		# every source of the largest catalogue on ONE declination: the edges collapse, one rank owns everything
		A, B, C = patch(120, 'A', 3., 0.1, 0.1), patch(2000, 'B', 1., 0.1, 0.1), patch(900, 'C', 2., 0.1 - 0.002, 0.1 + 0.002)
	else:
		# the third catalogue lies in the southern third only: two zones hold none of it; a primary with a NaN declination keeps its row
		A, B, C = patch(300, 'A', 3., -0.15, 0.15), patch(3000, 'B', 1., -0.15, 0.15), patch(700, 'C', 2., -0.15, -0.06)
		A['dec'][7] = np.nan
		B['dec'][11] = np.nan
	return A, B, C


@pytest.mark.parametrize('case', ['lopsided', 'one_declination'])
def test_zone_sharded_edge_cases(tmp_path, case):
	"""zones that hold nothing of a catalogue, a rank without input primaries, NaN declinations, collapsed zone edges (world 3)"""
	import nway_oracle as orc
	outfile = str(tmp_path / 'zones_edge.npz')
	mp.spawn(zone_edge_worker, args=(3, free_port(), outfile, case), nprocs=3, join=True)
	got = np.load(outfile)
	A, B, C = zone_edge_catalogues(case)
	want = orc.nway_match([A, B, C], 20., 0.85)
	zs = got['zone_sizes']
	assert zs[:, 0].sum() == len(A['ra'])
	if case == 'lopsided':
		assert (zs[:, 2] == 0).sum() >= 1 and (zs[:, 0] > 0).all()  # (a zone with primaries and nothing of catalogue C)
	else:
		assert (zs[:, 0] > 0).sum() == 1  # (one rank owns every primary)
	assert int(got['total']) == len(want['ncat'])
	for key in want:
		if key.startswith('_'):
			continue
		np.testing.assert_array_equal(got[key], want[key], err_msg=key)


# ---- magnitude priors on a sharded match (distributed.MagnitudePriors): selection gathered, histograms on every rank, rows local ----

def mag_worker(rank, world, port, outfile, mode, which):
	os.environ['MASTER_ADDR'] = '127.0.0.1'
	os.environ['MASTER_PORT'] = str(port)
	dist.init_process_group('gloo', rank=rank, world_size=world)
	try:
		from nway_amd import distributed
		from cpu_engines import OracleShardedMatch, OracleZoneShardedMatch
		from goldenutil import mag3_tables, magmix_tables
		tabs = mag3_tables() if which == 'mag3' else magmix_tables()
		radius, comp, kw = (20., 0.9, {}) if which == 'mag3' else (12., np.array([1.0, 0.8, 0.7]), dict(mag_include_radius=1.5, mag_exclude_radius=6.0))
		bounds = [distributed.shard_bounds(len(t['ra']), world) for t in tabs]
		if world == 2:
			bounds[0] = [0, len(tabs[0]['ra']) // 3, len(tabs[0]['ra'])]  # (uneven shards)
		def rows(t, b):
			return dict(t, ra=t['ra'][b[rank]:b[rank + 1]], dec=t['dec'][b[rank]:b[rank + 1]], error=t['error'][b[rank]:b[rank + 1]], mags=[], magnames=[], maghists=[])
		cls = OracleShardedMatch if mode == 'rows' else OracleZoneShardedMatch
		eng = cls(rows(tabs[0], bounds[0]), [rows(t, b) for t, b in zip(tabs[1:], bounds[1:])], radius, comp, device=torch.device('cpu'))
		eng.step()
		mags = [[(n, np.array(v[b[rank]:b[rank + 1]]), h) for n, v, h in zip(t['magnames'], t['mags'], t['maghists'])] for t, b in zip(tabs, bounds)]
		local = eng.magnitude_priors(mags, **kw)
		table = eng.gather_magnitude_table(local, dst=0)
		if rank == 0:
			np.savez(outfile, **table)
	finally:
		dist.destroy_process_group()


@pytest.mark.parametrize('mode,world', [('rows', 2), ('zones', 3)])
def test_sharded_magnitude_priors_three_way_golden(tmp_path, mode, world):
	"""the shape of BASELINE configs[1] -- XMM x OPT x IRAC, three magnitude columns over two catalogues, histograms learned from the
	posterior-selected matches -- over several ranks (primary rows sharded / declination zones): the gathered table equals the
	REFERENCE's table of the one-process run (tests/golden/mag3.npz)"""
	from goldenutil import golden, assert_table_matches, assert_checksums_match
	outfile = str(tmp_path / 'mag.npz')
	mp.spawn(mag_worker, args=(world, free_port(), outfile, mode, 'mag3'), nprocs=world, join=True)
	t = dict(np.load(outfile))
	g = golden('mag3')
	names = ['XMM', 'OPT', 'IRAC']
	assert_checksums_match(t, g, 'm3_', names)
	rows = g['m3_sub_rows']
	assert_table_matches(t, g, 'm3_sub_', names, rows=rows, rtol=1e-9, atol=1e-13)
	for b in ('bias_OPT_R', 'bias_OPT_I', 'bias_IRAC_CH1'):
		np.testing.assert_allclose(np.asarray(t[b])[rows], g['m3_sub_' + b], rtol=1e-9, err_msg=b)
		np.testing.assert_allclose(np.sum(t[b]), g['m3_sum_' + b][0], rtol=1e-9, err_msg=b)


def test_sharded_magnitude_priors_on_every_catalogue_golden(tmp_path):
	"""magnitude columns on the PRIMARY (supplied histogram) and on both secondaries (learned by radius, exclusion radius apart),
	primary rows sharded over two ranks: the reference's table (tests/golden/magmix.npz)"""
	from goldenutil import golden, assert_table_matches
	outfile = str(tmp_path / 'magmix.npz')
	mp.spawn(mag_worker, args=(2, free_port(), outfile, 'rows', 'magmix'), nprocs=2, join=True)
	t = dict(np.load(outfile))
	g = golden('magmix')
	assert_table_matches(t, g, 'rad_', ['P', 'A', 'B'], rtol=1e-9, atol=1e-13)
	for b in ('bias_P_F', 'bias_A_M', 'bias_B_M'):
		np.testing.assert_allclose(t[b], g['rad_' + b], rtol=1e-9, err_msg=b)


def test_bench_plain_form_starts_the_ranks_itself():
	"""`python bench.py --gpus N` without a launcher and without WORLD_SIZE (round 6): bench.py becomes the launcher of N ranks
	(torch.distributed.run, 127.0.0.1) instead of running one GPU -- here over gloo and only as far as the rendezvous (no GPU):
	rank 0 reports how many ranks answered; with the real backend and fewer GPUs than ranks the one-rank run says so in `error`"""
	import json
	import subprocess
	bench = os.path.join(ROOT, 'bench.py')
	env = dict(os.environ, NWAY_BENCH_BACKEND='gloo')
	env.pop('WORLD_SIZE', None)
	env.pop('RANK', None)
	res = subprocess.run([sys.executable, bench, '--gpus', '3', '--rendezvous-only'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True,
		timeout=300, env=env, cwd=ROOT)
	assert res.returncode == 0, res.stderr[-2000:]
	out = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])
	assert out['world'] == 3 and out['ranks_seen'] == 3 and out['gpus_asked'] == 3 and out['error'] is None
	assert 'torch.distributed.run' in res.stderr and '--nproc-per-node 3' in res.stderr
	if not torch.cuda.is_available():
		env['NWAY_BENCH_BACKEND'] = 'nccl'
		res = subprocess.run([sys.executable, bench, '--gpus', '2', '--rendezvous-only'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True,
			timeout=300, env=env, cwd=ROOT)
		out = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])
		assert out['world'] == 1 and 'GPU(s) visible' in out['error']
