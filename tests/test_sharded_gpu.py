"""The primary-row sharding on the device (what BASELINE configs[3] / [4] name): two ranks (gloo,
sharing the one GPU of the test box) each own a contiguous range of primaries, all-gather their
slices of the secondary catalogues and run the HIP pipeline on their shard; the rank-order
concatenation must equal the single-GPU table of the same catalogues AND the oracle's.
Also: bench.py itself under torch.distributed.run with two ranks, both scaling modes."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from goldenutil import ROOT, RTOL, ATOL, cat

sys.path.insert(0, os.path.join(ROOT, 'oracle'))

pytestmark = pytest.mark.gpu


def free_port():
	s = socket.socket()
	s.bind(('127.0.0.1', 0))
	port = s.getsockname()[1]
	s.close()
	return port


def catalogues(k, flat):
	rng = np.random.RandomState(91)
	n0, n1, n2 = 20000, 300000, 200000
	if flat:
		sky = lambda n: (rng.uniform(200.0, 220.0, n), rng.uniform(-15.0, 15.0, n))
		area = 20.0 * 30.0
	else:
		sky = lambda n: (rng.uniform(0, 360, n), np.degrees(np.arcsin(rng.uniform(-1, 1, n))))
		area = 41252.96
	a = cat('A', *sky(n0), rng.uniform(0.5, 2, n0), area)
	b = cat('B', *sky(n1), rng.uniform(0.2, 0.4, n1), area)
	c = cat('C', *sky(n2), 0.5 * np.ones(n2), area)
	for t, m in ((b, 14000), (c, 9000)):
		t['ra'][:m] = a['ra'][:m] + rng.normal(0, 1, m) / 3600.
		t['dec'][:m] = np.clip(a['dec'][:m] + rng.normal(0, 1, m) / 3600., -90, 90)
	return [a, b, c][:k]


def worker(rank, world, port, outfile, k, flat, cut):
	os.environ['MASTER_ADDR'] = '127.0.0.1'
	os.environ['MASTER_PORT'] = str(port)
	os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
	dist.init_process_group('gloo', rank=rank, world_size=world)
	try:
		sys.path.insert(0, ROOT)
		from nway_amd import distributed
		tabs = catalogues(k, flat)
		dev = torch.device('cuda', 0)
		torch.cuda.set_device(dev)
		n0 = len(tabs[0]['ra'])
		pb = [0, int(cut * n0), n0]  # cut = 1.0: the second rank owns no primary at all
		def rows(t, lo, hi):
			return dict(t, ra=t['ra'][lo:hi], dec=t['dec'][lo:hi], error=t['error'][lo:hi])
		secs = []
		for c in range(1, k):
			n = len(tabs[c]['ra'])
			sc = [0, int(0.41 * n), n]  # uneven slices of the secondaries as well
			secs.append(rows(tabs[c], sc[rank], sc[rank + 1]))
		sm = distributed.ShardedMatch(rows(tabs[0], pb[rank], pb[rank + 1]), secs, 10., 0.9, device=dev)
		assert sm.primary_offset == pb[rank]
		assert [int(f['ra'].shape[0]) for f in sm.full_secondaries] == [len(t['ra']) for t in tabs[1:]]
		for _ in range(3):  # (repeated steps recycle the scratch copies)
			sm.step()
		total = sm.total_rows()
		table = sm.gather_table(dst=0)
		if rank == 0:
			np.savez(outfile, total=total, **table)
	finally:
		dist.destroy_process_group()


@pytest.mark.parametrize('k,flat,cut', [(2, False, 0.7), (3, False, 0.35), (2, True, 0.5), (3, True, 0.6), (2, False, 1.0)])
def test_primary_shards_on_device(tmp_path, k, flat, cut):
	import nway_amd as nw
	import nway_oracle_c as orc_c
	outfile = str(tmp_path / 'sharded.npz')
	mp.spawn(worker, args=(2, free_port(), outfile, k, flat, cut), nprocs=2, join=True)
	got = np.load(outfile)
	tabs = catalogues(k, flat)
	names = [t['name'] for t in tabs]
	# the single-GPU table of the library: identical, bit for bit
	want = nw.nway_match(tabs, 10., 0.9, logger=nw.NullOutputLogger())
	assert int(got['total']) == len(want) > len(tabs[0]['ra'])
	for key in want.columns:
		np.testing.assert_array_equal(got[key], want[key].values, err_msg=key)
	# and the oracle's
	o = orc_c.nway_match(tabs, 10., 0.9)
	for n in names:
		np.testing.assert_array_equal(got[n], o[n])
	np.testing.assert_array_equal(got['ncat'], o['ncat'])
	np.testing.assert_array_equal(got['match_flag'], o['match_flag'])
	for c in ('Separation_max', 'dist_bayesfactor', 'dist_post', 'p_single', 'prob_has_match', 'prob_this_match'):
		np.testing.assert_allclose(got[c], o[c], rtol=RTOL, atol=ATOL, err_msg=c)


def lopsided_catalogues():
	"""3-way all-sky; the third catalogue lies in the southern sky only (a northern zone holds none of it), two sources without a declination"""
	tabs = catalogues(3, False)
	c = tabs[2]
	south = c['dec'] < -20.0
	tabs[2] = dict(c, ra=c['ra'][south].copy(), dec=c['dec'][south].copy(), error=c['error'][south].copy())
	tabs[0]['dec'][5] = np.nan
	tabs[1]['dec'][17] = np.nan
	return tabs


def zone_worker(rank, world, port, outfile, k, flat):
	os.environ['MASTER_ADDR'] = '127.0.0.1'
	os.environ['MASTER_PORT'] = str(port)
	os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
	dist.init_process_group('gloo', rank=rank, world_size=world)
	try:
		sys.path.insert(0, ROOT)
		from nway_amd import distributed
		tabs = catalogues(k, flat) if k > 0 else lopsided_catalogues()
		dev = torch.device('cuda', 0)
		torch.cuda.set_device(dev)
		def rows(t, lo, hi):
			return dict(t, ra=t['ra'][lo:hi], dec=t['dec'][lo:hi], error=t['error'][lo:hi])
		parts = []
		for c, t in enumerate(tabs):
			n = len(t['ra'])
			cut = [0, int((0.7, 0.41, 0.55)[c] * n), n]  # uneven input shards of every catalogue
			parts.append(rows(t, cut[rank], cut[rank + 1]))
		zm = distributed.ZoneShardedMatch(parts[0], parts[1:], 10., 0.9, device=dev)
		# the zone of a rank: about half of every catalogue (the edges are quantiles of the largest secondary catalogue)
		assert 0.3 * len(tabs[1]['ra']) < zm.cats[1].n < 0.7 * len(tabs[1]['ra'])
		if k == 0:
			assert zm.cats[2].n == (0 if rank == 1 else len(tabs[2]['ra']))  # (the northern zone holds nothing of the third catalogue)
		for _ in range(3):  # (repeated steps recycle the scratch copies)
			zm.step()
		total = zm.total_rows()
		table = zm.gather_table(dst=0)
		if rank == 0:
			np.savez(outfile, total=total, zone_rows=zm.local_rows(), **table)
	finally:
		dist.destroy_process_group()


@pytest.mark.parametrize('k,flat', [(2, False), (3, False), (2, True), (3, True), (0, False)])
def test_declination_zones_on_device(tmp_path, k, flat):
	"""both sides sharded by declination zones (ZoneShardedMatch), two gloo ranks on the one GPU through the HIP pipeline: the
	ranks' tables, concatenated and sorted by primary, equal the single-GPU table bit for bit (and the C oracle's)"""
	import nway_amd as nw
	import nway_oracle_c as orc_c
	outfile = str(tmp_path / 'zones.npz')
	mp.spawn(zone_worker, args=(2, free_port(), outfile, k, flat), nprocs=2, join=True)
	got = np.load(outfile)
	tabs = catalogues(k, flat) if k > 0 else lopsided_catalogues()  # (k = 0: a zone without a catalogue, sources without a declination)
	want = nw.nway_match(tabs, 10., 0.9, logger=nw.NullOutputLogger())
	assert int(got['total']) == len(want) > len(tabs[0]['ra'])
	assert 0.3 * len(want) < int(got['zone_rows']) < 0.7 * len(want)
	for key in want.columns:
		np.testing.assert_array_equal(got[key], want[key].values, err_msg=key)
	o = orc_c.nway_match(tabs, 10., 0.9)
	for n in [t['name'] for t in tabs] + ['ncat', 'match_flag']:
		np.testing.assert_array_equal(got[n], o[n])
	for c in ('Separation_max', 'dist_bayesfactor', 'dist_post', 'p_single', 'prob_has_match', 'prob_this_match'):
		np.testing.assert_allclose(got[c], o[c], rtol=RTOL, atol=ATOL, err_msg=c)


@pytest.mark.parametrize('scaling', ['weak', 'strong'])
def test_bench_under_torchrun_two_ranks(tmp_path, scaling):
	"""bench.py as the driver launches it for N > 1 (one process per rank, torch.distributed.run), here
	with two gloo ranks on the one GPU and a reduced workload: the JSON line of rank 0"""
	env = dict(os.environ, NWAY_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', NWAY_BENCH_EXTRAS='0')
	cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
		'--master-port', str(free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '2', '--prewarm', '3',
		'--n-primary', '20000', '--n-secondary', '2000000', '--scaling', scaling]
	res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600, env=env, cwd=ROOT)
	assert res.returncode == 0, res.stderr[-3000:]
	line = [l for l in res.stdout.splitlines() if l.startswith('{')][-1]
	out = json.loads(line)
	assert out['n_gpus'] == 2 and out['steps'] == 5 and out['scaling'] == scaling
	assert out['metric'] == 'candidate Bayes-factor evals/s' and out['value'] > 0 and out['ms_per_step'] > 0
	rows = out['config']['rows_per_step']
	if scaling == 'weak':
		# every rank owns 20 000 primaries (80 % of them with a counterpart): its rows, twice
		assert 2 * 20000 * 1.7 < rows < 2 * 20000 * 1.9
	else:
		assert 20000 * 1.7 < rows < 20000 * 1.9
	assert out['roofline']['frac'] > 0 and out['roofline']['pass_frac'] > 0
	(tmp_path / ('bench_x2_%s.json' % scaling)).write_text(line)
	keep = os.path.join(ROOT, 'gpurun_out')
	if os.path.isdir(keep):
		with open(os.path.join(keep, 'bench_x2_%s.json' % scaling), 'w') as f:
			f.write(line + '\n')


def rccl_worker(rank, world, port, outfile):
	os.environ['MASTER_ADDR'] = '127.0.0.1'
	os.environ['MASTER_PORT'] = str(port)
	os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
	dev = torch.device('cuda', 0)
	torch.cuda.set_device(dev)
	dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
	try:
		sys.path.insert(0, ROOT)
		from nway_amd import distributed
		x = torch.arange(5, dtype=torch.float64, device=dev)
		# every collective the engines and bench.py issue, through RCCL itself
		got = [torch.zeros(1, dtype=torch.int64, device=dev)]
		dist.all_gather(got, torch.tensor([7], dtype=torch.int64, device=dev))
		full = torch.empty(5, dtype=torch.float64, device=dev)
		dist.all_gather_into_tensor(full, x)
		a2a = torch.empty(64, dtype=torch.uint8, device=dev)
		dist.all_to_all_single(a2a, torch.arange(64, dtype=torch.uint8, device=dev))
		m = torch.tensor([3.5], dtype=torch.float64, device=dev)
		dist.all_reduce(m, op=dist.ReduceOp.MAX)
		dist.broadcast(x, src=0)
		dist.barrier()
		torch.cuda.synchronize(dev)
		ok = int(got[0].item()) == 7 and torch.equal(full, x) and torch.equal(a2a.cpu(), torch.arange(64, dtype=torch.uint8)) and float(m.item()) == 3.5
		# and the engine on top of it (one rank: the collectives of set-up and total_rows)
		tabs = catalogues(2, False)
		sm = distributed.ShardedMatch(tabs[0], [tabs[1]], 10., 0.9, device=dev)
		sm.step()
		rows = sm.total_rows()
		# the same exchanges through the library's OWN RCCL calls (nwayhip_comm_*: comm='rccl'), both engines
		comm = distributed.make_comm(dev)
		col = torch.arange(1000, dtype=torch.float64, device=dev) * 0.5
		got_col = comm.allgatherv(col, [1000])
		blocks = torch.arange(4096, dtype=torch.uint8, device=dev)
		back = torch.zeros_like(blocks)
		comm.exchange(blocks, back)
		torch.cuda.synchronize(dev)
		ok = ok and torch.equal(got_col, col) and torch.equal(back, blocks)
		sm2 = distributed.ShardedMatch(tabs[0], [tabs[1]], 10., 0.9, device=dev, comm=comm)
		sm2.step()
		sp = distributed.SecondarySplitMatch(tabs[0], [tabs[1]], 10., 0.9, device=dev, comm=comm)
		for _ in range(3):
			sp.step()
		t_split = sp.local_table()
		t_shard = sm.local_table()
		same = all(np.array_equal(t_split[key], t_shard[key], equal_nan=True) for key in t_shard)
		np.savez(outfile, ok=ok, rows=rows, backend=dist.get_backend(), rows_rccl=sm2.total_rows(), rows_split=sp.total_rows(), same=same)
		comm.close()
	finally:
		dist.destroy_process_group()


def test_rccl_is_there_and_carries_the_engines_collectives(tmp_path):
	"""one rank, backend "nccl" (= RCCL on ROCm): the process group comes up on the GPU box and every kind of
	collective nway_amd.distributed and bench.py issue goes through the library (more than one rank needs more
	than one GPU: the driver's scaling run)"""
	import nway_amd as nw
	outfile = str(tmp_path / 'rccl.npz')
	mp.spawn(rccl_worker, args=(1, free_port(), outfile), nprocs=1, join=True)
	got = np.load(outfile)
	assert bool(got['ok']) and str(got['backend']) == 'nccl'
	want = nw.nway_match(catalogues(2, False), 10., 0.9, logger=nw.NullOutputLogger())
	assert int(got['rows']) == len(want)
	# through nwayhip_comm_* (RCCL behind the C ABI): the same rows from both engines, the split mode's table equal to the sharded one
	assert int(got['rows_rccl']) == len(want) and int(got['rows_split']) == len(want) and bool(got['same'])


@pytest.mark.parametrize('scaling,comm', [('weak', 'torch'), ('strong', 'torch'), ('weak', 'rccl'), ('strong', 'rccl')])
def test_bench_one_rank_through_rccl(scaling, comm):
	"""bench.py's N > 1 code -- engines, barriers, the MAX over ranks -- with ONE rank on the real backend ("nccl" = RCCL):
	NWAY_BENCH_FORCE_DIST=1 (the driver's 2 / 4 / 8-GPU runs take the same lines with more ranks)"""
	env = dict(os.environ, NWAY_BENCH_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_PORT=str(free_port()), NWAY_BENCH_EXTRAS='0')
	cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '5', '--warmup', '2', '--prewarm', '3', '--n-primary', '20000',
		'--n-secondary', '2000000', '--scaling', scaling, '--cpu-sample', '0', '--comm', comm]
	res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600, env=env, cwd=ROOT)
	assert res.returncode == 0, res.stderr[-3000:]
	out = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])
	assert out['n_gpus'] == 1 and out['scaling'] == scaling and 20000 * 1.7 < out['config']['rows_per_step'] < 20000 * 1.9
	assert out['config']['exchanges'].startswith('nwayhip_comm' if comm == 'rccl' else 'torch')
	assert out['config']['parallelism'].startswith('secondary-stream' if scaling == 'strong' else 'primary-row')


EXTRA_JOBS = ['c5_zones', 'c4s_rows', 'c5_rows', 'c3s_split', 'c5_split', 'c3s_zones', 'c4s_zones']  # (bench.py: the two jobs the north star names first)


def check_extra_configs(out, world, comms, scale):
	"""bench.py's extra_configs block: the fixed-size jobs BASELINE names for several GPUs, one record per job and carrier"""
	recs = out['extra_configs']
	assert [(r['job'], r['exchanges'].split(' ')[0]) for r in recs] == [(j, c) for j in EXTRA_JOBS for c in (comms if not j.endswith('zones') else comms[:1])], [
		(r['job'], r['exchanges']) for r in recs]
	for r in recs:
		assert 'error' not in r, r
		assert r['n_gpus'] == world and r['ranks_seen'] == world and r['flags'] == 0 and r['scaling'] == 'strong'
		assert r['ms_per_step'] > 0 and r['value'] > 0 and 0 < r['pass_frac'] < 1 and 0 < r['rank0_pass_frac'] < 1
		n0 = r['sizes'][0]
		assert n0 == max(int({'c3s': 1e5, 'c4s': 1e5, 'c5': 5e5}[r['job'].split('_')[0]] * scale), 8 * world)
		# rows of the WHOLE job: every primary once + its counterparts (80 %; the 3-way job: (1 + 0.8)(1 + 0.6) rows per primary)
		per = 2.88 if r['job'].startswith('c4s') else 1.8
		assert 0.9 * per * n0 < r['rows'] < 1.1 * per * n0 + 50, r
		if r['job'].endswith('split'):
			assert r['mode'].startswith('secondary-stream') and 0 < r['exchange_block_records_used'] <= r['exchange_block_records']
			assert r["exchange_block_records"] <= 3 * r["exchange_block_records_used"] + 96   # (sized by the settling step: twice the fullest block, unless the first guess was within 1.5 x of that)
		elif r['job'].endswith('zones'):
			# every rank streams about 1 / world of every secondary catalogue (+ the seams), and sent its input shard once
			assert r['mode'].startswith('declination zones')
			for zone_n, n in zip(r['rank0_zone_sizes'][1:], r['sizes'][1:]):
				assert 0.5 * n / world <= zone_n <= 1.5 * n / world + 64, r
			assert r['setup_exchange_bytes'] >= 24 * sum(r['sizes'][1:]) // world
		else:
			assert r['mode'].startswith('primary-row') and r['setup_exchange_bytes'] >= 16 * sum(r['sizes'][1:])


def test_bench_extra_configs_two_ranks(tmp_path):
	"""the block the first multi-GPU run of the driver will carry: two gloo ranks on the one GPU,
	catalogue sizes scaled to 2 %"""
	env = dict(os.environ, NWAY_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', NWAY_BENCH_EXTRA_SCALE='0.02')
	cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
		'--master-port', str(free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '2', '--prewarm', '3',
		'--n-primary', '20000', '--n-secondary', '2000000']
	res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=900, env=env, cwd=ROOT)
	assert res.returncode == 0, res.stderr[-3000:]
	line = [l for l in res.stdout.splitlines() if l.startswith('{')][-1]
	out = json.loads(line)
	assert out['n_gpus'] == 2 and out['scaling'] == 'weak'
	check_extra_configs(out, 2, ['torch.distributed'], 0.02)
	keep = os.path.join(ROOT, 'gpurun_out')
	if os.path.isdir(keep):
		with open(os.path.join(keep, 'bench_x2_extras.json'), 'w') as f:
			f.write(line + '\n')


def test_bench_extra_configs_one_rank_both_carriers():
	"""the same block with ONE rank on the real backend: every job through torch.distributed AND through the library's own
	RCCL calls (nwayhip_comm_*)"""
	env = dict(os.environ, NWAY_BENCH_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_PORT=str(free_port()), NWAY_BENCH_EXTRA_SCALE='0.02')
	cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '4', '--warmup', '2', '--prewarm', '3', '--n-primary', '20000',
		'--n-secondary', '2000000', '--cpu-sample', '0']
	res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=900, env=env, cwd=ROOT)
	assert res.returncode == 0, res.stderr[-3000:]
	out = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])
	check_extra_configs(out, 1, ['torch.distributed', 'nwayhip_comm_*'], 0.02)


def test_bench_eight_gloo_ranks_share_the_gpu(tmp_path):
	"""bench.py as the driver will launch it on an 8-GPU node -- torch.distributed.run, eight ranks -- here with the eight ranks on
	the ONE GPU over gloo and sizes scaled down: the weak headline, ranks_seen, one job per sharding mode in extra_configs, and the
	fixed-size summary with the one-GPU references rank 0 measures in the same launch"""
	env = dict(os.environ, NWAY_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', NWAY_BENCH_EXTRA_SCALE='0.02',
		NWAY_BENCH_EXTRA_ONLY='c5_zones,c4s_rows,c3s_split')
	# (round 6: the PLAIN form `python bench.py --gpus 8`, no launcher and no WORLD_SIZE -- bench.py starts the eight ranks itself)
	env.pop('WORLD_SIZE', None)
	cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '4', '--warmup', '2', '--prewarm', '3',
		'--n-primary', '5000', '--n-secondary', '800000']
	res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=1200, env=env, cwd=ROOT)
	assert res.returncode == 0, res.stderr[-3000:]
	line = [l for l in res.stdout.splitlines() if l.startswith('{')][-1]
	out = json.loads(line)
	assert out['n_gpus'] == 8 and out['ranks_seen'] == 8 and out['scaling'] == 'weak' and 'error' not in out and 'supplementary_aborted' not in out
	assert 8 * 5000 * 1.7 < out['config']['rows_per_step'] < 8 * 5000 * 1.9
	recs = out['extra_configs']
	assert [r['job'] for r in recs][:2] == ['c5_zones', 'c4s_rows'], 'the two jobs the north star names for N GPUs are measured first'
	assert sorted(r['job'] for r in recs) == ['c3s_split', 'c4s_rows', 'c5_zones']
	for r in recs:
		assert 'error' not in r and r['ranks_seen'] == 8 and r['n_gpus'] == 8 and r['flags'] == 0 and r['value'] > 0, r
	fixed = out['fixed_size_jobs']
	for name in ('c3s', 'c4s', 'c5'):
		f = fixed[name]
		assert f['n_gpus'] == 8 and f['best']['ranks_seen'] == 8 and f['one_gpu']['value'] > 0 and f['speedup_vs_one_gpu'] > 0, f
		assert abs(f['best']['rows'] - f['one_gpu']['rows']) < 0.05 * f['one_gpu']['rows'], f
	keep = os.path.join(ROOT, 'gpurun_out')
	if os.path.isdir(keep):
		with open(os.path.join(keep, 'bench_x8_gloo.json'), 'w') as f:
			f.write(line + '\n')


def test_bench_headline_survives_hung_extras(tmp_path):
	"""the blocks after the headline (fixed-size jobs in the multi-GPU modes) run collectives no multi-GPU machine has executed yet:
	when they do not finish within --extras-watchdog seconds, rank 0 prints the measured headline with an error record and every rank
	leaves with status 0 (here: two gloo ranks on the one GPU and a watchdog that fires at once)"""
	env = dict(os.environ, NWAY_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', NWAY_BENCH_EXTRA_SCALE='0.02')
	cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
		'--master-port', str(free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '2', '--prewarm', '3',
		'--n-primary', '5000', '--n-secondary', '800000', '--extras-watchdog', '0.05']
	res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600, env=env, cwd=ROOT)
	assert res.returncode == 0, res.stderr[-3000:]
	lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
	assert len(lines) == 1
	out = json.loads(lines[0])
	assert out['n_gpus'] == 2 and out['ranks_seen'] == 2 and out['value'] > 0 and out['ms_per_step'] > 0
	# (either rank 0's own timer, or the broken collective it was in when the other rank left); the abort is flagged at the top level,
	# and whatever records were finished by then stand in front of the error record
	err = out['extra_configs'][-1]['error']
	assert 'watchdog' in err or 'measured before that' in err, err
	assert out['supplementary_aborted'] == err
	assert all('job' in r for r in out['extra_configs'][:-1])  # (records finished -- or broken off by the other rank's leaving -- before the cut)


def mag_worker(rank, world, port, outfile, mode):
	os.environ['MASTER_ADDR'] = '127.0.0.1'
	os.environ['MASTER_PORT'] = str(port)
	os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
	dist.init_process_group('gloo', rank=rank, world_size=world)
	try:
		sys.path.insert(0, ROOT)
		from nway_amd import distributed
		from goldenutil import mag3_tables
		tabs = mag3_tables()
		dev = torch.device('cuda', 0)
		torch.cuda.set_device(dev)
		bounds = [distributed.shard_bounds(len(t['ra']), world) for t in tabs]
		bounds[0] = [0, len(tabs[0]['ra']) // 3, len(tabs[0]['ra'])]  # (uneven shards of the primaries)
		def rows(t, b):
			return dict(t, ra=t['ra'][b[rank]:b[rank + 1]], dec=t['dec'][b[rank]:b[rank + 1]], error=t['error'][b[rank]:b[rank + 1]], mags=[], magnames=[], maghists=[])
		cls = distributed.ShardedMatch if mode == 'rows' else distributed.ZoneShardedMatch
		eng = cls(rows(tabs[0], bounds[0]), [rows(t, b) for t, b in zip(tabs[1:], bounds[1:])], 20., 0.9, dev)
		eng.step()
		mags = [[(n, np.array(v[b[rank]:b[rank + 1]]), h) for n, v, h in zip(t['magnames'], t['mags'], t['maghists'])] for t, b in zip(tabs, bounds)]
		local = eng.magnitude_priors(mags)
		table = eng.gather_magnitude_table(local, dst=0)
		if rank == 0:
			np.savez(outfile, **table)
	finally:
		dist.destroy_process_group()


@pytest.mark.parametrize('mode', ['rows', 'zones'])
def test_sharded_magnitude_priors_on_device(tmp_path, mode):
	"""BASELINE configs[1]'s shape (XMM x OPT x IRAC, three learned magnitude priors) over two ranks sharing the GPU, primary rows
	sharded and declination zones: the HIP pipeline per rank, the selection gathered, nwayhip_bias_lookup + nwayhip_group_stats on
	every rank's own rows -- the gathered table against the REFERENCE's one-process table (tests/golden/mag3.npz)"""
	from goldenutil import golden, assert_table_matches, assert_checksums_match
	outfile = str(tmp_path / 'mag.npz')
	mp.spawn(mag_worker, args=(2, free_port(), outfile, mode), nprocs=2, join=True)
	t = dict(np.load(outfile))
	g = golden('mag3')
	names = ['XMM', 'OPT', 'IRAC']
	assert_checksums_match(t, g, 'm3_', names, rtol=1e-7)
	rows = g['m3_sub_rows']
	assert_table_matches(t, g, 'm3_sub_', names, rows=rows)
	for b in ('bias_OPT_R', 'bias_OPT_I', 'bias_IRAC_CH1'):
		np.testing.assert_allclose(np.asarray(t[b])[rows], g['m3_sub_' + b], rtol=RTOL, err_msg=b)
		np.testing.assert_allclose(np.sum(t[b]), g['m3_sum_' + b][0], rtol=1e-7, err_msg=b)


def local_zone_worker(rank, world, port, outfile, k, flat, zpr, streams):
	os.environ['MASTER_ADDR'] = '127.0.0.1'
	os.environ['MASTER_PORT'] = str(port)
	os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
	dist.init_process_group('gloo', rank=rank, world_size=world)
	try:
		sys.path.insert(0, ROOT)
		from nway_amd import distributed
		tabs = catalogues(k, flat)
		dev = torch.device('cuda', 0)
		torch.cuda.set_device(dev)
		bounds = [distributed.shard_bounds(len(t['ra']), world) for t in tabs]
		parts = [dict(t, ra=t['ra'][b[rank]:b[rank + 1]], dec=t['dec'][b[rank]:b[rank + 1]], error=t['error'][b[rank]:b[rank + 1]]) for t, b in zip(tabs, bounds)]
		zm = distributed.ZoneShardedMatch(parts[0], parts[1:], 10., 0.9, device=dev, zones_per_rank=zpr, streams=streams)
		assert len(zm.zones) == zpr and len(zm.edges) == world * zpr - 1
		for _ in range(3):
			zm.step()
		total = zm.total_rows()
		table = zm.gather_table(dst=0)
		if rank == 0:
			np.savez(outfile, total=total, zone_rows=zm.local_rows(), **table)
		zm.close()
	finally:
		dist.destroy_process_group()


@pytest.mark.parametrize('world,k,flat,zpr,streams', [(1, 2, False, 4, 1), (1, 2, False, 4, 2), (1, 3, True, 3, 3), (2, 3, False, 2, 2), (2, 2, False, 4, 1), (2, 2, True, 3, 1)])
def test_several_zones_per_rank_on_device(tmp_path, world, k, flat, zpr, streams):
	"""several declination zones per rank (ZoneShardedMatch(zones_per_rank=, streams=), round 5): a rank runs its zones one after the other, or
	round robin on several HIP streams -- down to ONE rank whose zones keep each cell table inside the LDS: the table equals the
	single-GPU table of the whole job bit for bit"""
	import nway_amd as nw
	outfile = str(tmp_path / 'zones.npz')
	mp.spawn(local_zone_worker, args=(world, free_port(), outfile, k, flat, zpr, streams), nprocs=world, join=True)
	got = np.load(outfile)
	tabs = catalogues(k, flat)
	want = nw.nway_match(tabs, 10., 0.9, logger=nw.NullOutputLogger())
	assert int(got['total']) == len(want) > len(tabs[0]['ra'])
	for key in want.columns:
		np.testing.assert_array_equal(got[key], want[key].values, err_msg=key)


def test_zones_holding_one_source_or_none():
	"""eight zones over catalogues of a few dozen sources: zones with ONE source of a catalogue (its columns must not be views into the
	packed row: the library wants 16-byte aligned columns -- found by tools/dev/soak_zones.py in round 5) and with none"""
	import nway_amd as nw
	from nway_amd import distributed
	rng = np.random.RandomState(3)
	sky = lambda n: (rng.uniform(0, 360, n), np.degrees(np.arcsin(rng.uniform(-1, 1, n))))
	a = cat('A', *sky(40), rng.uniform(0.5, 2, 40), 41252.96)
	b = cat('B', *sky(13), 0.3 * np.ones(13), 41252.96)
	c = cat('C', *sky(300), 0.5 * np.ones(300), 41252.96)
	b['ra'][:8], b['dec'][:8] = a['ra'][:8] + 1e-4, a['dec'][:8]
	c['ra'][:20], c['dec'][:20] = a['ra'][:20], np.clip(a['dec'][:20] + 2e-4, -90, 90)
	dev = torch.device('cuda', 0)
	want = nw.nway_match([a, b, c], 10., 0.9, logger=nw.NullOutputLogger())
	for zpr, streams in ((8, 1), (8, 3), (5, 2)):
		eng = distributed.ZoneShardedMatch(a, [b, c], 10., 0.9, dev, zones_per_rank=zpr, streams=streams, local_only=True)
		assert min(len(z['secondaries'][0]['ra']) for z in eng.zones) <= 1
		eng.step()
		got = eng.gather_table()
		eng.close()
		assert len(got['ncat']) == len(want) > 40
		for key in want.columns:
			np.testing.assert_array_equal(got[key], want[key].values, err_msg=key)
	# the same two-way (round 6: such zones go out as ONE launch set -- zones without a secondary, with one, tables of 2^10 positions in
	# one slice of the owner-computes registration)
	want2 = nw.nway_match([a, b], 10., 0.9, logger=nw.NullOutputLogger())
	for zpr, registration in ((16, 'atomics'), (16, 'owner'), (5, 'owner')):
		eng = distributed.ZoneShardedMatch(a, [b], 10., 0.9, dev, zones_per_rank=zpr, local_only=True, registration=registration)
		assert len(eng.zones) == zpr and min(len(z['secondaries'][0]['ra']) for z in eng.zones) <= 3  # (13 secondaries, 40 primaries: zones of a few sources)
		for _ in range(3):
			eng.step()
		assert eng.batched and eng.owner_computes == (registration == 'owner')
		got = eng.gather_table()
		eng.close()
		assert len(got['ncat']) == len(want2) > 40
		for key in want2.columns:
			np.testing.assert_array_equal(got[key], want2[key].values, err_msg=key)


@pytest.mark.parametrize('flat,zpr,registration', [(False, 4, 'atomics'), (True, 3, 'atomics'), (False, 8, 'owner'), (True, 3, 'owner'), (False, 5, 'owner')])
def test_zones_of_a_rank_as_one_launch_set(flat, zpr, registration):
	"""round 6 (include/nwayhip.h: nwayhip_zones_*): the zones of a rank go out as ONE registration, ONE sweep and ONE tail launch --
	every zone's table, status words and counters are those of the zones enqueued one after the other, and the gathered table
	is the single-GPU table of the whole job bit for bit; step after step (the two frames of argument blocks alternate with the
	two scratch copies of the plans)"""
	import nway_amd as nw
	from nway_amd import distributed, _hip
	tabs = catalogues(2, flat)
	dev = torch.device('cuda', 0)
	want = nw.nway_match(tabs, 10., 0.9, logger=nw.NullOutputLogger())
	serial = distributed.ZoneShardedMatch(tabs[0], tabs[1:], 10., 0.9, dev, zones_per_rank=zpr, local_only=True, one_launch=False)
	serial.step()
	assert not serial.batched
	st_serial = [np.asarray(serial._zone_status(z)).copy() for z in serial.zones]
	# (registration: every claim of a table position an atomic in memory, or owner-computes -- records, buckets, one workgroup per
	# slice of a table claiming in LDS, csrc/zones.inc; the default takes the second from 200 000 primaries per set on)
	eng = distributed.ZoneShardedMatch(tabs[0], tabs[1:], 10., 0.9, dev, zones_per_rank=zpr, local_only=True, registration=registration)
	for step in range(5):
		eng.step()
		assert eng.batched, 'the zones of a 2-way sparse job qualify for one launch set'
		assert eng.owner_computes == (registration == 'owner')
		for z, st in zip(eng.zones, st_serial):
			got = np.asarray(eng._zone_status(z))
			assert int(got[_hip.ST_FLAGS]) == 0
			# (rows, registrations and links; the survivor counters depend on which of two colliding registrations was displaced)
			for w in (_hip.ST_ROWS, _hip.ST_REGISTRATIONS, _hip.ST_PAIRS):
				assert int(got[w]) == int(st[w]), 'status word %d of a zone, step %d' % (w, step)
		if step in (0, 1, 4):
			got = eng.gather_table()
			assert len(got['ncat']) == len(want) > len(tabs[0]['ra'])
			for key in want.columns:
				np.testing.assert_array_equal(got[key], want[key].values, err_msg='%s, step %d' % (key, step))
	serial.close()
	eng.close()


def test_zone_launch_set_through_the_c_abi():
	"""nwayhip_zones_enqueue on plans of the caller's own: a set of ONE plan, a plan on a workspace it has not seen (its first run
	clears it) and a 3-way plan are enqueued one after the other (nwayhip_zones_batched() == 0) with the same result; a null
	argument buffer likewise; two sparse 2-way plans that have run once go out as one launch set"""
	import nway_amd as nw
	from nway_amd import _hip
	dev = torch.device('cuda', 0)
	tabs = catalogues(2, False)
	half = len(tabs[0]['ra']) // 2
	parts = [dict(tabs[0], ra=tabs[0]['ra'][:half], dec=tabs[0]['dec'][:half], error=tabs[0]['error'][:half]),
		dict(tabs[0], ra=tabs[0]['ra'][half:], dec=tabs[0]['dec'][half:], error=tabs[0]['error'][half:])]
	dens, dens_plus = nw._densities_from_sizes(['A', 'B'], [len(tabs[0]['ra']), len(tabs[1]['ra'])], [tabs[0]['area'], tabs[1]['area']], nw.NullOutputLogger())
	params = _hip.make_params(2, _hip.SCHEME_SPHERE, 10., 10. / 3600, dens, dens_plus, nw._prior_table(dens, dens_plus, nw._completeness_vector(0.9, 2)))
	plans, cats = [], []
	for prim in parts:
		c = [_hip.DeviceCatalogue(t['ra'], t['dec'], t['error'], dev) for t in (prim, tabs[1])]
		plan, st = _hip.run_plan([c[0].n, c[1].n], type(params).from_buffer_copy(params), c, 200000, 200000, dev, lean=True)
		assert int(st[_hip.ST_FLAGS]) == 0
		plans.append(plan)
		cats.append(c)
	rows_alone = [int(p.read_status()[_hip.ST_ROWS]) for p in plans]
	tables_alone = [[_hip.to_host(p.cols[c][:m]) for c in ('log_bf', 'p_any', 'p_i')] + [_hip.to_host(p.cols['idx'][1][:m])] for p, m in zip(plans, rows_alone)]
	batch = _hip.ZoneBatch(plans)
	for step in range(3):
		batch.enqueue(cats)
		assert batch.batched
		for p, m, alone in zip(plans, rows_alone, tables_alone):
			assert int(p.read_status()[_hip.ST_ROWS]) == m and int(p.read_status()[_hip.ST_FLAGS]) == 0
			for got, want in zip([_hip.to_host(p.cols[c][:m]) for c in ('log_bf', 'p_any', 'p_i')] + [_hip.to_host(p.cols['idx'][1][:m])], alone):
				np.testing.assert_array_equal(got, want)
	# an ordinary enqueue of one of the plans in between (its scratch copies change parity against the other's): still one launch set, same tables
	plans[1].enqueue(cats[1])
	batch.enqueue(cats)
	assert batch.batched
	for p, m, alone in zip(plans, rows_alone, tables_alone):
		assert int(p.read_status()[_hip.ST_ROWS]) == m and int(p.read_status()[_hip.ST_FLAGS]) == 0
		np.testing.assert_array_equal(_hip.to_host(p.cols['p_i'][:m]), alone[2])
	batch.close()
	one = _hip.ZoneBatch(plans[:1])
	one.enqueue(cats[:1])
	assert not one.batched and int(plans[0].read_status()[_hip.ST_ROWS]) == rows_alone[0]
	one.close()
	for p in plans:
		p.close()


def _short_runs_worker(outfile):
	"""(a process of its own: the library reads its development switches once)"""
	os.environ['NWAYHIP_DEV'] = '1'
	os.environ['NWAYHIP_ZONES_RUN_ROOM'] = '8'
	sys.path.insert(0, ROOT)
	import nway_amd as nw
	from nway_amd import distributed
	tabs = catalogues(2, False)
	dev = torch.device('cuda', 0)
	eng = distributed.ZoneShardedMatch(tabs[0], tabs[1:], 10., 0.9, dev, zones_per_rank=4, local_only=True, registration='owner')
	for _ in range(3):
		eng.step()
	assert eng.batched and eng.owner_computes
	got = eng.gather_table()
	np.savez(outfile, **got)
	eng.close()


def test_owner_registration_with_runs_too_short(tmp_path):
	"""owner-computes registration (csrc/zones.inc) where a workgroup's run of records has room for 8 of its ~50 per slice: the rest
	takes the old road -- claims as atomics in memory during k_register_pre_zones -- and every owner then starts from the table in
	memory instead of from nothing; the table is the single-GPU table of the whole job bit for bit all the same"""
	import nway_amd as nw
	outfile = str(tmp_path / 'short.npz')
	mp.spawn(_short_runs_worker_entry, args=(outfile,), nprocs=1, join=True)
	got = np.load(outfile)
	tabs = catalogues(2, False)
	want = nw.nway_match(tabs, 10., 0.9, logger=nw.NullOutputLogger())
	assert len(got['ncat']) == len(want) > len(tabs[0]['ra'])
	for key in want.columns:
		np.testing.assert_array_equal(got[key], want[key].values, err_msg=key)


def _short_runs_worker_entry(rank, outfile):
	_short_runs_worker(outfile)
