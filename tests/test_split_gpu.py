"""The secondary-split mode on the device: two ranks (gloo, sharing the one GPU of the test box)
run ONE job -- every rank registers all primaries and sweeps its slice of the secondaries, the
candidates travel through the export / import buffers, the owners finish their rows -- and the
rank-order concatenation must equal the single-GPU table of the same catalogues."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from goldenutil import ROOT, cat

pytestmark = pytest.mark.gpu


def free_port():
	s = socket.socket()
	s.bind(('127.0.0.1', 0))
	port = s.getsockname()[1]
	s.close()
	return port


def catalogues(k, flat):
	rng = np.random.RandomState(77)
	n0, n1, n2 = 30000, 400000, 250000
	if flat:
		sky = lambda n: (rng.uniform(100.0, 130.0, n), rng.uniform(-20.0, 20.0, n))
		area = 30.0 * 40.0
	else:
		sky = lambda n: (rng.uniform(0, 360, n), np.degrees(np.arcsin(rng.uniform(-1, 1, n))))
		area = 41252.96
	a = cat('A', *sky(n0), rng.uniform(0.5, 2, n0), area)
	b = cat('B', *sky(n1), rng.uniform(0.2, 0.4, n1), area)
	c = cat('C', *sky(n2), 0.5 * np.ones(n2), area)
	for t, m in ((b, 20000), (c, 15000)):
		t['ra'][:m] = a['ra'][:m] + rng.normal(0, 1, m) / 3600.
		t['dec'][:m] = np.clip(a['dec'][:m] + rng.normal(0, 1, m) / 3600., -90, 90)
	# one secondary that is a candidate of two primaries owned by different ranks
	a['ra'][29999], a['dec'][29999] = a['ra'][3], a['dec'][3] + 2.0 / 3600.
	return [a, b, c][:k]


def worker(rank, world, port, outfile, k, flat, capacity=None, tuning=None):
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
		pb = [0, 11000, len(tabs[0]['ra'])]
		def rows(t, lo, hi):
			return dict(t, ra=t['ra'][lo:hi], dec=t['dec'][lo:hi], error=t['error'][lo:hi])
		secs = []
		for c in range(1, k):
			n = len(tabs[c]['ra'])
			cut = [0, int(0.37 * n), n]
			secs.append(rows(tabs[c], cut[rank], cut[rank + 1]))
		sm = distributed.SecondarySplitMatch(rows(tabs[0], pb[rank], pb[rank + 1]), secs, 10., 0.9, device=dev, capacity=capacity, tuning=tuning)
		if capacity is not None:
			assert sm.capacity > capacity  # the export blocks overflowed and were enlarged, on every rank alike
		for _ in range(3):  # (repeated steps: the export headers and the scratch copies are recycled)
			sm.step()
		total = sm.total_rows()
		table = sm.gather_table(dst=0)
		if rank == 0:
			np.savez(outfile, total=total, **table)
	finally:
		dist.destroy_process_group()


@pytest.mark.parametrize('k,flat', [(2, False), (3, False), (2, True)])
def test_secondary_split_on_device(tmp_path, k, flat):
	import nway_amd as nw
	outfile = str(tmp_path / 'split.npz')
	mp.spawn(worker, args=(2, free_port(), outfile, k, flat), nprocs=2, join=True)
	got = np.load(outfile)
	tabs = catalogues(k, flat)
	want = nw.nway_match(tabs, 10., 0.9, logger=nw.NullOutputLogger())
	assert int(got['total']) == len(want) > 30000
	for key in want.columns:
		np.testing.assert_array_equal(got[key], want[key].values, err_msg=key)


def test_secondary_split_grows_its_export_blocks(tmp_path):
	"""export blocks that are too small for the candidates of one peer are flagged by the receiver
	(NWAYHIP_FLAG_PAIR_OVERFLOW) and the engine comes back with larger ones on every rank"""
	import nway_amd as nw
	outfile = str(tmp_path / 'split.npz')
	mp.spawn(worker, args=(2, free_port(), outfile, 2, False, 64), nprocs=2, join=True)
	got = np.load(outfile)
	tabs = catalogues(2, False)
	want = nw.nway_match(tabs, 10., 0.9, logger=nw.NullOutputLogger())
	assert int(got['total']) == len(want)
	for key in want.columns:
		np.testing.assert_array_equal(got[key], want[key].values, err_msg=key)


@pytest.mark.parametrize('k,flat', [(2, False), (3, True)])
def test_secondary_split_with_a_table_beyond_the_lds(tmp_path, k, flat):
	"""the large-table sweep (sweepbig.inc) in split mode: survivors flushed by their waves, routed
	and exported by the workgroup at the end"""
	import nway_amd as nw
	outfile = str(tmp_path / 'split.npz')
	mp.spawn(worker, args=(2, free_port(), outfile, k, flat, None, dict(direct_log2=21)), nprocs=2, join=True)
	got = np.load(outfile)
	tabs = catalogues(k, flat)
	want = nw.nway_match(tabs, 10., 0.9, logger=nw.NullOutputLogger())
	assert int(got['total']) == len(want) > 30000
	for key in want.columns:
		np.testing.assert_array_equal(got[key], want[key].values, err_msg=key)
