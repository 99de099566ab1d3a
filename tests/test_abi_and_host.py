"""CPU-side tests: the C-ABI library loads and exports every symbol include/nwayhip.h
declares (no compute without a GPU), and the host logic around it."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

from goldenutil import ROOT, golden, cat

sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import nway_oracle as orc  # noqa: E402


@pytest.fixture(scope='module')
def built():
	from nway_amd import build
	return build.build_library()


def header_symbols():
	text = open(os.path.join(ROOT, 'include', 'nwayhip.h')).read()
	text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
	return sorted(set(re.findall(r'\b(nwayhip_[a-z_0-9]+)\s*\(', text)))


def test_library_exports_every_declared_symbol(built):
	from nway_amd import _hip
	lib = ctypes.CDLL(built)
	declared = header_symbols()
	assert len(declared) >= 12
	for name in declared:
		assert hasattr(lib, name), 'libnwayhip.so does not export %s' % name
	assert sorted(_hip.SYMBOLS) == declared, 'ctypes binding and header disagree'
	bound = _hip.load()
	assert bound.nwayhip_version() == _hip.ABI_VERSION == 3
	assert isinstance(bound.nwayhip_last_error(), bytes)


def test_struct_layout_matches_header(built):
	from nway_amd import _hip
	# sizes implied by include/nwayhip.h with natural alignment
	assert ctypes.sizeof(_hip.Catalogue) == 40
	# ..., sphere_cell_factor, bitmap_bits, table_slots, link_region_min, f32_roundtrip; direct_log2, fold_log2, disable, reserved
	assert ctypes.sizeof(_hip.MatchParams) == 6 * 4 + 3 * 8 + 8 * 8 * 2 + 128 * 8 + 5 * 8 + 4 * 4
	assert ctypes.sizeof(_hip.Table) == 8 + 8 * 8 + 28 * 8 + 11 * 8


def test_plan_rejects_bad_arguments(built):
	from nway_amd import _hip
	lib = _hip.load()
	p = _hip.make_params(2, _hip.SCHEME_SPHERE, 5.0, 5.0 / 3600, [1., 1.], [1., 1.], [1., 1.], radius_filter=False)
	handle = ctypes.c_void_p(0)
	n = (ctypes.c_int64 * 2)(10, 10)
	assert lib.nwayhip_plan_create(ctypes.byref(handle), ctypes.byref(p), n, 100, 100) != 0
	assert b'all-sky' in lib.nwayhip_last_error()
	p = _hip.make_params(2, _hip.SCHEME_FLAT, 5.0, 5.0 / 3600, [1., 1.], [1., 1.], [1., 1.])
	assert lib.nwayhip_plan_create(ctypes.byref(handle), ctypes.byref(p), n, 100, 100) == 0
	assert lib.nwayhip_plan_workspace_bytes(handle) > 0
	assert lib.nwayhip_plan_destroy(handle) == 0
	n = (ctypes.c_int64 * 2)(0, 10)
	assert lib.nwayhip_plan_create(ctypes.byref(handle), ctypes.byref(p), n, 100, 100) != 0


def test_zone_launch_set_rejects_bad_arguments(built):
	"""nwayhip_zones_create (round 6): plan count, null plans, zones of different jobs; no device is touched before an enqueue"""
	from nway_amd import _hip
	lib = _hip.load()
	assert ctypes.sizeof(_hip.ZoneRun) == 5 * 8
	n = (ctypes.c_int64 * 2)(10, 10)
	handles = []
	for scheme in (_hip.SCHEME_FLAT, _hip.SCHEME_FLAT, _hip.SCHEME_SPHERE):
		h = ctypes.c_void_p(0)
		p = _hip.make_params(2, scheme, 5.0, 5.0 / 3600, [1., 1.], [1., 1.], [1., 1.])
		assert lib.nwayhip_plan_create(ctypes.byref(h), ctypes.byref(p), n, 100, 100) == 0
		handles.append(h)
	zs = ctypes.c_void_p(0)
	arr = (ctypes.c_void_p * 2)(handles[0], handles[1])
	assert lib.nwayhip_zones_create(ctypes.byref(zs), arr, 0) != 0 and b'plans' in lib.nwayhip_last_error()
	assert lib.nwayhip_zones_create(ctypes.byref(zs), arr, _hip.MAXZONES + 1) != 0
	assert lib.nwayhip_zones_create(ctypes.byref(zs), (ctypes.c_void_p * 2)(handles[0], None), 2) != 0 and b'null' in lib.nwayhip_last_error()
	assert lib.nwayhip_zones_create(ctypes.byref(zs), (ctypes.c_void_p * 2)(handles[0], handles[2]), 2) != 0 and b'scheme' in lib.nwayhip_last_error()
	assert lib.nwayhip_zones_create(ctypes.byref(zs), arr, 2) == 0
	assert lib.nwayhip_zones_args_bytes(zs) > 0 and lib.nwayhip_zones_args_bytes(zs) % 512 == 0
	assert lib.nwayhip_zones_batched(zs) == 0
	assert lib.nwayhip_zones_enqueue(zs, None, None, 0, None) != 0
	assert lib.nwayhip_zones_destroy(zs) == 0
	for h in handles:
		assert lib.nwayhip_plan_destroy(h) == 0


def test_no_cpu_fallback_without_gpu(built):
	import torch
	if torch.cuda.is_available():
		pytest.skip('a GPU is present')
	import nway_amd
	with pytest.raises(nway_amd.NwayHipError):
		nway_amd.match.dist((1., 2.), (3., 4.))
	with pytest.raises(nway_amd.NwayHipError):
		nway_amd.bayesdist.log_bf([[None, 0.3]], [0.1, 0.2])
	t = cat('A', [10.], [10.], [1.], 1.0)
	with pytest.raises(nway_amd.NwayHipError):
		nway_amd.nway_match([t, t], 5., 0.9, logger=nway_amd.NullOutputLogger())


def test_scheme_decision_matches_oracle():
	import nway_amd
	rng = np.random.RandomState(0)
	ok = (rng.uniform(10, 20, 50), rng.uniform(-40, 40, 50))
	for tables, err in (([ok, ok], 0.01), ([ok, (np.r_[ok[0], 0.05], np.r_[ok[1], 0.])], 0.01),
			([ok, (ok[0], ok[1] + 10)], 0.01), ([ok], 1.5), ([(ok[0] + 340.95, ok[1])], 0.01)):
		assert nway_amd.choose_scheme(tables, err) == orc.choose_scheme(tables, err)


def test_prior_table_and_densities_match_oracle():
	import nway_amd
	log = nway_amd.NullOutputLogger()
	tabs = [cat('A', np.zeros(120), np.zeros(120), np.ones(120), 0.01), cat('B', np.zeros(5000), np.zeros(5000), np.ones(5000), 0.02),
		cat('C', np.zeros(77), np.zeros(77), np.ones(77), 0.5)]
	dens, dens_plus = nway_amd._compute_source_densities(tabs, log)
	od, odp = orc.source_densities(tabs)
	np.testing.assert_array_equal(dens, od)
	np.testing.assert_array_equal(dens_plus, odp)
	comp = nway_amd._completeness_vector(0.81, 3)
	np.testing.assert_array_equal(comp, orc.completeness_vector(0.81, 3))
	table = nway_amd._prior_table(dens, dens_plus, comp)
	# presence patterns via the oracle's case loop: separations[0][i] NaN <=> absent
	for pattern in range(4):
		nan = np.array([np.nan])
		sep = [[nan, np.array([1.0 if pattern & 1 else np.nan]), np.array([1.0 if pattern & 2 else np.nan])],
			[nan, nan, np.array([1.0 if pattern == 3 else np.nan])], [nan, nan, nan]]
		prior, _ = orc.single_log_bf(3, od, odp, 1, sep, [np.ones(1)] * 3, 0.81)
		assert prior[0] == table[pattern]
	with pytest.raises(Exception):
		nway_amd._completeness_vector(np.array([1.0, 0.5]), 3)


def test_fits_roundtrip(tmp_path):
	from nway_amd import _fits
	rng = np.random.RandomState(1)
	n = 1234
	cols = [('ID', 'J', np.arange(n)), ('RA', 'D', rng.uniform(0, 360, n)), ('DEC', 'D', rng.uniform(-90, 90, n)),
		('pos_err', 'E', rng.uniform(0.1, 2, n)), ('flag', 'I', rng.randint(0, 3, n))]
	path = str(tmp_path / 'cat.fits')
	_fits.write_table(path, cols, 'TESTCAT', table_header={'SKYAREA': 2.0})
	t = _fits.read_table(path)
	assert t.name == 'TESTCAT' and t.header['SKYAREA'] == 2.0 and t.formats == ['J', 'D', 'D', 'E', 'I']
	np.testing.assert_array_equal(t.data['ID'], cols[0][2])
	np.testing.assert_array_equal(t.data['RA'], cols[1][2])
	np.testing.assert_array_equal(t.data['pos_err'], cols[3][2].astype(np.float32))
	assert os.path.getsize(path) % 2880 == 0


def test_xmm_fixture_is_the_shipped_catalogue():
	g = golden('xmm_inputs')
	assert len(g['RA']) == 1797 and g['pos_err'].dtype == np.float32 and float(g['area'][0]) == 2.0


def test_bench_live_traffic_degrades_gracefully(monkeypatch, tmp_path):
	"""bench.py measures roofline.traffic itself (child runs under rocprofv3 --pmc); where that cannot work -- no profiler, no GPU, a
	counter pass that fails -- it says why and the caller falls back to the recorded figure: never an exception, never a guess"""
	import bench
	# (a) no rocprofv3 at all
	monkeypatch.setattr('shutil.which', lambda name: None)
	monkeypatch.setattr('os.path.exists', lambda p, _real=os.path.exists: False if p == '/opt/rocm/bin/rocprofv3' else _real(p))
	traffic, why, detail = bench.live_traffic([], 1000)
	assert traffic is None and 'not found' in why
	monkeypatch.undo()
	# (b) a profiler that runs and fails (stand-in script): the reason is reported
	fake = tmp_path / 'rocprofv3'
	fake.write_text('#!/bin/sh\necho no device >&2\nexit 7\n')
	fake.chmod(0o755)
	monkeypatch.setattr('shutil.which', lambda name: str(fake))
	traffic, why, detail = bench.live_traffic([], 1000, budget_s=30.0)
	assert traffic is None and 'rc 7' in why
	# (c) a profiler whose passes report 100 000 KiB fetched and 5 000 KiB written per k_sweep dispatch: the guide's correction
	fake.write_text('#!/bin/sh\nwhile [ "$1" != "--pmc" ]; do shift; done; c=$2; while [ "$1" != "-d" ]; do shift; done; d=$2\n'
		'mkdir -p $d/host; v=100000; [ "$c" = WRITE_SIZE ] && v=5000\n'
		"printf '\"Kernel_Name\",\"Counter_Name\",\"Counter_Value\"\\n\"void k_sweep<1, true, true>(SweepArgs)\",\"%s\",%s\\n\"k_tail2(Tail2Args)\",\"%s\",1\\n' $c $v $c > $d/host/1_counter_collection.csv\n")
	traffic, why, detail = bench.live_traffic([], 10000000, budget_s=30.0)
	assert traffic == 100000 * 1024 + 80e6 + 5000 * 1024 and why.startswith('counters measured in this run'), why
	# ... and the two parts apart: what the counters said, what the model adds
	assert detail['counters_raw_bytes'] == 100000 * 1024 + 5000 * 1024 and detail['correction_model_bytes'] == 80e6 and 'MODEL' in detail['correction_note']
	# (d) a pass that hangs: killed with its process group when the budget is spent
	fake.write_text('#!/bin/sh\nsleep 300 &\nwait\n')
	import time
	t0 = time.time()
	traffic, why, detail = bench.live_traffic([], 1000, budget_s=22.0)
	assert traffic is None and 'did not finish' in why and time.time() - t0 < 40
