#!/usr/bin/env python
"""bench.py -- throughput of the nway hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

Workload (BASELINE.json configs[2], the "HBM-roofline config"; configs[1] needs the two
COSMOS catalogues that are missing from the reference checkout): synthetic 2-way match,
1e5 primaries x 1e7 secondaries uniform on the sphere, 5 arcsec radius, 80 % of the
primaries with a true counterpart ("C3-S").

A step = one pass of the whole hot path over the resident catalogues: primary cell
registration, secondary sweep, separations, Bayes factors / priors / posteriors, per-primary
p_any / p_i / match_flag.  Inputs are resident in HBM when the timed region starts; nothing
inside the region synchronises with the host.  The steps alternate over --sec-buffers distinct
copies of the secondary catalogue (3 x 160 MB by default) so that no step finds its stream in
the 256 MiB Infinity Cache.

metric: candidate Bayes-factor evaluations per second = rows of the match table produced
per second (one row = one match hypothesis incl. the no-counterpart rows), whole job.

--scaling weak (default): every rank owns --n-primary primaries and matches them against the
whole secondary catalogue (all-gathered once at set-up); strong: the job is fixed at
--n-primary x --n-secondary, every rank sweeps a 1/N slice of the secondaries against all the
primaries and the candidates are routed to the primaries' owners (nway_amd.distributed).

Prints ONE JSON line on rank 0 (contract in the task description) with these extra objects:
  roofline     dominant kernel (the secondary sweep, HBM bound): algorithmic bytes per launch
               (16 B per secondary: ra + dec read once) / mean launch duration measured with HIP
               events on the pipeline's stream on every 8th launch of the timed region; plus
               the WHOLE pass by SURVEY 8(d): pass_bytes / ms_per_step / peak = pass_frac.
  cpu_baseline the C restatement of the oracle (oracle/nway_oracle.c, "port") timed on this host:
               all cores (OpenMP build), one core, and the numpy oracle on one thread.
  check        the table of the last timed step against the CPU table of the same workload.
  io           host -> device upload of the inputs and device -> host download of the table; e2e_ms:
               host arrays in -> table on the host, everything included (upload + pass + download).
roofline also carries two ceilings measured on this box in this run: read_peak (the sweep's own
access pattern with nothing behind the loads, nwayhip_read_probe) and copy_peak (device-to-device
copy of one column, bytes read + written).
"""
from __future__ import division, print_function

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
	sys.path.insert(0, ROOT)

SKY_AREA = 4 * np.pi * (180 / np.pi)**2
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def uniform_sphere(rng, n):
	ra = rng.uniform(0.0, 360.0, size=n)
	dec = np.degrees(np.arcsin(rng.uniform(-1.0, 1.0, size=n)))
	return ra, dec


def make_workload(n_primary, n_secondary, seed, true_fraction=0.8, sigma_secondary=0.1):
	"""C3-S of SURVEY.md 8d: uniform-sky catalogues; the first ``true_fraction`` of the
	primaries get a counterpart in the secondary catalogue, offset by N(0, sigma_p) per axis."""
	rng = np.random.default_rng(seed)
	pra, pdec = uniform_sphere(rng, n_primary)
	psig = rng.uniform(0.3, 1.5, size=n_primary)
	sra, sdec = uniform_sphere(rng, n_secondary)
	ntrue = min(int(true_fraction * n_primary), n_secondary)
	slots = rng.choice(n_secondary, size=ntrue, replace=False)
	ddec = rng.normal(0.0, 1.0, size=ntrue) * psig[:ntrue] / 3600.0
	dra = rng.normal(0.0, 1.0, size=ntrue) * psig[:ntrue] / 3600.0 / np.maximum(np.cos(np.radians(pdec[:ntrue])), 1e-6)
	sdec[slots] = np.clip(pdec[:ntrue] + ddec, -90.0, 90.0)
	sra[slots] = (pra[:ntrue] + dra) % 360.0
	primary = dict(name='PRIM', ra=pra, dec=pdec, error=psig, area=SKY_AREA, mags=[], maghists=[], magnames=[])
	secondary = dict(name='SEC', ra=sra, dec=sdec, error=sigma_secondary, area=SKY_AREA, mags=[], maghists=[], magnames=[])
	return primary, secondary


def kernel_source_hash():
	"""sha256 over the kernel sources, as tools/summarize_profile.py stamps it into profiles/sweep_traffic.json"""
	import glob
	import hashlib
	h = hashlib.sha256()
	files = sorted(glob.glob(os.path.join(ROOT, 'nway_amd', 'csrc', '*.inc')) + glob.glob(os.path.join(ROOT, 'nway_amd', 'csrc', '*.hip'))) + [
		os.path.join(ROOT, 'include', 'nwayhip.h')]
	for f in files:
		h.update(os.path.basename(f).encode())
		h.update(open(f, 'rb').read())
	return h.hexdigest()[:16]


def pass_bytes(tables, rows):
	"""algorithmic bytes of one pass, SURVEY.md 8(d): every input column read once (ra, dec, and the
	positional error where it is a column) + every output column written once (k = 2: 66 B per row,
	k = 3: 94 B per row)"""
	k = len(tables)
	b = 0.0
	for t in tables:
		b += len(t['ra']) * (16.0 + (8.0 if np.ndim(t['error']) > 0 else 0.0))
	per_row = 4 * k + 8 * (k * (k - 1) // 2) + 8 + 1 + 8 * 5 + 1
	return b + per_row * rows


def cpu_baseline(primary, secondary, radius, completeness, numpy_sample):
	"""the CPU legs of SURVEY 8(d): the C port on all the cores this process may use (OpenMP
	build of the same source) and on one core, whole workload; the numpy oracle (the like-for-like
	stand-in of the reference's own numpy path) on one thread, on a bounded sample"""
	sys.path.insert(0, os.path.join(ROOT, 'oracle'))
	import nway_oracle_c
	n = len(secondary['ra'])
	sec = dict(secondary, error=secondary['error'] * np.ones(n))
	cores = nway_oracle_c.host_cores()
	legs = {}
	table = None
	for name, threads in (('all_cores', 0), ('one_core', 1)):
		if threads == 0 and cores == 1:
			continue
		dt = None
		for rep in range(2 if threads == 0 else 1):  # (all cores: the better of two -- the first call also starts the thread pool)
			t0 = time.perf_counter()
			table = nway_oracle_c.nway_match([primary, sec], radius, completeness, threads=threads)
			dt = min(dt, time.perf_counter() - t0) if dt is not None else time.perf_counter() - t0
		legs[name] = dict(value=len(table['ncat']) / dt, cores=(cores if threads == 0 else 1), seconds=dt, rows=len(table['ncat']))
	import nway_oracle
	m = min(numpy_sample, n)
	t0 = time.perf_counter()
	tn = nway_oracle.nway_match([primary, dict(sec, ra=sec['ra'][:m], dec=sec['dec'][:m], error=sec['error'][:m])], radius, completeness)
	dt = time.perf_counter() - t0
	legs['numpy_one_thread'] = dict(value=len(tn['ncat']) / dt, cores=1, seconds=dt, rows=len(tn['ncat']),
		sample='all %d primaries x first %d secondaries' % (len(primary['ra']), m))
	best = legs.get('all_cores', legs['one_core'])
	out = dict(value=best['value'], unit='candidate evaluations/s', cores=best['cores'], kind='port',
		sample='oracle/nway_oracle.c built with -fopenmp, %d threads: the whole workload, %d primaries x %d secondaries, '
			'%d rows in %.3f s (the better of two calls) -- a declination-sorted sweep: parallel sample sort of the secondaries (splitters, per-thread bucket '
			'counts, scatter, independent bucket sorts), then every thread a contiguous range of primaries (binary search + exact test of the band); '
			'the one-core leg beside it runs the same code on one thread' % (
			best['cores'], len(primary['ra']), n, best['rows'], best['seconds']),
		speedup_all_cores_vs_one=(legs['all_cores']['value'] / legs['one_core']['value'] if 'all_cores' in legs else None),
		cores_note='cores = what this process can use: %d logical CPUs visible (os.sched_getaffinity), cgroup CPU quota %s' % (
			nway_oracle_c.visible_cores(), ('%.1f CPUs' % nway_oracle_c.cpu_quota()) if nway_oracle_c.cpu_quota() is not None else 'none'),
		one_core=legs['one_core'], numpy_one_thread=legs['numpy_one_thread'],
		reference_note='the reference itself (pure Python, one thread; it cannot travel to the GPU box) measured in the build container '
			'on its own fixtures: 8.4e4 rows/s (tests/elltest 2-way, 37 706 rows in 0.45 s), 5.3e4 rows/s (3-way, 450 435 rows in 8.5 s)')
	return out, table


def measured_ceilings(sec_copies, device, reps=30):
	"""what this box's memory system gives, measured in this run ('also measure an on-box copy
	kernel'): the sweep's stream with nothing behind the loads (read only, alternating over the copies of
	the secondary catalogue so that the Infinity Cache does not serve it), and a device-to-device copy"""
	import torch
	from nway_amd import _hip
	lib = _hip.load()
	n = sec_copies[0].n
	out = torch.zeros(1024, dtype=torch.float64, device=device)
	stream = _hip.current_stream_ptr(device)

	def probe(i):
		c = sec_copies[i % len(sec_copies)]
		_hip.check(lib.nwayhip_read_probe(_hip.ptr(c.ra), _hip.ptr(c.dec), n, _hip.ptr(out), 256, stream))
	for i in range(5):
		probe(i)
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	for i in range(reps):
		probe(i)
	e1.record()
	e1.synchronize()
	read_gbs = 16.0 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
	dst = torch.empty_like(sec_copies[0].ra)
	for i in range(3):
		dst.copy_(sec_copies[i % len(sec_copies)].ra)
	e0.record()
	for i in range(reps):
		dst.copy_((sec_copies[i % len(sec_copies)].ra, sec_copies[i % len(sec_copies)].dec)[i & 1])
	e1.record()
	e1.synchronize()
	copy_gbs = 16.0 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
	return read_gbs, copy_gbs


def end_to_end(tables, radius, completeness, device, reps=3):
	"""host arrays in -> match table on the host: upload, one pass, download of every column; the best of
	`reps` (the first carries the allocations)"""
	import torch
	import nway_amd
	best = None
	for _ in range(reps):
		torch.cuda.synchronize(device)
		t0 = time.perf_counter()
		res = nway_amd.run_match(tables, radius, completeness, device=device, lean=True)
		t1 = time.perf_counter()
		m = res.nrows
		cols = [res.to_host('idx', c) for c in range(len(tables))] + [res.to_host('sep', 0)]
		cols += [res.to_host(nme) for nme in ('sep_max', 'log_bf', 'dist_post', 'p_single', 'p_any', 'p_i', 'ncat', 'match_flag')]
		t2 = time.perf_counter()
		res.plan.close()
		rec = dict(e2e_ms=(t2 - t0) * 1e3, upload_and_pass_ms=(t1 - t0) * 1e3, download_ms=(t2 - t1) * 1e3, rows=int(m), attempts=int(res.plan.attempts))
		if best is None or rec['e2e_ms'] < best['e2e_ms']:
			best = rec
	return best


def table_check(plan, names, cpu_table):
	"""the device table of the last timed step against the CPU table of the same workload: index
	columns, ncat and match_flag must be identical, the floating columns within 1e-6 relative"""
	from nway_amd import _hip
	m = int(plan.read_status()[_hip.ST_ROWS])
	out = dict(rows_gpu=m, rows_cpu=len(cpu_table['ncat']))
	if m != out['rows_cpu']:
		out['ok'] = False
		return out
	ok = True
	for c, nme in enumerate(names):
		same = bool(np.array_equal(_hip.to_host(plan.cols['idx'][c][:m]).astype(np.int64), cpu_table[nme]))
		out['idx_%s_equal' % nme] = same
		ok &= same
	flag = _hip.to_host(plan.cols['match_flag'][:m]).astype(np.int64)
	out['match_flag_equal'] = bool(np.array_equal(flag, cpu_table['match_flag']))
	out['match_flag_histogram'] = [int(x) for x in np.bincount(flag, minlength=3)]
	ok &= out['match_flag_equal']
	worst = 0.0
	for src, dst in (('log_bf', 'dist_bayesfactor'), ('dist_post', 'dist_post'), ('p_single', 'p_single'), ('p_any', 'prob_has_match'), ('p_i', 'prob_this_match')):
		got, want = _hip.to_host(plan.cols[src][:m]), cpu_table[dst]
		rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-12)
		rel = np.where(np.abs(got - want) <= 1e-12, 0.0, rel)
		worst = max(worst, float(np.nanmax(rel)) if m else 0.0)
	out['max_relative_difference'] = worst
	ok &= worst <= 1e-6
	out['ok'] = bool(ok)
	return out


def make_workload3(n0, n1, n2, seed):
	"""C4-S of SURVEY.md 8d (BASELINE configs[3]): uniform-sky 3-way; 80 % of the primaries have a counterpart in the first
	secondary catalogue (sigma 0.1), 60 % in the second (0.5); the primaries' error is 1 arcsec"""
	rng = np.random.default_rng(seed)
	pra, pdec = uniform_sphere(rng, n0)
	psig = np.ones(n0)
	out = [dict(name='PRIM', ra=pra, dec=pdec, error=psig, area=SKY_AREA, mags=[], maghists=[], magnames=[])]
	for name, n, sigma, frac in (('A', n1, 0.1, 0.8), ('B', n2, 0.5, 0.6)):
		ra, dec = uniform_sphere(rng, n)
		m = min(int(frac * n0), n)
		slots = rng.choice(n, size=m, replace=False)
		dec[slots] = np.clip(pdec[:m] + rng.normal(0, 1, size=m) * psig[:m] / 3600., -90, 90)
		ra[slots] = (pra[:m] + rng.normal(0, 1, size=m) * psig[:m] / 3600. / np.maximum(np.cos(np.radians(pdec[:m])), 1e-6)) % 360
		out.append(dict(name=name, ra=ra, dec=dec, error=sigma, area=SKY_AREA, mags=[], maghists=[], magnames=[]))
	return out


def live_traffic(argv_tail, n_secondary, budget_s=150.0):
	"""HBM bytes per launch of the sweep, MEASURED in this run: two child processes of this very script (a few steps, no CPU legs) under
	`rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes: the two TCC counters do not fit one, and --pmc is never combined
	with other traces), corrected as MI355X_MICROARCH.md prescribes -- KiB * 1024; gfx950 reports exactly half of a wide (16 B per lane)
	coalesced streaming read, so the stream's uncounted half is added back (the kernel's gathers are counted in full).  Returns (bytes,
	source) or (None, why not)."""
	import csv
	import glob
	import shutil
	import subprocess
	import tempfile
	prof = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
	if prof is None:
		return None, 'rocprofv3 not found', None
	out = tempfile.mkdtemp(prefix='nway_bench_pmc_')
	t0 = time.perf_counter()
	vals = {}
	try:
		for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
			left = budget_s - (time.perf_counter() - t0)
			if left < 20:
				return None, 'time budget of the counter passes spent', None
			cmd = [prof, '--pmc', counter, '--output-format', 'csv', '-d', os.path.join(out, counter), '--', sys.executable, os.path.abspath(__file__),
				'--steps', '6', '--warmup', '2', '--prewarm', '20', '--cpu-sample', '0', '--two-pipelines', '0', '--live-traffic', '0'] + argv_tail
			env = dict(os.environ, TMPDIR='/tmp')
			# (its own process group: a pass that overruns is killed with everything it started, nothing else)
			proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, cwd='/tmp', env=env, start_new_session=True)
			try:
				stdout, _ = proc.communicate(timeout=left)
			except subprocess.TimeoutExpired:
				import signal
				try:
					os.killpg(proc.pid, signal.SIGKILL)
				except OSError:
					pass
				proc.communicate()
				return None, 'rocprofv3 --pmc %s did not finish in %.0f s' % (counter, left), None
			files = glob.glob(os.path.join(out, counter, '*', '*counter_collection.csv'))
			if proc.returncode != 0 or not files:
				return None, 'rocprofv3 --pmc %s failed (rc %d): %s' % (counter, proc.returncode, stdout[-200:].replace('\n', ' ')), None
			per = []
			for r in csv.DictReader(open(files[0])):
				if 'k_sweep' in r['Kernel_Name'] and r['Counter_Name'] == counter:
					per.append(float(r['Counter_Value']))
			if not per:
				return None, 'no k_sweep dispatch in the %s pass' % counter, None
			vals[counter] = sum(per) / len(per)
	except Exception as e:
		return None, '%s: %s' % (type(e).__name__, e), None
	finally:
		shutil.rmtree(out, ignore_errors=True)
	raw = vals['FETCH_SIZE'] * 1024
	stream = 16.0 * n_secondary
	fetch = raw + stream / 2 if raw >= stream / 2 else 2 * raw
	detail = dict(fetch_raw_bytes=raw, write_raw_bytes=vals['WRITE_SIZE'] * 1024, counters_raw_bytes=raw + vals['WRITE_SIZE'] * 1024,
		correction_model_bytes=fetch - raw,
		correction_note='MODEL, not a counter: gfx950 reports half of a wide (16 B per lane) coalesced streaming read (MI355X_MICROARCH.md, HBM/rocprofv3 '
			'section); the uncounted half of the 16 B x n_secondary stream is added to the raw FETCH_SIZE (the gathers are counted in full)')
	return fetch + vals['WRITE_SIZE'] * 1024, ('counters measured in this run (child processes of this script under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, '
		'separate passes, mean per k_sweep dispatch: FETCH %.0f KiB raw, WRITE %.0f KiB) + the guide\'s gfx950 correction, a model (traffic_detail); %.0f s'
		% (vals['FETCH_SIZE'], vals['WRITE_SIZE'], time.perf_counter() - t0)), detail


# FP64 vector peak of the part: 256 CUs x 4 SIMDs x 16 FP64 lanes per clock x 2 flops (FMA) x 2.4 GHz (half the FP32 vector rate of
# MI355X_MICROARCH.md, 157.3 TFLOP/s spec; SURVEY.md 8d: "FP64 vector ALU, ~79 TFLOP/s nominal")
FP64_PEAK_TFLOPS = 78.6
# FP64 flops (an FMA counts two) of the elementary functions as csrc/fastmath.inc evaluates them on their short roads, counted off the
# source: sincos 49, x/180*pi 6, atan2 25 and hypot 14 (a division and a square root at ~10 each), log 46, log10 56, 10^x 34
FLOPS_TEST = 135.0   # one distance test: the trig-free bound (~15), radians of both longitudes, ONE sincos (the longitude difference), the
                     # Vincenty numerators and denominator (~16), hypot, atan2, degrees and arc seconds (fastskymatch.py:36-47)
FLOPS_POINT = 55.0   # sin / cos of a latitude, once per primary (registration) and once per link (routing)
FLOPS_GROUP = 146.0  # per primary: two log10 and one 10^x of the log-sum-exp (__init__.py:423-457)


def flops_row(n_present):
	"""a row with n_present catalogues (bayesdistance.py:64-86 + :26-32 + its two terms of the group statistics): n weights s**-2 and
	their logarithms, log of their sum, three operations per pair, two divisions, posterior (10^x, a division), p_i and the LSE term (2 x 10^x)"""
	n = float(n_present)
	return 0.0 if n < 2 else 57.0 * n + 117.0 + 1.5 * n * (n - 1) + 68.0


def alu_model(k, n_primary, rows, tests, ms):
	"""The FP64 roofline beside the HBM one ("report both rooflines honestly"): algorithmic FP64 flops of a pass --
	distance tests, latitudes, rows, group statistics as counted above; a row with a counterpart is taken to hold (2 + k) / 2
	catalogues for k >= 3 (the status words do not count rows by ncat) -- over the pass time, against the FP64 vector peak."""
	links = max(rows - n_primary, 0)
	n_bar = 2.0 if k == 2 else (2.0 + k) / 2.0
	flops = tests * FLOPS_TEST + (n_primary + links) * FLOPS_POINT + n_primary * FLOPS_GROUP + links * flops_row(n_bar)
	tf = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
	return dict(fp64_flops=flops, achieved=tf, peak=FP64_PEAK_TFLOPS, unit='TFLOP/s', frac=tf / FP64_PEAK_TFLOPS,
		model='tests x %g + (primaries + links) x %g + primaries x %g + rows with a counterpart x %g flops (bench.py: alu_model)' % (FLOPS_TEST, FLOPS_POINT, FLOPS_GROUP, flops_row(n_bar)))


def bound_of(pass_frac, alu_frac):
	"""what a pass is bound by, from its two roofline fractions: neither above a fifth = the chains of dependent round trips and the
	launch boundaries of its kernels, not a throughput of the machine"""
	if pass_frac >= 0.2 and pass_frac >= alu_frac:
		return 'hbm'
	if alu_frac >= 0.2:
		return 'fp64'
	return 'latency (launch boundaries + dependent round trips + FP64 chains at 1-2 waves per SIMD: neither roofline above 0.2)'


def job_bytes(sizes, error_columns, rows):
	"""algorithmic bytes of a whole JOB: every input column once, every output column once"""
	k = len(sizes)
	b = sum(n * (16.0 + (8.0 if col else 0.0)) for n, col in zip(sizes, error_columns))
	return b + (4 * k + 8 * (k * (k - 1) // 2) + 8 + 1 + 8 * 5 + 1) * rows


def extra_configs(args, world, rank, device, dist, backend, records=None, only_jobs=None, t_start=None, deadline=None):
	"""The jobs BASELINE names for several GPUs, measured in the SAME launch as the headline (whose default, weak scaling of
	C3-S, is N x by construction): every job is FIXED in size and divided over the ranks, so value(N) / value(1) is its
	strong-scaling curve.  One record per job, mode and carrier of the exchanges:
	  c3s_split      C3-S as ONE job (1e5 x 1e7, 5"), secondary stream split (SecondarySplitMatch)
	  c4s_rows       BASELINE configs[3]: 3-way 1e5 x 1e6 x 1e6, 10", primary rows sharded (ShardedMatch)
	  c5_rows        BASELINE configs[4]: 5e5 x 1e8, 5", primary rows sharded
	  c5_split       the same job, secondary stream split
	  c3s_zones, c4s_zones, c5_zones   the three jobs with BOTH sides sharded by declination zones (ZoneShardedMatch: one
	                 all-to-all-v of rows at set-up, no collective per step, every rank streams 1/N of the secondaries)
	Each rank generates ITS shard of the primaries and ITS slices of the secondaries (the counterparts of its primaries lie
	in its own slices), so no rank ever holds a whole 1e8-row catalogue on the host.  NWAY_BENCH_EXTRA_SCALE (tests) scales
	every catalogue size."""
	import torch
	from nway_amd import distributed, _hip
	scale = float(os.environ.get('NWAY_BENCH_EXTRA_SCALE', '1'))
	sz = lambda n: max(int(n * scale), 8 * world)
	# (in the order of what a first multi-GPU run should not miss if the time budget below runs out)
	jobs = [('c5_zones', 'zones', [sz(5e5), sz(1e8)], 5.0), ('c5_rows', 'rows', [sz(5e5), sz(1e8)], 5.0), ('c4s_rows', 'rows', [sz(1e5), sz(1e6), sz(1e6)], 10.0),
		('c3s_split', 'split', [sz(1e5), sz(1e7)], 5.0), ('c5_split', 'split', [sz(5e5), sz(1e8)], 5.0),
		('c3s_zones', 'zones', [sz(1e5), sz(1e7)], 5.0), ('c4s_zones', 'zones', [sz(1e5), sz(1e6), sz(1e6)], 10.0)]
	only = os.environ.get('NWAY_BENCH_EXTRA_ONLY')
	comms = ['torch'] + (['rccl'] if backend == 'nccl' else [])
	steps, warm = min(args.steps, 20), min(max(args.warmup, 2), 5)
	# (records: the caller's list, appended to record by record -- what has been measured is the caller's whatever happens later;
	# only_jobs: this call's share of the list above; t_start: when the budget's clock started -- the calls of one run share it)
	records = [] if records is None else records
	budget_s = float(os.environ.get('NWAY_BENCH_EXTRA_BUDGET', '200'))  # the whole block is skipped job by job once this is spent
	t_start = time.perf_counter() if t_start is None else t_start

	def agreed(ok):
		"""the same decision on every rank (a rank that failed alone would leave the others in a collective)"""
		flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=device)
		dist.all_reduce(flag, op=dist.ReduceOp.MIN)
		return bool(flag.item())
	for name, mode, sizes, radius in jobs:
		if only and name not in only.split(','):
			continue
		if only_jobs is not None and name not in only_jobs:
			continue
		if not agreed(time.perf_counter() - t_start < budget_s and (deadline is None or time.perf_counter() < deadline)):
			records.append(dict(job=name, skipped='time budget of the extra configurations (%g s) spent' % budget_s))
			continue
		local = [distributed.shard_bounds(n, world) for n in sizes]
		mine = [int(b[rank + 1] - b[rank]) for b in local]
		tabs, gen_error = None, None
		try:
			if len(sizes) == 2:
				tabs = list(make_workload(mine[0], mine[1], args.seed + 77 + 1000 * rank))
			else:
				tabs = make_workload3(mine[0], mine[1], mine[2], args.seed + 77 + 1000 * rank)
			for t, n in zip(tabs, sizes):
				t['area'] = SKY_AREA  # (of the whole catalogue: the engines take the densities from the global sizes)
		except Exception as e:
			gen_error = '%s: %s' % (type(e).__name__, e)
		if not agreed(gen_error is None):
			records.append(dict(job=name, error='a rank could not generate its shard: %s' % gen_error))
			continue
		for comm in (comms if mode != 'zones' else comms[:1]):  # (the zone mode exchanges at set-up only: one carrier)
			rec = dict(job=name, mode={'split': 'secondary-stream slices + candidate routing', 'rows': 'primary-row shards',
				'zones': 'declination zones (both sides sharded)'}[mode],
				sizes=sizes, radius_arcsec=radius, exchanges=('nwayhip_comm_* (RCCL behind the C ABI)' if comm == 'rccl' else 'torch.distributed (%s)' % backend),
				scaling='strong', n_gpus=world)
			engine = None
			try:
				torch.cuda.synchronize(device)
				t0 = time.perf_counter()
				cls = {'split': distributed.SecondarySplitMatch, 'rows': distributed.ShardedMatch, 'zones': distributed.ZoneShardedMatch}[mode]
				engine = cls(tabs[0], tabs[1:], radius, args.completeness, device, comm=('rccl' if comm == 'rccl' else None))
				torch.cuda.synchronize(device)
				rec['setup_s'] = time.perf_counter() - t0
				for _ in range(warm):
					engine.step()
				torch.cuda.synchronize(device)
				dist.barrier()
				torch.cuda.synchronize(device)
				t0 = time.perf_counter()
				for _ in range(steps):
					engine.step()
				torch.cuda.synchronize(device)
				el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
				dist.barrier()
				dist.all_reduce(el, op=dist.ReduceOp.MAX)
				st = engine.read_status()
				flags = torch.tensor([int(st[_hip.ST_FLAGS])], dtype=torch.int64, device=device)
				dist.all_reduce(flags, op=dist.ReduceOp.MAX)
				seen = torch.ones(1, dtype=torch.int64, device=device)
				dist.all_reduce(seen)
				rows = engine.total_rows()
				ms = float(el.item()) * 1e3 / steps
				jb = job_bytes(sizes, [True] + [False] * (len(sizes) - 1), rows)
				rec.update(ms_per_step=ms, steps=steps, rows=rows, value=rows / (ms * 1e-3), ranks_seen=int(seen.item()), flags=int(flags.item()),
					job_bytes=jb, pass_frac=jb / (ms * 1e-3) / 1e9 / (HBM_PEAK_GBS * world),
					pass_frac_note='algorithmic bytes of the WHOLE job (SURVEY 8d) / step time / (N x 8 TB/s)',
					rank0_rows=engine.local_rows(), rank0_pass_frac=engine.pass_bytes(engine.local_rows()) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
					path=(engine.plan.description if engine.plan is not None else None))
				if mode == 'split':
					rec['exchange_block_records'] = engine.capacity
					rec['exchange_block_records_used'] = engine.block_records_used
					rec['exchange_bytes_per_peer_per_step'] = 32 * (engine.capacity + 1) * (len(sizes) - 1)
				elif mode == 'zones':
					rec['setup_exchange_bytes'] = engine.moved_bytes   # (what this rank sent, its own zone's rows included)
					rec['rank0_zone_sizes'] = [int(c.n) for c in engine.cats]
				else:
					rec['setup_exchange_bytes'] = engine.gathered_bytes
			except Exception as e:  # (a job that does not fit a mode is a record, not the end of the run; every rank raises alike)
				rec['error'] = '%s: %s' % (type(e).__name__, e)
			finally:
				if engine is not None:
					if hasattr(engine, 'close'):
						engine.close()
					elif getattr(engine, 'plan', None) is not None:
						engine.plan.close()
					if getattr(engine, 'comm', None) is not None:
						engine.comm.close()
				del engine
				torch.cuda.empty_cache()
			records.append(rec)
	return records


FIXED_JOBS = [('c3s', [1e5, 1e7], 5.0, 'BASELINE configs[2]: 2-way 1e5 x 1e7, 5 arcsec'),
	('c4s', [1e5, 1e6, 1e6], 10.0, 'BASELINE configs[3]: 3-way 1e5 x 1e6 x 1e6, 10 arcsec'),
	('c5', [5e5, 1e8], 5.0, 'BASELINE configs[4]: 2-way 5e5 x 1e8, 5 arcsec')]


def single_gpu_jobs(args, device, names, budget_s=240.0, out=None, deadline=None):
	"""The fixed-size jobs BASELINE names, each as ONE job on ONE GPU (this process's): the N = 1 point of their strong-scaling
	curves, measured in the same launch and on the same hardware as the N > 1 points of `extra_configs`, so that
	value(N) / value(1) needs no second run.  Same generators, same seeds' family, same step definition (one pass of the whole
	path over the resident catalogues).  NWAY_BENCH_EXTRA_SCALE scales the sizes (tests)."""
	import torch
	import nway_amd
	from nway_amd import _hip
	scale = float(os.environ.get('NWAY_BENCH_EXTRA_SCALE', '1'))
	out = {} if out is None else out  # (the caller's dict: a job's record is the caller's as soon as it is measured)
	t_start = time.perf_counter()
	steps, warm = min(args.steps, 20), min(max(args.warmup, 2), 5)
	for name, sizes, radius, what in FIXED_JOBS:
		if name not in names:
			continue
		if time.perf_counter() - t_start > budget_s or (deadline is not None and time.perf_counter() > deadline):
			out[name] = dict(skipped='time budget of the single-GPU references (%g s) or of the supplementary blocks spent' % budget_s)
			continue
		sizes = [max(int(n * scale), 8) for n in sizes]
		rec = dict(job=what, sizes=sizes, radius_arcsec=radius, n_gpus=1)
		plan = None
		try:
			tabs = list(make_workload(sizes[0], sizes[1], args.seed + 77)) if len(sizes) == 2 else make_workload3(sizes[0], sizes[1], sizes[2], args.seed + 77)
			k = len(tabs)
			err = radius / 60. / 60
			scheme = nway_amd.choose_scheme([(t['ra'], t['dec']) for t in tabs], err)
			dens, dens_plus = nway_amd._compute_source_densities(tabs, nway_amd.NullOutputLogger())
			comp = nway_amd._completeness_vector(args.completeness, k)
			params = _hip.make_params(k, scheme, radius, err, dens, dens_plus, nway_amd._prior_table(dens, dens_plus, comp))
			cats = [_hip.DeviceCatalogue(t['ra'], t['dec'], np.asarray(t['error'], dtype=float), device) for t in tabs]
			cap_pairs, cap_rows = nway_amd._estimate_capacities([c.n for c in cats], [SKY_AREA] * k, radius, scheme, True)
			plan, st = _hip.run_plan([c.n for c in cats], params, cats, cap_pairs, cap_rows, device, lean=True)
			t0 = time.perf_counter()
			while time.perf_counter() - t0 < 0.03:  # (at least 30 ms of passes first: the clocks of an idle GPU take ~10 ms to come up)
				for _ in range(warm):
					plan.enqueue(cats)
				torch.cuda.synchronize(device)
			t0 = time.perf_counter()
			for _ in range(steps):
				plan.enqueue(cats)
			torch.cuda.synchronize(device)
			ms = (time.perf_counter() - t0) * 1e3 / steps
			st = plan.read_status()
			rows = int(st[_hip.ST_ROWS])
			jb = job_bytes(sizes, [True] + [False] * (k - 1), rows)
			rec.update(ms_per_step=ms, steps=steps, rows=rows, value=rows / (ms * 1e-3), flags=int(st[_hip.ST_FLAGS]), job_bytes=jb,
				pass_frac=jb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, path=plan.description)
			rec['alu'] = alu_model(k, sizes[0], rows, int(st[_hip.ST_TESTS]), ms)
			rec['bound'] = bound_of(rec['pass_frac'], rec['alu']['frac'])
			if name == 'c5' and os.environ.get('NWAY_BENCH_LOCAL_ZONES', '1') != '0':
				# the same job with the catalogues bucketed into declination zones at set-up (ZoneShardedMatch on ONE rank, round 5): every
				# zone's cell table fits the LDS of a sweep workgroup, where the job as one zone needs the large-table sweep; round 6: the
				# zones of a step go out as ONE launch set (one registration, one sweep, one tail launch: nwayhip_zones_enqueue)
				plan.close()
				plan = None
				cats = None
				torch.cuda.empty_cache()
				from nway_amd import distributed
				eng = distributed.ZoneShardedMatch(tabs[0], tabs[1:], radius, args.completeness, device, zones_per_rank=8, local_only=True)
				try:
					t0 = time.perf_counter()
					while time.perf_counter() - t0 < 0.03:  # (24 short launches per pass: the clocks of an idle GPU take ~10 ms to come up)
						eng.step()
						torch.cuda.synchronize(device)
					t0 = time.perf_counter()
					for _ in range(steps):
						eng.step()
					torch.cuda.synchronize(device)
					msz = (time.perf_counter() - t0) * 1e3 / steps
					stz = eng.read_status()
					alz = alu_model(k, sizes[0], int(stz[_hip.ST_ROWS]), int(stz[_hip.ST_TESTS]), msz)
					rec['zones'] = dict(zones=8, one_launch_set=bool(eng.batched), alu=alz, bound=bound_of(jb / (msz * 1e-3) / 1e9 / HBM_PEAK_GBS, alz['frac']), ms_per_step=msz, rows=int(stz[_hip.ST_ROWS]), value=int(stz[_hip.ST_ROWS]) / (msz * 1e-3), flags=int(stz[_hip.ST_FLAGS]),
						pass_frac=jb / (msz * 1e-3) / 1e9 / HBM_PEAK_GBS, setup_s=eng.setup_seconds,
						note='ZoneShardedMatch(zones_per_rank=8) on one rank, the zones of a step as one launch set: the one-time bucketing of the catalogues by zone is set-up, like the exchanges of the multi-GPU modes')
				finally:
					eng.close()
		except Exception as e:
			rec['error'] = '%s: %s' % (type(e).__name__, e)
		finally:
			if plan is not None:
				plan.close()
			cats = tabs = None
			torch.cuda.empty_cache()
		out[name] = rec
	return out


def fixed_size_summary(extras, n1, world):
	"""per job BASELINE names: its best N-GPU record of `extra_configs`, the one-GPU run of the same job from the same launch,
	and their ratio -- the job's strong-scaling speed-up at this N"""
	out = {}
	for name, _, _, what in FIXED_JOBS:
		recs = [r for r in (extras or []) if r.get('job', '').split('_')[0] == name and 'value' in r and not r.get('flags')]
		best = max(recs, key=lambda r: r['value']) if recs else None
		one = (n1 or {}).get(name)
		entry = dict(job=what)
		if one is not None:
			entry['one_gpu'] = dict((key, one.get(key)) for key in ('ms_per_step', 'value', 'rows', 'pass_frac', 'alu', 'bound', 'error', 'skipped') if key in one)
			if one.get('zones'):
				entry['one_gpu_zones'] = one['zones']
		if best is not None:
			entry['n_gpus'] = world
			entry['best'] = dict(mode=best['mode'], exchanges=best['exchanges'], ms_per_step=best['ms_per_step'], value=best['value'], rows=best['rows'],
				pass_frac=best['pass_frac'], ranks_seen=best['ranks_seen'])
			if one is not None and one.get('value'):
				entry['speedup_vs_one_gpu'] = best['value'] / one['value']  # (rows per second of the same-sized job; the shards are seeded per rank, so the row counts agree to a fraction of a per cent, not exactly)
				if one.get('zones') and not one['zones'].get('flags'):
					entry['speedup_vs_best_one_gpu'] = best['value'] / max(one['value'], one['zones']['value'])  # (against the faster of the two one-GPU runs)
		out[name] = entry
	return out


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument('--gpus', type=int, default=1)
	ap.add_argument('--steps', type=int, default=40)
	ap.add_argument('--warmup', type=int, default=6)
	ap.add_argument('--n-primary', type=int, default=100000)
	ap.add_argument('--n-secondary', type=int, default=10000000)
	ap.add_argument('--radius', type=float, default=5.0)
	ap.add_argument('--completeness', type=float, default=0.9)
	ap.add_argument('--seed', type=int, default=1)
	ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak')
	ap.add_argument('--sec-buffers', type=int, default=3, help='distinct device copies of the secondary catalogue the steps alternate over')
	ap.add_argument('--cpu-sample', type=int, default=1000000, help='secondaries in the numpy leg of the CPU baseline (0 = no CPU baseline at all)')
	ap.add_argument('--event-every', type=int, default=0, help='every n-th sweep launch of the timed region carries a HIP event pair (0 = choose: 2 for --steps <= 24, 4 below 96, 8 from there: a short run still has >= 10 timed launches)')
	ap.add_argument('--profile-stages', action='store_true', help='also time every stage (adds event records to the region)')
	ap.add_argument('--prewarm', type=int, default=200, help='untimed steps before the W warm-up steps: the first ~10 ms after an idle period run at lower clocks (20 steps right after start-up: 82 us each, after 200: 78.5)')
	ap.add_argument('--two-pipelines', type=int, default=1, help='also time the steps alternating over two (or this many, if > 2) independent pipelines (reported beside, never as, `value`); 0 = skip')
	ap.add_argument('--comm', choices=['torch', 'rccl'], default=os.environ.get('NWAY_BENCH_COMM', 'torch'),
		help='who carries the exchanges of the multi-GPU modes: torch.distributed (default) or the library\'s own RCCL calls behind the C ABI (nwayhip_comm_*)')
	ap.add_argument('--extras', type=int, default=int(os.environ.get('NWAY_BENCH_EXTRAS', '1')),
		help='N > 1: also measure, in the same launch, the fixed-size jobs BASELINE names for several GPUs (extra_configs: C3-S as one job, configs[3], configs[4]; both sharding modes, both carriers of the exchanges); 0 = skip')
	ap.add_argument('--extras-watchdog', type=float, default=float(os.environ.get('NWAY_BENCH_EXTRAS_WATCHDOG', '0')),
		help='N > 1: seconds after which hung supplementary blocks are abandoned and what has been measured is printed (0 = from their budget: '
		'NWAY_BENCH_SUPP_BUDGET, 330 s, after which no further job starts, + 120 s for the job in flight)')
	ap.add_argument('--rendezvous-only', action='store_true', help='(tests) every rank joins the process group, rank 0 prints how many answered, nothing is measured')
	ap.add_argument('--fixed-jobs', type=int, default=int(os.environ.get('NWAY_BENCH_FIXED_JOBS', '1')),
		help='also measure, as ONE job on ONE GPU, the fixed-size jobs BASELINE names (configs[3], configs[4]; with N > 1 also configs[2]): '
		'the N = 1 point of their strong-scaling curves, in this launch (rank 0, the other ranks wait); 0 = skip')
	ap.add_argument('--live-traffic', type=int, default=int(os.environ.get('NWAY_BENCH_LIVE_TRAFFIC', '1')),
		help='N = 1: measure roofline.traffic in this run (two short child runs under rocprofv3 --pmc, ~30 s); 0: the recorded profiles/sweep_traffic.json')
	ap.add_argument('--streams', type=int, default=int(os.environ.get('NWAY_BENCH_STREAMS', '1')),
		help='independent pipelines (own workspace, own output table, own HIP stream) the steps alternate over')
	args = ap.parse_args()
	# ONE budget for everything after the headline (N > 1): no job of the supplementary blocks starts once it is spent; the watchdog
	# allows the job then in flight two more minutes
	supp_budget = float(os.environ.get('NWAY_BENCH_SUPP_BUDGET', '330'))
	if args.extras_watchdog <= 0:
		args.extras_watchdog = supp_budget + 120.0

	import torch

	# `python bench.py --gpus N` without a launcher: this process becomes the launcher of N ranks (one per GPU, RCCL) -- the very command
	# line the bench contract gives for N > 1.  With fewer than N GPUs (and the real backend) nothing sensible can be started: the one-GPU
	# run goes ahead and its line says so in a top-level `error`.
	spawn_note = None
	if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and os.environ.get('NWAY_BENCH_FORCE_DIST') != '1':
		shared = os.environ.get('NWAY_BENCH_BACKEND', 'nccl') != 'nccl'   # (gloo: ranks may share a GPU -- functional tests)
		have = torch.cuda.device_count() if torch.cuda.is_available() else 0
		if have >= args.gpus or shared:
			import socket
			import subprocess
			with socket.socket() as sock:
				sock.bind(('127.0.0.1', 0))
				port = sock.getsockname()[1]
			cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
				'--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
			sys.stderr.write('bench.py: --gpus %d without a launcher: starting %s\n' % (args.gpus, ' '.join(cmd)))
			sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))))
		spawn_note = '--gpus %d asked for, %d GPU(s) visible: ONE rank ran (n_gpus below says 1)' % (args.gpus, have)
		sys.stderr.write('bench.py: %s\n' % spawn_note)

	world = int(os.environ.get('WORLD_SIZE', '1'))
	if args.rendezvous_only:
		import torch.distributed as dist
		rank0 = int(os.environ.get('RANK', '0'))
		seen = 1
		if world > 1:
			dist.init_process_group(os.environ.get('NWAY_BENCH_BACKEND', 'nccl'))
			t = torch.ones(1, dtype=torch.int64)
			if dist.get_backend() == 'nccl':
				torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % max(torch.cuda.device_count(), 1))
				t = t.cuda()
			dist.all_reduce(t)
			seen = int(t.item())
			dist.destroy_process_group()
		if rank0 == 0:
			print(json.dumps(dict(rendezvous_only=True, gpus_asked=args.gpus, world=world, ranks_seen=seen, error=spawn_note)))
		return

	import nway_amd
	from nway_amd import _hip

	rank = int(os.environ.get('RANK', '0'))
	local_rank = int(os.environ.get('LOCAL_RANK', '0'))
	ngpu = max(torch.cuda.device_count(), 1)
	force_dist = world == 1 and os.environ.get('NWAY_BENCH_FORCE_DIST') == '1'  # (one rank through the engines and RCCL: a dry run of the N > 1 code)
	if world > 1 or force_dist:
		import torch.distributed as dist
		if force_dist:
			os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
			os.environ.setdefault('MASTER_PORT', '29533')
			os.environ.setdefault('RANK', '0')
			os.environ.setdefault('WORLD_SIZE', '1')
		# "nccl" is RCCL on ROCm.  NWAY_BENCH_BACKEND=gloo lets several ranks share one GPU
		# (functional testing of the sharded path on a 1-GPU box only).
		backend = os.environ.get('NWAY_BENCH_BACKEND', 'nccl')
		torch.cuda.set_device(local_rank % ngpu)
		if backend == 'nccl':
			dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank % ngpu))
		else:
			dist.init_process_group(backend)
	tuning = None
	if args.gpus != world:
		if rank == 0:
			sys.stderr.write('note: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE (the line carries a top-level `error`)\n' % (args.gpus, world))
	device = torch.device('cuda', (local_rank % ngpu) if world > 1 else 0)
	torch.cuda.set_device(device)

	io = None
	engine = None
	strong = args.scaling == 'strong' and (world > 1 or force_dist)
	if (world > 1 or force_dist) and not strong:
		# weak scaling: every rank owns n_primary primaries (a contiguous row shard of the global
		# primary catalogue, N x n_primary rows) and loads a 1/world slice of the ONE secondary
		# catalogue (n_secondary rows in total); the slices are all-gathered once at set-up so that
		# every GPU holds the whole secondary catalogue, then each step matches the rank's primary
		# shard against it -- per-GPU work is fixed, no collective on the per-step path.
		n_sec_local = args.n_secondary // world + (args.n_secondary % world if rank == world - 1 else 0)
		primary, secondary = make_workload(args.n_primary, n_sec_local, args.seed + 1000 * rank)
		from nway_amd import distributed
		engine = distributed.ShardedMatch(primary, [secondary], args.radius, args.completeness, device, tuning=tuning, comm=('rccl' if args.comm == 'rccl' else None))
	elif strong:
		# strong scaling: ONE job of n_primary x n_secondary.  Every rank generates the same
		# catalogues (same seed) and keeps its slice of the secondaries and its shard of the primaries
		from nway_amd import distributed
		primary, secondary = make_workload(args.n_primary, args.n_secondary, args.seed)
		sb = distributed.shard_bounds(args.n_secondary, world)
		pb = distributed.shard_bounds(args.n_primary, world)
		sec_slice = dict(secondary, ra=secondary['ra'][sb[rank]:sb[rank + 1]], dec=secondary['dec'][sb[rank]:sb[rank + 1]])
		prim_shard = dict(primary, ra=primary['ra'][pb[rank]:pb[rank + 1]], dec=primary['dec'][pb[rank]:pb[rank + 1]], error=primary['error'][pb[rank]:pb[rank + 1]])
		engine = distributed.SecondarySplitMatch(prim_shard, [sec_slice], args.radius, args.completeness, device, tuning=tuning, comm=('rccl' if args.comm == 'rccl' else None))
	else:
		primary, secondary = make_workload(args.n_primary, args.n_secondary, args.seed)

	if engine is None:
		log = nway_amd.NullOutputLogger()
		tables = [primary, secondary]
		err = args.radius / 60. / 60
		scheme = nway_amd.choose_scheme([(t['ra'], t['dec']) for t in tables], err)
		dens, dens_plus = nway_amd._compute_source_densities(tables, log)
		comp = nway_amd._completeness_vector(args.completeness, 2)
		params = _hip.make_params(2, scheme, args.radius, err, dens, dens_plus, nway_amd._prior_table(dens, dens_plus, comp), tuning=tuning)
		torch.cuda.synchronize(device)
		t0 = time.perf_counter()
		cats = [_hip.DeviceCatalogue(t['ra'], t['dec'], np.asarray(t['error'], dtype=float), device) for t in tables]
		torch.cuda.synchronize(device)
		h2d_first_s = time.perf_counter() - t0
		# the same once more: the first upload of a process also pays for the context, the allocator's first blocks and
		# the first touch of the freshly generated host arrays
		del cats
		torch.cuda.synchronize(device)
		t0 = time.perf_counter()
		cats = [_hip.DeviceCatalogue(t['ra'], t['dec'], np.asarray(t['error'], dtype=float), device) for t in tables]
		torch.cuda.synchronize(device)
		h2d_s = time.perf_counter() - t0
		h2d_bytes = sum(c.ra.numel() * 8 * (2 + (1 if c.sigma is not None else 0)) for c in cats)
		sizes = [c.n for c in cats]
		# further copies of the secondary catalogue at other addresses: a 160 MB stream that is read
		# again and again would otherwise be served by the 256 MiB Infinity Cache
		sec_copies = [cats[1]]
		for _ in range(max(args.sec_buffers, 1) - 1):
			cp = _hip.DeviceCatalogue.__new__(_hip.DeviceCatalogue)
			cp.ra, cp.dec = cats[1].ra.clone(), cats[1].dec.clone()
			cp.sigma = None if cats[1].sigma is None else cats[1].sigma.clone()
			cp.sigma_const, cp.n = cats[1].sigma_const, cats[1].n
			sec_copies.append(cp)
		# settle capacities with one untimed run
		cap_pairs, cap_rows = nway_amd._estimate_capacities(sizes, [SKY_AREA, SKY_AREA], args.radius, scheme, True)
		plan, st = _hip.run_plan(sizes, params, cats, cap_pairs, cap_rows, device, lean=True)
		rows_per_step = int(st[_hip.ST_ROWS])
		# every step is a complete, independent pass; with --streams S the steps alternate over S
		# pipelines (workspace + output table + HIP stream each) so that the latency-bound stages of
		# one pass overlap the HBM-bound sweep of another
		plans = [plan] + [_hip.MatchPlan(sizes, plan.params, plan.cap_pairs, plan.cap_rows, device, lean=True) for _ in range(args.streams - 1)]  # (plan.params: what the settling run ended up with)
		from nway_amd import distributed as _ds
		streams = _ds.side_streams(device, len(plans)) if len(plans) > 1 else [None]
		counter = [0]

		def step():
			i = counter[0] % len(plans)
			these = [cats[0], sec_copies[counter[0] % len(sec_copies)]]
			counter[0] += 1
			if streams[i] is None:
				plans[i].enqueue(these)
			else:
				with torch.cuda.stream(streams[i]):
					plans[i].enqueue(these)
		read_status = plan.read_status
		# device -> host download of the whole table, once, outside the timed region
		torch.cuda.synchronize(device)
		t0 = time.perf_counter()
		d2h_bytes = 0
		for name in ('sep_max', 'log_bf', 'dist_post', 'p_single', 'p_any', 'p_i', 'ncat', 'match_flag'):
			d2h_bytes += _hip.to_host(plan.cols[name][:rows_per_step]).nbytes
		for col in plan.cols['idx'] + plan.cols['sep']:
			d2h_bytes += _hip.to_host(col[:rows_per_step]).nbytes
		d2h_s = time.perf_counter() - t0
		io = dict(h2d_ms=h2d_s * 1e3, h2d_first_ms=h2d_first_s * 1e3, h2d_bytes=int(h2d_bytes), h2d_mode=_hip.upload_mode['last'], d2h_ms=d2h_s * 1e3, d2h_bytes=int(d2h_bytes),
			note='host arrays -> HBM before the timed region (page-locked in place for the copy engine), table -> host after it; never part of `value`')
	else:
		step = engine.step
		n_sec_copies = 1
		if not strong and engine.plan is not None:
			# as on one GPU: the passes alternate over distinct device copies of the secondary catalogue, so that no pass
			# finds its 160 MB stream in the 256 MiB Infinity Cache (every rank holds the whole catalogue in this mode)
			def clone(c):
				cp = _hip.DeviceCatalogue.__new__(_hip.DeviceCatalogue)
				cp.ra, cp.dec = c.ra.clone(), c.dec.clone()
				cp.sigma = None if c.sigma is None else c.sigma.clone()
				cp.sigma_const, cp.n = c.sigma_const, c.n
				return cp
			copies = [engine.cats] + [[engine.cats[0]] + [clone(c) for c in engine.cats[1:]] for _ in range(max(args.sec_buffers, 1) - 1)]
			n_sec_copies = len(copies)
			turn = [0]

			def step():
				engine.step(copies[turn[0] % len(copies)])
				turn[0] += 1
		read_status = engine.read_status
		engine.step()
		rows_per_step = engine.total_rows()
		plan = engine.plan
		plans = [plan]

	def barrier():
		torch.cuda.synchronize(device)
		if world > 1 or force_dist:
			dist.barrier()
			torch.cuda.synchronize(device)

	for _ in range(max(args.prewarm, 0) + args.warmup):
		step()
	mask = (1 << _hip.STAGES) - 1 if args.profile_stages else (1 << 1)
	event_every = args.event_every if args.event_every > 0 else (2 if args.steps <= 24 else (4 if args.steps < 96 else 8))  # (>= 10 timed sweeps from 20 steps on)
	for pl in plans:
		pl.profile(mask, 1 if args.profile_stages else event_every)
	barrier()
	t0 = time.perf_counter()
	for _ in range(args.steps):
		step()
	# a rank's clock stops when ITS K steps have finished on the device; the closing barrier follows,
	# and the reported time is the maximum over the ranks (below) -- the moment the slowest rank was
	# done, without the latency of the barrier collective itself
	torch.cuda.synchronize(device)
	elapsed = time.perf_counter() - t0
	barrier()
	launches, ms = [0] * _hip.STAGES, [0.0] * _hip.STAGES
	sweep_samples = []
	for pl in plans:
		sweep_samples += pl.profile_samples(1)
		n_, ms_ = pl.profile_read()
		launches = [a + b for a, b in zip(launches, n_)]
		ms = [a + b for a, b in zip(ms, ms_)]
		pl.profile(0)
	if world > 1 or force_dist:
		tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
		dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
		elapsed = float(tmax.item())
		rows_per_step = engine.total_rows()
	st = read_status()
	assert int(st[_hip.ST_FLAGS]) == 0, 'overflow flags set: %d' % int(st[_hip.ST_FLAGS])
	for pl in plans[1:]:  # (--streams > 1: the flags of every pipeline, not of the first alone)
		assert int(pl.read_status()[_hip.ST_FLAGS]) == 0, 'overflow flags set on a sibling pipeline: %d' % int(pl.read_status()[_hip.ST_FLAGS])
	ms_per_step = elapsed * 1e3 / args.steps

	if rank == 0:
		n_sec_swept = int(plan.sizes[1])
		sweep_ms = ms[1] / max(launches[1], 1)
		alg_bytes = 16.0 * n_sec_swept
		achieved = alg_bytes / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
		traffic, traffic_source, traffic_detail = None, None, None
		live_note = None
		under_profiler = any(k.startswith(('ROCPROF', 'ROCP_', 'ROCTX')) for k in os.environ)  # (a run that is itself being profiled does not start profilers)
		if args.live_traffic and engine is None and world == 1 and args.cpu_sample != 0 and not under_profiler:
			tail = ['--n-primary', str(args.n_primary), '--n-secondary', str(args.n_secondary), '--radius', str(args.radius), '--completeness', str(args.completeness),
				'--seed', str(args.seed), '--sec-buffers', str(args.sec_buffers)]
			traffic, traffic_source, traffic_detail = live_traffic(tail, n_sec_swept)
			if traffic is None:
				live_note, traffic_source = traffic_source, None
		tf = os.path.join(ROOT, 'profiles', 'sweep_traffic.json')
		if traffic is None and os.path.exists(tf):
			try:
				rec = json.load(open(tf))
				if rec.get('n_secondary') == n_sec_swept:
					traffic = rec.get('hbm_bytes_per_launch')
					same = rec.get('kernel_source_sha16') == kernel_source_hash()
					traffic_source = ('NOT measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same command, '
						'recorded in profiles/sweep_traffic.json (%s, git %s%s, kernel sources %s): %s' % (rec.get('round', 'earlier round'),
						rec.get('git_head'), ' + uncommitted changes' if rec.get('git_dirty') else '', rec.get('kernel_source_sha16'),
						'the kernel sources of this tree are the ones that were measured' if same else
						'STALE -- the kernel sources of this tree differ from the measured build (rerun tools/profile_round.sh)'))
			except Exception:
				traffic = None
			if traffic_source and live_note:
				traffic_source += ' [live measurement not available: %s]' % live_note
		# the whole pass, SURVEY 8(d): rank 0's pass (its primaries, the secondaries it streams, its rows)
		local_rows = int(st[_hip.ST_ROWS])
		if engine is None:
			p_bytes = pass_bytes([primary, secondary], local_rows)
		else:
			p_bytes = engine.pass_bytes(local_rows)
		if strong:
			workload = ('C3-S synthetic 2-way, ONE job: %d primaries x %d secondaries over %d GPUs (every rank sweeps its 1/%d slice of the '
				'secondaries against all the primaries; candidates routed to the owners of the primaries)' % (args.n_primary, args.n_secondary, world, world))
		else:
			workload = ('C3-S synthetic 2-way: %d primaries per GPU (%d in total) x %d secondaries (whole catalogue resident on every GPU)'
				% (args.n_primary, args.n_primary * world, n_sec_swept))
		workload += ', uniform sky, radius %g arcsec, completeness %g, seed %d' % (args.radius, args.completeness, args.seed)
		out = dict(metric='candidate Bayes-factor evals/s', value=rows_per_step / (ms_per_step * 1e-3), unit='candidate evaluations/s',
			n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=ms_per_step, higher_is_better=True,
			scaling=('strong' if strong else 'weak'), vs_baseline=None, dtype='f64', data='synthetic',
			config=dict(workload=workload,
				rows_per_step=rows_per_step, distance_tests_per_step_rank0=int(st[_hip.ST_TESTS]),
				survivors_per_step_rank0=int(st[_hip.ST_SURVIVORS]), registrations_rank0=int(st[_hip.ST_REGISTRATIONS]),
				parallelism=('secondary-stream slices x%d + candidate routing' % world) if strong else ('primary-row shards x%d' % world),
				exchanges=(None if engine is None else ('nwayhip_comm_* (RCCL behind the C ABI)' if args.comm == 'rccl' else 'torch.distributed')),
				streams=len(plans), secondary_buffers=(len(sec_copies) if engine is None else n_sec_copies), prewarm_steps=max(args.prewarm, 0),
				setup_exchange=(None if engine is None else dict(seconds=engine.setup_seconds, bytes=engine.gathered_bytes,
					note='one-time exchange at set-up (RCCL), outside the timed steps'))),
			roofline=dict(bound='hbm', kernel='k_sweep', achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s',
				frac=achieved / HBM_PEAK_GBS, traffic=traffic, traffic_source=traffic_source, traffic_detail=traffic_detail, algorithmic_bytes_per_launch=alg_bytes,
				launch_ms=sweep_ms, launches_timed=int(launches[1]), event_every=(1 if args.profile_stages else event_every),
				launch_ms_min=(min(sweep_samples) if sweep_samples else None), launch_ms_median=(float(np.median(sweep_samples)) if sweep_samples else None),
				launch_ms_max=(max(sweep_samples) if sweep_samples else None),
				pass_bytes=p_bytes, pass_achieved=p_bytes / (ms_per_step * 1e-3) / 1e9, pass_frac=p_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
				pass_note='SURVEY 8(d): every input column once + 66 B per row, divided by the WHOLE step (all launches and the gaps between them); rank 0'))
		if io is not None:
			io.update(end_to_end([primary, secondary], args.radius, args.completeness, device))
			io['e2e_note'] = ('nway_amd.run_match from host arrays to the table on the host: upload, one pass (with its capacity estimate holding: '
				'attempts = 1), download of every column; best of 3')
			out['io'] = io
			read_gbs, copy_gbs = measured_ceilings(sec_copies, device)
			out['roofline'].update(read_peak=read_gbs, copy_peak=copy_gbs, frac_of_read_peak=achieved / read_gbs if read_gbs > 0 else None,
				peaks_note='measured on this box in this run: read_peak = the sweep\'s access pattern with nothing behind the loads (k_read_probe, '
					'16 B per secondary, three buffers alternating); copy_peak = device-to-device copy of one 80 MB column, bytes read + written')
		if engine is None and len(plans) == 1 and args.two_pipelines:
			# supplementary, never `value`: the same steps alternating over TWO independent pipelines
			# (plan + workspace + table + stream each), so that the latency-bound registration and tail
			# of one pass run beside those of the other (the sweeps cannot share a CU: 156 KB of LDS each)
			params2 = _hip.make_params(2, scheme, args.radius, err, dens, dens_plus, nway_amd._prior_table(dens, dens_plus, comp))
			second = _hip.MatchPlan(sizes, params2, plan.cap_pairs, plan.cap_rows, device, lean=True)
			first = plan
			npipes = max(2, int(args.two_pipelines))
			more = [_hip.MatchPlan(sizes, params2, plan.cap_pairs, plan.cap_rows, device, lean=True) for _ in range(npipes - 2)]
			from nway_amd import distributed as _d
			pair, lanes = [first, second] + more, _d.side_streams(device, npipes)  # (the package's own side streams: see there)

			def two(n):
				for j in range(n):
					with torch.cuda.stream(lanes[j % npipes]):
						pair[j % npipes].enqueue([cats[0], sec_copies[j % len(sec_copies)]])
			two(args.warmup + 2)
			torch.cuda.synchronize(device)
			t1 = time.perf_counter()
			two(args.steps)
			torch.cuda.synchronize(device)
			ms2 = (time.perf_counter() - t1) * 1e3 / args.steps
			assert int(second.read_status()[_hip.ST_FLAGS]) == 0 and int(second.read_status()[_hip.ST_ROWS]) == rows_per_step
			out['two_pipelines'] = dict(streams=npipes, ms_per_step=ms2, value=rows_per_step / (ms2 * 1e-3),
				pass_frac=p_bytes / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
				note='supplementary: the same passes, %d in flight on separate HIP streams (measured: 2 -> 62, 3 -> 65, 4 -> 68 us per pass); `value` above is one pass at a time' % npipes)
			second.close()
			for extra in more:
				extra.close()
			if first is not plan:
				first.close()
		if engine is None and len(plans) == 1 and args.two_pipelines and args.cpu_sample != 0:
			# supplementary, never `value`: the SAME catalogues cut into four declination zones at set-up (the bucketing of the
			# catalogues by zone is outside the pass, as an upload is) and the zones of a pass as ONE launch set -- one registration,
			# one sweep, one tail launch (nwayhip_zones_enqueue): smaller cell tables (a quarter of the bitmap to copy into every sweep
			# workgroup's LDS), every zone's workgroups on the XCDs whose L2 holds its tags
			try:
				from nway_amd import distributed as _dz
				zeng = _dz.ZoneShardedMatch(primary, [secondary], args.radius, args.completeness, device, zones_per_rank=4, local_only=True)
				try:
					for _ in range(args.warmup + 10):
						zeng.step()
					torch.cuda.synchronize(device)
					t1 = time.perf_counter()
					for _ in range(args.steps):
						zeng.step()
					torch.cuda.synchronize(device)
					msz = (time.perf_counter() - t1) * 1e3 / args.steps
					stz = zeng.read_status()
					# (the like-for-like one-zone time: the headline's plan on ONE secondary buffer too -- 160 MB that the Infinity Cache holds)
					for _ in range(args.warmup):
						plan.enqueue(cats)
					torch.cuda.synchronize(device)
					t1 = time.perf_counter()
					for _ in range(args.steps):
						plan.enqueue(cats)
					torch.cuda.synchronize(device)
					ms1 = (time.perf_counter() - t1) * 1e3 / args.steps
					out['zones_one_launch_set'] = dict(zones=4, one_zone_same_buffer_ms=ms1, one_launch_set=bool(zeng.batched), ms_per_step=msz, rows=int(stz[_hip.ST_ROWS]), flags=int(stz[_hip.ST_FLAGS]),
						value=int(stz[_hip.ST_ROWS]) / (msz * 1e-3), pass_frac=p_bytes / (msz * 1e-3) / 1e9 / HBM_PEAK_GBS, setup_s=zeng.setup_seconds,
						note='supplementary: the headline\'s catalogues as four declination zones (cut once at set-up, ONE secondary buffer) in one launch set per pass; `value` above is the job as one zone')
					assert out['zones_one_launch_set']['rows'] == rows_per_step and out['zones_one_launch_set']['flags'] == 0
				finally:
					zeng.close()
			except Exception as e:
				out['zones_one_launch_set'] = dict(error='%s: %s' % (type(e).__name__, e))
		if args.profile_stages:
			out['stages_ms'] = dict((name, ms[i] / max(launches[i], 1) * (launches[i] / float(args.steps)))
				for i, name in enumerate(_hip.STAGE_NAMES))
		if world == 1 and args.cpu_sample > 0:
			out['cpu_baseline'], cpu_table = cpu_baseline(primary, secondary, args.radius, args.completeness, args.cpu_sample)
			out['check'] = table_check(plan, [primary['name'], secondary['name']], cpu_table)
			assert out['check']['ok'], 'the timed table differs from the CPU table: %r' % (out['check'],)
		else:
			out['cpu_baseline'] = None
	ranks_seen = 1
	if world > 1 or force_dist:
		seen = torch.ones(1, dtype=torch.int64, device=device)
		dist.all_reduce(seen)
		ranks_seen = int(seen.item())
	extras = [] if (world > 1 or force_dist) and args.extras else None   # filled record by record
	n1 = {} if args.fixed_jobs else None                                   # filled job by job
	watchdog = None
	multi = world > 1 or force_dist

	def finish(aborted=None):
		"""rank 0's line with everything measured so far; `aborted`: why the supplementary blocks did not run to their end"""
		out['ranks_seen'] = ranks_seen
		if args.gpus != world or ranks_seen != world:
			out['error'] = '--gpus %d, WORLD_SIZE %d, ranks that answered the all-reduce: %d' % (args.gpus, world, ranks_seen)
		if spawn_note:
			out['error'] = (out.get('error', '') + '; ' if out.get('error') else '') + spawn_note
		if extras is not None and (extras or aborted):
			out['extra_configs'] = list(extras) + ([dict(error=aborted)] if aborted else [])
		if aborted:
			out['supplementary_aborted'] = aborted
		if n1 or extras:
			out['fixed_size_jobs'] = fixed_size_summary(extras, n1, world)
			out['fixed_size_jobs_note'] = ('the jobs BASELINE names, fixed in size: `best` = the fastest of this launch\'s extra_configs records of the job over '
				'%d GPU(s), `one_gpu` = the same job as one job on one GPU measured in this launch, `speedup_vs_one_gpu` their ratio (strong scaling); '
				'the headline `value` above is %s' % (world, 'the weak-scaling run the bench contract asks for (per-GPU work fixed), N x by construction'
				if (world > 1 and not strong) else 'the one-GPU run of configs[2]'))
		sys.stdout.write(json.dumps(out) + '\n')
		sys.stdout.flush()

	if multi and (args.extras or args.fixed_jobs):
		# The blocks below (fixed-size jobs in several multi-GPU modes, the one-GPU references) run collectives that no machine
		# with more than one GPU has executed yet.  The headline above is measured: whatever happens to them -- a hang, an error on
		# one rank -- it must still come out, and so must every record of theirs that was finished by then.  After --extras-watchdog
		# seconds the other ranks leave and rank 0 prints the line as it stands; an exception in the blocks does the same at once.
		import threading
		import traceback
		once = threading.Lock()

		def give_up(reason):
			if not once.acquire(False):
				time.sleep(60)  # (the other thread is printing; it ends the process)
				return
			if rank == 0:
				finish(aborted=reason)
			os._exit(0)
		watchdog = threading.Timer(args.extras_watchdog + (2.0 if rank == 0 else 0.0), give_up,
			['the supplementary blocks did not finish within %g s: the watchdog printed what had been measured and ended the run' % args.extras_watchdog])
		watchdog.daemon = True
		watchdog.start()
	try:
		run_fixed = args.fixed_jobs and ((multi and args.extras) or (not multi and args.cpu_sample != 0))
		# (with N > 1 the one-GPU references belong to the extra_configs block; a profiling / A-B invocation -- --cpu-sample 0 on one GPU -- measures the headline only)
		if (multi and args.extras) or run_fixed:
			# (the headline's engine has been measured and is released first)
			if engine is not None and getattr(engine, 'plan', None) is not None:
				engine.plan.close()
			engine = None
			if not multi or args.extras:
				for pl in plans:
					if pl is not None and getattr(pl, 'handle', None):
						pl.close()
				plans, plan = [], None
				cats = sec_copies = None
			torch.cuda.empty_cache()

		deadline = (time.perf_counter() + supp_budget) if multi else None

		def one_gpu(names):
			if run_fixed:
				if rank == 0:
					single_gpu_jobs(args, device, names, out=n1, deadline=deadline)
				if multi:
					dist.barrier()  # (the other ranks wait here while rank 0 measures the one-GPU references)
		# Order = what a first run on N GPUs must not lose to a time budget: the two jobs the north star names for N GPUs
		# (configs[4] by declination zones, configs[3] by primary rows), then THEIR one-GPU references, then the other modes
		# and carriers, then the remaining reference.
		first = ['c5_zones', 'c4s_rows']
		t_extras = time.perf_counter()
		if multi and args.extras:
			extra_configs(args, world, rank, device, dist, backend, records=extras, only_jobs=first, t_start=t_extras, deadline=deadline)
		spent_first = time.perf_counter() - t_extras
		one_gpu(['c5', 'c4s'])
		if multi and args.extras:
			# (one budget for both calls; the one-GPU references in between have their own)
			extra_configs(args, world, rank, device, dist, backend, records=extras, only_jobs=['c5_rows', 'c3s_split', 'c5_split', 'c3s_zones', 'c4s_zones'],
				t_start=time.perf_counter() - spent_first, deadline=deadline)
			one_gpu(['c3s'])
	except Exception as e:
		if watchdog is None:
			raise
		traceback.print_exc()
		give_up('the supplementary blocks ended with %s: %s (rank %d); the headline and the records above were measured before that' % (type(e).__name__, e, rank))
	if watchdog is not None:
		watchdog.cancel()
	if rank == 0:
		finish()
	if world > 1 or force_dist:
		dist.destroy_process_group()


if __name__ == '__main__':
	main()
