#!/usr/bin/env python
"""bench.py -- throughput of the nway hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the "HBM-roofline config"; configs[1] needs the two
COSMOS catalogues that are missing from the reference checkout): synthetic 2-way match,
1e5 primaries x 1e7 secondaries uniform on the sphere, 5 arcsec radius, 80 % of the
primaries with a true counterpart ("C3-S").

A step = one pass of the whole hot path over the resident catalogues: primary cell
registration, secondary sweep, separations, neighbour lists, tuple expansion, Bayes
factors / priors / posteriors, per-primary p_any / p_i / match_flag.  Inputs are resident
in HBM when the timed region starts; nothing inside the region synchronises with the host.

metric: candidate Bayes-factor evaluations per second = rows of the match table produced
per second (one row = one match hypothesis incl. the no-counterpart rows), whole job.

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline     dominant kernel (the secondary sweep, HBM bound): algorithmic bytes per launch
               (16 B per secondary: ra + dec read once) / mean launch duration measured with
               HIP events on the pipeline's stream during the timed region.
  cpu_baseline the C restatement of the oracle (oracle/nway_oracle.c, "port") timed on this
               host, 1 thread, on a bounded sample of the same workload.
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


def cpu_baseline(primary, secondary, radius, completeness, sample_secondaries):
	"""C port of the oracle on a bounded sample (all primaries x the first
	``sample_secondaries`` secondaries), 1 thread"""
	sys.path.insert(0, os.path.join(ROOT, 'oracle'))
	import nway_oracle_c
	nway_oracle_c.load()
	n = min(sample_secondaries, len(secondary['ra']))
	sec = dict(secondary, ra=secondary['ra'][:n], dec=secondary['dec'][:n], error=secondary['error'] * np.ones(n))
	t0 = time.perf_counter()
	table = nway_oracle_c.nway_match([primary, sec], radius, completeness)
	dt = time.perf_counter() - t0
	rows = len(table['ncat'])
	return dict(value=rows / dt, unit='candidate evaluations/s', cores=1, kind='port',
		sample='oracle/nway_oracle.c, 1 thread: all %d primaries x first %d secondaries of the same workload, %d rows in %.2f s'
		% (len(primary['ra']), n, rows, dt),
		reference_note='the reference itself (pure Python, one thread; it cannot travel to the GPU box) measured in the build container '
			'on its own fixtures: 8.4e4 rows/s (tests/elltest 2-way, 37 706 rows in 0.45 s), 5.3e4 rows/s (3-way, 450 435 rows in 8.5 s)'), table


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument('--gpus', type=int, default=1)
	ap.add_argument('--steps', type=int, default=20)
	ap.add_argument('--warmup', type=int, default=3)
	ap.add_argument('--n-primary', type=int, default=100000)
	ap.add_argument('--n-secondary', type=int, default=10000000)
	ap.add_argument('--radius', type=float, default=5.0)
	ap.add_argument('--completeness', type=float, default=0.9)
	ap.add_argument('--seed', type=int, default=1)
	ap.add_argument('--cpu-sample', type=int, default=10000000, help='secondaries in the CPU baseline sample (0 = skip)')
	ap.add_argument('--profile-stages', action='store_true', help='also time every stage (adds event records to the region)')
	ap.add_argument('--streams', type=int, default=int(os.environ.get('NWAY_BENCH_STREAMS', '1')),
		help='independent pipelines (own workspace, own output table, own HIP stream) the steps alternate over')
	args = ap.parse_args()

	import torch
	import nway_amd
	from nway_amd import _hip

	world = int(os.environ.get('WORLD_SIZE', '1'))
	rank = int(os.environ.get('RANK', '0'))
	local_rank = int(os.environ.get('LOCAL_RANK', '0'))
	ngpu = max(torch.cuda.device_count(), 1)
	if world > 1:
		import torch.distributed as dist
		# "nccl" is RCCL on ROCm.  NWAY_BENCH_BACKEND=gloo lets several ranks share one GPU
		# (functional testing of the sharded path on a 1-GPU box only).
		backend = os.environ.get('NWAY_BENCH_BACKEND', 'nccl')
		torch.cuda.set_device(local_rank % ngpu)
		if backend == 'nccl':
			dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank % ngpu))
		else:
			dist.init_process_group(backend)
	if args.gpus != world:
		if rank == 0:
			sys.stderr.write('note: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE\n' % (args.gpus, world))
	device = torch.device('cuda', (local_rank % ngpu) if world > 1 else 0)
	torch.cuda.set_device(device)

	# weak scaling: every rank owns n_primary primaries (a contiguous row shard of the global
	# primary catalogue, N x n_primary rows) and loads a 1/world slice of the ONE secondary
	# catalogue (n_secondary rows in total); the slices are all-gathered once at set-up so that
	# every GPU holds the whole secondary catalogue, then each step matches the rank's primary
	# shard against it -- per-GPU work is fixed, no collective on the per-step path.
	n_sec_local = args.n_secondary // world + (args.n_secondary % world if rank == world - 1 else 0)
	primary, secondary = make_workload(args.n_primary, n_sec_local, args.seed + 1000 * rank)
	if world > 1:
		from nway_amd import distributed
		engine = distributed.ShardedMatch(primary, [secondary], args.radius, args.completeness, device)
	else:
		engine = None

	if engine is None:
		log = nway_amd.NullOutputLogger()
		tables = [primary, secondary]
		err = args.radius / 60. / 60
		scheme = nway_amd.choose_scheme([(t['ra'], t['dec']) for t in tables], err)
		dens, dens_plus = nway_amd._compute_source_densities(tables, log)
		comp = nway_amd._completeness_vector(args.completeness, 2)
		params = _hip.make_params(2, scheme, args.radius, err, dens, dens_plus, nway_amd._prior_table(dens, dens_plus, comp))
		cats = [_hip.DeviceCatalogue(t['ra'], t['dec'], np.asarray(t['error'], dtype=float), device) for t in tables]
		sizes = [c.n for c in cats]
		# settle capacities with one untimed run
		cap_pairs, cap_rows = nway_amd._estimate_capacities(sizes, [SKY_AREA, SKY_AREA], args.radius, scheme, True)
		plan, st = _hip.run_plan(sizes, params, cats, cap_pairs, cap_rows, device, lean=True)
		rows_per_step = int(st[_hip.ST_ROWS])
		# every step is a complete, independent pass; with --streams S the steps alternate over S
		# pipelines (workspace + output table + HIP stream each) so that the latency-bound stages of
		# one pass overlap the HBM-bound sweep of another
		plans = [plan] + [_hip.MatchPlan(sizes, params, plan.cap_pairs, plan.cap_rows, device, lean=True) for _ in range(args.streams - 1)]
		streams = [torch.cuda.Stream(device=device) for _ in plans] if len(plans) > 1 else [None]
		counter = [0]

		def step():
			i = counter[0] % len(plans)
			counter[0] += 1
			if streams[i] is None:
				plans[i].enqueue(cats)
			else:
				with torch.cuda.stream(streams[i]):
					plans[i].enqueue(cats)
		read_status = plan.read_status
	else:
		step = engine.step
		read_status = engine.read_status
		engine.step()
		rows_per_step = engine.total_rows()
		plan = engine.plan
		plans = [plan]

	def barrier():
		torch.cuda.synchronize(device)
		if world > 1:
			dist.barrier()
			torch.cuda.synchronize(device)

	for _ in range(args.warmup):
		step()
	mask = (1 << _hip.STAGES) - 1 if args.profile_stages else (1 << 1)
	for pl in plans:
		pl.profile(mask)
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
	for pl in plans:
		n_, ms_ = pl.profile_read()
		launches = [a + b for a, b in zip(launches, n_)]
		ms = [a + b for a, b in zip(ms, ms_)]
		pl.profile(0)
	if world > 1:
		tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
		dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
		elapsed = float(tmax.item())
		rows_per_step = engine.total_rows()
	st = read_status()
	assert int(st[_hip.ST_FLAGS]) == 0, 'overflow flags set: %d' % int(st[_hip.ST_FLAGS])
	ms_per_step = elapsed * 1e3 / args.steps

	if rank == 0:
		n_sec_swept = int(plan.sizes[1])
		sweep_ms = ms[1] / max(launches[1], 1)
		alg_bytes = 16.0 * n_sec_swept
		achieved = alg_bytes / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
		traffic = None
		tf = os.path.join(ROOT, 'profiles', 'sweep_traffic.json')
		if os.path.exists(tf):
			try:
				rec = json.load(open(tf))
				if rec.get('n_secondary') == n_sec_swept:
					traffic = rec.get('hbm_bytes_per_launch')
			except Exception:
				traffic = None
		out = dict(metric='candidate Bayes-factor evals/s', value=rows_per_step / (ms_per_step * 1e-3), unit='candidate evaluations/s',
			n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=ms_per_step, higher_is_better=True, scaling='weak',
			vs_baseline=None, dtype='f64', data='synthetic',
			config=dict(workload='C3-S synthetic 2-way: %d primaries per GPU (%d in total) x %d secondaries (whole catalogue resident on every GPU), '
				'uniform sky, radius %g arcsec, completeness %g, seed %d'
				% (args.n_primary, args.n_primary * world, n_sec_swept, args.radius, args.completeness, args.seed),
				rows_per_step=rows_per_step, distance_tests_per_step_rank0=int(st[_hip.ST_TESTS]),
				survivors_per_step_rank0=int(st[_hip.ST_SURVIVORS]), registrations_rank0=int(st[_hip.ST_REGISTRATIONS]),
				parallelism='primary-row shards x%d' % world, streams=len(plans),
				setup_allgatherv=(None if engine is None else dict(seconds=engine.setup_seconds, bytes=engine.gathered_bytes,
					note='one-time all-gatherv of the secondary columns (RCCL), outside the timed steps'))),
			roofline=dict(bound='hbm', kernel='k_sweep', achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s',
				frac=achieved / HBM_PEAK_GBS, traffic=traffic, algorithmic_bytes_per_launch=alg_bytes,
				launch_ms=sweep_ms, launches_timed=int(launches[1])))
		if args.profile_stages:
			out['stages_ms'] = dict((name, ms[i] / max(launches[i], 1) * (launches[i] / float(args.steps)))
				for i, name in enumerate(_hip.STAGE_NAMES))
		if world == 1 and args.cpu_sample > 0:
			out['cpu_baseline'], _ = cpu_baseline(primary, secondary, args.radius, args.completeness, args.cpu_sample)
		else:
			out['cpu_baseline'] = None
		print(json.dumps(out))
	if world > 1:
		dist.destroy_process_group()


if __name__ == '__main__':
	main()
