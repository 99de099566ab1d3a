"""What every rank of an N-GPU declination-zone job (nway_amd.distributed.ZoneShardedMatch) costs per step, measured on ONE GPU:
the zones of the N ranks are cut exactly as the engine cuts them (quantiles of the declinations of the largest secondary
catalogue, the secondaries within the match radius of a zone go to it), every zone's pass -- its primaries against its
secondaries, densities of the whole job -- runs on the one GPU, one after the other.  The mode has NO collective on the per-step
path, so a step of the real job takes what its slowest rank takes: the strong-scaling factor printed is
(pass of the whole job on one GPU) / (slowest zone), both measured here.  The tables of the zones together are checked against
the table of the whole job (row count, checksum of the index columns).

    python tools/zone_shard_costs.py [c3s|c4s|c5] ...      (on the GPU box; default: all three)
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import nway_amd  # noqa: E402
from nway_amd import _hip  # noqa: E402

dev = torch.device('cuda', 0)
log = nway_amd.NullOutputLogger()
JOBS = dict(c3s=('C3-S 2-way 1e5 x 1e7, 5"', [100000, 10000000], 5.0), c4s=('C4-S 3-way 1e5 x 1e6 x 1e6, 10" (BASELINE configs[3])', [100000, 1000000, 1000000], 10.0),
	c5=('C5 2-way 5e5 x 1e8, 5" (BASELINE configs[4])', [500000, 100000000], 5.0))


def timed_pass(tables, sizes_global, radius, steps=30):
	"""(us per pass, rows, checksum of the index columns with `gidx` applied) of the pass over `tables` with the densities of the whole job"""
	k = len(tables)
	err = radius / 3600.
	dens, dens_plus = nway_amd._densities_from_sizes([t['name'] for t in tables], sizes_global, [bench.SKY_AREA] * k, log)
	comp = nway_amd._completeness_vector(0.9, k)
	params = _hip.make_params(k, _hip.SCHEME_SPHERE, radius, err, dens, dens_plus, nway_amd._prior_table(dens, dens_plus, comp))
	cats = [_hip.DeviceCatalogue(t['ra'], t['dec'], np.asarray(t['error'], dtype=float), dev) for t in tables]
	sizes = [c.n for c in cats]
	areas = [bench.SKY_AREA * max(n, 1) / g for n, g in zip(sizes, sizes_global)]
	cap_pairs, cap_rows = nway_amd._estimate_capacities(sizes, areas, radius, _hip.SCHEME_SPHERE, True)
	plan, st = _hip.run_plan(sizes, params, cats, cap_pairs, cap_rows, dev, lean=True)
	for _ in range(10):
		plan.enqueue(cats)
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	for _ in range(steps):
		plan.enqueue(cats)
	torch.cuda.synchronize()
	us = (time.perf_counter() - t0) / steps * 1e6
	st = plan.read_status()
	assert int(st[_hip.ST_FLAGS]) == 0, st[:4]
	m = int(st[_hip.ST_ROWS])
	check = 0
	for c, t in enumerate(tables):
		idx = _hip.to_host(plan.cols['idx'][c][:m]).astype(np.int64)
		g = t.get('gidx')
		if g is not None:
			idx = np.where(idx >= 0, g[np.maximum(idx, 0)], -1)
		check += int(((idx + 2) * (1000003 + 7919 * c) % 2147483647).sum())
	desc = plan.description
	plan.close()
	del cats
	torch.cuda.empty_cache()
	return us, m, check, desc


for job in (sys.argv[1:] or ['c3s', 'c4s', 'c5']):
	title, sizes, radius = JOBS[job]
	tables = list(bench.make_workload(sizes[0], sizes[1], 1)) if len(sizes) == 2 else bench.make_workload3(sizes[0], sizes[1], sizes[2], 1)
	whole_us, whole_rows, whole_check, desc = timed_pass(tables, sizes, radius)
	print('## %s: the whole job on one GPU %.1f us per pass, %d rows (%s sweep, %s tail)' % (title, whole_us, whole_rows, desc['sweep'], desc['tail']))
	print('| GPUs N | zone passes (us, every rank of the job, one after the other on this GPU) | slowest | secondaries per zone (of %s) | rows of all zones | strong-scaling factor = whole / slowest |' % (
		' + '.join('%d' % n for n in sizes[1:])))
	print('|---|---|---|---|---|---|')
	margin = radius / 3600. * (1 + 1e-9) + 1e-12
	big = 1 + int(np.argmax(sizes[1:]))
	for world in (2, 4, 8):
		dec = tables[big]['dec']
		edges = np.quantile(dec[np.isfinite(dec)], [z / world for z in range(1, world)])
		times, rows, check, nsec = [], 0, 0, []
		for z in range(world):
			zt = []
			for c, t in enumerate(tables):
				m = margin if c > 0 else 0.0
				z_lo = np.searchsorted(edges, t['dec'] - m, side='right')
				z_hi = np.searchsorted(edges, t['dec'] + m, side='right')
				pick = np.flatnonzero((z_lo <= z) & (z <= z_hi))
				zt.append(dict(t, ra=np.ascontiguousarray(t['ra'][pick]), dec=np.ascontiguousarray(t['dec'][pick]),
					error=(t['error'] if np.ndim(t['error']) == 0 else np.ascontiguousarray(np.asarray(t['error'])[pick])), gidx=pick))
			us, m, ck, _ = timed_pass(zt, sizes, radius)
			times.append(us)
			rows += m
			check += ck
			nsec.append(sum(len(t['ra']) for t in zt[1:]))
		assert rows == whole_rows and check == whole_check, (rows, whole_rows, check, whole_check)
		print('| %d | %s | %.1f | %d .. %d | %d (= the whole job\'s, index columns equal) | %.2f |' % (world, ' '.join('%.0f' % x for x in times), max(times),
			min(nsec), max(nsec), rows, whole_us / max(times)))
	print()
