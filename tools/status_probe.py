"""Per-stage timings and status words of one workload on the GPU (development aid).

    python tools/status_probe.py [c3s|c3d|c4s|c4d] [n_primary] [n_secondary] [radius]

c3s: 2-way uniform sky (bench workload)      c3d: 2-way dense 6 deg^2 patch (flat cells)
c4s: 3-way uniform sky, 1e5 x 1e6 x 1e6     c4d: 3-way dense 8 deg^2 patch     c6d: 4-way dense (hybrid path)
c1x / c2x: BASELINE configs[0] / [1] on the stand-ins (C1', C2'): the reference's real COSMOS_XMM (1 797 sources)
x seeded OPT (560 536) [x IRAC (345 512)] in 2 deg^2, radius 20 arcsec
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import nway_amd
from nway_amd import _hip

args = sys.argv[1:]
config = args.pop(0) if args and not args[0].isdigit() else 'c3s'
n0 = int(args[0]) if len(args) > 0 else 100000
n1 = int(args[1]) if len(args) > 1 else (10000000 if config.startswith('c3') else 1000000)
radius = float(args[2]) if len(args) > 2 else (5.0 if config.startswith('c3') else (20.0 if config in ('c1x', 'c2x') else 10.0))
force_slots = int(os.environ.get('PROBE_LINK_SLOTS', '0'))
rng = np.random.default_rng(3)


def patch_catalogue(name, n, half, sigma, parents=None, frac=0.0, psig=None):
	ra = rng.uniform(150 - half, 150 + half, size=n)
	dec = rng.uniform(2 - half, 2 + half, size=n)
	if parents is not None:
		m = int(frac * len(parents['ra']))
		slots = rng.choice(n, size=m, replace=False)
		dec[slots] = parents['dec'][:m] + rng.normal(0, 1, size=m) * psig[:m] / 3600.
		ra[slots] = parents['ra'][:m] + rng.normal(0, 1, size=m) * psig[:m] / 3600. / np.cos(np.radians(parents['dec'][:m]))
	return dict(name=name, ra=ra, dec=dec, error=sigma, area=(2 * half)**2, mags=[], maghists=[], magnames=[])


def sphere_catalogue(name, n, sigma, parents=None, frac=0.0, psig=None):
	ra, dec = bench.uniform_sphere(rng, n)
	if parents is not None:
		m = int(frac * len(parents['ra']))
		slots = rng.choice(n, size=m, replace=False)
		dec[slots] = np.clip(parents['dec'][:m] + rng.normal(0, 1, size=m) * psig[:m] / 3600., -90, 90)
		ra[slots] = (parents['ra'][:m] + rng.normal(0, 1, size=m) * psig[:m] / 3600. / np.maximum(np.cos(np.radians(parents['dec'][:m])), 1e-6)) % 360
	return dict(name=name, ra=ra, dec=dec, error=sigma, area=bench.SKY_AREA, mags=[], maghists=[], magnames=[])


if config == 'c3s':
	prim, sec = bench.make_workload(n0, n1, 1)
	tables = [prim, sec]
elif config == 'c3d':
	psig = rng.uniform(0.3, 1.5, size=n0)
	prim = patch_catalogue('P', n0, 1.23, psig)
	tables = [prim, patch_catalogue('S', n1, 1.23, 0.1, prim, 0.8, psig)]
elif config == 'c4s':
	psig = np.ones(n0)
	prim = sphere_catalogue('P', n0, psig)
	tables = [prim, sphere_catalogue('A', n1, 0.1, prim, 0.8, psig), sphere_catalogue('B', n1, 0.5, prim, 0.6, psig)]
elif config == 'c4d':
	psig = np.ones(n0)
	prim = patch_catalogue('P', n0, 1.42, psig)
	tables = [prim, patch_catalogue('A', n1, 1.42, 0.1, prim, 0.8, psig), patch_catalogue('B', n1, 1.42, 0.5, prim, 0.6, psig)]
elif config == 'c6d':  # 4-way dense: 1e5 x 5e5 x 5e5 x 5e5 in 8 deg^2 (the hybrid path)
	psig = np.ones(n0)
	prim = patch_catalogue('P', n0, 1.42, psig)
	tables = [prim, patch_catalogue('A', n1 // 2, 1.42, 0.1, prim, 0.8, psig), patch_catalogue('B', n1 // 2, 1.42, 0.5, prim, 0.6, psig),
		patch_catalogue('C', n1 // 2, 1.42, 0.3, prim, 0.5, psig)]
elif config in ('c1x', 'c2x'):
	sys.path.insert(0, os.path.join(ROOT, 'tests'))
	from goldenutil import xmm_tables
	tables = list(xmm_tables())[:2 if config == 'c1x' else 3]
else:
	raise SystemExit('unknown config ' + config)

k = len(tables)
dev = torch.device('cuda', 0)
phases = None
if os.environ.get('NWAYHIP_PHASES'):  # development: wall-clock stamps of the last fused kernel's workgroups (common.inc: dbg_stamp)
	phases = torch.zeros(3 * 1024 * 8, dtype=torch.int64, device=dev)
	os.environ['NWAYHIP_DEV'] = '1'
	os.environ['NWAYHIP_DBG_PTR'] = str(phases.data_ptr())
log = nway_amd.NullOutputLogger()
err = radius / 3600.
scheme = nway_amd.choose_scheme([(t['ra'], t['dec']) for t in tables], err)
dens, dp = nway_amd._compute_source_densities(tables, log)
comp = nway_amd._completeness_vector(0.9, k)
params = _hip.make_params(k, scheme, radius, err, dens, dp, nway_amd._prior_table(dens, dp, comp), link_slots=force_slots,
	correction=int(os.environ.get('PROBE_CORRECTION', '0')))
cats = [_hip.DeviceCatalogue(t['ra'], t['dec'], np.asarray(t['error'], dtype=float), dev) for t in tables]
sizes = [c.n for c in cats]
cap_pairs, cap_rows = nway_amd._estimate_capacities(sizes, [t['area'] for t in tables], radius, scheme, True)
plan, st = _hip.run_plan(sizes, params, cats, cap_pairs, cap_rows, dev, lean=True)
print('path:', ('general', 'sparse', 'hybrid')[plan.path], 'link_slots', plan.link_slots)
print('plan:', ' '.join('%s=%s' % kv for kv in sorted(plan.description.items())), 'attempts=%d' % plan.attempts)
print(config, 'scheme', scheme, 'status', [int(x) for x in st[:4]], 'surv', [int(x) for x in st[8:8 + k - 1]],
	'pairs', [int(x) for x in st[16:16 + k - 1]], 'notflat', [int(x) for x in st[24:24 + k]])
import time
for _ in range(3):
	plan.enqueue(cats)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
	plan.enqueue(cats)
torch.cuda.synchronize()
print('wall (no stage events): %.1f us/step  rows/s=%.3g' % ((time.perf_counter() - t0) / 20 * 1e6, int(st[0]) / ((time.perf_counter() - t0) / 20)))
plan.profile(0xff)
reps = 10
for _ in range(reps):
	plan.enqueue(cats)
torch.cuda.synchronize()
n, ms = plan.profile_read()
stage_us = [1e3 * m / reps for m in ms]
print('stages us/step:', ' '.join('%s=%.1f' % (nm, v) for nm, v in zip(_hip.STAGE_NAMES, stage_us)),
	'| total=%.1f us  rows/s=%.3g' % (sum(stage_us), int(st[0]) / (sum(stage_us) * 1e-6)))

if phases is not None:
	phases.zero_()
	plan.enqueue(cats)
	torch.cuda.synchronize()
	t = phases.cpu().numpy().reshape(3, 1024, 8)[2].astype(np.float64) * 0.01
	used = t[:, 0] > 0
	t = t[used]
	n = int((t > 0).all(axis=0).sum())
	print('tail kernel, first %d workgroups: us between stamps, mean | max' % used.sum())
	for i in range(1, n):
		d = t[:, i] - t[:, i - 1]
		print('   stamp %d -> %d: %7.2f | %7.2f' % (i - 1, i, d.mean(), d.max()))
	print('   start of the first to end of the last: %.1f us' % (t[:, n - 1].max() - t[:, 0].min()))
