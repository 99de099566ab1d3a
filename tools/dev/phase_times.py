"""development aid: where the time of the sparse path's three kernels goes, from wall-clock stamps
left by thread 0 of every workgroup (common.inc: dbg_stamp; NWAYHIP_DBG_PTR).

    python tools/dev/phase_times.py            (on the GPU box)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
dev = torch.device('cuda', 0)
BLOCKS, STAMPS = 1024, 8
buf = torch.zeros(3 * BLOCKS * STAMPS, dtype=torch.int64, device=dev)
os.environ["NWAYHIP_DEV"] = "1"  # the library reads its development switches only then (plan.inc: env_int)
os.environ["NWAYHIP_DBG_PTR"] = str(buf.data_ptr())
os.environ.setdefault("NWAYHIP_LIBRARY", os.path.join(ROOT, "tools", "dev", "bin", "lib_dev.so"))  # a -DNWAYHIP_DEVBUILD build carries the stamps
import bench  # noqa: E402
import nway_amd  # noqa: E402
from nway_amd import _hip  # noqa: E402

three = len(sys.argv) > 1 and sys.argv[1] == 'c4s'   # BASELINE configs[3]: 3-way 1e5 x 1e6 x 1e6, 10 arcsec (tail: k_tail3q)
if three:
	tables = bench.make_workload3(100000, 1000000, 1000000, 3)
	radius = 10.
else:
	n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
	n1 = int(sys.argv[2]) if len(sys.argv) > 2 else 10000000
	primary, secondary = bench.make_workload(n0, n1, 1)
	tables = [primary, secondary]
	radius = 5.
k = len(tables)
log = nway_amd.NullOutputLogger()
err = radius / 3600
scheme = nway_amd.choose_scheme([(t['ra'], t['dec']) for t in tables], err)
dens, dens_plus = nway_amd._compute_source_densities(tables, log)
comp = nway_amd._completeness_vector(0.9, k)
params = _hip.make_params(k, scheme, radius, err, dens, dens_plus, nway_amd._prior_table(dens, dens_plus, comp))
cats = [_hip.DeviceCatalogue(t['ra'], t['dec'], np.asarray(t['error'], dtype=float), dev) for t in tables]
sizes = [c.n for c in cats]
cap_pairs, cap_rows = nway_amd._estimate_capacities(sizes, [bench.SKY_AREA] * k, radius, scheme, True)
plan, st = _hip.run_plan(sizes, params, cats, cap_pairs, cap_rows, dev, lean=True)
for _ in range(5):
	plan.enqueue(cats)
torch.cuda.synchronize()
fused = os.environ.get('NWAYHIP_FUSED_FRONT', '0') == '1'  # (front.inc: the registration inside the sweep launch)
names = {0: ('k_register_x', ['start', None, 'claims + stores landed', 'end']),
	1: ('k_sweep', ['start', 'bitmap in LDS', 'wave 0 done streaming', 'all waves done', 'probes landed', 'end', 'barrier passed', 'parked tiles tested']),
	2: ('k_tail2', ['start', 'cnt/slot/sigma landed', 'block scan', 'lookback', 'rows landed', 'group stats landed', 'items set up', 'separations done'])}
if three:
	names[2] = ('k_tail3q', ['start', 'exact tests done', 'counts ready (scan word)', 'rows + statistics in registers', 'look-back done', 'rows written'])
if fused:
	names[0] = ('registration inside the sweep launch (times since the first SWEEP workgroup started)', [None, 'slice parked (all waves)', 'claims + stores landed', 'announced'])
acc = {}
for rep in range(10):
	buf.zero_()
	plan.enqueue(cats)
	torch.cuda.synchronize()
	t = buf.cpu().numpy().reshape(3, BLOCKS, STAMPS)
	for k, (kname, labels) in names.items():
		ref = 1 if (fused and k == 0) else k
		used = t[ref][:, 0] > 0
		tk = t[k][used][:, :len(labels)].astype(np.float64) * 0.01  # 100 MHz -> us
		t0 = (t[ref][used][:, 0].astype(np.float64) * 0.01).min()
		acc.setdefault(k, []).append((tk - t0, used.sum()))
for k, (kname, labels) in names.items():
	rel = np.stack([a for a, _ in acc[k]])  # reps x blocks x stamps
	print('%s: %d workgroups; us since the first workgroup started (mean over workgroups | latest workgroup), mean of 10 runs' % (kname, acc[k][0][1]))
	for i, lab in enumerate(labels):
		if lab is not None and (fused or lab not in ('barrier passed', 'parked tiles tested')):
			print('    %-26s %7.2f | %7.2f' % (lab, rel[:, :, i].mean(), rel[:, :, i].max(axis=1).mean()))
t = buf.cpu().numpy().reshape(3, BLOCKS, STAMPS).astype(np.float64) * 0.01
# distribution over the workgroups of the last run (the scan of the tail waits for its SLOWEST predecessor)
for k, (kname, labels) in names.items():
	ref = 1 if (fused and k == 0) else k
	used = t[ref][:, 0] > 0
	t0 = t[ref][used][:, 0].min()
	for i, lab in enumerate(labels):
		if lab is None or i >= t[k].shape[1]:
			continue
		v = t[k][used][:, i] - t0
		if (t[k][used][:, i] > 0).sum() == 0:
			continue
		order = np.argsort(v)
		print('%-14s %-26s p10 %6.2f p50 %6.2f p90 %6.2f p99 %6.2f max %6.2f  slowest workgroups %s' % (kname[:14], lab, np.percentile(v, 10), np.percentile(v, 50),
			np.percentile(v, 90), np.percentile(v, 99), v.max(), list(np.flatnonzero(used)[order[-6:]])))
first = t[1][t[1][:, 0] > 0][:, 0].min() if fused else t[0][t[0][:, 0] > 0][:, 0].min()
print('plan: %s' % plan.description)
print('last run: %s start -> sweep start %.2f us, sweep start -> tail start %.2f us, tail start -> tail end %.2f us' % (
	'sweep' if fused else 'register', t[1][t[1][:, 0] > 0][:, 0].min() - first, t[2][t[2][:, 0] > 0][:, 0].min() - t[1][t[1][:, 0] > 0][:, 0].min(),
	t[2][t[2][:, 0] > 0][:, 5].max() - t[2][t[2][:, 0] > 0][:, 0].min()))
