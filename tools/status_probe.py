"""print the status words and per-stage timings of one bench-workload run (GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import nway_amd
from nway_amd import _hip
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
n1 = int(sys.argv[2]) if len(sys.argv) > 2 else 10000000
radius = float(sys.argv[3]) if len(sys.argv) > 3 else 5.0
prim, sec = bench.make_workload(n0, n1, 1)
tables = [prim, sec]
dev = torch.device('cuda', 0)
log = nway_amd.NullOutputLogger()
err = radius / 3600.
scheme = nway_amd.choose_scheme([(t['ra'], t['dec']) for t in tables], err)
dens, dp = nway_amd._compute_source_densities(tables, log)
comp = nway_amd._completeness_vector(0.9, 2)
params = _hip.make_params(2, scheme, radius, err, dens, dp, nway_amd._prior_table(dens, dp, comp))
cats = [_hip.DeviceCatalogue(t['ra'], t['dec'], np.asarray(t['error'], dtype=float), dev) for t in tables]
plan, st = _hip.run_plan([c.n for c in cats], params, cats, 500000, 500000, dev)
print('status', [int(x) for x in st[:4]], 'surv', int(st[8]), 'pairs', int(st[16]), 'notflat', [int(x) for x in st[24:26]])
plan.profile(0xff)
for _ in range(10):
	plan.enqueue(cats)
torch.cuda.synchronize()
n, ms = plan.profile_read()
print('stages us:', ' '.join('%s=%.1f' % (nm, 1e3 * m / max(k, 1)) for nm, k, m in zip(_hip.STAGE_NAMES, n, ms)))
