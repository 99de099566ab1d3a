"""development aid: does a captured hipGraph of the pass (three launches) run faster than the same launches enqueued one by one?
The bench workload (C3-S), three copies of the secondary catalogue; a graph holds 6 consecutive passes (the plan alternates over two
scratch copies, the bench over three secondary buffers).   python tools/dev/graph_probe.py [rounds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import nway_amd
from nway_amd import _hip

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
device = torch.device('cuda', 0)
torch.cuda.set_device(device)
primary, secondary = bench.make_workload(100000, 10000000, 1)
tables = [primary, secondary]
radius = 5.0
err = radius / 3600.
scheme = nway_amd.choose_scheme([(t['ra'], t['dec']) for t in tables], err)
log = nway_amd.NullOutputLogger()
dens, dens_plus = nway_amd._compute_source_densities(tables, log)
comp = nway_amd._completeness_vector(0.9, 2)
params = _hip.make_params(2, scheme, radius, err, dens, dens_plus, nway_amd._prior_table(dens, dens_plus, comp))
cats = [_hip.DeviceCatalogue(t['ra'], t['dec'], np.asarray(t['error'], dtype=float), device) for t in tables]
sizes = [c.n for c in cats]
sec_copies = [cats[1]]
for _ in range(2):
	cp = _hip.DeviceCatalogue.__new__(_hip.DeviceCatalogue)
	cp.ra, cp.dec = cats[1].ra.clone(), cats[1].dec.clone()
	cp.sigma = None
	cp.sigma_const, cp.n = cats[1].sigma_const, cats[1].n
	sec_copies.append(cp)
cap_pairs, cap_rows = nway_amd._estimate_capacities(sizes, [bench.SKY_AREA] * 2, radius, scheme, True)
plan, st = _hip.run_plan(sizes, params, cats, cap_pairs, cap_rows, device, lean=True)
rows = int(st[_hip.ST_ROWS])
print('plan:', plan.description, 'rows', rows)
stream = torch.cuda.Stream(device=device)
counter = [0]


def one_pass():
	plan.enqueue([cats[0], sec_copies[counter[0] % 3]])
	counter[0] += 1


with torch.cuda.stream(stream):
	for _ in range(300):
		one_pass()
	stream.synchronize()
	counter[0] = 0
	graph = torch.cuda.CUDAGraph()
	try:
		with torch.cuda.graph(graph, stream=stream):
			for _ in range(6):
				one_pass()
	except Exception as e:
		print('capture failed:', type(e).__name__, e)
		sys.exit(0)
	for r in range(rounds):
		for _ in range(30):
			one_pass()
		stream.synchronize()
		t0 = time.perf_counter()
		for _ in range(120):
			one_pass()
		stream.synchronize()
		plain = (time.perf_counter() - t0) / 120 * 1e6
		for _ in range(5):
			graph.replay()
		stream.synchronize()
		t0 = time.perf_counter()
		for _ in range(20):
			graph.replay()
		stream.synchronize()
		g = (time.perf_counter() - t0) / 120 * 1e6
		st = plan.read_status()
		print('round %d: plain %.2f us per pass, graph %.2f us per pass (rows %d, flags %d)' % (r, plain, g, int(st[_hip.ST_ROWS]), int(st[_hip.ST_FLAGS])))
