"""development aid: is a pass of the several-zones engine bound by the host's enqueue loop?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np, torch
import bench
from nway_amd import distributed
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
tabs = list(bench.make_workload(500000, 100000000, 78))
def measure(tag, eng):
	for _ in range(30): eng.step()
	torch.cuda.synchronize(dev)
	t0 = time.perf_counter()
	for _ in range(20): eng.step()
	t1 = time.perf_counter()
	torch.cuda.synchronize(dev)
	t2 = time.perf_counter()
	print('%s: host enqueue %.1f us per pass, until the device is done %.1f us per pass' % (tag, (t1 - t0) * 1e6 / 20, (t2 - t0) * 1e6 / 20), flush=True)
eng = distributed.ZoneShardedMatch(tabs[0], tabs[1:], 5., 0.9, dev, zones_per_rank=8, streams=2, local_only=True)
measure('fresh process', eng)
import nway_oracle_c
p, s = bench.make_workload(20000, 2000000, 1)
nway_oracle_c.nway_match([p, dict(s, error=s['error'] * np.ones(len(s['ra'])))], 5., 0.9, threads=0)
measure('after an OpenMP region in this process', eng)
time.sleep(0.5)
measure('half a second later', eng)
