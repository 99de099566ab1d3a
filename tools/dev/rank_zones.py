import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from nway_amd import distributed, _hip
dev = torch.device('cuda', 0)
for n0, n1 in ((62500, 12500000), (100000, 10000000)):
	tabs = list(bench.make_workload(n0, n1, 78))
	for zpr in (1, 2, 4, 8):
		eng = distributed.ZoneShardedMatch(tabs[0], tabs[1:], 5.0, 0.9, dev, zones_per_rank=zpr, local_only=True)
		for _ in range(30):
			eng.step()
		torch.cuda.synchronize(dev)
		best = 1e9
		for rep in range(3):
			t0 = time.perf_counter()
			for _ in range(40):
				eng.step()
			torch.cuda.synchronize(dev)
			best = min(best, (time.perf_counter() - t0) * 1e6 / 40)
		print('%d x %d, %d zones: %.1f us per pass (best of 3 x 40), batched %s' % (n0, n1, zpr, best, eng.batched))
		eng.close()
