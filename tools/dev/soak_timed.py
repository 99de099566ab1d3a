import os, sys, time
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import numpy as np
import test_hip_fuzz as fz
import nway_amd as nw
lo, hi = int(sys.argv[1]), int(sys.argv[2])
for seed in range(lo, hi):
	rng = np.random.default_rng(1000 + seed)
	k = int(rng.integers(2, 6))
	tabs, radius = (fz.flat_case if seed % 2 == 0 else fz.sphere_case)(rng, k)
	if seed % 2 == 1 and k > 4:
		tabs = tabs[:4]
	comp = float(rng.choice([1.0, 0.9, 0.5]))
	t0=time.time()
	try:
		rows = fz.compare(nw, tabs, radius, comp, 'cli' if seed % 3 == 0 else 'api', f32=(seed % 5 == 0))
		print(seed, k, len(tabs), rows, '%.2f s' % (time.time()-t0), flush=True)
	except Exception as e:
		print(seed, k, 'FAILED', str(e).strip().splitlines()[0][:150], '%.2f s' % (time.time()-t0), flush=True)
