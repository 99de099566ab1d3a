"""development aid: the randomised parity test of tests/test_hip_fuzz.py over a seed range
    python tools/dev/soak.py 40 400        (on the GPU box)
"""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import test_hip_fuzz as fz
import nway_amd as nw

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad, t0, rows = [], time.time(), 0
for seed in range(lo, hi):
	rng = np.random.default_rng(1000 + seed)
	k = int(rng.integers(2, 6))
	tabs, radius = (fz.flat_case if seed % 2 == 0 else fz.sphere_case)(rng, k)
	if seed % 2 == 1 and k > 4:
		tabs = tabs[:4]
	comp = float(rng.choice([1.0, 0.9, 0.5]))
	excused0 = fz.TIE_EXCUSES['rows']
	try:
		rows += fz.compare(nw, tabs, radius, comp, 'cli' if seed % 3 == 0 else 'api', f32=(seed % 5 == 0))
	except Exception as e:
		bad.append(seed)
		print('seed %d FAILED: %s' % (seed, str(e).strip().splitlines()[0][:200]))
		if not isinstance(e, AssertionError):
			traceback.print_exc()
print('match_flag differences excused by a rounding-level tie: %d rows in %d configurations' % (fz.TIE_EXCUSES['rows'], fz.TIE_EXCUSES['configurations']))
print('%d configurations, %d rows, %d failures %s in %.0f s' % (hi - lo, rows, len(bad), bad, time.time() - t0))
