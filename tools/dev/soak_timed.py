"""development aid: tools/dev/soak.py with a time limit per configuration (a dense k = 5 case keeps
the numpy oracle busy for minutes: skipped) and the time of every one
    python tools/dev/soak_timed.py 400 800 [seconds per configuration]        (on the GPU box; SOAK_MISSING=1: NaN coordinates in the all-sky cases)
"""
import os
import signal
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import test_hip_fuzz as fz
import nway_amd as nw

lo, hi = int(sys.argv[1]), int(sys.argv[2])
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 20


class TooSlow(Exception):
	pass


def alarm(signum, frame):
	raise TooSlow()


signal.signal(signal.SIGALRM, alarm)
bad, skipped, rows, t_all = [], [], 0, time.time()
for seed in range(lo, hi):
	rng = np.random.default_rng(1000 + seed)
	k = int(rng.integers(2, 6))
	tabs, radius = (fz.flat_case if seed % 2 == 0 else fz.sphere_case)(rng, k)
	if seed % 2 == 1 and k > 4:
		tabs = tabs[:4]
	comp = float(rng.choice([1.0, 0.9, 0.5]))
	if os.environ.get('SOAK_MISSING') and seed % 2 == 1:
		# sources without a coordinate (they match nothing and must not disturb the others): a few per catalogue, all-sky cases
		for t in tabs:
			for col in ('ra', 'dec'):
				n = len(t[col])
				if n > 8:
					t[col] = np.array(t[col], dtype=float)
					t[col][rng.choice(n, size=int(rng.integers(0, 4)), replace=False)] = np.nan
	excused0 = fz.TIE_EXCUSES['rows']
	t0 = time.time()
	signal.alarm(limit)
	try:
		rows += fz.compare(nw, tabs, radius, comp, 'cli' if seed % 3 == 0 else 'api', f32=(seed % 5 == 0))
	except TooSlow:
		skipped.append(seed)
	except Exception as e:
		bad.append(seed)
		print('seed %d FAILED: %s' % (seed, str(e).strip().splitlines()[0][:200]), flush=True)
	finally:
		signal.alarm(0)
	if fz.TIE_EXCUSES['rows'] != excused0:
		print('seed %d: %d match_flag difference(s) excused by a rounding-level tie' % (seed, fz.TIE_EXCUSES['rows'] - excused0), flush=True)
	if os.environ.get('SOAK_VERBOSE'):
		print(seed, k, '%.2f s' % (time.time() - t0), flush=True)
print('match_flag differences excused by a rounding-level tie: %d rows in %d configurations' % (fz.TIE_EXCUSES['rows'], fz.TIE_EXCUSES['configurations']))
print('%d configurations, %d rows, %d failures %s, %d skipped as too slow for the oracle %s in %.0f s' % (hi - lo, rows, len(bad), bad, len(skipped), skipped, time.time() - t_all))
