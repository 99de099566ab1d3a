"""development aid: where the host time of nway_amd.nway_match goes (cProfile, cumulative), per configuration
    python tools/dev/prof_api.py [c1|c2|ell2|ell3]
"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import nway_amd
from goldenutil import ell_tables, xmm_tables, mag3_tables

log = nway_amd.NullOutputLogger()
X, R, O = ell_tables()
XM, OP, IR = xmm_tables()
cases = dict(c1=([XM, OP], 20., 0.9), c2=([XM, OP, IR], 20., 0.9), ell2=([X, O], 10., 1.0), ell3=([X, R, O], 10., 1.0), c2m=(None, 20., 0.9))
nway_amd.nway_match([X, O], 10., 1.0, logger=log)
for name in (sys.argv[1:] or ['c1', 'c2', 'ell3']):
	tabs, radius, c = cases[name]
	kw = {}
	if tabs is None:  # (configs[1] with its three magnitude priors: nway_match edits the magnitude columns in place -- fresh tables per call)
		kw = dict(store_mag_hists=False)
		fresh = [mag3_tables() for _ in range(12)]

	for _ in range(2):
		nway_amd.nway_match(tabs if tabs is not None else fresh.pop(), radius, c, logger=log, **kw)
	best = 1e9
	for _ in range(5):
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		df = nway_amd.nway_match(tabs if tabs is not None else fresh.pop(), radius, c, logger=log, **kw)
		best = min(best, time.perf_counter() - t0)
	print('==== %s: %d rows, best of 5: %.2f ms' % (name, len(df), best * 1e3))
	pr = cProfile.Profile()
	pr.enable()
	for _ in range(5):
		nway_amd.nway_match(tabs if tabs is not None else fresh.pop(), radius, c, logger=log, **kw)
	pr.disable()
	st = pstats.Stats(pr, stream=sys.stdout)
	st.sort_stats('cumulative').print_stats(45)
