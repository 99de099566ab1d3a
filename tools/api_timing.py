"""Wall-clock of the drop-in API (host arrays in, pandas DataFrame out) on the reference's own
fixtures (development aid):  python tools/api_timing.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
import nway_amd
from goldenutil import ell_tables, xmm_tables, mag3_tables

log = nway_amd.NullOutputLogger()
X, R, O = ell_tables()
XM, OP, IR = xmm_tables()
cases = [('elltest 2-way r=10', [X, O], 10., 1.0), ('elltest 3-way r=10', [X, R, O], 10., 1.0),
	('xmm stand-in 2-way r=20', [XM, OP], 20., 0.9), ('xmm stand-in 3-way r=20', [XM, OP, IR], 20., 0.9),
	('xmm 3-way + 3 mag priors r=20', None, 20., 0.9)]
nway_amd.nway_match([X, O], 10., 1.0, logger=log)  # warm-up: library load, context
for name, tabs, radius, c in cases:
	best = 1e9
	for _ in range(3):
		tt = tabs if tabs is not None else mag3_tables()  # nway_match edits magnitude columns in place
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		df = nway_amd.nway_match(tt, radius, c, logger=log, store_mag_hists=False)
		best = min(best, time.perf_counter() - t0)
	print('%-28s %8d rows  %7.1f ms' % (name, len(df), best * 1e3))
