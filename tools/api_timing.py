"""Wall-clock of the drop-in API -- host arrays in, pandas DataFrame out, everything included (densities,
scheme choice, upload, capacity estimate and its repeats, the pass, download, DataFrame) -- on the
reference's own fixtures and on the stand-ins of BASELINE configs[0] / [1], as a markdown table next to
what the reference itself took in the survey container (one thread of an 8-core
Xeon @ 2.1 GHz; it cannot travel to the GPU box).

    python tools/api_timing.py > profiles/api_timing_r03.md      (on the GPU box)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import nway_amd
from goldenutil import ell_tables, xmm_tables, mag3_tables

log = nway_amd.NullOutputLogger()
X, R, O = ell_tables()
XM, OP, IR = xmm_tables()
# (name, tables, radius, completeness, reference rows, reference seconds in the survey container)
cases = [('tests/elltest X x O, 2-way, 10"', [X, O], 10., 1.0, 37706, 0.45),
	('tests/elltest X x R x O, 3-way, 10"', [X, R, O], 10., 1.0, 450435, 8.5),
	("C1' (configs[0] stand-in) COSMOS_XMM x OPT, 2-way, 20\"", [XM, OP], 20., 0.9, 44909, 2.38),
	("C2' (configs[1] stand-in) COSMOS_XMM x OPT x IRAC, 3-way, 20\"", [XM, OP, IR], 20., 0.9, 449459, 17.9),
	("C2' with three magnitude priors (auto)", None, 20., 0.9, None, None)]
nway_amd.nway_match([X, O], 10., 1.0, logger=log)  # warm-up: library load, context
print('| input | rows | nway_amd.nway_match, host arrays -> DataFrame (ms, best of 3) | reference nwaylib.nway_match (s; BASELINE.md section 2) | ratio |')
print('|---|---|---|---|---|')
for name, tabs, radius, c, ref_rows, ref_s in cases:
	best = 1e9
	for _ in range(3):
		tt = tabs if tabs is not None else mag3_tables()  # nway_match edits magnitude columns in place
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		df = nway_amd.nway_match(tt, radius, c, logger=log, store_mag_hists=False)
		best = min(best, time.perf_counter() - t0)
	assert ref_rows is None or len(df) == ref_rows, (name, len(df), ref_rows)
	print('| %s | %d | %.1f | %s | %s |' % (name, len(df), best * 1e3, '%.2f' % ref_s if ref_s else 'not measured', '%.0fx' % (ref_s / best) if ref_s else ''))
