import os, sys, time
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import numpy as np, torch
import nway_amd
from goldenutil import ell_tables, xmm_tables
log = nway_amd.NullOutputLogger()
X, R, O = ell_tables()
XM, OP, IR = xmm_tables()
print([len(t['ra']) for t in (X,R,O)], [t['area'] for t in (X,R,O)])
nway_amd.nway_match([X, O], 10., 1.0, logger=log)
for name, tabs, radius, c in [('ell2',[X,O],10.,1.0),('ell3',[X,R,O],10.,1.0),('xmm2',[XM,OP],20.,0.9),('xmm3',[XM,OP,IR],20.,0.9)]:
	for slots in (0, -1):
		best=1e9
		for _ in range(3):
			torch.cuda.synchronize(); t0=time.perf_counter()
			res = nway_amd.run_match(tabs, radius, c, logger=log, link_slots=slots, lean=True)
			torch.cuda.synchronize(); best=min(best,time.perf_counter()-t0)
			path, att, ls = res.plan.path, res.plan.attempts, res.plan.link_slots
			res.plan.close()
		print(name, 'slots', slots, 'path', path, 'link_slots', ls, 'attempts', att, '%.2f ms' % (best*1e3))
