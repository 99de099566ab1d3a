"""development aid: phase stamps (common.inc: dbg_stamp) of the three launches of a zone launch set -- the c5 job (5e5 x 1e8)
as Z zones on one GPU.  Needs a -DNWAYHIP_DEVBUILD library (tools/dev/build_variants.sh dev "-DNWAYHIP_DEVBUILD").

    python tools/dev/phase_zones.py [zones] [n_primary] [n_secondary]          (on the GPU box)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
dev = torch.device('cuda', 0)
BLOCKS, STAMPS = 1024, 8
buf = torch.zeros(4 * BLOCKS * STAMPS, dtype=torch.int64, device=dev)
os.environ["NWAYHIP_DEV"] = "1"
os.environ["NWAYHIP_DBG_PTR"] = str(buf.data_ptr())
os.environ.setdefault("NWAYHIP_LIBRARY", os.path.join(ROOT, "tools", "dev", "bin", "lib_dev.so"))
import bench  # noqa: E402
from nway_amd import distributed  # noqa: E402

zpr = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n0 = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
n1 = int(sys.argv[3]) if len(sys.argv) > 3 else 100000000
tabs = list(bench.make_workload(n0, n1, 78))
eng = distributed.ZoneShardedMatch(tabs[0], tabs[1:], 5.0, 0.9, dev, zones_per_rank=zpr, local_only=True)
for _ in range(5):
	eng.step()
torch.cuda.synchronize()
assert eng.batched
names = {0: (('k_register_pre_zones', ['start', 'ra / dec landed', 'trig done, barrier', 'records staged', 'records written', 'end']) if os.environ.get('PHASE_OWNER', '1') == '1' else ('k_register_x_zones', ['start', None, 'claims + stores landed', 'end'])),
	1: ('k_sweep_zones', ['start', 'bitmap in LDS', 'wave 0 done streaming', 'all waves done', 'probes landed', 'end']),
	3: ('k_claim_zones', ['start', 'run lengths scanned, slice set up', 'records claimed (this thread)', 'all claimed', 'slice written back']),
	2: ('k_tail2_zones', ['start', 'cnt/slot/sigma landed', 'block scan', 'lookback', 'rows landed', 'group stats landed', 'items set up', 'separations done'])}
acc = {}
for rep in range(6):
	buf.zero_()
	eng.step()
	torch.cuda.synchronize()
	t = buf.cpu().numpy().reshape(4, BLOCKS, STAMPS)
	for k, (kname, labels) in names.items():
		used = t[k][:, 0] > 0
		tk = t[k][used][:, :len(labels)].astype(np.float64) * 0.01
		t0 = tk[:, 0].min()
		acc.setdefault(k, []).append((tk - t0, used.sum()))
for k, (kname, labels) in names.items():
	rel = np.stack([a for a, _ in acc[k]])
	print('%s: %d workgroups stamped (the first %d of the launch); us since the first started (mean | p90 | latest), mean of 6 runs' % (kname, acc[k][0][1], BLOCKS))
	for i, lab in enumerate(labels):
		if lab is not None:
			print('    %-26s %7.2f | %7.2f | %7.2f' % (lab, rel[:, :, i].mean(), np.percentile(rel[:, :, i], 90, axis=1).mean(), rel[:, :, i].max(axis=1).mean()))
t = buf.cpu().numpy().reshape(4, BLOCKS, STAMPS).astype(np.float64) * 0.01
f = lambda k, i, red: red(t[k][t[k][:, 0] > 0][:, i])
print('last run: register start -> sweep start %.2f, sweep start -> tail start %.2f, tail start -> last stamped tail workgroup end %.2f us' % (
	f(1, 0, np.min) - f(0, 0, np.min), f(2, 0, np.min) - f(1, 0, np.min), f(2, 5, np.max) - f(2, 0, np.min)))
print('sweep workgroups: survivors flushed etc. are not stamped; zones %d, plan of zone 0: %s' % (zpr, eng.zones[0]['plan'].description))
