"""development aid: what the registration costs while OTHER compute units stream at the HBM rate.  Two HIP streams with CU masks
(hipExtStreamCreateWithCUMask): the registration (the front of a pass over a tiny secondary catalogue) on the last `reg_cus` CUs, the
read probe (the sweep's access pattern, nothing behind the loads) on the others.  Stage times from the plan's own events.
    python tools/dev/cumask_probe.py [reg_cus]        (on the GPU box)
"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import nway_amd
from nway_amd import _hip

reg_cus = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
ncu = torch.cuda.get_device_properties(dev).multi_processor_count
hip = ctypes.CDLL('libamdhip64.so')
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]


def masked_stream(cus):
	words = (ncu + 31) // 32
	mask = (ctypes.c_uint32 * words)()
	for c in cus:
		mask[c // 32] |= 1 << (c % 32)
	s = ctypes.c_void_p()
	rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), words, mask)
	assert rc == 0, rc
	return torch.cuda.ExternalStream(s.value, device=dev)


lib = _hip.load()
primary, secondary = bench.make_workload(100000, 10000000, 1)
small = dict(secondary, ra=secondary['ra'][:20000].copy(), dec=secondary['dec'][:20000].copy())
log = nway_amd.NullOutputLogger()
err = 5. / 3600
tables = [primary, small]
dens, dens_plus = nway_amd._densities_from_sizes(['P', 'S'], [100000, 10000000], [bench.SKY_AREA] * 2, log)
comp = nway_amd._completeness_vector(0.9, 2)
params = _hip.make_params(2, _hip.SCHEME_SPHERE, 5., err, dens, dens_plus, nway_amd._prior_table(dens, dens_plus, comp))
cats = [_hip.DeviceCatalogue(t['ra'], t['dec'], np.asarray(t['error'], dtype=float), dev) for t in tables]
sizes = [c.n for c in cats]
plan, st = _hip.run_plan(sizes, params, cats, 400000, 400000, dev, lean=True)
print('plan of the small pass:', plan.description, 'rows', int(st[_hip.ST_ROWS]))
big = [_hip.DeviceCatalogue(secondary['ra'], secondary['dec'], 0.1, dev)]
for _ in range(2):
	cp = _hip.DeviceCatalogue.__new__(_hip.DeviceCatalogue)
	cp.ra, cp.dec, cp.sigma, cp.sigma_const, cp.n = big[0].ra.clone(), big[0].dec.clone(), None, big[0].sigma_const, big[0].n
	big.append(cp)
out = torch.zeros(1 << 16, dtype=torch.float64, device=dev)
s_all_a, s_all_b = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
s_stream = masked_stream(range(0, ncu - reg_cus))
s_reg = masked_stream(range(ncu - reg_cus, ncu))


def probes(stream, n, blocks):
	for i in range(n):
		c = big[i % 3]
		_hip.check(lib.nwayhip_read_probe(_hip.ptr(c.ra), _hip.ptr(c.dec), c.n, _hip.ptr(out), blocks, ctypes.c_void_p(stream.cuda_stream)))


def run(label, reg_stream, probe_stream, probe_blocks, passes=12):
	torch.cuda.synchronize()
	plan.profile(0xff)
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	nprobe = 0
	if probe_stream is not None:
		nprobe = 60
		e0.record(probe_stream)
		probes(probe_stream, nprobe, probe_blocks)
		e1.record(probe_stream)
		time.sleep(0.0002)
	for _ in range(passes):
		plan.enqueue(cats, stream=ctypes.c_void_p(reg_stream.cuda_stream))
	torch.cuda.synchronize()
	nl, ms = plan.profile_read()
	plan.profile(0)
	stage = dict((nm, 1e3 * m / max(n, 1)) for nm, m, n in zip(_hip.STAGE_NAMES, ms, nl))
	rate = ''
	if nprobe:
		us = e0.elapsed_time(e1) * 1e3 / nprobe
		rate = '   probe: %.1f us per 160 MB = %.2f TB/s' % (us, 160e6 / us / 1e6)
	print('%-78s register %.1f us, sweep (tiny) %.1f, tail %.1f%s' % (label, stage['register'], stage['sweep'], stage['rows'], rate))


for rep in range(2):
	run('registration alone, all CUs', s_all_a, None, 0)
	run('registration alone on its %d CUs' % reg_cus, s_reg, None, 0)
	run('registration on all CUs, the probe on all CUs too (shared)', s_all_a, s_all_b, ncu)
	run('registration on its %d CUs, the probe on the other %d' % (reg_cus, ncu - reg_cus), s_reg, s_stream, ncu - reg_cus)
	run('probe alone on %d CUs (registration stream idle but for the passes after it)' % (ncu - reg_cus), s_all_a, s_stream, ncu - reg_cus, passes=0)
