"""development aid: the read probe of the library (nwayhip_read_probe, the sweep's access pattern) timed the way tools/dev/ubench_read.hip
times its kernels -- many launches back to back -- to tell the kernel's rate from the way bench.py measures it"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nway_amd import _hip
lib = _hip.load()
dev = torch.device('cuda', 0)
n = 10000000
bufs = [(torch.zeros(n, dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.float64, device=dev)) for _ in range(3)]
out = torch.zeros(1024, dtype=torch.float64, device=dev)
stream = _hip.current_stream_ptr(dev)
ptrs = [(_hip.ptr(a), _hip.ptr(b)) for a, b in bufs]
po = _hip.ptr(out)
for blocks in (256, 512, 1024):
	for reps in (12, 30, 200):
		for i in range(50):
			lib.nwayhip_read_probe(ptrs[i % 3][0], ptrs[i % 3][1], n, po, blocks, stream)
		torch.cuda.synchronize()
		e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
		t0 = time.perf_counter()
		e0.record()
		for i in range(reps):
			lib.nwayhip_read_probe(ptrs[i % 3][0], ptrs[i % 3][1], n, po, blocks, stream)
		t1 = time.perf_counter()
		e1.record()
		e1.synchronize()
		us = e0.elapsed_time(e1) * 1e3 / reps
		print('blocks %4d reps %3d: %.2f us per launch = %.3f TB/s (host: %.2f us per call)' % (blocks, reps, us, 16.0 * n / us * 1e-6, (t1 - t0) * 1e6 / reps))
