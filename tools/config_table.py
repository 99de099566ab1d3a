"""All synthetic configurations of SURVEY.md section 8(d) on one GPU: rows, distance tests, time
per pass, rows/s and the algorithmic-byte rate of the WHOLE pass (every input
column read once -- ra, dec, and sigma where it is a column: the primaries here; the secondaries
carry a scalar error -- plus 66 / 94 B per output row for k = 2 / 3) as a markdown table.

    python tools/config_table.py > profiles/configs_r01.md      (on the GPU box)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # (the FP64 model beside the byte model: bench.alu_model)
CONFIGS = [
	('C3-S 2-way 1e5 x 1e7, uniform sky, 5"', ['c3s'], 2, [100000, 10000000]),
	('C3-D 2-way 1e5 x 1e7, 6 deg^2 patch (flat cells), 5"', ['c3d'], 2, [100000, 10000000]),
	('C4-S 3-way 1e5 x 1e6 x 1e6, uniform sky, 10"', ['c4s'], 3, [100000, 1000000, 1000000]),
	('C4-D 3-way 1e5 x 1e6 x 1e6, 8 deg^2 patch, 10"', ['c4d'], 3, [100000, 1000000, 1000000]),
	('C5 shard (1 of 8 GPUs) 2-way 62500 x 1e8, uniform sky, 5"', ['c3s', '62500', '100000000'], 2, [62500, 100000000]),
	('C5 whole on ONE GPU 2-way 5e5 x 1e8, uniform sky, 5"', ['c3s', '500000', '100000000'], 2, [500000, 100000000]),
	("C1' BASELINE configs[0] stand-in: real COSMOS_XMM 1 797 x seeded OPT 560 536, 2 deg^2, 20\"", ['c1x'], 2, [1797, 560536]),
	("C2' BASELINE configs[1] stand-in: the same x seeded IRAC 345 512 (3-way), 20\"", ['c2x'], 3, [1797, 560536, 345512]),
]
print('| configuration | path | rows M | distance tests | us per pass | rows/s | B_alg (MB) | B_alg / t (GB/s) | of 8 TB/s | FP64 model (Mflop) | of %g TFLOP/s | bound | stages (us, each bracketed by events) |' % bench.FP64_PEAK_TFLOPS)
print('|---|---|---|---|---|---|---|---|---|---|---|---|---|')
for name, args, k, sizes in CONFIGS:
	out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'status_probe.py')] + args, stdout=subprocess.PIPE,
		stderr=subprocess.STDOUT, universal_newlines=True).stdout
	st = re.search(r'status \[(\d+), (\d+), (\d+), (\d+)\]', out)
	tot = re.search(r'wall \(no stage events\): ([0-9.]+) us/step', out)
	path = re.search(r'path: (\w+)', out)
	plan = re.search(r'plan: (.*)', out)
	stages = re.search(r'stages us/step: (.*?) \|', out)
	if not st or not tot:
		print('| %s | failed | | | | | | | | | | | |' % name)
		sys.stderr.write(out)
		continue
	rows, tests, us = int(st.group(1)), int(st.group(4)), float(tot.group(1))
	b_alg = 24.0 * sizes[0] + 16.0 * sum(sizes[1:]) + (66.0 if k == 2 else 94.0) * rows
	rate = b_alg / (us * 1e-6) / 1e9
	alu = bench.alu_model(k, sizes[0], rows, tests, us * 1e-3)
	print('| %s | %s | %d | %d | %.1f | %.3g | %.1f | %.0f | %.2f | %.0f | %.3f | %s | %s |' % (name, (path.group(1) if path else '?') + (' (%s)' % ' '.join(x for x in plan.group(1).split() if x.split('=')[0] in ('sweep', 'tail', 'link_slots', 'direct_log2')) if plan else ''), rows, tests, us, rows / (us * 1e-6), b_alg / 1e6, rate, rate / 8000.,
		alu['fp64_flops'] / 1e6, alu['frac'], bench.bound_of(rate / 8000., alu['frac']).split(' (')[0],
		' '.join(x for x in (stages.group(1) if stages else '').split() if not x.endswith('=0.0'))))
