"""profiles/alu_<tag>.md from the output of tools/profile_alu.sh: per configuration the pass (rows, tests, time, both roofline
fractions, bench.alu_model / bench.bound_of) and per kernel of the pass its mean duration (kernel trace), its vector instructions
and the share of its duration the vector units were issuing (SQ counters).

    python tools/summarize_alu.py gpurun_out/alu_r06 > profiles/alu_r06.md
"""
import csv
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

out = sys.argv[1]
SIMDS, CLOCK = 1024, 2.4e9
NAMES = dict(c3s='C3-S 2-way 1e5 x 1e7, 5" (configs[2])', c4s='C4-S 3-way 1e5 x 1e6 x 1e6, 10" (configs[3])', c3d='C3-D 2-way 1e5 x 1e7 in 6 deg^2',
	c4d='C4-D 3-way in 8 deg^2', c1x="C1' configs[0] stand-in", c2x="C2' configs[1] stand-in")
SIZES = dict(c3s=(2, [100000, 10000000]), c4s=(3, [100000, 1000000, 1000000]), c3d=(2, [100000, 10000000]), c4d=(3, [100000, 1000000, 1000000]),
	c1x=(2, [1797, 560536]), c2x=(3, [1797, 560536, 345512]))
short = lambda n: n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
print('# FP64 / issue side of the configurations (tools/profile_alu.sh; kernel trace and ONE pass of SQ counters per configuration, tools/status_probe.py)')
print('# VALU busy = SQ_ACTIVE_INST_VALU (quad-cycles, summed over waves) x 4 / %d SIMDs / %.1f GHz: the time the average SIMD spent issuing vector' % (SIMDS, CLOCK / 1e9))
print('# instructions; its share of the kernel\'s duration is the issue fraction.  FP64 model and bound: bench.py alu_model / bound_of.')
print()
for cfg in ('c3s', 'c4s', 'c3d', 'c4d', 'c1x', 'c2x'):
	log = os.path.join(out, cfg + '.trace.log')
	if not os.path.exists(log):
		continue
	text = open(log).read()
	st = re.search(r'status \[(\d+), (\d+), (\d+), (\d+)\]', text)
	tot = re.search(r'wall \(no stage events\): ([0-9.]+) us/step', text)
	if not st or not tot:
		print('## %s: failed\n' % cfg)
		continue
	rows, tests, us = int(st.group(1)), int(st.group(4)), float(tot.group(1))
	k, sizes = SIZES[cfg]
	b_alg = 24.0 * sizes[0] + 16.0 * sum(sizes[1:]) + (66.0 if k == 2 else 94.0) * rows
	hbm = b_alg / (us * 1e-6) / 8e12
	alu = bench.alu_model(k, sizes[0], rows, tests, us * 1e-3)
	print('## %s: %d rows, %d distance tests, %.1f us per pass under the profiler; HBM fraction %.3f, FP64 model %.0f Mflop = %.3f of %g TFLOP/s -> bound: %s'
		% (NAMES[cfg], rows, tests, us, hbm, alu['fp64_flops'] / 1e6, alu['frac'], bench.FP64_PEAK_TFLOPS, bench.bound_of(hbm, alu['frac']).split(' (')[0]))
	dur = {}
	for f in glob.glob(os.path.join(out, cfg, 'trace', '*', '*kernel_stats.csv')):
		for r in csv.DictReader(open(f)):
			dur[short(r['Name'])] = (int(r['Calls']), float(r['AverageNs']) / 1e3)
	acc = {}
	for f in glob.glob(os.path.join(out, cfg, 'sq', '*', '*counter_collection.csv')):
		for r in csv.DictReader(open(f)):
			acc.setdefault((short(r['Kernel_Name']), r['Counter_Name']), []).append(float(r['Counter_Value']))
	mean = lambda kn, c: (sum(acc[(kn, c)]) / len(acc[(kn, c)])) if (kn, c) in acc else float('nan')
	print('| kernel | calls | us (trace) | waves | vector instructions | VALU busy us | share of the duration | waiting for an instruction (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES) |')
	print('|---|---|---|---|---|---|---|---|')
	for kn, (calls, avg) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
		if not kn.startswith('k_'):
			continue
		busy = mean(kn, 'SQ_ACTIVE_INST_VALU') * 4 / SIMDS / CLOCK * 1e6
		print('| %s | %d | %.1f | %.0f | %.3g | %.1f | %.2f | %.2f |' % (kn, calls, avg, mean(kn, 'SQ_WAVES'), mean(kn, 'SQ_INSTS_VALU'), busy, busy / avg,
			mean(kn, 'SQ_WAIT_INST_ANY') / max(mean(kn, 'SQ_WAVE_CYCLES'), 1.0)))
	print()
